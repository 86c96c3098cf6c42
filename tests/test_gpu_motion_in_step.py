"""SLAM.update(z, u) with its motion-model sample inside (SLAM.java:80-131, :90 included) as one call, gms_slam_update_u_dev: the
sample is taken inside the scoring launch.  Against the two calls it stands for -- gms_pf_sample_motion, then gms_slam_update_dev
on the poses in place -- step after step, bit for bit: poses, weights, log-weights, resampling indices, map and field."""
import numpy as np
import pytest
import torch

from gridmap_slam_robot_amd import GridMap, ParticleFilter, synth

pytestmark = pytest.mark.gpu


def _drive(one_call, N, B, n_maps, steps, refine=False, order=None, monkeypatch=None):
    ext, res = 6.4, 0.05
    if order is not None:
        monkeypatch.setenv("GMS_SCORE_ORDER", order)
    tr = synth.make_trace(ext, res, B, T=steps + 4, seed=6, n_scans=steps + 4)
    m = GridMap(ext, ext, res, (-ext / 2, -ext / 2), n_maps=n_maps)
    pf = ParticleFilter(m, N)
    for t in range(2):
        m.update(np.stack([tr.scans[t]] * n_maps) if n_maps > 1 else tr.scans[t], np.stack([tr.poses[t]] * n_maps) if n_maps > 1 else tr.poses[t])
    P = synth.make_particles(tr.poses[1], N, seed=3, sigma_xy=0.02, sigma_theta_deg=1.0)
    pf.set_poses(np.stack([P] * n_maps) if n_maps > 1 else P)
    if refine:
        pf.set_refine(True)
    d = tr.poses[2].astype(np.float64) - tr.poses[1].astype(np.float64)
    odo = (float(np.hypot(d[0], d[1])), float(d[2]))
    dev = torch.device("cuda", 0)
    out = []
    rng = np.random.default_rng(2)
    for k in range(steps):
        scan = np.stack([tr.scans[2 + k]] * n_maps) if n_maps > 1 else tr.scans[2 + k]
        beams = torch.from_numpy(scan.view(np.uint8).reshape(-1).copy()).to(dev)
        r01 = rng.random(n_maps)
        torch.cuda.synchronize()
        if one_call:
            pf.slam_update_u_dev(odo[0], odo[1], 77, k, beams.data_ptr(), B, r01, 0.5, True)
        else:
            pf.sample_motion(odo[0], odo[1], 77, k)
            pf.slam_update_dev(0, beams.data_ptr(), B, r01, 0.5, True)
        m.synchronize()
        rec = dict(poses=pf.get_poses().copy(), w=pf.get_weights().copy(), lw=pf.get_log_weights().copy(), idx=pf.last_resample_indices().copy())
        if k % 3 == 2 or k == steps - 1:
            rec["log"] = m.download_log().copy(); rec["lik"] = m.download_likelihood().copy()
        out.append(rec)
    pf.close(); m.close()
    return out


@pytest.mark.parametrize("N,B,n_maps,refine,order", [(1500, 120, 1, False, None), (300, 33, 1, False, None), (2048, 360, 1, False, None),
                                                      (512, 90, 2, False, None), (700, 120, 1, True, None), (1024, 120, 1, False, "1")])
def test_motion_inside_the_scoring_launch(monkeypatch, N, B, n_maps, refine, order):
    a = _drive(True, N, B, n_maps, 9, refine, order, monkeypatch)
    b = _drive(False, N, B, n_maps, 9, refine, order, monkeypatch)
    moved = False
    for k, (x, y) in enumerate(zip(a, b)):
        for key in x:
            assert np.array_equal(x[key], y[key], equal_nan=True), f"step {k}: {key}"
        moved = moved or (k > 0 and not np.array_equal(x["poses"], a[k - 1]["poses"]))
    assert moved
