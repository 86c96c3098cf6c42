"""The locality order of the particles ahead of a scoring launch (k_order, gms_pf_kernels.hip) only decides which lane
forms which particle's product: every result must be bit-identical with the order forced on (GMS_SCORE_ORDER=1),
forced off (=0), and equal to the oracle.  Covers particle counts that are not a multiple of the workgroup, fewer
particles than one workgroup, batched maps, poses that are not finite, clouds that straddle the +-pi seam, the poses
entering through the scan step (device and host buffers) and the stand-alone scoring entry points."""
import numpy as np
import pytest

from gridmap_slam_robot_amd import GridMap, ParticleFilter, synth
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def filt(monkeypatch, m, n, mode):
    monkeypatch.setenv("GMS_SCORE_ORDER", str(mode))          # read when the filter is created
    return ParticleFilter(m, n)


def built_map(n_maps=1, beams=180, seed=3, scans=6):
    tr = synth.make_trace(12.8, 0.05, beams, T=16, seed=seed, n_scans=scans + 4)
    m = GridMap(12.8, 12.8, 0.05, (-6.4, -6.4), n_maps=n_maps)
    for t in range(scans):
        if n_maps == 1:
            m.update(tr.scans[t], tr.poses[t])
        else:
            m.update(np.stack([tr.scans[t]] * n_maps), np.stack([tr.poses[t]] * n_maps))
    return tr, m


@pytest.mark.parametrize("n", [1, 63, 300, 1024, 1500, 5000])
def test_weights_identical_with_and_without_the_order_and_equal_to_the_oracle(monkeypatch, n):
    tr, m = built_map()
    g = orc.Grid(12.8, 12.8, 0.05, -6.4, -6.4)
    lik = m.download_likelihood()
    P = synth.make_particles(tr.poses[6], n, seed=n, sigma_xy=0.10, sigma_theta_deg=5.0)
    out = {}
    for mode in (0, 1):
        pf = filt(monkeypatch, m, n, mode)
        pf.set_poses(P)
        pf.score(tr.scans[6])
        out[mode] = (pf.get_weights().copy(), pf.get_log_weights().copy(), pf.get_poses().copy())
        pf.close()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    assert np.array_equal(out[1][2], P)                                   # the filter's own pose order is untouched
    k = min(n, 200)
    want = g.score(lik, tr.scans[6], P[:k])                               # GridMap.java:260-291
    ok = want > 1e-290
    assert ok.any() and np.max(np.abs(out[1][0][:k][ok] - want[ok]) / want[ok]) <= 1e-11      # (segment products: tests/test_gpu_parity.py's TIGHT)


def test_scan_steps_identical_with_and_without_the_order(monkeypatch):
    import torch
    tr, _ = built_map()
    n, T = 2500, 5
    res = {}
    for mode in (0, 1):
        _, m = built_map()
        m.set_stream(torch.cuda.current_stream().cuda_stream)
        pf = filt(monkeypatch, m, n, mode)
        rows = []
        for t in range(T):
            P = synth.make_particles(tr.poses[6 + t % 3], n, seed=50 + t, sigma_xy=0.05, sigma_theta_deg=2.0)
            beams = tr.scans[6 + t % 3]
            if t % 2 == 0:                                                # poses and scan as device buffers ...
                Pd = torch.from_numpy(P).cuda()
                bd = torch.from_numpy(beams.view(np.uint8).copy()).cuda()
                pf.slam_update_dev(Pd.data_ptr(), bd.data_ptr(), len(beams), 0.37 + 0.1 * t, 0.9)
                torch.cuda.synchronize()
            else:                                                         # ... and as host arrays
                pf.slam_update(P, beams, 0.37 + 0.1 * t, 0.9)
            rows.append((pf.get_poses().copy(), pf.get_weights().copy(), dict(pf.last_step())))
        res[mode] = (rows, m.download_log().copy(), m.download_likelihood().copy())
        pf.close(); m.close()
    for (p0, w0, s0), (p1, w1, s1) in zip(res[0][0], res[1][0]):
        assert np.array_equal(p0, p1) and np.array_equal(w0, w1)
        assert s0.keys() == s1.keys() and all(np.array_equal(s0[k], s1[k], equal_nan=True) for k in s0)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


def test_batched_maps_and_odd_poses(monkeypatch):
    M, n = 3, 1300
    tr, m = built_map(n_maps=M)
    P = np.stack([synth.make_particles(tr.poses[6], n, seed=9 + i, sigma_xy=0.2, sigma_theta_deg=20.0) for i in range(M)])
    P[0, 5] = (np.nan, 0.0, 0.0)                       # not finite: scored as the reference scores them, never a bad index
    P[1, 7] = (0.0, np.inf, 1.0)
    P[2, 11] = (1e30, -1e30, 0.5)
    P[1, 100:200, 2] += np.float32(np.pi)              # a cloud on both sides of the +-pi seam
    P[1, 100:200, 2] = np.where(P[1, 100:200, 2] > np.pi, P[1, 100:200, 2] - np.float32(2 * np.pi), P[1, 100:200, 2])
    scans = np.stack([tr.scans[6]] * M)
    out = {}
    for mode in (0, 1):
        pf = filt(monkeypatch, m, n, mode)
        pf.set_poses(P)
        pf.score(scans)
        out[mode] = (pf.get_weights().copy(), pf.get_log_weights().copy())
        pf.close()
    assert np.array_equal(out[0][0], out[1][0], equal_nan=True) and np.array_equal(out[0][1], out[1][1], equal_nan=True)
    assert out[1][0].shape == (M, n)


def test_the_order_is_booked_under_its_own_profile_class(monkeypatch):
    tr, m = built_map()
    n = 2048
    P = synth.make_particles(tr.poses[6], n, seed=1)
    for mode, want in ((0, 0), (1, 3), (-1, 0)):       # -1: the launcher's rule; 2048 particles x 180 beams is far below it
        pf = filt(monkeypatch, m, n, mode)
        pf.set_poses(P)
        m.profile_reset(); m.profile(True)
        for _ in range(3):
            pf.score(tr.scans[6])
        prof = m.profile_get(); m.profile(False)
        assert prof["order"][1] == want and prof["score"][1] >= 3
        pf.close()
