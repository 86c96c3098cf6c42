"""The reference-order audit path (gms_pf_set_reference_order): the three chains the default kernels re-associate -- the product of
a scan's factors (GridMap.java:262-288), weightSum (SLAM.java:100) and the cumulative weights of resample() (SLAM.java:137-144) -- each
as ONE chain in the reference's own order.  Against the oracle that is EQUALITY: every raw weight bit for bit, the zeros and the
denormal band included, the weight sum, Neff, the strongest particle, and every resampling index with no allowance.  Then the default
path against the audit path: what differs between the two is association and nothing else, and how little that is gets recorded
(gpurun_out/reference_order_audit.json)."""
import json
import os

import numpy as np
import pytest

from gridmap_slam_robot_amd import GridMap, ParticleFilter, synth
from gridmap_slam_robot_amd._lib import GMS_ERR_STATE, GmsError
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c3():
    c = synth.CONFIGS["C3"]
    ext, res, B, N = c["extent"], c["resolution"], c["beams"], c["particles"]
    T = 64
    tr = synth.make_trace(ext, res, B, T=T, seed=1234, n_scans=T // 2 + 2)
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    m = GridMap(ext, ext, res, (-ext / 2, -ext / 2), max_beams=2048)
    for t in range(T // 2):
        m.update(tr.scans[t], tr.poses[t])
    lik = m.download_likelihood().reshape(-1)
    return tr, g, m, lik, N, T


def test_c3_at_the_bench_cloud_equals_the_oracle_bit_for_bit_in_reference_order():
    tr, g, m, lik, N, T = _c3()
    audit, plain = ParticleFilter(m, N), ParticleFilter(m, N)
    audit.set_reference_order(True)
    record = []
    for s in range(2):
        t = T // 2 + s
        P = synth.make_particles(tr.poses[t], N, seed=99 + s)             # bench.py's cloud: most raw products underflow
        want = g.score(lik, tr.scans[t], P)
        n_zero, n_denormal = int((want == 0).sum()), int(((want > 0) & (want < 2.2250738585072014e-308)).sum())
        assert n_zero > N // 2 and n_denormal >= 1, "this cloud must exercise the zeros and the denormal band"
        for pf in (audit, plain):
            pf.set_poses(P)
            pf.score(tr.scans[t])
        w = audit.get_weights()
        assert np.array_equal(w, want)                                    # ALL 16 384 products, zeros and denormals included
        wp = plain.get_weights()
        big = want > 1e-290
        dev_rel = float(np.max(np.abs(wp[big] - want[big]) / want[big]))
        assert dev_rel <= 1e-11
        st = audit.normalize()
        wn = want.copy()
        ws, strongest = orc.normalize(wn)
        assert st["weight_sum"] == ws and st["strongest"] == strongest and st["n_zero"] == n_zero
        assert np.array_equal(audit.get_weights(), wn)                    # weight /= weightSum: the same division of the same operands
        assert st["neff"] == orc.neff(wn)
        assert np.array_equal(audit.weighted_pose(), orc.weighted_pose(P, wn))
        idx, amb = audit.resample(0.37, want_indices=True)
        want_idx, clamped = orc.resample_indices(wn.copy(), 0.37)
        assert clamped == 0 and amb == 0 and np.array_equal(idx, want_idx)   # no allowance
        assert np.array_equal(audit.get_poses(), P[want_idx]) and np.array_equal(audit.get_weights(), wn[want_idx])
        # the default path against the audit path
        stp = plain.normalize()
        wnp = plain.get_weights()
        idxp, ambp = plain.resample(0.37, want_indices=True)
        differing = int((idxp != idx).sum())
        assert differing <= ambp and (np.abs(idxp.astype(np.int64) - idx) <= 1).all()
        record.append({"scan": int(t), "particles": int(N), "zero_products": n_zero, "denormal_products": n_denormal,
                       "raw_weight_max_rel_diff_default_vs_reference_order": dev_rel,
                       "weight_sum_rel_diff": float(abs(stp["weight_sum"] - ws) / ws),
                       "normalised_weight_max_rel_diff": float(np.max(np.abs(wnp[big] - wn[big]) / wn[big])),
                       "neff_rel_diff": float(abs(stp["neff"] - st["neff"]) / st["neff"]),
                       "resample_slots_differing": differing, "resample_slots_flagged_ambiguous": int(ambp)})
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "reference_order_audit.json"), "w") as f:
            json.dump({"what": "default path vs gms_pf_set_reference_order(1) at C3's bench cloud; the audit path itself equals the oracle bit for bit",
                       "scans": record}, f, indent=1)
    audit.close(); plain.close()


def test_the_scan_step_in_reference_order_and_the_options_it_excludes():
    """the scan step of an audit filter takes the separate launches and stays equal to the oracle; shards and log-normalisation refuse"""
    ext, res, B, N = 10.24, 0.05, 360, 3000
    tr = synth.make_trace(ext, res, B, T=16, seed=5)
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    m = GridMap(ext, ext, res, (-ext / 2, -ext / 2), max_beams=512)
    log = g.new_log()
    for t in range(6):
        g.integrate(log, tr.scans[t], tr.poses[t])
        m.update(tr.scans[t], tr.poses[t])
    lik = g.build_likelihood(log)
    pf = ParticleFilter(m, N)
    pf.set_reference_order(True)
    P = synth.make_particles(tr.poses[6], N, seed=3, sigma_xy=0.03, sigma_theta_deg=1.5)
    st = pf.slam_update(P, tr.scans[6], 0.73, 2.0, True, fetch=True)       # fraction 2: neff < 2 N always -> resample
    want = g.score(lik, tr.scans[6], P)
    wn = want.copy()
    ws, strongest = orc.normalize(wn)
    assert st["weight_sum"] == ws and st["strongest"] == strongest and st["neff"] == orc.neff(wn)
    want_idx, _ = orc.resample_indices(wn.copy(), 0.73)
    assert np.array_equal(pf.last_resample_indices().reshape(-1), want_idx)
    assert np.array_equal(pf.get_poses(), P[want_idx]) and np.array_equal(pf.get_weights(), wn[want_idx])
    # the map was updated at the weighted pose of the SCORED population (SLAM.java:165-178 on the normalised weights)
    wp = orc.weighted_pose(P, wn)
    assert np.array_equal(np.asarray(pf.last_step()["weighted_pose"], dtype=np.float32).reshape(-1)[:3], wp)
    g.integrate(log, tr.scans[6], wp)
    assert np.max(np.abs(m.download_log().reshape(-1) - log)) < 1e-10
    with pytest.raises(GmsError) as e:
        pf.set_log_normalize(True)
    assert e.value.code == GMS_ERR_STATE
    sh = ParticleFilter(m, 256)
    sh.set_shard(256, 1024)
    with pytest.raises(GmsError) as e:
        sh.set_reference_order(True)
    assert e.value.code == GMS_ERR_STATE
    pf.set_reference_order(False)                                          # off again: the default kernels, within their bars
    pf.set_poses(P)
    pf.score(tr.scans[6])
    want = g.score(g.build_likelihood(log), tr.scans[6], P)                # (the step above has integrated scan 6 into the map)
    big = want > 1e-290
    assert np.max(np.abs(pf.get_weights()[big] - want[big]) / want[big]) <= 1e-11


def test_a_batched_handle_in_reference_order_two_maps_of_config_5():
    """Equality with the oracle is not a single-map statement: two independent maps of config 5's shape (1024 x 1024 cells, 4096
    particles, 1080 beams each) in ONE batched handle, the audit path on: every raw weight of both maps bit for bit, each map's weight
    sum, strongest particle, Neff and weighted pose, and every resampling index of both draws with no allowance; then the batched scan
    step of that filter (separate launches) with the map updates at the two weighted poses."""
    c = synth.CONFIGS["C5"]
    M, B, N, ext, res = 2, c["beams"], c["particles"], c["extent"], c["resolution"]
    traces = [synth.make_trace(ext, res, B, T=8, seed=700 + i, n_scans=6) for i in range(M)]
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    mb = GridMap(ext, ext, res, (-ext / 2, -ext / 2), n_maps=M, max_beams=B)
    assert (mb.W, mb.H, mb.n_maps) == (1024, 1024, 2)
    logs = [g.new_log() for _ in range(M)]
    for t in range(3):
        mb.update(np.stack([tr.scans[t] for tr in traces]), np.stack([tr.poses[t] for tr in traces]))
        for i in range(M):
            g.integrate(logs[i], traces[i].scans[t], traces[i].poses[t])
    liks = mb.download_likelihood().reshape(M, -1)
    for i in range(M):
        assert np.array_equal(liks[i], g.build_likelihood(mb.download_log().reshape(M, -1)[i]))
    pf = ParticleFilter(mb, N)
    pf.set_reference_order(True)
    # a cloud wide enough that many products underflow (1080 factors) and a tight one, one per map
    Ph = np.stack([synth.make_particles(traces[0].poses[3], N, seed=1), synth.make_particles(traces[1].poses[3], N, seed=2, sigma_xy=0.02, sigma_theta_deg=0.3)])
    scans3 = np.stack([tr.scans[3] for tr in traces])
    pf.set_poses(Ph)
    pf.score(scans3)
    w = pf.get_weights()
    wants = [g.score(liks[i], scans3[i], Ph[i]) for i in range(M)]
    assert int((wants[0] == 0).sum()) > 0, "the wide cloud must exercise exact zeros"
    for i in range(M):
        assert np.array_equal(w[i], wants[i]), f"map {i}: {int((w[i] != wants[i]).sum())} raw weights differ"
    sts = pf.normalize()
    wn = pf.get_weights()
    wps = pf.weighted_pose()
    r01 = np.array([0.37, 0.81])
    idx, amb = pf.resample(r01, want_indices=True)
    for i in range(M):
        x = wants[i].copy()
        ws, strongest = orc.normalize(x)
        assert sts[i]["weight_sum"] == ws and sts[i]["strongest"] == strongest and sts[i]["neff"] == orc.neff(x)
        assert np.array_equal(wn[i], x)
        assert np.array_equal(wps[i], orc.weighted_pose(Ph[i], x))
        want_idx, clamped = orc.resample_indices(x.copy(), float(r01[i]))
        assert clamped == 0 and np.array_equal(idx[i], want_idx) and int(np.asarray(amb).reshape(-1)[i]) == 0
    assert np.array_equal(pf.get_poses(), np.stack([Ph[i][idx[i]] for i in range(M)]))
    # the batched scan step of the audit filter: scored, normalised and resampled in the reference's order, each map updated at its own weighted pose
    P4 = np.stack([synth.make_particles(traces[i].poses[4], N, seed=10 + i, sigma_xy=0.03, sigma_theta_deg=0.5) for i in range(M)])
    scans4 = np.stack([tr.scans[4] for tr in traces])
    liks_now = mb.download_likelihood().reshape(M, -1)
    st = pf.slam_update(P4, scans4, np.array([0.11, 0.52]), 2.0, True, fetch=True)       # fraction 2: always resample
    last_idx = pf.last_resample_indices()
    logs_after = mb.download_log().reshape(M, -1)
    for i in range(M):
        x = g.score(liks_now[i], scans4[i], P4[i])
        ws, strongest = orc.normalize(x)
        assert st[i]["weight_sum"] == ws and st[i]["strongest"] == strongest and st[i]["neff"] == orc.neff(x)
        want_idx, _ = orc.resample_indices(x.copy(), float([0.11, 0.52][i]))
        assert np.array_equal(last_idx[i], want_idx)
        wp = orc.weighted_pose(P4[i], x)
        g.integrate(logs[i], scans4[i], wp)
        assert np.array_equal(logs_after[i] != 0, logs[i] != 0) and np.max(np.abs(logs_after[i] - logs[i])) < 1e-10
    pf.close(); mb.close()
