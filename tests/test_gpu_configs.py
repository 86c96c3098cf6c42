"""BASELINE.json's remaining configurations at FULL size on the GPU, against the CPU oracle:

  C4  65 536 particles as 8 shards x 8 192 on one GPU (2048^2 @ 2 cm, 720 beams) through
      gms_slam_update_sharded_begin_dev / _end_dev, the all-gathers being device copies between the shards' gather
      buffers (what RCCL does over xGMI)                                         (SLAM.java:87-153 at that scale)
  C5  64 maps x 4 096 particles x 1 080 beams, 1024^2 @ 5 cm, one batched handle  (every map against the oracle)
  C3  at bench.py's OWN particle cloud (sigma 0.10 m / 5 deg): the cloud the headline number is measured on

The oracle scores 65 536 x 720 in about half a second, so these are parity tests proper, not property tests.
"""
import ctypes as C

import numpy as np
import pytest

from gridmap_slam_robot_amd import GridMap, ParticleFilter, synth
from oracle import oracle as orc

from _checks import assert_resample_indices

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)) if a.size else 0.0


def _hip_memcpy_dtod(dst: int, src: int, nbytes: int):
    hip = C.CDLL(None)                                        # the HIP runtime the library and torch share
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), 3) == 0      # hipMemcpyDeviceToDevice


def _check_filter_against_oracle(g, lik, scan, P, w_raw, st, wpose, tight=1e-11):
    """raw weights, weight sum, strongest, Neff, weighted pose of one population vs the sequential oracle"""
    want = g.score(lik, scan, P)                                  # GridMap.java:261-294 x N
    ok = want > 1e-290
    assert ok.sum() > len(P) // 10, "the cloud must leave most products representable"
    assert rel_err(w_raw[ok], want[ok]) <= tight                  # bar 1e-5
    assert (w_raw[~ok] <= 1e-289).all()
    wn = want.copy()
    ws, strongest = orc.normalize(wn)                             # SLAM.java:100-124
    assert st["strongest"] == strongest
    assert abs(st["weight_sum"] - ws) <= 1e-11 * ws
    ne = orc.neff(wn)                                             # SLAM.java:180-190
    assert abs(st["neff"] - ne) <= 1e-9 * ne
    assert np.allclose(wpose, orc.weighted_pose(P, wn), rtol=0, atol=2e-6)     # SLAM.java:165-178 (f32 result)
    return wn, ok


def test_c4_65536_particles_as_8_shards_against_the_oracle_and_the_standalone_filter():
    import torch
    dev = torch.device("cuda", 0)
    c = synth.CONFIGS["C4"]
    ext, res, B, N = c["extent"], c["resolution"], c["beams"], c["particles"]
    world, n = 8, N // 8
    assert (N, n, B) == (65536, 8192, 720)
    tr = synth.make_trace(ext, res, B, T=16, seed=1234, n_scans=7)
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    ref_map = GridMap(ext, ext, res, (-ext / 2, -ext / 2))
    assert (ref_map.W, ref_map.H) == (2048, 2048)
    maps = [GridMap(ext, ext, res, (-ext / 2, -ext / 2)) for _ in range(world)]
    log = g.new_log()
    for t in range(4):
        g.integrate(log, tr.scans[t], tr.poses[t])
        for m in [ref_map] + maps:
            m.update(tr.scans[t], tr.poses[t])
    ref = ParticleFilter(ref_map, N)
    pfs = []
    for r, m in enumerate(maps):
        pf = ParticleFilter(m, n)
        pf.set_shard(r * n, N)
        pfs.append(pf)

    for t, (frac, sig_xy, sig_th) in ((4, (0.9, res, 0.3)), (5, (0.9, 2 * res, 0.5)), (6, (-1.0, res, 0.3))):
        Ph = synth.make_particles(tr.poses[t], N, seed=300 + t, sigma_xy=sig_xy, sigma_theta_deg=sig_th)
        P = torch.from_numpy(Ph).to(dev)
        beams = torch.from_numpy(tr.scans[t].view(np.uint8).copy()).to(dev)
        r01 = 0.2718 + 0.1 * t
        lik_before = ref_map.download_likelihood().reshape(-1)
        assert np.array_equal(lik_before, g.build_likelihood(ref_map.download_log().reshape(-1)))

        # ---- stand-alone filter of the whole population: one fused step
        if frac >= 0:
            ref.slam_update_dev(P.data_ptr(), beams.data_ptr(), B, r01, frac, True)
        else:
            ref.set_poses_dev(P.data_ptr()); ref.score_dev(beams.data_ptr(), B); ref.normalize(fetch=False)
            ref_map.update_at_dev(beams.data_ptr(), B, ref)

        # ---- 8 shards: begin -> all-gather of both buffers (device copies) -> end
        for r, pf in enumerate(pfs):
            pf.slam_update_sharded_begin_dev(P[r * n:(r + 1) * n].data_ptr(), beams.data_ptr(), B)
        for m in maps:
            m.synchronize()
        bufs = [pf.gather_buffers() for pf in pfs]
        assert bufs[0][1] == n * 24 and bufs[0][3] == (n // 256) * 9
        # raw weights as they travel: the packed payload of every shard against the oracle
        raw = np.empty(N, dtype=np.float64)
        for r in range(world):
            pk, nb, _, _ = bufs[r]
            host = torch.empty(nb, dtype=torch.uint8)
            hip = C.CDLL(None)
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            assert hip.hipMemcpy(C.c_void_p(host.data_ptr()), C.c_void_p(pk + r * nb), C.c_size_t(nb), 2) == 0
            rec = host.numpy().view(np.dtype([("w", "<f8"), ("x", "<f4"), ("y", "<f4"), ("t", "<f4"), ("pad", "<u4")]))
            raw[r * n:(r + 1) * n] = rec["w"]
            assert np.array_equal(np.stack([rec["x"], rec["y"], rec["t"]], axis=1), Ph[r * n:(r + 1) * n])
        for r in range(world):
            for q in range(world):
                if q != r:
                    pk, nb, pt, nd = bufs[r]
                    _hip_memcpy_dtod(pk + q * nb, bufs[q][0] + q * nb, nb)
                    _hip_memcpy_dtod(pt + q * nd * 8, bufs[q][2] + q * nd * 8, nd * 8)
        for pf in pfs:
            pf.slam_update_sharded_end_dev(beams.data_ptr(), B, r01, frac, True)

        # ---- oracle: weights, statistics, resample indices, map
        st = ref.stats()
        last = ref.last_step()
        wpose = last["weighted_pose"]             # getWeightedPose of the scored population = the pose integrated at
        wn, ok = _check_filter_against_oracle(g, lik_before, tr.scans[t], Ph, raw, st, wpose)
        assert np.array_equal(last["strongest_pose"], Ph[st["strongest"]])
        poses, weights = ref.get_poses(), ref.get_weights()
        if frac >= 0:
            did = last["did_resample"]
            assert did == (orc.neff(wn) < frac * N)               # GridMapApp.java:185-186
            if did:
                # the device normalises w / weightSum with ITS weight sum; the sequential oracle scan runs on the same
                # values it would produce: feed it the device's normalised source weights (same weights in, same slots out)
                src_w = raw / st["weight_sum"]
                want_idx, _ = orc.resample_indices(np.ascontiguousarray(src_w), r01)      # SLAM.java:133-153
                got_src = _source_indices(poses, Ph)
                assert_resample_indices(got_src, want_idx, last["n_ambiguous"])
                assert np.array_equal(weights, src_w[got_src])    # copies keep their weight (SLAM.java:42)
            else:
                assert np.array_equal(poses, Ph)
        else:
            assert np.array_equal(poses, Ph)
            assert rel_err(weights[ok], wn[ok]) <= 1e-11          # particles whose raw product is a normal double
        # map update at the filter's own weighted pose (SLAM.java:93,102-105)
        g.integrate(log, tr.scans[t], wpose)
        got_log = ref_map.download_log().reshape(-1)
        assert np.array_equal(got_log != 0, log != 0)
        nz = log != 0
        assert rel_err(got_log[nz], log[nz]) <= 1e-13

        # ---- every shard == the stand-alone filter, bit for bit
        ref_lik = ref_map.download_likelihood()
        ref_log = ref_map.download_log()
        for r, (pf, m) in enumerate(zip(pfs, maps)):
            assert pf.stats() == st
            assert np.array_equal(pf.get_poses(), poses[r * n:(r + 1) * n])
            assert np.array_equal(pf.get_weights(), weights[r * n:(r + 1) * n])
            assert np.array_equal(m.download_log(), ref_log)
            assert np.array_equal(m.download_likelihood(), ref_lik)
    for pf in pfs + [ref]:
        pf.close()


def _source_indices(poses_after, P_before):
    """source slot of every resampled particle, recovered from the copied pose (poses of the cloud are distinct)"""
    key = {}
    for i, p in enumerate(map(bytes, P_before)):
        key.setdefault(p, i)
    out = np.array([key[bytes(p)] for p in poses_after], dtype=np.int32)
    return out


def test_c5_full_size_64_maps_against_the_oracle():
    import torch
    dev = torch.device("cuda", 0)
    c = synth.CONFIGS["C5"]
    M, B, N, ext, res = c["n_maps"], c["beams"], c["particles"], c["extent"], c["resolution"]
    assert (M, B, N) == (64, 1080, 4096)
    traces = [synth.make_trace(ext, res, B, T=8, seed=600 + i, n_scans=5) for i in range(M)]
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    mb = GridMap(ext, ext, res, (-ext / 2, -ext / 2), n_maps=M, max_beams=B)
    assert (mb.W, mb.H) == (1024, 1024)
    for t in range(3):
        mb.update(np.stack([tr.scans[t] for tr in traces]), np.stack([tr.poses[t] for tr in traces]))
    logs = mb.download_log().reshape(M, -1)
    liks = mb.download_likelihood().reshape(M, -1)
    # every map's log-odds and likelihood field
    for i in range(M):
        log = g.new_log()
        for t in range(3):
            g.integrate(log, traces[i].scans[t], traces[i].poses[t])
        assert np.array_equal(logs[i] != 0, log != 0)
        nz = log != 0
        assert rel_err(logs[i][nz], log[nz]) <= 1e-13
        assert np.array_equal(liks[i], g.build_likelihood(logs[i]))

    # one batched fused scan step on device-resident inputs (what tools/bench and bench.py --config C5 time)
    Ph = np.stack([synth.make_particles(traces[i].poses[3], N, seed=i, sigma_xy=0.05, sigma_theta_deg=0.5) for i in range(M)])
    scans3 = np.stack([tr.scans[3] for tr in traces])
    P = torch.from_numpy(Ph).to(dev)
    beams = torch.from_numpy(scans3.view(np.uint8).reshape(M, -1).copy()).to(dev)
    r01 = np.random.default_rng(3).random(M)
    pf = ParticleFilter(mb, N)
    # weights first, through the separate entry points (raw weights are observable there)
    pf.set_poses_dev(P.data_ptr()); pf.score_dev(beams.data_ptr(), B)
    w_raw = pf.get_weights()
    sts = pf.normalize()
    wposes = pf.weighted_pose()
    wns = []
    for i in range(M):
        wns.append(_check_filter_against_oracle(g, liks[i], scans3[i], Ph[i], w_raw[i], sts[i], wposes[i])[0])
    # the fused batched step from the same state
    pf.slam_update_dev(P.data_ptr(), beams.data_ptr(), B, r01, 0.9, True)
    st2 = pf.stats()
    poses, weights = pf.get_poses(), pf.get_weights()
    last = pf.last_step()
    did, amb = last["did_resample"], last["n_ambiguous"]
    assert np.array_equal(last["weighted_pose"], wposes)
    logs2 = mb.download_log().reshape(M, -1)
    liks2 = mb.download_likelihood().reshape(M, -1)
    for i in range(M):
        assert st2[i] == sts[i]
        ne = orc.neff(wns[i])
        assert bool(did[i]) == (ne < 0.9 * N)
        if did[i]:
            src_w = w_raw[i] / sts[i]["weight_sum"]
            want_idx, _ = orc.resample_indices(np.ascontiguousarray(src_w), float(r01[i]))
            got_src = _source_indices(poses[i], Ph[i])
            assert_resample_indices(got_src, want_idx, amb[i])
            assert np.array_equal(weights[i], src_w[got_src])
        else:
            assert np.array_equal(poses[i], Ph[i])
        log = logs[i].copy()
        g.integrate(log, scans3[i], wposes[i])                    # the device integrated at its own weighted pose
        assert np.array_equal(logs2[i] != 0, log != 0)
        nz = log != 0
        assert rel_err(logs2[i][nz], log[nz]) <= 1e-13
        assert np.array_equal(liks2[i], g.build_likelihood(logs2[i]))
    pf.close()


def test_c3_at_the_bench_cloud_against_the_oracle():
    """bench.py's own operating point: C3 map pre-built from 32 scans, particles = synth.make_particles defaults
    (sigma 0.10 m / 5 deg).  Most raw products underflow there (the reference's plain product, reproduced); the
    log-weights do not, and are compared for ALL 16 384 particles; raw weights on the representable subset."""
    c = synth.CONFIGS["C3"]
    ext, res, B, N = c["extent"], c["resolution"], c["beams"], c["particles"]
    T = 64
    tr = synth.make_trace(ext, res, B, T=T, seed=1234, n_scans=T // 2 + 2)
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    m = GridMap(ext, ext, res, (-ext / 2, -ext / 2), max_beams=2048)
    for t in range(T // 2):
        m.update(tr.scans[t], tr.poses[t])
    lik = m.download_likelihood().reshape(-1)
    assert np.array_equal(lik, g.build_likelihood(m.download_log().reshape(-1)))
    pf = ParticleFilter(m, N)
    for s in range(2):
        t = T // 2 + s
        P = synth.make_particles(tr.poses[t], N, seed=99 + s)             # bench.py's cloud
        pf.set_poses(P)
        pf.score(tr.scans[t])
        lw = pf.get_log_weights()
        want_lw = g.score_log(lik, tr.scans[t], P)
        assert np.isfinite(want_lw).all()
        assert np.max(np.abs(lw - want_lw) / np.abs(want_lw)) <= 1e-12   # every particle
        w = pf.get_weights()
        want = g.score(lik, tr.scans[t], P)
        ok = want > 1e-290
        assert ok.sum() >= 1
        assert rel_err(w[ok], want[ok]) <= 1e-11
        # below the normal range the sequential product may stick at a denormal where the device says 0
        assert (w[want == 0.0] == 0.0).all() and (w[~ok] <= 1e-289).all()
        st = pf.normalize()
        wn = want.copy()
        ws, strongest = orc.normalize(wn)
        assert st["strongest"] == strongest
        assert abs(st["weight_sum"] - ws) <= 1e-11 * ws
        assert abs(st["neff"] - orc.neff(wn)) <= 1e-9 * orc.neff(wn)
        assert st["n_zero"] == int((w == 0.0).sum())
        assert abs(st["max_log_weight"] - want_lw.max()) <= 1e-12 * abs(want_lw.max())
        assert np.allclose(pf.weighted_pose(), orc.weighted_pose(P, wn), rtol=0, atol=2e-6)
        gw = pf.get_weights()
        idx, amb = pf.resample(0.37, want_indices=True)
        want_idx, _ = orc.resample_indices(np.ascontiguousarray(gw), 0.37)
        assert_resample_indices(idx, want_idx, amb)
    pf.close()
