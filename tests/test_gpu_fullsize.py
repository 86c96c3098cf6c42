"""BASELINE.json's full sizes on the GPU (configs C2, C3, and a slice of C5): parity against the oracle where
it finishes in seconds (it does: the C port scores 16 384 x 720 in ~0.1 s), plus the size-independent
properties the domain offers -- order independence of two scans, dirty rebuild == full rebuild, count
checksum of the ray cast, permutation equivariance of the weights, sortedness and copy counts of the
systematic resample -- and the edge cases (empty scan, one particle, all-miss scan, ragged grid)."""
import os

import numpy as np
import pytest

from gridmap_slam_robot_amd import GridMap, Observation, ParticleFilter, _lib, synth
from oracle import oracle as orc

from _checks import assert_resample_indices

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)) if a.size else 0.0


@pytest.mark.parametrize("cfg", ["C2", "C3"])
def test_full_size_scan_step_against_the_oracle(cfg):
    c = synth.CONFIGS[cfg]
    ext, res, B, N = c["extent"], c["resolution"], c["beams"], c["particles"]
    tr = synth.make_trace(ext, res, B, T=16, seed=1234, n_scans=6)
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    m = GridMap(ext, ext, res, (-ext / 2, -ext / 2))
    assert (m.W, m.H) == (g.W, g.H) == ((1024, 1024) if cfg == "C2" else (2048, 2048))
    log = g.new_log()
    visits = 0
    for t in range(4):
        visits += g.integrate(log, tr.scans[t], tr.poses[t])
        m.update(tr.scans[t], tr.poses[t])                       # integrate + dirty-rect rebuild
    got_log = m.download_log().reshape(-1)
    assert np.array_equal(got_log != 0, log != 0)
    nz = log != 0
    assert rel_err(got_log[nz], log[nz]) <= 1e-13                 # bar 1e-5
    lik = g.build_likelihood(got_log)                            # same map from here on
    assert np.array_equal(m.download_likelihood().reshape(-1), lik)          # dirty rebuilds == the reference's full rebuild
    m.compute_likelihood_map()
    assert np.array_equal(m.download_likelihood().reshape(-1), lik)          # idempotent

    P = synth.make_particles(tr.poses[4], N, seed=99, sigma_xy=res, sigma_theta_deg=0.3)
    pf = ParticleFilter(m, N)
    pf.set_poses(P)
    pf.score(tr.scans[4])
    w = pf.get_weights()
    want = g.score(lik, tr.scans[4], P)
    ok = want > 1e-290
    assert ok.sum() > N // 10
    assert rel_err(w[ok], want[ok]) <= 1e-11                      # bar 1e-5
    st = pf.normalize()
    wn = want.copy()
    ws, strongest = orc.normalize(wn)
    assert st["strongest"] == strongest
    assert abs(st["weight_sum"] - ws) <= 1e-11 * ws
    assert abs(st["neff"] - orc.neff(wn)) <= 1e-9 * orc.neff(wn)
    assert np.allclose(pf.weighted_pose(), orc.weighted_pose(P, wn), rtol=0, atol=2e-6)
    got_n = pf.get_weights()
    assert abs(got_n.sum() - 1.0) <= 1e-12
    idx, amb = pf.resample(0.61803, want_indices=True)
    want_idx, _ = orc.resample_indices(np.ascontiguousarray(got_n), 0.61803)     # same weights in, same slots out
    assert_resample_indices(idx, want_idx, amb)
    # systematic resampling: non-decreasing sources, copy count within 1 of N*w
    assert (np.diff(idx) >= 0).all() and idx.min() >= 0 and idx.max() < N
    copies = np.bincount(idx, minlength=N)
    assert np.max(np.abs(copies - N * got_n)) <= 1.0 + 1e-9


def test_two_scans_commute_and_counts_add_up():
    c = synth.CONFIGS["C3"]
    ext, res, B = c["extent"], c["resolution"], c["beams"]
    tr = synth.make_trace(ext, res, B, T=16, seed=77, n_scans=2)
    a = GridMap(ext, ext, res, (-ext / 2, -ext / 2))
    b = GridMap(ext, ext, res, (-ext / 2, -ext / 2))
    a.integrate_observation(tr.scans[0], tr.poses[0]); a.integrate_observation(tr.scans[1], tr.poses[1])
    b.integrate_observation(tr.scans[1], tr.poses[1]); b.integrate_observation(tr.scans[0], tr.poses[0])
    la = a.download_log()
    assert np.array_equal(la, b.download_log())                    # x + y == y + x: the update is order-independent
    # checksum: every visited cell whose class is free or occupied moved the map by exactly one constant
    _, cls0, n0 = a.trace_scan(tr.scans[0], tr.poses[0])
    _, cls1, n1 = a.trace_scan(tr.scans[1], tr.poses[1])
    n_free = sum(int((cls[i, :n[i]] == 0).sum()) for cls, n in ((cls0, n0), (cls1, n1)) for i in range(B))
    n_occ = sum(int((cls[i, :n[i]] == 2).sum()) for cls, n in ((cls0, n0), (cls1, n1)) for i in range(B))
    lf, lo = a.params.l_free, a.params.l_occ
    assert abs(la.sum() - (n_free * lf + n_occ * lo)) <= 1e-9 * abs(n_free * lf)


def test_weights_are_equivariant_under_particle_permutation():
    c = synth.CONFIGS["C2"]
    ext, res, B, N = c["extent"], c["resolution"], c["beams"], 4096
    tr = synth.make_trace(ext, res, B, T=16, seed=5, n_scans=5)
    m = GridMap(ext, ext, res, (-ext / 2, -ext / 2))
    for t in range(4):
        m.update(tr.scans[t], tr.poses[t])
    P = synth.make_particles(tr.poses[4], N, seed=3, sigma_xy=0.05, sigma_theta_deg=1.0)
    P[100] = P[7]                                                 # duplicates score identically
    perm = np.random.default_rng(0).permutation(N)
    pf = ParticleFilter(m, N)
    pf.set_poses(P); pf.score(tr.scans[4]); w = pf.get_weights()
    pf.set_poses(P[perm]); pf.score(tr.scans[4]); wp = pf.get_weights()
    assert np.array_equal(wp, w[perm])
    assert w[100] == w[7]


def test_edge_cases_empty_scan_single_particle_all_miss_ragged_grid():
    m = GridMap(3.3, 2.1, 0.07, (0.0, 1.0))                       # 48 x 30: W % 4 == 0 but ragged tiles
    g = orc.Grid(3.3, 2.1, 0.07, 0.0, 1.0)
    assert (m.W, m.H) == (g.W, g.H)
    m2 = GridMap(3.0, 2.1, 0.07, (0.0, 1.0))                      # 43 x 30: W % 4 != 0 (scalar apply path)
    g2 = orc.Grid(3.0, 2.1, 0.07, 0.0, 1.0)
    rng = np.random.default_rng(4)
    for mm, gg in ((m, g), (m2, g2)):
        B = 50
        ang = rng.uniform(-np.pi, np.pi, B)
        dist = rng.uniform(0.2, 4.0, B)
        hits = rng.uniform(0, 1, B) < 0.7
        obs = Observation.from_polar(ang, np.where(hits, dist, 10.0), hits)
        pose = np.array([1.5, 2.0, 0.4], dtype=np.float32)
        log = gg.new_log()
        for _ in range(3):
            gg.integrate(log, obs.beams, pose)
            mm.update(obs, pose)
        got = mm.download_log().reshape(-1)
        assert np.array_equal(got != 0, log != 0) and np.max(np.abs(got - log)) <= 1e-12
        assert np.array_equal(mm.download_likelihood().reshape(-1), gg.build_likelihood(got))
        # one particle (N = 1), and the same through probabilityOf
        lik = gg.build_likelihood(got)
        assert abs(mm.probability_of(obs, pose) - gg.probability_of(lik, obs.beams, pose)) <= 1e-13 * gg.probability_of(lik, obs.beams, pose)
        # empty scan: nothing touched, every weight is the empty product
        before = mm.download_log()
        empty = Observation(np.zeros(0, dtype=obs.beams.dtype))
        mm.update(empty, pose)
        assert np.array_equal(mm.download_log(), before)
        pf = ParticleFilter(mm, 5)
        pf.score(empty)
        assert np.array_equal(pf.get_weights(), np.ones(5))
        # all-miss scan: free cells only, and probabilityOf skips every beam (weight 1)
        miss = Observation.from_polar(ang, np.full(B, 10.0), np.zeros(B, dtype=bool))
        log2 = got.copy()
        gg.integrate(log2, miss.beams, pose)
        mm.integrate_observation(miss, pose)
        assert np.max(np.abs(mm.download_log().reshape(-1) - log2)) <= 1e-12
        pf.score(miss)
        assert np.array_equal(pf.get_weights(), np.ones(5))
        st = pf.normalize()
        assert st["weight_sum"] == 5.0 and st["strongest"] == 0 and abs(st["neff"] - 5.0) < 1e-12
        # the fused scan step (paired launches, deferred apply) on the same awkward inputs == the separate calls
        twin = GridMap(*((3.3, 2.1) if mm is m else (3.0, 2.1)), 0.07, (0.0, 1.0))
        twin.upload_log(mm.download_log()); twin.compute_likelihood_map(); mm.compute_likelihood_map()
        for n_p in (1, 5, 300):
            fa, fb = ParticleFilter(mm, n_p), ParticleFilter(twin, n_p)
            for k, scan in enumerate((obs, miss, empty, obs)):
                P = synth.make_particles(pose, n_p, seed=k, sigma_xy=0.05, sigma_theta_deg=2.0)
                sa = fa.slam_update(P, scan, 0.6, 0.5, True, fetch=True)
                fb.set_poses(P); fb.score(scan); sb = fb.normalize(); fb.resample_if(0.6, 0.5); twin.update_at(scan, fb)
                assert sa == sb
                assert np.array_equal(fa.get_poses(), fb.get_poses()) and np.array_equal(fa.get_weights(), fb.get_weights())
            assert np.array_equal(mm.download_log(), twin.download_log())
            assert np.array_equal(mm.download_likelihood(), twin.download_likelihood())


def test_config5_slice_batched_against_the_oracle():
    c = synth.CONFIGS["C5"]
    M, B, N = 4, c["beams"], 1024
    ext, res = c["extent"], c["resolution"]
    traces = [synth.make_trace(ext, res, B, T=8, seed=60 + i, n_scans=4) for i in range(M)]
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    mb = GridMap(ext, ext, res, (-ext / 2, -ext / 2), n_maps=M, max_beams=2048)    # > 4096 rays: the 16-ray workgroups
    logs = [g.new_log() for _ in range(M)]
    for t in range(3):
        mb.update(np.stack([tr.scans[t] for tr in traces]), np.stack([tr.poses[t] for tr in traces]))
        for i in range(M):
            g.integrate(logs[i], traces[i].scans[t], traces[i].poses[t])
    lb = mb.download_log().reshape(M, -1)
    kb = mb.download_likelihood().reshape(M, -1)
    P = np.stack([synth.make_particles(traces[i].poses[3], N, seed=i, sigma_xy=0.05, sigma_theta_deg=0.5) for i in range(M)])
    pf = ParticleFilter(mb, N)
    pf.set_poses(P)
    pf.score(np.stack([tr.scans[3] for tr in traces]))
    w = pf.get_weights()
    for i in range(M):
        assert np.array_equal(lb[i] != 0, logs[i] != 0)
        assert rel_err(lb[i][logs[i] != 0], logs[i][logs[i] != 0]) <= 1e-13
        lik = g.build_likelihood(lb[i])
        assert np.array_equal(kb[i], lik)
        want = g.score(lik, traces[i].scans[3], P[i])
        ok = want > 1e-290
        assert rel_err(w[i][ok], want[ok]) <= 1e-11


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("GMS_FUZZ_SEEDS", "16")))))
def test_fused_step_random_shapes_equal_the_separate_calls(seed):
    """Random map sizes / resolutions (both blur half-widths and the generic one), beam and particle counts, resample
    thresholds: a few fused scan steps back to back must leave exactly what the separate entry points leave."""
    rng = np.random.default_rng(1000 + seed)
    res = float(rng.choice([0.05, 0.02, 0.1, 0.035]))
    W = int(rng.integers(40, 400)); H = int(rng.integers(40, 400))
    ext_x, ext_y = W * res * 0.999, H * res * 0.999
    pos = (-ext_x / 2 + float(rng.uniform(-0.3, 0.3)), -ext_y / 2 + float(rng.uniform(-0.3, 0.3)))
    B = int(rng.integers(1, 400)); N = int(rng.integers(1, 3000))
    a, b = GridMap(ext_x, ext_y, res, pos), GridMap(ext_x, ext_y, res, pos)
    pa, pb = ParticleFilter(a, N), ParticleFilter(b, N)
    for step in range(5):
        ang = rng.uniform(-np.pi, np.pi, B)
        dist = rng.uniform(0.1, 0.6 * max(ext_x, ext_y), B)
        hits = rng.uniform(0, 1, B) < 0.8
        obs = Observation.from_polar(ang, np.where(hits, dist, 10.0), hits)
        centre = np.array([pos[0] + ext_x / 2, pos[1] + ext_y / 2, 0.0], dtype=np.float32)
        P = synth.make_particles(centre + rng.normal(0, 0.05, 3).astype(np.float32), N, seed=step, sigma_xy=0.05, sigma_theta_deg=3.0)
        r01, frac = float(rng.random()), float(rng.choice([0.5, 0.9, 2.0, -1.0]))
        integrate = bool(rng.uniform() < 0.85)
        sa = pa.slam_update(P, obs, r01, frac, integrate, fetch=True)
        pb.set_poses(P); pb.score(obs); sb = pb.normalize()
        if frac >= 0:
            pb.resample_if(r01, frac)
        if integrate:
            b.update_at(obs, pb)
        assert all(sa[k] == sb[k] or (sa[k] != sa[k] and sb[k] != sb[k]) for k in sa)      # (NaN Neff when every weight is 0)
        if step in (2, 4):
            assert np.array_equal(pa.get_poses(), pb.get_poses()) and np.array_equal(pa.get_weights(), pb.get_weights(), equal_nan=True)
            assert np.array_equal(a.download_log(), b.download_log())
            assert np.array_equal(a.download_likelihood(), b.download_likelihood())


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("GMS_FUZZ_SEEDS", "8")))))
def test_random_shapes_against_the_oracle(seed):
    """The same random shapes against the CPU oracle: ray-cast map (cells touched bit-exact, log-odds <= 1e-12 rel.),
    likelihood field (==), weights (<= 1e-10 rel. where representable), strongest, resample indices."""
    rng = np.random.default_rng(5000 + seed)
    res = float(rng.choice([0.05, 0.02, 0.1, 0.035]))
    W = int(rng.integers(40, 300)); H = int(rng.integers(40, 300))
    ext_x, ext_y = W * res * 0.999, H * res * 0.999
    pos = (-ext_x / 2 + float(rng.uniform(-0.3, 0.3)), -ext_y / 2 + float(rng.uniform(-0.3, 0.3)))
    B = int(rng.integers(1, 300)); N = int(rng.integers(1, 1500))
    m = GridMap(ext_x, ext_y, res, pos)
    g = orc.Grid(ext_x, ext_y, res, pos[0], pos[1])
    assert (m.W, m.H) == (g.W, g.H)
    log = g.new_log()
    centre = np.array([pos[0] + ext_x / 2, pos[1] + ext_y / 2, 0.0], dtype=np.float32)
    for step in range(3):
        ang = rng.uniform(-np.pi, np.pi, B)
        dist = rng.uniform(0.1, 0.6 * max(ext_x, ext_y), B)
        hits = rng.uniform(0, 1, B) < 0.8
        obs = Observation.from_polar(ang, np.where(hits, dist, 10.0), hits)
        pose = (centre + np.array([rng.normal(0, 0.2), rng.normal(0, 0.2), rng.uniform(-3, 3)])).astype(np.float32)
        g.integrate(log, obs.beams, pose)
        m.update(obs, pose)
    got = m.download_log().reshape(-1)
    assert np.array_equal(got != 0, log != 0)
    nz = log != 0
    assert not nz.any() or np.max(np.abs(got[nz] - log[nz]) / np.abs(log[nz])) <= 1e-12
    lik = g.build_likelihood(got)
    assert np.array_equal(m.download_likelihood().reshape(-1), lik)
    P = synth.make_particles(centre, N, seed=seed, sigma_xy=0.08, sigma_theta_deg=4.0)
    pf = ParticleFilter(m, N)
    pf.set_poses(P); pf.score(obs)
    st = pf.normalize()
    w = g.score(lik, obs.beams, P)
    ws, strongest = orc.normalize(w)
    if ws > 1e-280:
        gw = pf.get_weights()
        ok = w > 1e-290
        assert np.max(np.abs(gw[ok] - w[ok]) / w[ok]) <= 1e-10
        assert st["strongest"] == strongest or w[st["strongest"]] == w[strongest]
        r01 = float(rng.random())
        idx, amb = pf.resample(r01, want_indices=True)
        want, _ = orc.resample_indices(gw, r01)
        assert_resample_indices(idx, want, amb)


def dense_log(W, H, seed=0):
    """Worst case for the likelihood kernel: every 64 x 32 tile holds all three codes (an 8 x 8 checkerboard of
    occupied / free cells with unexplored specks), so no tile takes the uniform short cut."""
    y, x = np.mgrid[0:H, 0:W]
    log = np.where(((x >> 3) + (y >> 3)) & 1, 2.197224312426715, -0.8472978036208759)
    rng = np.random.default_rng(seed)
    log[rng.random((H, W)) < 0.02] = 0.0
    return log.reshape(-1)


def test_likelihood_full_rebuild_on_a_dense_map_equals_the_oracle():
    """2048^2, all 4096 tiles non-uniform: the kernel's worst case (the synthetic room leaves ~90 % of the tiles uniform).
    Compared with == against the oracle; then a dirty rebuild after one more scan == a full rebuild."""
    c = synth.CONFIGS["C3"]
    ext, res, B = c["extent"], c["resolution"], c["beams"]
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    m = GridMap(ext, ext, res, (-ext / 2, -ext / 2))
    log = dense_log(m.W, m.H)
    m.upload_log(log)
    m.compute_likelihood_map()
    want = g.build_likelihood(log)
    assert np.array_equal(m.download_likelihood().reshape(-1), want)
    m.compute_likelihood_map()                                     # idempotent (tile states stay "not uniform")
    assert np.array_equal(m.download_likelihood().reshape(-1), want)
    tr = synth.make_trace(ext, res, B, T=16, seed=3, n_scans=1)
    m.update(tr.scans[0], tr.poses[0])                             # dirty rebuild on top of the dense field
    log2 = log.copy()
    g.integrate(log2, tr.scans[0], tr.poses[0])
    got = m.download_log().reshape(-1)
    # (counts times constant vs. up to ~700 sequential additions near the robot: relative bar, 1e-5 in north_star)
    assert np.max(np.abs(got - log2) / np.maximum(np.abs(log2), 1.0)) <= 1e-12
    assert np.array_equal(np.sign(got), np.sign(log2))
    assert np.array_equal(m.download_likelihood().reshape(-1), g.build_likelihood(got))


@pytest.mark.parametrize("order_mode", [0, 1])
def test_a_million_particles(monkeypatch, order_mode):
    """2^20 particles on one handle (1024 scoring groups, 4096 reduction blocks, three scan levels): weights of a sample against
    the oracle, the normalised population sums to 1, the strongest particle is the oracle's among the sample's candidates,
    systematic resampling returns non-decreasing sources with copy counts within 1 of N w; with and without the locality order."""
    N, B = 1 << 20, 96                                             # GMS_MAX_PARTICLES
    ext, res = 12.8, 0.05
    tr = synth.make_trace(ext, res, B, T=10, seed=8, n_scans=6)
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    m = GridMap(ext, ext, res, (-ext / 2, -ext / 2))
    for t in range(4):
        m.update(tr.scans[t], tr.poses[t])
    lik = g.build_likelihood(m.download_log().reshape(-1))
    P = synth.make_particles(tr.poses[4], N, seed=3, sigma_xy=0.05, sigma_theta_deg=2.0)
    monkeypatch.setenv("GMS_SCORE_ORDER", str(order_mode))
    pf = ParticleFilter(m, N)
    pf.set_poses(P)
    pf.score(tr.scans[4])
    w = pf.get_weights()
    sample = np.random.default_rng(0).choice(N, 4000, replace=False)
    sample = np.concatenate([sample, [0, 1023, 1024, N - 1025, N - 1]])
    want = g.score(lik, tr.scans[4], P[sample])
    ok = want > 1e-290
    assert ok.sum() > 1000 and rel_err(w[sample][ok], want[ok]) <= 1e-11
    st = pf.normalize()
    assert st["weight_sum"] > 0 and abs(st["weight_sum"] - w.sum()) <= 1e-9 * w.sum()
    assert w[st["strongest"]] == w.max() and st["strongest"] == int(np.argmax(w))          # first maximum (SLAM.java:110-115)
    wn = pf.get_weights()
    assert abs(wn.sum() - 1.0) <= 1e-9
    assert abs(st["neff"] - 1.0 / np.sum(wn * wn)) <= 1e-6 * st["neff"]
    idx, amb = pf.resample(0.4142, want_indices=True)
    assert idx.shape == (N,) and (np.diff(idx) >= 0).all() and idx.min() >= 0 and idx.max() < N
    copies = np.bincount(idx, minlength=N)
    assert np.max(np.abs(copies - N * wn)) <= 1.0 + 1e-6
    assert np.array_equal(pf.get_poses(), P[idx])
    pf.close()
    with pytest.raises(_lib.GmsError) as e:                       # one more does not fit the resampling kernels' LDS: refused at creation
        ParticleFilter(m, N + 1)
    assert e.value.code == _lib.GMS_ERR_INVALID
    m.close()


def test_a_large_map_far_from_the_origin():
    """8192 x 8192 cells (16 384 x 16 384 was run once by hand), the robot near the far corner, so that every cell index is
    large: the ray-cast map, the likelihood field and the weights against the oracle."""
    W, res = 8192, 0.05
    ext = W * res
    tr = synth.make_trace(40.0, res, 360, T=8, seed=3, n_scans=6)
    m = GridMap(ext, ext, res, (-ext / 2, -ext / 2))
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    assert (m.W, m.H) == (W, W)
    log = g.new_log()
    off = np.array([ext / 2 - 25.0, -(ext / 2 - 25.0), 0.0], dtype=np.float32)
    for t in range(4):
        pose = (tr.poses[t] + off).astype(np.float32)
        m.update(tr.scans[t], pose)
        g.integrate(log, tr.scans[t], pose)
    got = m.download_log().reshape(-1)
    assert np.array_equal(got != 0, log != 0)
    nz = log != 0
    assert rel_err(got[nz], log[nz]) <= 1e-13
    lik = g.build_likelihood(got)
    assert np.array_equal(m.download_likelihood().reshape(-1), lik)
    N = 2048
    P = synth.make_particles((tr.poses[4] + off).astype(np.float32), N, seed=1, sigma_xy=0.05, sigma_theta_deg=2.0)
    pf = ParticleFilter(m, N)
    pf.set_poses(P); pf.score(tr.scans[4])
    want = g.score(lik, tr.scans[4], P)
    ok = want > 1e-290
    assert ok.sum() > N // 4 and rel_err(pf.get_weights()[ok], want[ok]) <= 1e-11
    pf.close(); m.close()
