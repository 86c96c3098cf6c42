"""The reference's own filter shape on the device -- SLAM (J/slam/SLAM.java) with one GridMapData per particle: update() scores a
particle against ITS OWN likelihood field and integrates the scan into ITS OWN map at ITS OWN pose (:88-107), resample() deep-copies
the surviving particles' maps (:41-45 -> GridMap.java:106-124) -- against the oracle's literal restatement of the loop
(orc_slam_update / orc_slam_resample, tests/test_oracle_slam.py), frame by frame over a recording: every particle's pose, weight and
both arrays of its map.  The motion-model draw is not the reference's (unseeded Well1024a): both sides take Philox variates; the
device's double-precision log / sin / cos may round the last ulp of a float pose differently from glibc, so after that is checked the
oracle continues from the device's poses (as tests/test_gpu_trace_replay.py does)."""
import os

import numpy as np
import pytest

from gridmap_slam_robot_amd import SLAMParticleMaps, synth
from gridmap_slam_robot_amd.trace import read_trace
from oracle import oracle as orc

from _checks import assert_resample_indices

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
THREADS = min(16, os.cpu_count() or 1)


def _frames_to_scans(frames):
    return [(orc.deskew(f.angle, f.distance, f.hit, f.d_center, f.d_theta), (f.d_center, f.d_theta)) for f in frames]


def _compare_weights(w, wo, where):
    """normalised weights: the raw products are the reference's bits (one lane multiplies in beam order), the weight sum is a blocked
    sum on the device: zeros are zeros, everything else agrees to 1e-13"""
    assert np.isfinite(wo).all(), f"{where}: the oracle's weights must be finite for this check"
    assert np.array_equal(w == 0, wo == 0), f"{where}: {int((w == 0).sum())} zero weights on the device, {int((wo == 0).sum())} in the oracle"
    big = wo > 1e-290
    assert big.any() and np.max(np.abs(w[big] - wo[big]) / wo[big]) <= 1e-13, f"{where}: weights off by {np.max(np.abs(w[big] - wo[big]) / wo[big]):.3e}"
    assert (np.abs(w[~big] - wo[~big]) <= 1e-300).all()


def _compare_maps(dev, o, where):
    """logData: the same cells changed, values to 1e-13 relative (the device adds a scan's increments of one cell as one sum:
    DESIGN.md section 5); likelihoodData: equal."""
    n = o.n
    logs, liks = dev.maps().reshape(n, -1), dev.maps(likelihood=True).reshape(n, -1)
    for i in range(n):
        lo, ko = o.log(i), o.lik(i)
        assert np.array_equal(logs[i] != 0, lo != 0), f"{where}: particle {i}: another set of cells touched"
        err = np.abs(logs[i] - lo)
        assert (err <= 1e-13 * np.maximum(np.abs(lo), 1.0)).all(), f"{where}: particle {i}: log-odds off by {err.max():.3e}"
        assert np.array_equal(liks[i], ko), f"{where}: particle {i}: likelihood field differs in {int((liks[i] != ko).sum())} cells"


def _run(dev, o, scans, start, seed, rng, check_maps_at, resample_rule=True, label=""):
    """frames through both; returns the number of resampling steps"""
    n = o.n
    P0 = np.tile(np.asarray(start, np.float32), (n, 1))
    dev.set_poses(P0)
    o.set_poses(P0)
    resampled = 0
    for k, (z, u) in enumerate(scans):
        prev = o.poses
        neff = dev.update(z, u, seed=seed, sequence=k)                                   # SLAM.update(z, u): SLAM.java:80-131
        P = dev.get_particles()[0]
        Po = orc.sample_motion(prev, u[0], u[1], seed=seed, sequence=k)                  # :90 -> Odometry.java:77-96
        assert (np.all(P == Po, axis=1)).mean() > 0.99 and np.max(np.abs(P - Po)) <= 2e-6, f"{label} frame {k}: motion samples"
        o.set_poses(P)
        neff_o = o.update(z, u, sample_motion=False, threads=THREADS)
        st = dev.last_stats
        w, wo = dev.get_particles()[1], o.weights
        _compare_weights(w, wo, f"{label} frame {k}")
        assert st["strongest"] == o.strongest and st["n_zero"] == int((wo == 0).sum())
        assert abs(neff - neff_o) <= 1e-11 * neff_o
        assert np.allclose(dev.get_weighted_pose(), o.weighted_pose(), rtol=0, atol=2e-6)
        if k in check_maps_at:
            _compare_maps(dev, o, f"{label} frame {k}")
        if resample_rule and neff_o < n // 2:                                            # GridMapApp.java:185-186
            r01 = float(rng.random())
            idx, amb = dev.resample(r01, want_indices=True)
            want, clamped = o.resample(r01)
            assert clamped == 0
            assert_resample_indices(idx, want, amb)
            assert np.array_equal(idx, want), f"{label} frame {k}: the draw {r01} sits on a rounding boundary; pick another seed"
            resampled += 1
            assert np.array_equal(dev.get_particles()[0], o.poses)
            _compare_weights(dev.get_particles()[1], o.weights, f"{label} frame {k} after resampling")
            if k in check_maps_at or resampled == 1:
                _compare_maps(dev, o, f"{label} frame {k} after the resampling copy")
    return resampled


def test_the_reference_operating_point_500_particles_of_120x120_cells():
    """SLAM.java:50,57: 500 particles, GridMap(6 m, 6 m, 0.05 m, (-3, -3)); 90 beams per revolution; thirty revolutions of a drive
    through a 4.8 m room, the caller's `if (neff < N / 2) resample()` (GridMapApp.java:185-186) included."""
    N, B, T = 500, 90, 30
    frames, truth = synth.make_recording(6.0, B, T=48, seed=77, n_frames=T)
    scans = _frames_to_scans(frames)
    start = synth.true_pose(synth.make_world(6.0, 77), -1, 48)
    dev = SLAMParticleMaps(6.0, 6.0, 0.05, (-3.0, -3.0), num_particles=N, max_beams=128)
    assert (dev.W, dev.H) == (120, 120)
    g = orc.Grid(6.0, 6.0, 0.05, -3.0, -3.0)
    o = orc.Slam(g, N)
    # reset(): uniform weights, poses 0, blank maps (SLAM.java:65-77)
    P, w = dev.get_particles()
    assert np.array_equal(P, np.zeros((N, 3), np.float32)) and np.array_equal(w, np.full(N, 1.0 / N))
    assert not dev.maps().any() and not dev.maps(likelihood=True).any()
    resampled = _run(dev, o, scans, start, seed=2024, rng=np.random.default_rng(11), check_maps_at={0, 1, 7, 15, 22, T - 1}, label="500x120^2")
    assert resampled >= 1, "the drive must trigger at least one resampling step"
    assert dev.maps_copied() == resampled * N
    # getWeightedPose follows the drive (the filter works, not only matches)
    wp = dev.get_weighted_pose()
    assert np.hypot(wp[0] - truth[T - 1][0], wp[1] - truth[T - 1][1]) < 1.0          # (the reference filter without its pose refinement drifts: a sanity bound, not a quality claim)
    # calculateCombined over the particles' maps (GridMapApp.java:439-458)
    comb = dev.calculate_combined().reshape(-1)
    want = orc.combine_maps(o.logs())
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(comb), fin) and np.max(np.abs(comb[fin] - want[fin])) <= 1e-9 * max(1.0, np.abs(want[fin]).max())


def test_4096_particles_of_256x256_cells_on_the_committed_recording():
    """Thirty revolutions of tests/golden/recording_360.bin (the reference's DataRecorder format; synthetic: the reference ships no
    recording), every other measurement of its 360 per revolution -- the reference's scans hold 90 to 180 (one per 2-4 degrees); at
    360 the plain product over a blank map, 0.1^360, is 0 for every particle and update() divides 0 by 0 -- through
    4096 particles with a 12.8 m map each (8.6 GB of GridMapData on the device, both generations), with the caller's resampling rule.
    Poses, weights, Neff, strongest and weighted pose every frame; all 4096 maps (both arrays) at five of them and after the first
    resampling copy."""
    N, T = 4096, 30
    frames = read_trace(os.path.join(HERE, "golden", "recording_360.bin"))[:T]
    for f in frames:
        f.angle, f.distance, f.hit = f.angle[::2].copy(), f.distance[::2].copy(), f.hit[::2].copy()
    scans = _frames_to_scans(frames)
    start = synth.true_pose(synth.make_world(25.6, 4321), -1, 64)
    ext = 12.8
    dev = SLAMParticleMaps(ext, ext, 0.05, (-ext / 2, -ext / 2), num_particles=N, max_beams=512)
    assert (dev.W, dev.H) == (256, 256)
    g = orc.Grid(ext, ext, 0.05, -ext / 2, -ext / 2)
    o = orc.Slam(g, N)
    resampled = _run(dev, o, scans, start, seed=7, rng=np.random.default_rng(3), check_maps_at={0, 9, 19, T - 1}, label="4096x256^2")
    assert resampled >= 1 and dev.maps_copied() == resampled * N
    # resample() whatever Neff says (the GUI's button: GridMapApp.java:306)
    r01 = 0.4242
    idx, amb = dev.resample(r01, want_indices=True)
    want, _ = o.resample(r01)
    assert_resample_indices(idx, want, amb)
    assert np.array_equal(idx, want)
    _compare_maps(dev, o, "4096x256^2 after the last resampling copy")


def test_edge_scans_the_band_walk_and_the_skip_rule(monkeypatch):
    """no beam at all; a scan of misses only; particles outside the map (a ray whose start cell is outside emits nothing,
    RayIterator.java:108); |dTheta| > 30 degrees skips integrateObservation but not computeLikelihoodMap (SLAM.java:82,102); and the
    count tile forced down to a few rows, so that the scan's box is walked in bands."""
    ext, res, B, N = 6.4, 0.05, 72, 24
    tr = synth.make_trace(ext, res, B, T=8, seed=21)
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    for tile_cells in (0, 1000):
        if tile_cells:
            monkeypatch.setenv("GMS_SLAM_TILE_CELLS", str(tile_cells))
        dev = SLAMParticleMaps(ext, ext, res, (-ext / 2, -ext / 2), num_particles=N, max_beams=128)
        o = orc.Slam(g, N)
        P = synth.make_particles(tr.poses[0], N, seed=8, sigma_xy=0.05, sigma_theta_deg=3.0)
        P[3] = [40.0, 1.0, 0.3]                   # far outside the map
        P[4] = [-3.3, 0.0, 0.0]                   # just outside: rays would enter the map, but the walk never starts
        P[5] = [3.19, 3.19, 0.7]                  # in the corner cell
        dev.set_poses(P); o.set_poses(P)
        z = tr.scans[0]
        dev.update(z, None); o.update(z, None)
        _compare_maps(dev, o, f"tile {tile_cells}: first scan")
        _compare_weights(dev.get_particles()[1], o.weights, f"tile {tile_cells}: first scan")
        # misses only
        zm = tr.scans[1].copy()
        zm["hit"] = 0
        zm["distance"] = 10.0
        dev.update(zm, None); o.update(zm, None)
        assert np.array_equal(dev.get_particles()[1], np.full(N, 1.0 / N))     # no beam hit: every product is 1
        _compare_maps(dev, o, f"tile {tile_cells}: misses only")
        # no beam at all
        dev.update(z[:0], None); o.update(z[:0], None)
        _compare_maps(dev, o, f"tile {tile_cells}: empty scan")
        # the skip rule
        before = dev.maps()
        dev.update(tr.scans[2], (0.0, np.radians(31.0)), seed=5, sequence=9)
        o.set_poses(dev.get_particles()[0]); o.update(tr.scans[2], (0.0, np.radians(31.0)), sample_motion=False)
        assert np.array_equal(dev.maps(), before)
        _compare_maps(dev, o, f"tile {tile_cells}: skipped update")
        dev.close()


@pytest.mark.parametrize("case", ["2cm_11_taps", "ragged_map", "one_particle", "three_ray_groups", "signed_taps", "tiny_tap_9", "fifteen_taps", "one_tap", "zero_end_taps"])
def test_other_geometries(case):
    """what the two sizes above do not reach: the 11-tap kernel of a 2 cm map (k_slam_likelihood<5>), a map that is no multiple of the
    64 x 32 likelihood tiles and narrower than one, a filter of one particle, a scan of more beams than one group of producer lanes
    holds (128): several groups of rays per band.  Blur kernels given by the caller: one with negative taps and one with a tap of
    1e-300 (the on-demand field's literal form: tap * value sums in which nothing may be rescaled), and fifteen plain taps (the widest
    kernel the class planes are kept for; the scaled form, sixteen lanes per end point), a kernel of one tap (no blur) and one whose
    outer taps are 0.0."""
    W, H, res, B, N, T = {"2cm_11_taps": (3.2, 3.2, 0.02, 72, 12, 5), "ragged_map": (2.6, 4.45, 0.05, 64, 10, 5),
                          "one_particle": (4.0, 4.0, 0.05, 90, 1, 5), "three_ray_groups": (4.0, 4.0, 0.05, 300, 6, 4),
                          "signed_taps": (4.0, 4.0, 0.05, 90, 8, 5), "tiny_tap_9": (4.0, 4.0, 0.05, 90, 8, 5), "fifteen_taps": (4.0, 3.0, 0.05, 90, 8, 5),
                          "one_tap": (4.0, 4.0, 0.05, 90, 8, 4), "zero_end_taps": (4.0, 4.0, 0.05, 90, 8, 4)}[case]
    taps = {"signed_taps": [-0.03, 0.11, 0.26, 0.32, 0.26, 0.11, -0.03], "tiny_tap_9": [1e-300, 0.02, 0.1, 0.23, 0.3, 0.23, 0.1, 0.02, 1e-300],
            "one_tap": [1.0], "zero_end_taps": [0.0, 0.25, 0.5, 0.25, 0.0],
            "fifteen_taps": [0.002, 0.006, 0.016, 0.035, 0.065, 0.1, 0.13, 0.292, 0.13, 0.1, 0.065, 0.035, 0.016, 0.006, 0.002]}.get(case)
    ext = min(W, H)
    tr = synth.make_trace(ext, res, B, T=T, seed=31)
    g = orc.Grid(W, H, res, -W / 2, -H / 2)
    if taps is not None:
        g.set_kernel(taps)
    dev = SLAMParticleMaps(W, H, res, (-W / 2, -H / 2), num_particles=N, max_beams=max(128, B), kernel=taps)
    o = orc.Slam(g, N)
    assert (dev.W, dev.H) == (g.W, g.H)
    P = synth.make_particles(tr.poses[0], N, seed=4, sigma_xy=0.03, sigma_theta_deg=2.0)
    dev.set_poses(P); o.set_poses(P)
    rng = np.random.default_rng(2)
    for k in range(T):
        z = tr.scans[k]
        dev.update(z, None); o.update(z, None, threads=THREADS)
        if case == "signed_taps":                          # (factors, and with them weights, of either sign)
            w, wo = dev.get_particles()[1], o.weights
            assert np.isfinite(wo).all() and np.max(np.abs(w - wo) / np.abs(wo)) <= 1e-12, f"{case} frame {k}"
        else:
            _compare_weights(dev.get_particles()[1], o.weights, f"{case} frame {k}")
        _compare_maps(dev, o, f"{case} frame {k}")
        if N > 1 and k == T - 2:
            r01 = float(rng.random())
            idx, amb = dev.resample(r01, want_indices=True)
            want, _ = o.resample(r01)
            assert_resample_indices(idx, want, amb)
            assert np.array_equal(idx, want)
    dev.close()


def test_sixteen_bit_count_tiles_and_their_fallback(monkeypatch):
    """A map whose scan box does not fit a workgroup's LDS as 32-bit count cells is counted in 16-bit cells (n_free | n_occ << 8) when
    no count of the scan can pass 255: every ray visits a cell once, except a ray of zero length, which emits its one cell
    1 + additionalSteps times (RayIterator.java:75) -- beams + 2 x (zero-length beams) <= 255.  Both sides of that rule against the
    oracle, with the tile forced small (so that the 16-bit form is on offer and the box is walked in bands): 128 beams of which 10
    have no length (148: narrow; the robot's own cell collects 148 visits), and of which 70 have none (268: the 32-bit form)."""
    ext, res, B, N = 6.4, 0.05, 128, 16
    monkeypatch.setenv("GMS_SLAM_TILE_CELLS", "3000")
    tr = synth.make_trace(ext, res, B, T=3, seed=41)
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    for n_zero in (10, 70):
        dev = SLAMParticleMaps(ext, ext, res, (-ext / 2, -ext / 2), num_particles=N, max_beams=128)
        o = orc.Slam(g, N)
        P = synth.make_particles(tr.poses[0], N, seed=6, sigma_xy=0.04, sigma_theta_deg=2.0)
        dev.set_poses(P); o.set_poses(P)
        for k in range(3):
            z = tr.scans[k].copy()
            z["local_x"][:n_zero] = 0.0; z["local_y"][:n_zero] = 0.0; z["distance"][:n_zero] = 0.0; z["hit"][:n_zero] = 1
            dev.update(z, None); o.update(z, None, threads=THREADS)
            _compare_maps(dev, o, f"{n_zero} zero-length beams, scan {k}")
            _compare_weights(dev.get_particles()[1], o.weights, f"{n_zero} zero-length beams, scan {k}")
        # the robot's own cell: every ray's first step and all three steps of the zero-length ones, as free visits
        dev.close()


@pytest.mark.parametrize("lazy", ["1", "0"])
def test_resample_copies_likelihood_data_late_or_at_once(lazy, monkeypatch):
    """resample() copies logData at once and likelihoodData when it is asked for (the next update's computeLikelihoodMap overwrites
    every cell of it first; GMS_SLAM_LAZY_LIK_COPY=0: both at once).  What a caller can see is the reference's deep copy either way:
    after one resample(), after two in a row (the second must move the first one's fields), after an upload into one slot while the
    copies are still owed, and across an update that makes them moot."""
    monkeypatch.setenv("GMS_SLAM_LAZY_LIK_COPY", lazy)
    ext, res, B, N = 4.0, 0.05, 60, 32
    tr = synth.make_trace(ext, res, B, T=6, seed=13)
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    dev = SLAMParticleMaps(ext, ext, res, (-ext / 2, -ext / 2), num_particles=N, max_beams=128)
    o = orc.Slam(g, N)
    P = synth.make_particles(tr.poses[0], N, seed=9, sigma_xy=0.05, sigma_theta_deg=3.0)
    dev.set_poses(P); o.set_poses(P)
    for k in range(2):
        dev.update(tr.scans[k], None); o.update(tr.scans[k], None, threads=THREADS)
    for r01 in (0.31, 0.77):                                                   # two in a row
        idx, _ = dev.resample(r01, want_indices=True)
        want, _ = o.resample(r01)
        assert np.array_equal(idx, want)
    _compare_maps(dev, o, "two resamples in a row")
    dev.update(tr.scans[2], None); o.update(tr.scans[2], None, threads=THREADS)
    idx, _ = dev.resample(0.5, want_indices=True)
    want, _ = o.resample(0.5)
    assert np.array_equal(idx, want)
    field = np.full((dev.H, dev.W), 0.25)
    dev.set_map(3, lik=field)                                                  # an upload while the copies are owed
    liks = dev.maps(likelihood=True)
    for i in range(N):
        assert np.array_equal(liks[i].reshape(-1), field.reshape(-1) if i == 3 else o.lik(i)), f"slot {i}"
    dev.update(tr.scans[3], None); o.update(tr.scans[3], None, threads=THREADS)      # ... and an update makes every field current
    _compare_maps(dev, o, "update after the resample")
    dev.close()


def test_the_resampling_rule_decided_on_the_device():
    """update (nothing read back) + resample_if(r, 0.5) per revolution -- GridMapApp.java:185-186 without the host round trip -- against
    the oracle's `if (update(z, u) < n / 2) resample()`: the same steps resample (the indices say which), the same maps at the end.  A
    second filter whose weights never collapse (every scan empty: Neff = n) must never resample: identity indices, maps as they were."""
    ext, res, B, N, T = 4.0, 0.05, 60, 64, 14
    tr = synth.make_trace(ext, res, B, T=T, seed=23)
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    dev = SLAMParticleMaps(ext, ext, res, (-ext / 2, -ext / 2), num_particles=N, max_beams=128)
    o = orc.Slam(g, N)
    P = synth.make_particles(tr.poses[0], N, seed=2, sigma_xy=0.05, sigma_theta_deg=3.0)
    dev.set_poses(P); o.set_poses(P)
    rng = np.random.default_rng(3)
    did = 0
    for k in range(T):
        z = tr.scans[k]
        dev.update(z, None, fetch=False)
        neff_o = o.update(z, None, threads=THREADS)
        r01 = float(rng.random())
        dev.resample_if(r01, 0.5)
        idx = dev.pf.last_resample_indices().reshape(-1)
        if neff_o < N / 2:
            want, _ = o.resample(r01)
            assert np.array_equal(idx, want), f"scan {k}"
            did += 1
        else:
            assert np.array_equal(idx, np.arange(N)), f"scan {k}: no resampling step was due"
    assert 0 < did
    _compare_maps(dev, o, "after the last revolution")
    assert np.array_equal(dev.get_particles()[0], o.poses)
    assert dev.maps_copied() == did * N                   # the revolutions whose rule said no moved nothing (the draws are counted on the device)
    # never due
    dev.reset(); dev.set_poses(P)
    dev.update(tr.scans[0], None)
    logs = dev.maps().copy()
    liks = dev.maps(likelihood=True).copy()
    copied = dev.maps_copied()
    for k in range(3):
        dev.update(tr.scans[0][:0], None, fetch=False)
        dev.resample_if(0.3, 0.5)
        assert np.array_equal(dev.pf.last_resample_indices().reshape(-1), np.arange(N))
        assert dev.maps_copied() == copied, "no resampling step was due: no map may be copied"
    assert np.array_equal(dev.maps(), logs)
    # ... and the filter goes on from the generation it is in: one more real revolution against the oracle
    o2 = orc.Slam(g, N)
    o2.set_poses(dev.get_particles()[0])
    for i in range(N):
        o2.set_log(i, logs[i].reshape(-1)); o2.set_lik(i, liks[i].reshape(-1))
    dev.update(tr.scans[1], None); o2.update(tr.scans[1], None, threads=THREADS)
    _compare_maps(dev, o2, "after revolutions without a resampling step")
    # the filter of a gms_slam refuses what would move its particles without their maps
    from gridmap_slam_robot_amd._lib import GMS_ERR_STATE, GmsError
    with pytest.raises(GmsError) as e:
        dev.pf.resample(0.5)
    assert e.value.code == GMS_ERR_STATE
    dev.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_call_sequences_against_the_oracle(seed):
    """forty calls drawn at random -- update (with and without the motion sample, sometimes with a turn that skips the integration),
    resample, download of one or of all maps, upload of a log or of a field into a slot, reset, the combined map -- on the device
    and on the oracle: the handle's state machine (which generation is current, which copies are owed) against the plain loop."""
    rng = np.random.default_rng(seed)
    ext, res, B, N = 3.2, 0.05, 48, 12
    tr = synth.make_trace(ext, res, B, T=8, seed=50 + seed)
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    dev = SLAMParticleMaps(ext, ext, res, (-ext / 2, -ext / 2), num_particles=N, max_beams=64)
    o = orc.Slam(g, N)
    P = synth.make_particles(tr.poses[0], N, seed=seed, sigma_xy=0.04, sigma_theta_deg=2.0)
    dev.set_poses(P); o.set_poses(P)
    normalised = False
    for step in range(40):
        op = rng.choice(["update", "update", "update", "resample", "get_one", "get_all", "put_log", "put_lik", "reset", "combined"])
        where = f"seed {seed} call {step} ({op})"
        if op == "update":
            z = tr.scans[int(rng.integers(0, 8))]
            u = None if rng.random() < 0.5 else (0.01, float(np.radians(rng.choice([1.0, 40.0]))))
            dev.update(z, u, seed=seed, sequence=step)
            o.set_poses(dev.get_particles()[0])
            o.update(z, u, sample_motion=False, threads=THREADS)
            _compare_weights(dev.get_particles()[1], o.weights, where)
            normalised = True
        elif op == "resample" and normalised:
            r01 = float(rng.random())
            idx, amb = dev.resample(r01, want_indices=True)
            want, _ = o.resample(r01)
            assert_resample_indices(idx, want, amb)
            if not np.array_equal(idx, want):                                  # (a draw on a rounding boundary: follow the device)
                pytest.skip(f"{where}: the draw sits on a rounding boundary")
        elif op == "get_one":
            i = int(rng.integers(0, N))
            assert np.array_equal(dev.map_of(i, likelihood=True).reshape(-1), o.lik(i)), where
            assert np.max(np.abs(dev.map_of(i).reshape(-1) - o.log(i))) <= 1e-12, where
        elif op == "get_all":
            _compare_maps(dev, o, where)
        elif op == "put_log":
            i = int(rng.integers(0, N))
            lg = rng.choice([-1.7, 0.0, 0.0, 2.2], size=g.W * g.H)
            dev.set_map(i, log=lg.reshape(g.H, g.W)); o.set_log(i, lg)
        elif op == "put_lik":
            i = int(rng.integers(0, N))
            lk = rng.random(g.W * g.H)
            dev.set_map(i, lik=lk.reshape(g.H, g.W)); o.set_lik(i, lk)
        elif op == "reset":
            dev.reset(); o.reset()
            dev.set_poses(P); o.set_poses(P)
            normalised = False
        elif op == "combined":
            want = orc.combine_maps(o.logs())                                  # GridMapApp.calculateCombined (GridMapApp.java:439-458)
            got = dev.calculate_combined().reshape(-1)
            fin = np.isfinite(want)
            assert np.array_equal(np.isfinite(got), fin), where
            assert not fin.any() or np.max(np.abs(got[fin] - want[fin])) <= 1e-9 * max(1.0, np.abs(want[fin]).max()), where
    _compare_maps(dev, o, f"seed {seed}: at the end")
    dev.close()


def test_properties_that_need_no_oracle_at_1024_particles_of_256x256_cells():
    """size-independent properties of the per-particle-map path on 2 GB of GridMapData: (1) with equal weights and r = 0.5 the systematic
    draw is the identity, so resample()'s deep copies must reproduce every map bit for bit in the other generation; (2) an update with
    an empty scan integrates nothing and rebuilds every likelihood field from unchanged log-odds: both arrays stay as they are
    (computeLikelihoodMap is idempotent); (3) the same scan integrated into two filters whose particles are permutations of each other
    gives permuted maps (no cross-talk between workgroups sharing a CU)."""
    N, ext, res, B = 1024, 12.8, 0.05, 120
    tr = synth.make_trace(ext, res, B, T=4, seed=77)
    dev = SLAMParticleMaps(ext, ext, res, (-ext / 2, -ext / 2), num_particles=N, max_beams=128)
    P = synth.make_particles(tr.poses[0], N, seed=3, sigma_xy=0.2, sigma_theta_deg=8.0)
    dev.set_poses(P)
    for k in range(3):
        dev.update(tr.scans[k], None)
    logs = dev.maps().copy()
    assert (logs != 0).any(axis=(1, 2)).all()
    # (2) empty scans: the first brings likelihoodData up to the log-odds of the third scan (the field in memory was computed in front
    # of its integration, SLAM.java:93 before :105), the second finds nothing to change
    dev.update(tr.scans[0][:0], None)
    liks = dev.maps(likelihood=True).copy()
    dev.update(tr.scans[0][:0], None)
    assert np.array_equal(dev.maps(), logs) and np.array_equal(dev.maps(likelihood=True), liks)
    # (1) identity resampling: equal weights (an empty scan leaves every product at 1), r = 0.5
    w = dev.get_particles()[1]
    assert np.array_equal(w, np.full(N, 1.0 / N))
    before = dev.maps_copied()
    idx, amb = dev.resample(0.5, want_indices=True)
    assert np.array_equal(idx, np.arange(N)) and dev.maps_copied() == before + N
    assert np.array_equal(dev.maps(), logs) and np.array_equal(dev.maps(likelihood=True), liks)
    # (3) a permuted filter
    perm = np.random.default_rng(5).permutation(N)
    dev2 = SLAMParticleMaps(ext, ext, res, (-ext / 2, -ext / 2), num_particles=N, max_beams=128)
    dev2.set_poses(P[perm])
    for k in range(3):
        dev2.update(tr.scans[k], None)
    dev2.update(tr.scans[0][:0], None)
    assert np.array_equal(dev2.maps(), logs[perm]) and np.array_equal(dev2.maps(likelihood=True), liks[perm])
    dev.close(); dev2.close()


def test_squared_thresholds_classify_like_the_square_root():
    """inverseSensorModel compares (float) Math.sqrt(s) with measured -+ 1 (GridMap.java:217, SensorModel.java:31-41); the per-particle
    ray cast compares s with two thresholds per ray instead (gms_device.h: sq_lower / sq_upper).  For thousands of thresholds t --
    random, integers and halves (cell distances cluster there), tiny, huge, zero, negative, infinite, NaN -- the device's thresholds are
    checked against their definition with numpy's correctly rounded float32 sqrt: d < t <=> s < sq_lower(t), d > t <=> s > sq_upper(t)
    for every float s within a few ulps of the boundary (and at the domain's ends)."""
    from gridmap_slam_robot_amd import GridMap
    m = GridMap(3.2, 3.2, 0.05, (-1.6, -1.6))
    rng = np.random.default_rng(12)
    t = np.concatenate([
        rng.uniform(0, 600, 4000), np.arange(0, 400, 0.5), rng.uniform(0, 2, 500), 10.0 ** rng.uniform(-30, 19, 500),
        [0.0, -0.0, -1.0, -1e-30, 1e-45, 1.17549435e-38, 3.4e38, 1.8446744e19, 1.8446743e19, 1.8446746e19, np.inf, -np.inf, np.nan]]).astype(np.float32)
    lo, hi = m.debug_f32(4, t), m.debug_f32(5, t)
    one = np.uint32(1)

    def nbrs(c):
        """floats within 3 ulps of c (c >= 0 finite), plus 0, the largest float and inf"""
        out = [np.float32(0), np.float32(3.4028235e38), np.float32(np.inf)]
        if np.isfinite(c) and c >= 0:
            u = np.float32(c).view(np.uint32)
            for d in range(-3, 4):
                v = int(u) + d
                if 0 <= v <= 0x7f800000:
                    out.append(np.uint32(v).view(np.float32))
        return np.array(out, dtype=np.float32)

    with np.errstate(invalid="ignore"):
        for ti, l, h in zip(t, lo, hi):
            for c in (l, h):
                s = nbrs(c)
                d = np.sqrt(s)                                   # float32 in, float32 out: correctly rounded
                assert np.array_equal(d < ti, s < l), (ti, l, s)
                assert np.array_equal(d > ti, s > h), (ti, h, s)
    # a NaN s (a NaN pose) is neither below nor above anything, as NaN distance in the reference
    assert np.isnan(lo[-1]) and np.isnan(hi[-1])
