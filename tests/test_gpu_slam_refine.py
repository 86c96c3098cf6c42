"""SLAM.update's pose refinement in the reference's own filter shape (J/slam/SLAM.java:96-97 -> GridMap.findBestPose,
J/slam/GridMap.java:319-346): every particle's pose is replaced by the best pose of an 11 x 11 x 10 lattice around its motion-model
sample, searched against THE PARTICLE'S OWN likelihood field, before the particle is weighted and its map updated at that pose.  The
device (gms_slam_set_refine: k_slam_refine, the particle's field computed in the CU's LDS from the particle's class plane, or staged
there from memory, or read from memory: GMS_SLAM_REFINE_LDS) against the oracle's literal loop
(orc_slam_update(..., refine = 1)): poses EQUAL (the argmax is over products of doubles taken in beam order: any other association
could pick another pose), weights to 1e-13 (the weight sum is a blocked sum on the device), maps as in
tests/test_gpu_slam_particle_maps.py.  The motion-model samples are set by hand (Philox variates computed on the host) so that the
device's double-precision log / sin / cos cannot round a pose differently before the search starts; one test runs the draw inside
the refinement launch and compares it with the two separate calls."""
import os

import numpy as np
import pytest

from gridmap_slam_robot_amd import SLAMParticleMaps, synth
from gridmap_slam_robot_amd.trace import read_trace
from oracle import oracle as orc

from test_gpu_slam_particle_maps import _compare_maps, _compare_weights, _frames_to_scans

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
THREADS = min(16, os.cpu_count() or 1)


def _run_refined(dev, o, scans, start, seed, rng, check_maps_at, label):
    n = o.n
    P0 = np.tile(np.asarray(start, np.float32), (n, 1))
    dev.set_poses(P0); o.set_poses(P0)
    dev.set_refine(True)
    moved, resampled = 0, 0
    for k, (z, u) in enumerate(scans):
        P_in = orc.sample_motion(o.poses, u[0], u[1], seed=seed, sequence=k)              # SLAM.java:90
        dev.set_poses(P_in); o.set_poses(P_in)
        neff = dev.update(z, u, sample_motion=False)                                       # :93, :96, :99, :105
        neff_o = o.update(z, u, sample_motion=False, refine=True, threads=THREADS)
        P, w = dev.get_particles()
        assert np.array_equal(P, o.poses), f"{label} frame {k}: {int((P != o.poses).any(axis=1).sum())} of {n} refined poses differ"
        moved += int((P != P_in).any(axis=1).sum())
        _compare_weights(w, o.weights, f"{label} frame {k}")
        assert dev.last_stats["strongest"] == o.strongest
        assert abs(neff - neff_o) <= 1e-11 * neff_o
        if k in check_maps_at:
            _compare_maps(dev, o, f"{label} frame {k}")
        if neff_o < n // 2:                                                                # GridMapApp.java:185-186
            r01 = float(rng.random())
            idx, _ = dev.resample(r01, want_indices=True)
            want, clamped = o.resample(r01)
            assert clamped == 0 and np.array_equal(idx, want), f"{label} frame {k}: the draw {r01} sits on a rounding boundary; pick another seed"
            resampled += 1
    return moved, resampled


def test_refined_update_at_the_reference_operating_point():
    """500 particles x 120 x 120 cells x 90 beams (SLAM.java:50,57), twelve revolutions of the drive, the caller's resampling rule
    included: the field of a particle is 115 KB and lives in its workgroup's LDS."""
    N, B, T = 500, 90, 12
    frames, truth = synth.make_recording(6.0, B, T=48, seed=77, n_frames=T)
    scans = _frames_to_scans(frames)
    start = synth.true_pose(synth.make_world(6.0, 77), -1, 48)
    dev = SLAMParticleMaps(6.0, 6.0, 0.05, (-3.0, -3.0), num_particles=N, max_beams=128)
    o = orc.Slam(orc.Grid(6.0, 6.0, 0.05, -3.0, -3.0), N)
    moved, resampled = _run_refined(dev, o, scans, start, seed=2024, rng=np.random.default_rng(11), check_maps_at={0, 1, 5, T - 1}, label="refine 500x120^2")
    assert moved > N * T // 2, "the search must move most poses"
    assert resampled >= 1
    wp = dev.get_weighted_pose()
    assert np.hypot(wp[0] - truth[T - 1][0], wp[1] - truth[T - 1][1]) < 1.0
    dev.close()


@pytest.mark.parametrize("case", ["field_in_memory_forced", "field_staged_from_memory", "field_from_class_plane", "short_bands", "one_strip_rows", "wide_map", "nine_taps", "256x256_does_not_fit", "2cm_11_taps", "ragged_map", "long_scan", "underflow_700_beams"])
def test_refined_update_other_shapes(case, monkeypatch):
    """the form that reads the field from memory (forced on a small map; a 256 x 256 map, 512 KB, which no LDS holds), the form that
    stages a field written by k_slam_likelihood in the LDS (the default computes it there from the class plane), maps of 20 x 10 cells
    (the column march's bands are as short as they may be, the last one shorter than the kernel's half width) and of 7 x 40 (a row
    is one strip, and that one not full) and of 1100 x 12 (wider than the workgroup has threads: the field fits the LDS but is staged), a blur kernel of nine taps given by the caller (no compile-time kernel: k_slam_likelihood's generic
    form writes the field, the refinement stages it), the field in
    front of the refinement written from the particles' class planes (what a filter does whose logData exceeds the infinity cache:
    gms_slam::refine_field), the 11-tap
    kernel of a 2 cm map, a map whose width is odd (the staging's scalar form), a scan of 300 beams (whose rotation table does not
    fit beside the field: rotated per look-up), and a scan of 700 beams in a map so much larger than the room that no lattice pose
    puts an end point outside it: every product underflows to 0, maxProb stays 0 and the start pose is kept (GridMap.java:320-321,
    334) -- and update() then divides 0 by 0, on both sides."""
    W, H, res, B, N, T = {"field_in_memory_forced": (4.0, 4.0, 0.05, 72, 24, 4), "field_staged_from_memory": (4.0, 4.0, 0.05, 72, 24, 4),
                          "short_bands": (1.0, 0.5, 0.05, 48, 10, 4), "one_strip_rows": (0.35, 2.0, 0.05, 48, 10, 4), "wide_map": (55.0, 0.6, 0.05, 48, 6, 3), "nine_taps": (4.0, 4.0, 0.05, 72, 16, 4), "field_from_class_plane": (4.0, 4.0, 0.05, 72, 24, 5),
                          "256x256_does_not_fit": (12.8, 12.8, 0.05, 120, 24, 3),
                          "2cm_11_taps": (2.4, 2.4, 0.02, 72, 12, 3), "ragged_map": (2.55, 3.35, 0.05, 64, 10, 4),
                          "long_scan": (6.0, 6.0, 0.05, 300, 8, 3), "underflow_700_beams": (12.8, 12.8, 0.05, 700, 8, 2)}[case]
    if case == "field_in_memory_forced":
        monkeypatch.setenv("GMS_SLAM_REFINE_LDS", "0")
    if case == "field_staged_from_memory":
        monkeypatch.setenv("GMS_SLAM_REFINE_LDS", "2")
    if case == "field_from_class_plane":
        monkeypatch.setenv("GMS_SLAM_REFINE_FIELD", "codes")
    ext = min(W, H)
    tr = synth.make_trace(min(ext, 6.4), res, B, T=T + 1, seed=61)
    g = orc.Grid(W, H, res, -W / 2, -H / 2)
    taps = [0.01, 0.05, 0.12, 0.2, 0.24, 0.2, 0.12, 0.05, 0.01] if case == "nine_taps" else None      # (none of the compile-time kernels: the field is written to memory first)
    if taps is not None:
        g.set_kernel(taps)
    dev = SLAMParticleMaps(W, H, res, (-W / 2, -H / 2), num_particles=N, max_beams=max(128, B), kernel=taps)
    o = orc.Slam(g, N)
    assert (dev.W, dev.H) == (g.W, g.H)
    dev.set_refine(True)
    for k in range(T):
        P = synth.make_particles(tr.poses[k], N, seed=4 + k, sigma_xy=0.05, sigma_theta_deg=3.0)
        dev.set_poses(P); o.set_poses(P)
        z = tr.scans[k]
        dev.update(z, None); o.update(z, None, refine=True, threads=THREADS)
        Pd, w = dev.get_particles()
        assert np.array_equal(Pd, o.poses), f"{case} frame {k}"
        if case == "underflow_700_beams" and k == 0:       # (blank maps: every factor is 1 / range)
            assert np.array_equal(Pd, P) and np.isnan(o.weights).all() and np.isnan(w).all()
        else:
            assert (Pd != P).any()
            _compare_weights(w, o.weights, f"{case} frame {k}")
        _compare_maps(dev, o, f"{case} frame {k}")
    dev.close()


def test_the_motion_sample_drawn_in_the_refinement_launch():
    """update(z, u) with refinement on draws the motion-model sample (SLAM.java:90) inside the refinement launch: the same poses,
    weights and maps as gms_pf_sample_motion followed by update(sample_motion = False) on a second handle -- the same Philox
    counters, the same bits."""
    ext, res, B, N, T = 4.0, 0.05, 72, 40, 5
    frames, _ = synth.make_recording(ext, B, T=32, seed=5, n_frames=T)
    scans = _frames_to_scans(frames)
    start = synth.true_pose(synth.make_world(ext, 5), -1, 32)
    a = SLAMParticleMaps(ext, ext, res, (-ext / 2, -ext / 2), num_particles=N, max_beams=128)
    b = SLAMParticleMaps(ext, ext, res, (-ext / 2, -ext / 2), num_particles=N, max_beams=128)
    P0 = np.tile(np.asarray(start, np.float32), (N, 1))
    for h in (a, b):
        h.set_poses(P0); h.set_refine(True)
    for k, (z, u) in enumerate(scans):
        a.update(z, u, seed=99, sequence=k)
        b.pf.sample_motion(u[0], u[1], 99, k)
        b.update(z, u, sample_motion=False)
        Pa, wa = a.get_particles()
        Pb, wb = b.get_particles()
        assert np.array_equal(Pa, Pb) and np.array_equal(wa, wb), f"frame {k}"
    assert np.array_equal(a.maps(), b.maps()) and np.array_equal(a.maps(likelihood=True), b.maps(likelihood=True))
    a.close(); b.close()


def test_refinement_edge_cases():
    """a scan without a hit (every product is 1: the first lattice pose wins, GridMap.java:334 -- on the device as in the oracle); no
    beam at all; particles whose lattice lies outside the map (every beam skipped, :276: again the first pose) or straddles its
    edge; refinement switched off again.  (Products that are exactly 0: test_refined_update_other_shapes[underflow_700_beams].)"""
    ext, res, B, N = 4.0, 0.05, 64, 16
    tr = synth.make_trace(ext, res, B, T=4, seed=71)
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    dev = SLAMParticleMaps(ext, ext, res, (-ext / 2, -ext / 2), num_particles=N, max_beams=128)
    o = orc.Slam(g, N)
    dev.set_refine(True)
    P = synth.make_particles(tr.poses[0], N, seed=3, sigma_xy=0.05, sigma_theta_deg=3.0)
    P[2] = [30.0, 2.0, 0.4]                                                    # far outside
    P[3] = [-2.1, 0.0, 0.0]                                                    # the lattice straddles the map's edge
    dev.set_poses(P); o.set_poses(P)
    for name, z in (("first scan", tr.scans[0]), ("second scan", tr.scans[1])):
        dev.update(z, None); o.update(z, None, refine=True, threads=THREADS)
        assert np.array_equal(dev.get_particles()[0], o.poses), name
        _compare_maps(dev, o, name)
    zm = tr.scans[2].copy(); zm["hit"] = 0; zm["distance"] = 10.0
    before = dev.get_particles()[0]
    dev.update(zm, None); o.update(zm, None, refine=True, threads=THREADS)
    Pm = dev.get_particles()[0]
    assert np.array_equal(Pm, o.poses)
    assert not np.array_equal(Pm, before), "on equal products the FIRST lattice pose wins, not the start pose"
    dev.update(tr.scans[2][:0], None); o.update(tr.scans[2][:0], None, refine=True, threads=THREADS)
    assert np.array_equal(dev.get_particles()[0], o.poses)
    n_before = dev.get_particles()[0]
    dev.set_refine(False)
    dev.update(tr.scans[3], None); o.update(tr.scans[3], None, threads=THREADS)
    assert np.array_equal(dev.get_particles()[0], n_before) and np.array_equal(n_before, o.poses)
    _compare_maps(dev, o, "refinement off again")
    dev.close()


def test_both_forms_of_the_field_at_4096_maps_of_256_x_256(monkeypatch):
    """at 4096 x 256^2 logData is 2 GB and the field in front of the refinement is written from the class planes (the launcher's own
    rule); a second handle is told to read logData: poses, weights and the maps of a few particles EQUAL over three refined updates
    with a resampling step between them."""
    N, B, ext, T = 4096, 180, 12.8, 3
    frames, _ = synth.make_recording(ext / 2, B, T=48, seed=78, n_frames=T)
    scans = _frames_to_scans(frames)
    start = synth.true_pose(synth.make_world(ext / 2, 78), -1, 48)
    a = SLAMParticleMaps(ext, ext, 0.05, (-ext / 2, -ext / 2), num_particles=N, max_beams=256)
    monkeypatch.setenv("GMS_SLAM_REFINE_FIELD", "log")
    b = SLAMParticleMaps(ext, ext, 0.05, (-ext / 2, -ext / 2), num_particles=N, max_beams=256)
    P0 = np.tile(np.asarray(start, np.float32), (N, 1))
    for h in (a, b):
        h.set_poses(P0); h.set_refine(True)
    for k, (z, u) in enumerate(scans):
        for h in (a, b):
            h.update(z, u, seed=5, sequence=k)
        Pa, wa = a.get_particles()
        Pb, wb = b.get_particles()
        assert np.array_equal(Pa, Pb) and np.array_equal(wa, wb), f"frame {k}"
        assert (Pa != P0).any()
        ia, _ = a.resample(0.37, want_indices=True)
        ib, _ = b.resample(0.37, want_indices=True)
        assert np.array_equal(ia, ib)
    for i in (0, 1, 2047, 4095):
        assert np.array_equal(a.map_of(i), b.map_of(i)) and np.array_equal(a.map_of(i, likelihood=True), b.map_of(i, likelihood=True))
    a.close(); b.close()
