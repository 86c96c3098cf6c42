"""A recorded trace through the device path frame by frame -- read_trace -> gms_map_deskew -> gms_pf_sample_motion -> fused scan
step, and the same as one gms_slam_frame call per revolution (GridMapApp.onHandleData, J/app/GridMapApp.java:133-192; DataRecorder.load, J/app/DataRecorder.java:403-436) -- against
the oracle step by step, the oracle being fed the device's state of the stage before.  The recording is the committed synthetic
one (tests/golden/recording_360.bin, tools/make_recording.py; the reference ships none)."""
import os

import numpy as np
import pytest

from gridmap_slam_robot_amd import GridMap, ParticleFilter, synth
from gridmap_slam_robot_amd.replay import TraceReplay
from gridmap_slam_robot_amd.trace import read_trace
from oracle import oracle as orc

from _checks import assert_resample_indices, near_boundary_slots

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
REC = os.path.join(HERE, "golden", "recording_360.bin")


def test_recording_replays_against_the_oracle():
    frames = read_trace(REC)
    truth = np.load(os.path.join(HERE, "golden", "recording_360_poses.npy"))
    assert len(frames) == 64 and all(len(f.angle) == 360 for f in frames)
    ext, res, N, BOOT, STEPS, SEED = 25.6, 0.05, 1500, 6, 26, 99
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    a = GridMap(ext, ext, res, (-ext / 2, -ext / 2), max_beams=512)          # driven stage by stage, with read-backs
    b = GridMap(ext, ext, res, (-ext / 2, -ext / 2), max_beams=512)          # driven by TraceReplay, nothing read back until the end
    pa, pb = ParticleFilter(a, N), ParticleFilter(b, N)
    start = synth.true_pose(synth.make_world(ext, 4321), -1, 64)              # where the drive of the recording begins
    ra, rb = TraceReplay(a, pa, start, seed=SEED), TraceReplay(b, pb, start, seed=SEED)
    log = g.new_log()
    pose = np.asarray(start, dtype=np.float32)
    for f in frames[:BOOT]:
        ra.bootstrap(f); rb.bootstrap(f)
        pose = synth.dead_reckon(pose, f.d_center, f.d_theta)
        g.integrate(log, orc.deskew(f.angle, f.distance, f.hit, f.d_center, f.d_theta), pose)
    assert np.array_equal(a.download_log().reshape(-1) != 0, log != 0)
    log = a.download_log().reshape(-1).copy()                                 # the oracle's copy of the map, carried along from here
    lik = g.build_likelihood(log)
    assert np.array_equal(a.download_likelihood().reshape(-1), lik)
    assert np.max(np.abs(pose[:2] - truth[BOOT - 1][:2])) < 0.05              # dead reckoning follows the drive
    rng = np.random.default_rng(5)
    resampled = 0
    for k, f in enumerate(frames[BOOT:BOOT + STEPS]):
        r01 = float(rng.random())
        before = pa.get_poses()
        # stage 1: de-skew (GridMapApp.java:143-175)
        obs = a.deskew(f.angle, f.distance, f.hit, f.d_center, f.d_theta)
        want = orc.deskew(f.angle, f.distance, f.hit, f.d_center, f.d_theta)
        assert np.array_equal(obs.beams["hit"], want["hit"])
        for key in ("local_x", "local_y", "distance"):
            assert np.max(np.abs(obs.beams[key] - want[key])) <= 1e-13
        dev, B = a.deskew_dev(f.angle, f.distance, f.hit, f.d_center, f.d_theta)
        # stage 2: motion-model sample per particle (SLAM.java:90 -> Odometry.java:77-96; Philox variates)
        pa.sample_motion(f.d_center, f.d_theta, SEED, ra.frame_no)
        P = pa.get_poses()
        Po = orc.sample_motion(before, f.d_center, f.d_theta, seed=SEED, sequence=ra.frame_no)
        assert (np.all(P == Po, axis=1)).mean() > 0.995 and np.max(np.abs(P - Po)) <= 1e-6      # device libm vs glibc: last ulp of a float
        # stage 3: SLAM.update + conditional resample (SLAM.java:87-131, GridMapApp.java:185-186)
        pa.slam_update_dev(0, dev, B, r01, 0.5, True)
        ra.frame_no += 1
        rb.step(f, r01)                                                       # gms_slam_frame: the same three stages as ONE call
        st, last = pa.stats(), pa.last_step()
        w = g.score(lik, obs.beams, P)
        wn = w.copy()
        ws, strongest = orc.normalize(wn)
        assert ws > 0 and st["strongest"] == strongest and abs(st["weight_sum"] - ws) <= 1e-11 * ws
        neff = orc.neff(wn)
        assert abs(st["neff"] - neff) <= 1e-9 * neff
        assert np.allclose(last["weighted_pose"], orc.weighted_pose(P, wn), rtol=0, atol=2e-6)
        if abs(neff - 0.5 * N) > 1e-6 * N:
            assert last["did_resample"] == (neff < 0.5 * N)
        got = pa.get_poses()
        if last["did_resample"]:
            resampled += 1
            idx, _ = orc.resample_indices(np.ascontiguousarray(wn), r01)
            got_idx = pa.last_resample_indices()
            assert_resample_indices(got_idx, idx, last["n_ambiguous"] + near_boundary_slots(wn, r01))
            assert np.array_equal(got, P[got_idx])
        else:
            assert np.array_equal(got, P)
        g.integrate(log, obs.beams, last["weighted_pose"])                    # the map update happened at the filter's own pose
        lik = g.build_likelihood(log)
        assert np.array_equal(a.download_likelihood().reshape(-1), lik)
        if k % 5 == 4:
            gl = a.download_log().reshape(-1)
            assert np.array_equal(gl != 0, log != 0)
            nz = log != 0
            assert np.max(np.abs(gl[nz] - log[nz]) / np.abs(log[nz])) <= 1e-13
    assert resampled > 0
    # the one-call frame (de-skew and motion model in one launch, nothing read back) arrives at the same filter and the same map, bit for bit
    assert np.array_equal(pa.get_poses(), pb.get_poses()) and np.array_equal(pa.get_weights(), pb.get_weights())
    assert np.array_equal(a.download_log(), b.download_log()) and np.array_equal(a.download_likelihood(), b.download_likelihood())
    # ... and the filter still knows where the robot is
    est = pb.weighted_pose()
    assert np.hypot(*(est[:2] - truth[BOOT + STEPS - 1][:2])) < 0.25
    for h in (pa, pb):
        h.close()
    a.close(); b.close()


def test_frame_call_refuses_what_it_cannot_do():
    from gridmap_slam_robot_amd._lib import GmsError
    m = GridMap(6.4, 6.4, 0.05, (-3.2, -3.2), n_maps=2, max_beams=512)
    pf = ParticleFilter(m, 256)
    ang = np.linspace(0, 2 * np.pi, 90, endpoint=False); dist = np.full(90, 2.0); hit = np.ones(90, dtype=np.uint8)
    with pytest.raises(GmsError):
        pf.slam_frame(ang, dist, hit, 0.01, 0.0, 1, 0, 0.5)                   # batched handle: a frame is one robot's revolution
    pf.close(); m.close()
    m = GridMap(6.4, 6.4, 0.05, (-3.2, -3.2), max_beams=64)
    pf = ParticleFilter(m, 256)
    with pytest.raises(GmsError):
        pf.slam_frame(ang, dist, hit, 0.01, 0.0, 1, 0, 0.5)                   # more measurements than max_beams
    pf.close(); m.close()
