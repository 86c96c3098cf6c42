#!/usr/bin/env python3
"""Per-particle-map filter left running: `seconds` of SLAM.update on a drive round a room with the reference's resampling rule
(Neff < N / 2, GridMapApp.java:185-186), one host round trip per scan (the Neff).  Prints what a long run must keep: finite weights, a weighted
pose near the true one, the resampling count.  usage: pm_soak.py [seconds=5] [N=500] [host|device] [refine] [lognorm]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gridmap_slam_robot_amd import SLAMParticleMaps, synth
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 500
DEVICE_RULE = len(sys.argv) > 3 and sys.argv[3] == "device"        # the rule decided on the device (resample_if): no round trip per scan
REFINE = "refine" in sys.argv[4:]                                  # findBestPose of every particle against its own field (SLAM.java:96)
LOGNORM = "lognorm" in sys.argv[4:]                                # the opt-in log-normalisation (not the reference's arithmetic): a filter that stays alive
ext, res, B, T = 6.0, 0.05, 90, 48
frames, truth = synth.make_recording(ext, B, T=T, seed=77)
start = synth.true_pose(synth.make_world(ext, 77), -1, T)
s = SLAMParticleMaps(ext, ext, res, (-ext / 2, -ext / 2), num_particles=N, max_beams=128)
s.set_poses(np.tile(np.asarray(start, np.float32), (N, 1)))
if REFINE: s.set_refine(True)
if LOGNORM: s.pf.set_log_normalize(True)
scans = [(s.grid_map.deskew(f.angle, f.distance, f.hit, f.d_center, f.d_theta).beams, (f.d_center, f.d_theta)) for f in frames]
rng = np.random.default_rng(1)
t0 = time.perf_counter(); k = 0; resampled = 0; worst = 0.0
while time.perf_counter() - t0 < seconds:
    z, u = scans[k % T]
    if DEVICE_RULE:
        s.update(z, u, seed=3, sequence=k, fetch=False)
        s.resample_if(float(rng.random()), 0.5)
        if k % 1024 == 0:
            resampled += int(np.asarray(s.pf.did_resample()).reshape(-1)[0])           # (a sample: one scan in 1024)
    else:
        neff = s.update(z, u, seed=3, sequence=k)
        assert np.isfinite(neff) and 1.0 - 1e-9 <= neff <= N + 1e-6, (k, neff)
        if neff < N / 2:
            s.resample(float(rng.random())); resampled += 1
    if k % T == T - 1:
        wp = s.get_weighted_pose(); tp = truth[T - 1]
        worst = max(worst, float(np.hypot(wp[0] - tp[0], wp[1] - tp[1])))
    k += 1
el = time.perf_counter() - t0
w = s.get_particles()[1]
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(("refine " if REFINE else "") + ("lognorm " if LOGNORM else "") + f"{k} scans in {el:.1f} s ({el / k * 1e6:.0f} us per scan, {'rule decided on the device, no round trip' if DEVICE_RULE else 'host round trip included'}), {resampled} resampling steps{' (of the one scan in 1024 that was asked)' if DEVICE_RULE else ''}, weights finite: {bool(np.isfinite(w).all())}, "
      f"sum {w.sum():.15f}, worst end-of-lap distance of the weighted pose from the true one {worst:.3f} m, maps copied {s.maps_copied()}")
