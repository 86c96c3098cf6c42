#!/usr/bin/env python3
"""Where k_slam_particle's time goes (per-particle-map mode): the update with and without integrateObservation (|dTheta| > 30 degrees
skips it, SLAM.java:82), with and without the motion sample, at several beam counts.  Prints microseconds per kernel class."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from gridmap_slam_robot_amd import SLAMParticleMaps, synth

def run(N, ext, B, steps=40):
    T = 48
    frames, _ = synth.make_recording(ext, B, T=T, seed=77)
    start = synth.true_pose(synth.make_world(ext, 77), -1, T)
    dev = torch.device("cuda", 0)
    s = SLAMParticleMaps(ext, ext, 0.05, (-ext / 2, -ext / 2), num_particles=N, max_beams=max(128, B))
    s.grid_map.set_stream(torch.cuda.current_stream().cuda_stream)
    s.set_poses(np.tile(np.asarray(start, np.float32), (N, 1)))
    scans, odo = [], []
    for f in frames:
        obs = s.grid_map.deskew(f.angle, f.distance, f.hit, f.d_center, f.d_theta)
        scans.append(torch.from_numpy(obs.beams.view(np.uint8).reshape(-1).copy()).to(dev))
        odo.append((f.d_center, f.d_theta))
    for i in range(12):
        s.update_dev(scans[i].data_ptr(), B, odo[i], seed=11, sequence=i)
        if i % 4 == 3: s.resample(0.3)
    out = {}
    for name, skip, motion in (("full", False, True), ("no_integrate", True, True), ("no_motion", False, False), ("score_only", True, False)):
        s.grid_map.profile(True); s.grid_map.profile_reset()
        for i in range(steps):
            k = (12 + i) % T
            u = (odo[k][0], 1.0 if skip else odo[k][1])
            s.update_dev(scans[k].data_ptr(), B, u, seed=11, sequence=100 + i, sample_motion=motion)
            s.resample(0.3 + 0.01 * i)
        torch.cuda.synchronize()
        p = s.grid_map.profile_get(); s.grid_map.profile(False)
        out[name] = p["score"][0] / p["score"][1] * 1e3
    s.close()
    return out

for N, ext, B in ((500, 6.0, 90), (500, 6.0, 180), (500, 6.0, 360), (4096, 12.8, 180)):
    r = run(N, ext, B, steps=20 if N > 1000 else 40)
    print(f"N={N} ext={ext} B={B}: " + "  ".join(f"{k} {v:.1f} us" for k, v in r.items()), flush=True)
