#!/usr/bin/env python3
"""Refinement launch at 4096 x 256^2 x 180 (the field stays in memory) under event brackets (development tool): GMS_SLAM_REFINE_TAB_KB sweep."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gridmap_slam_robot_amd import SLAMParticleMaps, synth
N, ext, B, T = 4096, 12.8, 180, 48
frames, _ = synth.make_recording(ext / 2, B, T=T, seed=78)
start = synth.true_pose(synth.make_world(ext / 2, 78), -1, T)
dev = torch.device("cuda", 0)
s = SLAMParticleMaps(ext, ext, 0.05, (-ext / 2, -ext / 2), num_particles=N, max_beams=256)
s.grid_map.set_stream(torch.cuda.current_stream().cuda_stream)
s.set_poses(np.tile(np.asarray(start, np.float32), (N, 1)))
s.set_refine(True)
scans, odo = [], []
for f in frames:
    obs = s.grid_map.deskew(f.angle, f.distance, f.hit, f.d_center, f.d_theta)
    scans.append(torch.from_numpy(obs.beams.view(np.uint8).reshape(-1).copy()).to(dev)); odo.append((f.d_center, f.d_theta))
for i in range(12):
    s.update_dev(scans[i % T].data_ptr(), B, odo[i % T], seed=11, sequence=i)
    if i % 4 == 3: s.resample(0.3)
s.grid_map.profile(True); s.grid_map.profile_reset()
for i in range(10):
    s.update_dev(scans[(12 + i) % T].data_ptr(), B, odo[(12 + i) % T], seed=11, sequence=100 + i)
torch.cuda.synchronize()
rep = s.grid_map.profile_get()
print(os.environ.get("GMS_SLAM_REFINE_TAB_KB", "default"), {k: round(ms / n * 1e3, 1) for k, (ms, n) in rep.items() if n and k in ("refine", "likelihood", "score")})
