#!/usr/bin/env python3
"""A/B of two builds of the library on the per-particle-map update (development tool): alternating child processes, each timing
un-bracketed updates at one size; prints every run and the medians.
  python tools/ab_update.py build/exp/a.so build/exp/b.so [N EXT B] [rounds]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, numpy as np, torch
sys.path.insert(0, %r)
from gridmap_slam_robot_amd import SLAMParticleMaps, synth
N, ext, B = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
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
    scans.append(torch.from_numpy(obs.beams.view(np.uint8).reshape(-1).copy()).to(dev)); odo.append((f.d_center, f.d_theta))
for i in range(60):
    s.update_dev(scans[i %% T].data_ptr(), B, odo[i %% T], seed=11, sequence=i)
    if i %% 4 == 3: s.resample(0.3)
out = []
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(40):
        s.update_dev(scans[(12 + i) %% T].data_ptr(), B, odo[(12 + i) %% T], seed=11, sequence=100 + i)
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / 40 * 1e6)
print(sorted(out)[2])
''' % ROOT
libs = sys.argv[1:3]
N, ext, B = (sys.argv[3:6] if len(sys.argv) > 5 else ("500", "6.0", "90"))
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 4
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, GMS_LIBRARY=os.path.abspath(l))
        v = float(subprocess.run([sys.executable, "-c", CHILD, N, ext, B], env=env, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
        res[l].append(round(v, 2))
for l in libs:
    print(l, res[l], "median", sorted(res[l])[len(res[l]) // 2])
