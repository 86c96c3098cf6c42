#!/usr/bin/env python3
"""computeLikelihoodMap writing likelihoodData ONLY (likelihood_body's mode 1: what the per-particle maps and gms_ensure_lik run) on the
dense worst-case map of tools/kbench.py --dense -- every 64 x 32 tile blurred -- at 2048 x 2048 @ 2 cm (11 taps): N such maps as the
particles' maps of a gms_slam handle, an update with an empty scan (likelihood of every map; the particle kernel returns at once).
Algorithmic bytes = the bytes moved: 16 B per cell.  usage: lik_mode1_dense.py [N=8] [iters=40]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gridmap_slam_robot_amd import SLAMParticleMaps
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ext, res = 40.96, 0.02
s = SLAMParticleMaps(ext, ext, res, (-ext / 2, -ext / 2), num_particles=N, max_beams=128)
s.grid_map.set_stream(torch.cuda.current_stream().cuda_stream)
y, x = np.mgrid[0:s.H, 0:s.W]
log = np.where(((x >> 3) + (y >> 3)) & 1, 2.197224312426715, -0.8472978036208759)
log[np.random.default_rng(0).random((s.H, s.W)) < 0.02] = 0.0
for i in range(N):
    s.set_map(i, log=log)
z = np.zeros(0, dtype=s.grid_map.deskew(np.zeros(1), np.ones(1), np.ones(1, np.uint8), 0.0, 0.0).beams.dtype)
for _ in range(5):
    s.update(z, None)
s.grid_map.profile(True); s.grid_map.profile_reset()
for _ in range(iters):
    s.update(z, None)
torch.cuda.synchronize()
p = s.grid_map.profile_get(); s.grid_map.profile(False)
us = p["likelihood"][0] / p["likelihood"][1] * 1e3
b = 16.0 * s.W * s.H * N
print(f"{N} dense {s.W}x{s.H} maps, likelihoodData only: {us:.1f} us per launch (bracketed) = {us / N:.2f} us per map; {b / 1e6:.0f} MB -> {b / us / 1e6:.2f} TB/s = {b / us / 8e6:.3f} of the HBM peak")
