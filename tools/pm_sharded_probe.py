#!/usr/bin/env python3
"""Where the host time of one sharded update goes (one rank, development tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gridmap_slam_robot_amd import synth
from gridmap_slam_robot_amd.distributed import ShardedSlamParticleMaps, SlamShardOps, TorchCollectives
N, ext, B = 1024, 6.0, 90
frames, _ = synth.make_recording(ext, B, T=48, seed=77)
start = synth.true_pose(synth.make_world(ext, 77), -1, 48)
ops = SlamShardOps(ext, ext, 0.05, (-ext / 2, -ext / 2), N, 0, N, max_beams=128)
ops.slam.set_poses(np.tile(np.asarray(start, np.float32), (N, 1)))
f = ShardedSlamParticleMaps(N, ops, coll=TorchCollectives())
scans = [(ops.slam.grid_map.deskew(fr.angle, fr.distance, fr.hit, fr.d_center, fr.d_theta), (fr.d_center, fr.d_theta)) for fr in frames]
for i in range(10):
    f.update(scans[i][0], scans[i][1], seed=1, sequence=i)
def t(fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("update_local            %.1f us" % t(lambda i: ops.update_local(scans[i % 48][0], scans[i % 48][1], 1, 100 + i)))
print("weights.normalize       %.1f us" % t(lambda i: f.weights.normalize()))
print("weights.normalize_begin %.1f us" % t(lambda i: f.weights.normalize_begin()))
print("weights.normalize_end   %.1f us" % t(lambda i: f.weights.normalize_end()))
print("stats()                 %.1f us" % t(lambda i: f.weights.stats()))
print("update (all)            %.1f us" % t(lambda i: f.update(scans[i % 48][0], scans[i % 48][1], seed=1, sequence=200 + i)))
print("draw                    %.1f us" % t(lambda i: ops.draw(0.3, None), 20))
