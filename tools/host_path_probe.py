#!/usr/bin/env python3
"""Where the host time of gms_slam_update (host-resident inputs) goes: raw ctypes calls with pre-built arrays."""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gridmap_slam_robot_amd import GridMap, ParticleFilter, synth, _lib

cfg = synth.CONFIGS["C3"]; B, ext, res, N = cfg["beams"], cfg["extent"], cfg["resolution"], cfg["particles"]
tr = synth.make_trace(ext, res, B, T=40, seed=1234)
m = GridMap(ext, ext, res, (-ext / 2, -ext / 2), max_beams=2048)
for t in range(32): m.update(tr.scans[t], tr.poses[t])
pf = ParticleFilter(m, N)
P = np.ascontiguousarray(synth.make_particles(tr.poses[32], N, seed=99))
beams = np.ascontiguousarray(tr.scans[32])
L = _lib.load()
r = (C.c_double * 1)(0.37)
def run(poses, label, K=300):
    pp = C.c_void_p(poses.ctypes.data) if poses is not None else None
    for _ in range(20): L.gms_slam_update(pf._h, pp, C.c_void_p(beams.ctypes.data), B, r, 0.5, 1, None)
    m.synchronize(); t0 = time.perf_counter()
    for _ in range(K): L.gms_slam_update(pf._h, pp, C.c_void_p(beams.ctypes.data), B, r, 0.5, 1, None)
    t1 = time.perf_counter(); m.synchronize(); t2 = time.perf_counter()
    print(f"{label}: host issue {1e6 * (t1 - t0) / K:.1f} us/step, total {1e6 * (t2 - t0) / K:.1f} us/step", flush=True)
run(P, "poses + scan from host")
run(None, "scan from host only")

import torch
dev = torch.device("cuda", 0)
pose = np.ascontiguousarray(tr.poses[32], dtype=np.float32)
beams_dev = torch.from_numpy(beams.view(np.uint8).copy()).to(dev)
pose_dev = torch.from_numpy(pose).to(dev)
def probe(fn, label, K=300):
    for _ in range(20): fn()
    m.synchronize(); t0 = time.perf_counter()
    for _ in range(K): fn()
    t1 = time.perf_counter(); m.synchronize()
    print(f"{label}: host {1e6 * (t1 - t0) / K:.1f} us/call", flush=True)
probe(lambda: L.gms_map_integrate_dev(m._h, C.c_void_p(beams_dev.data_ptr()), B, C.c_void_p(pose_dev.data_ptr())), "integrate, device inputs (2 launches)")
probe(lambda: L.gms_map_integrate(m._h, C.c_void_p(beams.ctypes.data), B, C.c_void_p(pose.ctypes.data)), "integrate, host inputs (staging + 2 launches)")
probe(lambda: L.gms_pf_set_poses(pf._h, C.c_void_p(P.ctypes.data)), "set_poses from host (memcpy 196 KB + 1 launch)")
probe(lambda: L.gms_pf_set_poses_dev(pf._h, C.c_void_p(m._h.value)) if False else None, "noop python lambda")
