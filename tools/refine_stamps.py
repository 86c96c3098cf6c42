#!/usr/bin/env python3
"""Stage timeline of k_slam_refine (the per-particle pose refinement, SLAM.java:96) from an instrumented build (development tool).
  GMS_EXTRA_FLAGS=-DGMS_STAMPS python -c "from gridmap_slam_robot_amd import build as b; b.build()"; cp .../libgridmapslam.so build/exp/lib_stamps.so; rebuild
  GMS_LIBRARY=$PWD/build/exp/lib_stamps.so python tools/refine_stamps.py [N EXT B]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gridmap_slam_robot_amd import SLAMParticleMaps, synth, _lib
N, ext, B = (int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (500, 6.0, 90)
T = 48
frames, _ = synth.make_recording(ext, B, T=T, seed=77)
start = synth.true_pose(synth.make_world(ext, 77), -1, T)
dev = torch.device("cuda", 0)
s = SLAMParticleMaps(ext, ext, 0.05, (-ext / 2, -ext / 2), num_particles=N, max_beams=max(128, B))
s.grid_map.set_stream(torch.cuda.current_stream().cuda_stream)
s.set_poses(np.tile(np.asarray(start, np.float32), (N, 1)))
s.set_refine(True)
scans, odo = [], []
for f in frames:
    obs = s.grid_map.deskew(f.angle, f.distance, f.hit, f.d_center, f.d_theta)
    scans.append(torch.from_numpy(obs.beams.view(np.uint8).reshape(-1).copy()).to(dev)); odo.append((f.d_center, f.d_theta))
buf = torch.zeros(4 * 1024 * 16, dtype=torch.int64, device=dev)
_lib.check(_lib.load().gms_debug_set_stamps(s.grid_map._h, C.c_void_p(buf.data_ptr())))
for i in range(40):
    s.update_dev(scans[i % T].data_ptr(), B, odo[i % T], seed=11, sequence=i)
    if i % 2 == 1: s.resample(0.3 + 0.01 * i)
torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(4, 1024, 16).astype(np.float64)[3]
st[st == 0] = np.nan
n = min(N, 1024)
t0 = np.nanmin(st[:n, 0])
names = {0: "entered", 1: "pose + trig (wavefront 0)", 2: "field staged / horizontal sums (last wavefront)", 10: "past the first barrier", 11: "neighbours' rows read, second barrier", 12: "column march done (wavefront 0)",
         3: "field complete", 4: "tables built (last wavefront)",
         5: "past the tables' barrier", 6: "look-ups done (wavefront 0)", 7: "look-ups done (last wavefront)", 8: "look-ups done (wavefront 2)", 9: "left"}
print(f"k_slam_refine, {N} particles x {s.W}x{s.H}, {B} beams: microseconds after the first workgroup entered")
for k, nm in names.items():
    v = (st[:n, k] - t0) * 0.01
    v = v[~np.isnan(v)]
    if v.size: print(f"  {nm:34s} n={v.size:4d} first {v.min():7.2f} median {np.median(v):7.2f} last {v.max():7.2f}")
d = (st[:n] - st[:n, 0:1]) * 0.01
ORDER = (1, 2, 10, 11, 12, 3, 4, 5, 6, 7, 8, 9)
print("  per workgroup, microseconds after ITS OWN entry (median):", " | ".join(f"{names[k]} {np.nanmedian(d[:, k]):.2f}" for k in ORDER if np.isfinite(d[:, k]).any()))
print("  ids < 256:", " | ".join(f"{np.nanmedian(d[:256, k]):.2f}" for k in ORDER if np.isfinite(d[:, k]).any()), " ids >= 256:", " | ".join(f"{np.nanmedian(d[256:n, k]):.2f}" for k in ORDER if np.isfinite(d[:, k]).any()))
