#!/usr/bin/env python3
"""Stage timeline of k_slam_particle (per-particle-map mode) from an instrumented build (development tool).
  GMS_EXTRA_FLAGS=-DGMS_STAMPS python -m gridmap_slam_robot_amd.build --force; cp lib/libgridmapslam.so lib/exp_stamps.so; rebuild
  GMS_LIBRARY=$PWD/gridmap_slam_robot_amd/lib/exp_stamps.so python tools/pm_stamps.py [N EXT B]"""
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
scans, odo = [], []
for f in frames:
    obs = s.grid_map.deskew(f.angle, f.distance, f.hit, f.d_center, f.d_theta)
    scans.append(torch.from_numpy(obs.beams.view(np.uint8).reshape(-1).copy()).to(dev)); odo.append((f.d_center, f.d_theta))
buf = torch.zeros(4 * 1024 * 16, dtype=torch.int64, device=dev)
_lib.check(_lib.load().gms_debug_set_stamps(s.grid_map._h, C.c_void_p(buf.data_ptr())))
for i in range(20):
    s.update_dev(scans[i].data_ptr(), B, odo[i], seed=11, sequence=i)
    if i % 2 == 1: s.resample(0.3 + 0.01 * i)
torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(4, 1024, 16).astype(np.float64)[0]
st[st == 0] = np.nan
t0 = np.nanmin(st[:, 0])
names = {0: "entered", 1: "pose ready", 2: "factors + boxes ready", 3: "product chain done (its lane)", 4: "first group's rays set up", 5: "first round walked (producer)",
         8: "cell work begins (last wavefront: after the list)", 12: "cell work begins (first consumer wavefront)", 11: "cell work done (producer wavefront 0)",
         10: "cell work done (first consumer wavefront)", 9: "cell work done (last wavefront)", 6: "all rays counted", 7: "cells applied"}
print(f"k_slam_particle, {N} particles x {int(ext/0.05)}^2, {B} beams: microseconds after the first workgroup entered (workgroups 0..{min(N,1024)-1})")
for k, nm in names.items():
    v = (st[:min(N, 1024), k] - t0) * 0.01
    v = v[~np.isnan(v)]
    if v.size: print(f"  {nm:34s} n={v.size:4d} first {v.min():7.2f} median {np.median(v):7.2f} last {v.max():7.2f}")
d = (st[:min(N, 1024)] - st[:min(N, 1024), 0:1]) * 0.01
print("  per workgroup, microseconds after ITS OWN entry (median):", " | ".join(f"{names[k].split(' ')[0]} {np.nanmedian(d[:, k]):.2f}" for k in (1, 2, 3, 4, 5, 6, 7)))
# the slowest workgroups: which stage they lose their time in (microseconds spent per stage)
n = min(N, 1024)
tot = d[:n, 7]
order = np.argsort(-np.nan_to_num(tot))[:8]
print("  slowest workgroups (id: rays set up | walked | counted | applied, each after the previous stage):")
for i in order:
    print(f"    {i:4d}: {d[i, 4]:.1f} | {d[i, 5] - d[i, 4]:.1f} | {d[i, 6] - d[i, 5]:.1f} | {d[i, 7] - d[i, 6]:.1f}   total {tot[i]:.1f}")
med = np.nanmedian(d[:n], axis=0)
print(f"    median: {med[4]:.1f} | {med[5] - med[4]:.1f} | {med[6] - med[5]:.1f} | {med[7] - med[6]:.1f}   total {med[7]:.1f}")
for nm_, sl in (("ids < 256 ", slice(0, min(n, 256))), ("ids >= 256", slice(256, n))):
    if sl.stop > sl.start:
        m_ = np.nanmedian(d[sl], axis=0)
        print(f"  {nm_}: pose {m_[1]:.2f} | factors {m_[2]:.2f} | rays set up {m_[4]:.2f} | first round walked {m_[5]:.2f} | counted {m_[6]:.2f} | applied {m_[7]:.2f};"
              f" entered {np.nanmedian(st[sl, 0] - t0) * 0.01:.2f} after the first")
if np.isfinite(d[:n, 13]).any():
    print(f"  before the ray set-up's barrier (median, after own entry): log-weight wavefront done {np.nanmedian(d[:n, 13]):.2f} | producer 0 ready {np.nanmedian(d[:n, 14]):.2f} | wavefront 2 there {np.nanmedian(d[:n, 15]):.2f}")
print("  median total by XCD (id & 7):", " ".join(f"{np.nanmedian(tot[x::8]):.1f}" for x in range(8)), "| ids < 256:", f"{np.nanmedian(tot[:256]):.1f}", "ids >= 256:", f"{np.nanmedian(tot[256:n]):.1f}")
print("  max total by XCD:            ", " ".join(f"{np.nanmax(tot[x::8]):.1f}" for x in range(8)))
print("  workgroups slower than median + 3 us:", int((tot > med[7] + 3).sum()), "of", n, "; their ids mod 32:", sorted(set(int(i) % 32 for i in np.where(tot > med[7] + 3)[0])))
