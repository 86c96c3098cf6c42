#!/usr/bin/env python3
"""Per-segment look-up time of the scoring kernel, scan by scan (development tool; needs the instrumented build, see
tools/stamps.py): for each of the bench's scans the C3 step is run and the scoring workgroups' stamps (prologue done -> products
stored) are reduced to one median per beam segment.  Written as JSON: what a cost model for balanced segments is fitted to.

  GMS_LIBRARY=$PWD/gridmap_slam_robot_amd/lib/exp_stamps.so python tools/seg_times.py > gpurun_out/seg_times.json"""
import argparse, ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import bench
    from gridmap_slam_robot_amd import _lib
    args = argparse.Namespace(particles=0, exchange="auto", host_inputs=False, full_rebuild=False)
    wl = bench.Workload("C3", args, torch, dist, 0, 1, 0, False)
    buf = torch.zeros(4 * 1024 * 16, dtype=torch.int64, device=wl.dev)
    _lib.check(_lib.load().gms_debug_set_stamps(wl.m._h, C.c_void_p(buf.data_ptr())))
    out = []
    for i in range(16):
        wl.step(i)
    for rep in range(3):
        for s in range(wl.n_sets):
            for k in range(3):                      # the same scan three times: the last one's stamps are read
                wl.step(s)
            torch.cuda.synchronize()
            st = buf.cpu().numpy().reshape(4, 1024, 16).astype(np.float64)[0]
            n_wg = int((st[:, 2] > 0).sum())
            dur = (st[:n_wg, 2] - st[:n_wg, 1]) * 0.01
            tot = (st[:n_wg, 2].max() - st[:n_wg, 0].min()) * 0.01
            i = np.arange(n_wg)
            nseg = 16
            spx = nseg >> 3
            seg = (i & 7) * spx + (i >> 3) % spx
            t = wl.T // 2 + s
            scan = wl.tr.scans[t]
            rec = {"scan": int(t), "rep": rep, "launch_us": float(tot), "segments": []}
            for sg in range(nseg):
                d = dur[seg == sg]
                b = scan[sg * 45:(sg + 1) * 45]
                b = b[b["hit"] != 0]
                jumps = np.hypot(np.diff(b["local_x"]), np.diff(b["local_y"])) / wl.res if len(b) > 1 else np.zeros(0)
                rec["segments"].append({"seg": sg, "median_us": float(np.median(d)), "max_us": float(d.max()), "hits": int(len(b)),
                                        "jumps_cells": [round(float(x), 1) for x in jumps], "mean_range_m": float(b["distance"].mean()) if len(b) else 0.0})
            out.append(rec)
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
