#!/usr/bin/env python3
"""Per-kernel cost of the opt-in log-normalisation in the C3 closed loop: bench.Workload(loop=True) with and without
gms_pf_set_log_normalize, measure()'s bracketed per-class times."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import bench
args = argparse.Namespace(particles=0, exchange="auto", host_inputs=False, full_rebuild=False)
for ln in (False, True, False, True):
    wl = bench.Workload("C3", args, torch, dist, 0, 1, 0, False, loop=True)
    if ln:
        wl.pf.set_log_normalize(True)
    m = bench.measure(wl, 200, 20)
    prof = m["prof"]
    print("lognorm" if ln else "plain  ", "step %.2f us" % (m["steady"] * 1e6), {k: round(v[0] / v[1] * 1e3, 2) for k, v in prof.items() if v[1]}, flush=True)
    wl.pf.close(); wl.m.close()
