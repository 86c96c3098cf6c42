#!/usr/bin/env python3
"""Host time of the first scan steps after an idle queue: bench.py's Workload (C3), `warmup` steps, a barrier, then the
wall-clock time at which every one of the next `steps` calls returns, and the time the queue drains.  Answers: why a 20-step
timed region reads slower per step than a 200-step one (host-bound start? idle-queue launches? completion latency?).
usage: issue_probe.py [--warmup 5] [--steps 40] [--rounds 4]"""
import argparse
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)

ap = argparse.ArgumentParser()
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--rounds", type=int, default=4)
a = ap.parse_args()

import torch
import torch.distributed as dist

args = argparse.Namespace(particles=0, exchange="auto", host_inputs=False, full_rebuild=False)
torch.cuda.set_device(0)
wl = b.Workload("C3", args, torch, dist, 0, 1, 0, False)
for rnd in range(a.rounds):
    for i in range(a.warmup):
        wl.step(i)
    wl.barrier()
    if rnd == a.rounds - 1:
        time.sleep(0.5)            # the last round starts from a queue that has been idle for a while
    stamps = []
    t0 = time.perf_counter()
    for i in range(a.steps):
        wl.step(a.warmup + i)
        stamps.append(time.perf_counter())
    wl.barrier()
    t_end = time.perf_counter()
    d = [(stamps[0] - t0) * 1e6] + [(stamps[i] - stamps[i - 1]) * 1e6 for i in range(1, len(stamps))]
    print(f"round {rnd}: issue us per step: " + " ".join(f"{x:.0f}" for x in d))
    print(f"          issued {(stamps[-1] - t0) * 1e6:.0f} us, drained {(t_end - t0) * 1e6:.0f} us = {(t_end - t0) * 1e6 / a.steps:.1f} us/step;"
          f" first 20: issued at {(stamps[min(19, len(stamps) - 1)] - t0) * 1e6:.0f} us", flush=True)
