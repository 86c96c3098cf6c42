#!/usr/bin/env python3
"""bench.py -- particle-scan evals/s of the SLAM hot path on MI355X (BASELINE.json metric).

One "step" = one LIDAR scan through the whole path, inputs already resident in HBM:
  set poses (motion-model samples are an input) -> score N particles against the likelihood field
  -> normalise / Neff / weighted pose -> resample if neff < N/2 -> ray-cast the scan into the
  log-odds map at the weighted pose -> rebuild the likelihood field where it changed.

Workload at 1 GPU = BASELINE.json configs[2] ("C3"): 16384 particles, 720 beams, 2048x2048 grid @ 2 cm,
the configuration the metric is quoted on.  At N GPUs the particles are sharded, 16384 per GPU (weak
scaling; configs[3] is the 4-GPU point of that series), with an RCCL all-reduce for the weight
normaliser and an all-gather for resampling; every rank keeps a replica of the map.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  value = total particles x steps / max-over-ranks wall time of the K
steps.  roofline = the dominant kernel's algorithmic bytes per launch / its average launch duration
(HIP events on the library's stream, recorded inside the timed region).  cpu_baseline = the C oracle
(a port of the reference's Java loops; the reference itself cannot run here) on a bounded sample of
the same workload, one host thread.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(kind: str, *, n_particles=0, n_hit=0, n_beams=0, cells=0, visits=0) -> float:
    """SURVEY.md section 8(d) per-unit figures x the units one launch processes."""
    if kind == "score":        # 8 B per beam-eval + (12 B pose + 8 B weight) per particle + 17 B per beam once
        return 8.0 * n_particles * n_hit + 20.0 * n_particles + 17.0 * n_beams
    if kind == "likelihood":   # 16 B per cell (read log-odds, write likelihood)
        return 16.0 * cells
    if kind == "raycast":      # 16 B per visited cell (fp64 read + write)
        return 16.0 * visits
    if kind == "reduce":       # 16 B per particle
        return 16.0 * n_particles
    if kind == "resample":     # 8 B weight + 12 B pose read + 12 B pose write
        return 32.0 * n_particles
    if kind == "apply":
        return 16.0 * visits
    return 0.0


def pmc_traffic(kernel_class: str):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/*/pmc_traffic.json: separate `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs of this
    command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); None when not collected."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json"))):
        try:
            d = json.load(open(f))
            if kernel_class in d:
                best = d[kernel_class].get("hbm_bytes_per_launch")
        except Exception:
            pass
    return best


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3", help="C2 | C3 (per-GPU particles/beams/grid)")
    ap.add_argument("--particles", type=int, default=0, help="override particles per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-scans", type=int, default=0, help="scans of the CPU baseline sample (0 = auto)")
    ap.add_argument("--full-rebuild", action="store_true", help="rebuild the whole likelihood field every scan")
    ap.add_argument("--host-inputs", action="store_true", help="hand poses and scans over as HOST buffers every step (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--event-stride", type=int, default=8, help="timed region: HIP events around every n-th launch of the dominant kernel (a bracket costs microseconds of stream time)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "in-library", "torch-single", "torch-two"],
                    help="sharded runs: first exchange route to try (auto = in-library RCCL; later routes are fall-backs)")
    ap.add_argument("--torch-collectives", action="store_true", help="sharded runs: exchange through torch.distributed instead of the library's own RCCL communicator")
    ap.add_argument("--force-sharded", action="store_true", help="run the sharded (all-reduce / all-gather) code path even with one rank")
    args = ap.parse_args()

    # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner on C stdout when a
    # communicator is created, so everything but the result goes to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus N > 1 must be launched with torch.distributed.run", file=sys.stderr)
            return 2
    torch.cuda.set_device(local_rank)
    if world > 1 or (args.force_sharded and "RANK" in os.environ):
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from gridmap_slam_robot_amd import GridMap, ParticleFilter, _lib, synth
    from gridmap_slam_robot_amd.distributed import HipShardOps, RcclComm, ShardedParticleFilter

    cfg = dict(synth.CONFIGS[args.config])
    n_local = args.particles or cfg["particles"]
    n_global = n_local * world
    B, ext, res = cfg["beams"], cfg["extent"], cfg["resolution"]
    T = 64
    dev = torch.device("cuda", local_rank)

    # ---- inputs (identical on every rank: same seeds) ------------------------------------------------
    tr = synth.make_trace(ext, res, B, T=T, seed=1234)
    m = GridMap(ext, ext, res, (-ext / 2, -ext / 2), device=local_rank, max_beams=max(2048, B))
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    for t in range(T // 2):                      # pre-built map: likelihoods are informative
        m.update(tr.scans[t], tr.poses[t])
    m.synchronize()
    log0 = m.download_log().reshape(-1).copy() if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None

    n_sets = 8
    scans_dev = torch.from_numpy(tr.scans.view(np.uint8).reshape(T, -1).copy()).to(dev)
    pose_sets = []
    for s in range(n_sets):
        t = T // 2 + s
        allp = synth.make_particles(tr.poses[t], n_global, seed=99 + s)      # sigma 0.10 m / 5 deg
        pose_sets.append(torch.from_numpy(allp[rank * n_local:(rank + 1) * n_local].copy()).to(dev))
    pose_sets_host = [p.cpu().numpy() for p in pose_sets] if args.host_inputs else None
    n_hit = int(tr.scans[T // 2]["hit"].sum())
    r01 = np.random.default_rng(7).random(4096)

    if world > 1 or args.force_sharded:
        ops = HipShardOps(m, n_local, rank * n_local, n_global)
        spf = ShardedParticleFilter(n_global, ops)
        pf = ops.pf
        # default: both exchanges enqueued by the library on its own RCCL communicator (one C-ABI call per scan);
        # --torch-collectives routes them through torch.distributed instead
        comm = None
        if not args.torch_collectives:
            ok = 1
            try:
                comm = RcclComm(local_rank)
            except Exception as e:            # keep the run alive: the torch.distributed exchange does the same job
                print(f"bench.py: in-library RCCL communicator unavailable ({e}); using torch.distributed", file=sys.stderr)
                ok = 0
            if world > 1:                     # every rank must take the same path
                t_ok = torch.tensor([ok], dtype=torch.int32, device=dev)
                dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
                if int(t_ok.item()) == 0 and comm is not None:
                    comm.close()
                    comm = None
    else:
        pf = ParticleFilter(m, n_local)
        spf = comm = None

    two_collectives = [False]

    def step(i: int):
        s = i % n_sets
        t = T // 2 + s
        beams_ptr = scans_dev[t].data_ptr()
        if args.host_inputs and spf is None:
            pf.slam_update(pose_sets_host[s], tr.scans[t], r01[i % 4096], 0.5, True)
            return
        if spf is None and not args.full_rebuild:
            pf.slam_update_dev(pose_sets[s].data_ptr(), beams_ptr, B, r01[i % 4096], 0.5, True)   # one C-ABI call per scan
            return
        if comm is not None:
            pf.slam_update_sharded_dev(comm, pose_sets[s].data_ptr(), beams_ptr, B, r01[i % 4096], 0.5, True)
            return
        if spf is not None and not two_collectives[0]:
            # the same single-exchange step with the two all-gathers issued through torch.distributed
            spf.scan_step((pose_sets[s].data_ptr(), beams_ptr, B, True), r01[i % 4096], 0.5)
            return
        if spf is not None:
            # last resort: all-reduce of the partials, all-gather of the normalised particles (distributed.py)
            pf.set_poses_dev(pose_sets[s].data_ptr())
            pf.score_dev(beams_ptr, B)
            spf.normalize_begin()
            m.update_at_dev(beams_ptr, B, pf)
            spf.normalize_end()
            spf.resample(r01[i % 4096], 0.5)
            return
        # --full-rebuild: the separate entry points, likelihood field rebuilt everywhere as the reference does
        pf.set_poses_dev(pose_sets[s].data_ptr())
        pf.score_dev(beams_ptr, B)
        pf.normalize(fetch=False)
        pf.resample_if(r01[i % 4096], 0.5)
        if args.full_rebuild:
            m.integrate_at_dev(beams_ptr, B, pf)
            m.compute_likelihood_map()
        else:
            m.update_at_dev(beams_ptr, B, pf)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    bracket_us = m.profile_calibrate(200) * 1e3

    # ---- one untimed step first: if an exchange route cannot run here, every rank falls back together -----------
    # in-library RCCL (one grouped all-gather) -> torch.distributed scan_step (same protocol) -> the two-collective
    # protocol over torch.distributed
    def agree(ok: int) -> int:
        if world > 1:
            t_ok = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
            return int(t_ok.item())
        return ok

    if spf is not None:
        routes = ["in-library", "torch-single", "torch-two"]
        if args.exchange != "auto":
            routes = routes[routes.index(args.exchange):]
        for route in routes:
            if route == "in-library" and comm is None:
                continue
            if route == "torch-single":
                comm = None
            if route == "torch-two":
                comm, two_collectives[0] = None, True
            ok = 1
            try:
                step(0)
                torch.cuda.synchronize()
            except Exception as e:
                print(f"bench.py: sharded step via {route} failed ({e})", file=sys.stderr)
                ok = 0
            if agree(ok):
                break
        else:
            raise RuntimeError("no exchange route works on this node")

    # ---- warmup, with every kernel class bracketed: find the dominant one -----------------------------
    m.profile(True)
    m.profile_reset()
    for i in range(args.warmup):
        step(i)
    barrier()
    warm = m.profile_get()
    m.profile(False)
    if args.warmup > 0:
        dominant = max((k for k in warm if warm[k][1] > 0), key=lambda k: warm[k][0], default="score")
    else:
        dominant = "score"
    dom_bit = 1 << _lib.KERNEL_NAMES.index(dominant)

    # ---- timed region: exactly K steps, only the dominant kernel bracketed by events -------------------
    m.profile_reset()
    m.profile_sample(max(1, args.event_stride))
    m.profile(dom_bit)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    issue = time.perf_counter() - t0          # host time to enqueue K steps (the GPU must not be waiting on it)
    barrier()
    elapsed = time.perf_counter() - t0
    dom_ms, dom_n = m.profile_get()[dominant]
    m.profile(False)
    m.profile_sample(1)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- per-kernel breakdown (second pass, all classes bracketed; informational) ----------------------
    m.profile(True)
    m.profile_reset()
    nb = min(args.steps, 50)
    for i in range(nb):
        step(args.warmup + args.steps + i)
    barrier()
    prof = m.profile_get()
    m.profile(False)
    st = pf.stats()
    visits = None
    if rank == 0:
        cells, cls, counts = m.trace_scan(tr.scans[T // 2], tr.poses[T // 2])
        visits = int(counts.sum())

    if rank != 0:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return 0

    steps = args.steps
    value = n_global * steps / elapsed
    dom_avg_s = (dom_ms / max(dom_n, 1)) * 1e-3
    launches_per_step = {k: (prof[k][1] / nb if nb else 0) for k in prof}
    alg = algorithmic_bytes(dominant, n_particles=n_local, n_hit=n_hit, n_beams=B, cells=m.W * m.H, visits=visits or 0)
    # a class may launch several kernels per scan (reduce, resample): the figure is per scan step
    per_launch_scale = max(1.0, launches_per_step.get(dominant, 1.0))
    achieved = alg / per_launch_scale / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
    kernels = {}
    for k, (ms, n) in prof.items():
        if n:
            ab = algorithmic_bytes(k, n_particles=n_local, n_hit=n_hit, n_beams=B, cells=m.W * m.H, visits=visits or 0)
            if k == "likelihood" and not args.full_rebuild:
                ab = None        # the dirty-rect rebuild touches a scan-dependent part of the 16 B/cell field
            kernels[k] = {"ms_per_step": round(ms / nb, 5), "launches_per_step": round(n / nb, 2),
                          "algorithmic_gb_per_s": (round(ab / (ms / nb * 1e-3) / 1e9, 1) if ab else None)}
    # BASELINE metric (ii), map-update ms/scan = integrateObservation + computeLikelihoodMap: the map entry point by
    # itself (ray cast, apply, likelihood rebuild as three kernels at a fixed device-resident pose), wall time
    upd_pose = torch.from_numpy(np.ascontiguousarray(tr.poses[T // 2], dtype=np.float32)).to(dev)
    def map_update(i):
        bp = scans_dev[T // 2 + i % n_sets].data_ptr()
        if args.full_rebuild:
            m.integrate_dev(bp, B, upd_pose.data_ptr()); m.compute_likelihood_map()
        else:
            m.update_dev(bp, B, upd_pose.data_ptr())
    for i in range(5):
        map_update(i)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(nb):
        map_update(i)
    torch.cuda.synchronize()
    map_update_ms = (time.perf_counter() - t1) / nb * 1e3

    out = {
        "metric": "particle-scan evals/sec",
        "value": value,
        "unit": "particle-scan evals/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "host_issue_ms_per_step": issue / steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.config}: {n_local} particles/GPU x {B} beams ({n_hit} hits), {m.W}x{m.H} grid @ {res} m, "
                        f"full scan step (score+normalise+resample+ray-cast+likelihood rebuild)",
            "particles_total": n_global, "beams": B, "grid": [m.W, m.H], "resolution_m": res,
            "parallelism": f"particles sharded x{world}, map replicated" if world > 1 else "single GPU",
            "exchange": None if spf is None else (("torch.distributed, all-reduce + all-gather" if two_collectives[0] else "torch.distributed (RCCL), one exchange") if comm is None else "in-library RCCL: one grouped all-gather per scan (raw weights + block partials)"),
            "likelihood_rebuild": "full" if args.full_rebuild else "dirty-rect (bit-identical to full)",
            "inputs": "host buffers every step (PCIe-inclusive)" if args.host_inputs else "resident in HBM",
        },
        "beam_evals_per_s": value * n_hit,
        "scans_per_s": steps / elapsed,
        "map_update_ms_per_scan": map_update_ms,
        "kernels": kernels,
        "filter": {"neff": st["neff"], "n_zero_weights": st["n_zero"], "weight_sum": st["weight_sum"]},
        "roofline": {
            "kernel": dominant, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            # the committed PMC passes are of the default configuration: no figure for other modes
            "traffic": pmc_traffic(dominant) if (args.config == "C3" and not args.full_rebuild and not args.particles) else None,
            "algorithmic_bytes_per_launch": alg / per_launch_scale,
            "avg_launch_us": dom_avg_s * 1e6 / per_launch_scale, "launches_timed": dom_n, "event_stride": max(1, args.event_stride),
            # the same bracket around an empty kernel (dispatch latency + ~1 us): a kernel-trace profiler's duration
            # for the dominant kernel is about avg_launch_us minus this (profiles/ holds that trace)
            "event_bracket_empty_kernel_us": bracket_us,
        },
    }

    # ---- CPU baseline: the oracle (port of the Java loops), one thread, bounded sample ------------------
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
        log = log0
        lik = g.build_likelihood(log)
        n_cpu = n_local
        scans_cpu = args.cpu_scans or 10 ** 6          # bounded by time: ~12 s of single-thread CPU work
        P = synth.make_particles(tr.poses[T // 2], n_global, seed=99)[:n_cpu]
        c0 = time.perf_counter()
        done = 0
        for i in range(scans_cpu):
            t = T // 2 + (i % n_sets)
            w = g.score(lik, tr.scans[t], P)
            ws, strongest = orc.normalize(w)
            ne = orc.neff(w) if ws > 0 else float("nan")
            if ws > 0 and ne < n_cpu / 2:
                orc.resample_indices(w, float(r01[i]))
            wp = orc.weighted_pose(P, w) if ws > 0 else tr.poses[t]
            g.integrate(log, tr.scans[t], wp)
            lik = g.build_likelihood(log)
            done += 1
            if time.perf_counter() - c0 > (30.0 if args.cpu_scans else 12.0):
                break
        cpu_s = time.perf_counter() - c0
        # extra information: scoring alone (a pure function) over host threads with OpenMP; the thread count that does
        # best is reported (containers often expose more CPUs than they may use)
        try:
            navail = len(os.sched_getaffinity(0))
        except AttributeError:
            navail = os.cpu_count() or 1
        w_mt = g.score_mt(lik, tr.scans[T // 2], P, min(navail, 8))
        mt_s, ncore = float("inf"), 1
        th = 2
        while th <= navail:
            m0 = time.perf_counter()
            reps_mt = 0
            while time.perf_counter() - m0 < 0.4:
                g.score_mt(lik, tr.scans[T // 2 + reps_mt % n_sets], P, th)
                reps_mt += 1
            per = (time.perf_counter() - m0) / reps_mt
            if per < mt_s:
                mt_s, ncore = per, th
            th *= 2
        st_s = float("inf")
        for _ in range(3):
            s0 = time.perf_counter()
            w_st = g.score(lik, tr.scans[T // 2], P)
            st_s = min(st_s, time.perf_counter() - s0)
        assert np.array_equal(w_mt, w_st)
        out["cpu_baseline"] = {
            "value": n_cpu * done / cpu_s, "unit": "particle-scan evals/s", "cores": 1, "kind": "port",
            "sample": f"{done} scan steps of the same trace, all {n_cpu} particles "
                      f"(C oracle, gcc -O2 -ffp-contract=off, single thread; full likelihood rebuild per scan as the reference does)",
            "seconds": cpu_s,
            "score_only": {"single_thread_particle_evals_per_s": n_cpu / st_s, "openmp_particle_evals_per_s": n_cpu / mt_s,
                           "openmp_threads": ncore, "cpus_visible": navail,
                           "note": "probabilityOf over all particles only (no map update, no likelihood rebuild); the GPU scoring "
                                   "kernel alone does particles / kernels.score.ms_per_step"},
        }
    else:
        out["cpu_baseline"] = None

    os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
