#!/usr/bin/env python3
"""Summarises rocprofv3 PMC passes per kernel and writes profiles/<round>/pmc_traffic.json.

Usage: pmc_summary.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <out dir> [prefix] [config]
With a config name (C3, C5, ...) the classes are merged into <out dir>/pmc_traffic.json under that key (bench.py reads
roofline.traffic per configuration from there); without one the file is the classes themselves (round-1 layout).
Each input dir holds a `*_counter_collection.csv` of `rocprofv3 --kernel-trace --pmc <COUNTER> -- python3 bench.py ...`
(separate passes, kernel trace only, as MI355X_MICROARCH.md prescribes).  FETCH_SIZE / WRITE_SIZE are in KB;
FETCH_SIZE is doubled for gfx950 (128-byte requests are tallied at 64 B).  Kernel classes are bench.py's."""
import csv, glob, json, os, re, sys
from collections import defaultdict

CLASS_OF = [("k_raycast_tile", "raycast"), ("k_raycast<false, 16>", "raycast"), ("k_score_c", "score"), ("k_norm_raycast", "raycast"), ("k_raycast_norm_chunks", "raycast"), ("k_raycast_apply", "raycast"),
            ("k_raycast<false", "raycast"), ("k_lik_resample", "likelihood"), ("k_likelihood", "likelihood"),
            ("k_partials_apply", "reduce"), ("k_partials", "reduce"), ("k_normalize_pack", "reduce"), ("k_apply", "apply"),
            ("k_chunk_sums", "resample"), ("k_resample", "resample"), ("k_pose_trig", "pose_trig"), ("k_order", "order")]


STEP_KERNELS = ("k_score_c", "k_order", "k_partials", "k_normalize_pack", "k_norm_raycast", "k_raycast_tile", "k_lik_resample",
                "k_raycast_norm_chunks")


def per_kernel(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        acc[name].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v), min(v), max(v)) for k, v in acc.items()}


def main():
    fetch, write, out = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), sys.argv[3]
    prefix = sys.argv[4] if len(sys.argv) > 4 else "pmc"
    config = sys.argv[5] if len(sys.argv) > 5 else None
    os.makedirs(out, exist_ok=True)
    for tag, d in (("fetch_size", fetch), ("write_size", write)):
        with open(os.path.join(out, f"{prefix}_{tag}_per_kernel.csv"), "w") as fh:
            fh.write("kernel,dispatches,mean_counter_kb,min,max\n")
            for k, (n, mean, lo, hi) in d.items():
                fh.write(f'"{k}",{n},{mean:.1f},{lo:.1f},{hi:.1f}\n')
    # Traffic is kept PER KERNEL.  A class's per-launch figure is taken over the kernels that the scan step itself launches
    # (STEP_KERNELS): kernels of the same class that belong to another loop of the run -- the stand-alone map update that bench.py
    # times for BASELINE's second metric launches k_raycast_apply and k_likelihood<..., true> -- are listed but not added in.
    # A class with two launches per step (batched maps: k_partials and k_normalize_pack are both "reduce") gets the mean of the
    # two, as bench.py divides the class's algorithmic bytes by its launches per step.
    classes = {}
    for name in set(fetch) | set(write):
        cls = next((c for pat, c in CLASS_OF if name.startswith(pat.rstrip("("))), None)
        if cls is None or name.startswith("k_raycast<true"):
            continue
        n = max(fetch.get(name, (0,))[0], write.get(name, (0,))[0])
        classes.setdefault(cls, []).append((n, name))
    res = {}
    for cls, members in classes.items():
        top = max(n for n, _ in members)
        kern = {}
        for n, name in members:
            fk, wk = fetch.get(name, (0, 0.0))[1], write.get(name, (0, 0.0))[1]
            kern[name] = {"fetch_kb": round(fk, 1), "write_kb": round(wk, 1), "dispatches": n,
                          "hbm_bytes": int((2.0 * fk + wk) * 1024), "in_scan_step": any(name.startswith(pfx) for pfx in STEP_KERNELS)}
        step = [k for k, e in kern.items() if e["in_scan_step"] and 2 * e["dispatches"] >= top]
        if not step:                         # a run that is not the scan step (e.g. the dense likelihood rebuild): its busiest kernel
            step = [max(kern, key=lambda k: kern[k]["dispatches"])]
        per_step = sum(kern[k]["hbm_bytes"] for k in step)
        res[cls] = {"kernels": kern, "step_kernels": sorted(step), "launches_per_step": len(step), "hbm_bytes_per_step": per_step,
                    "hbm_bytes_per_launch": int(per_step / len(step))}
    res["_note"] = ("per scan step and kernel class, bench.py; separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes with "
                    "--kernel-trace only; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM (gfx950 tallies 128-B requests at "
                    "64 B for wide reads; for the 8-byte gathers of k_score_c the factor is uncalibrated, so its read side is an "
                    "upper bound); Infinity-Cache hits are counted, not excluded.  Paired launches are booked under the class "
                    "bench.py books them under (k_norm_raycast: raycast, k_lik_resample: likelihood, k_partials_apply: reduce); "
                    "hbm_bytes_per_launch is over the scan step's own kernels (step_kernels), per kernel figures under kernels.")
    path = os.path.join(out, "pmc_traffic.json")
    if config:
        allc = {}
        if os.path.exists(path):
            try:
                allc = json.load(open(path))
            except Exception:
                allc = {}
        allc[config] = res
        json.dump(allc, open(path, "w"), indent=1)
    else:
        json.dump(res, open(path, "w"), indent=1)
    for cls, e in res.items():
        if not cls.startswith("_"):
            print(cls, e["hbm_bytes_per_launch"], list(e["kernels"]))


if __name__ == "__main__":
    main()
