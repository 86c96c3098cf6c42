#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 counter passes (and kernel-trace stats) found under the given directories.
usage: pmc_kernels.py <out.json> <name filter (substring, '' = all)> <dir> [<dir> ...]
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled here for gfx950 (128-byte requests are tallied at 64 B:
MI355X_MICROARCH.md, HBM) and reported as fetch_bytes / write_bytes per dispatch."""
import csv, glob, json, os, re, sys
from collections import defaultdict

out, filt, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
res = defaultdict(lambda: defaultdict(list))
trace = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            if filt in name:
                res[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
            if filt in name:
                trace[name] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3}
summ = {}
for k in sorted(set(res) | set(trace)):
    e = {c: {"mean": sum(v) / len(v), "dispatches": len(v)} for c, v in res.get(k, {}).items()}
    if "FETCH_SIZE" in e:
        e["fetch_bytes"] = e["FETCH_SIZE"]["mean"] * 1024 * 2
    if "WRITE_SIZE" in e:
        e["write_bytes"] = e["WRITE_SIZE"]["mean"] * 1024
    if k in trace:
        e["kernel_trace"] = trace[k]
        if "fetch_bytes" in e or "write_bytes" in e:
            e["fabric_TBps"] = (e.get("fetch_bytes", 0) + e.get("write_bytes", 0)) / (trace[k]["avg_us"] * 1e-6) / 1e12
    summ[k] = e
json.dump(summ, open(out, "w"), indent=1)
for k, e in summ.items():
    print(k, {c: (round(v, 3) if isinstance(v, float) else v) for c, v in e.items() if c in ("fetch_bytes", "write_bytes", "fabric_TBps", "kernel_trace")})
