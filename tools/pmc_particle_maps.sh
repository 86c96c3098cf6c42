#!/bin/bash
# Counters of the per-particle-map kernels (bench.py --particle-maps 500,6,90), one rocprofv3 --pmc pass per counter group (kernel
# trace only), summarised per k_slam_* dispatch into gpurun_out/pmc_pm/summary.json.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_pm"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
K="python3 $ROOT/bench.py --particle-maps ${1:-500,6,90} --steps 30 --no-cpu-baseline --report $OUT/report.json"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_INSTS_VALU_INT32" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p$i" -- $K > /dev/null 2> "$OUT/p$i.stderr"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $K > "$OUT/kbench.stdout" 2> "$OUT/stats.stderr"
python3 - "$OUT" <<'PY'
import csv, glob, json, os, re, sys
from collections import defaultdict
out = sys.argv[1]
res = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if "k_slam" in name:
            res[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in d.items()} for k, d in res.items()}
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_slam" in r["Name"]:
            summ.setdefault(re.sub(r"\(.*", "", r["Name"]).replace("void ", ""), {})["kernel_trace"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
json.dump(summ, open(os.path.join(out, "summary.json"), "w"), indent=1)
for k, d in summ.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   ", c, v)
PY
