#!/bin/bash
# Counters of k_slam_refine (tools/pm_refine_probe.py, first case), one rocprofv3 --pmc pass per counter group (kernel trace only),
# summarised into gpurun_out/pmc_refine/summary.json.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_refine"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PM_CASES=${PM_CASES:-1}
K="python3 $ROOT/tools/pm_refine_probe.py"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/p$i" -- $K > /dev/null 2> "$OUT/p$i.stderr"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $K > "$OUT/probe.stdout" 2> "$OUT/stats.stderr"
python3 "$ROOT/tools/pmc_kernels.py" "$OUT/summary.json" "${PMC_FILTER:-k_slam_refine}" "$OUT"/p* "$OUT/stats" > /dev/null
python3 - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, e in d.items():
    print(k)
    for c, v in sorted(e.items()):
        print("   ", c, v if not isinstance(v, dict) or "mean" not in v else round(v["mean"], 1))
PY
