#!/bin/bash
# Runs kbench / bench.py once per experimental build gridmap_slam_robot_amd/lib/exp_<name>.so on ONE GPU box (GMS_LIBRARY selects the
# library; the product library is not touched).  usage: bash tools/exp_run.sh "<names>" [rounds] [kbench --only list]
cd "$(dirname "$0")/.."
L=$PWD/gridmap_slam_robot_amd/lib
NAMES=${1:-base}; ROUNDS=${2:-2}; ONLY=${3:-raycast}
for r in $(seq 1 $ROUNDS); do
  for n in $NAMES; do
    k=$(GMS_LIBRARY=$L/exp_$n.so python tools/kbench.py --only $ONLY --iters 200 2>/dev/null | tail -1)
    b=$(GMS_LIBRARY=$L/exp_$n.so python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step']*1e3,2), 'us/step', {k:v.get('avg_launch_us') for k,v in d['kernels'].items()})")
    echo "$n | kbench $k | bench $b"
  done
done
