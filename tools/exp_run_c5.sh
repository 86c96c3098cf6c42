#!/bin/bash
# like exp_run2.sh for another configuration: bash tools/exp_run_c5.sh "<names>" [rounds] [config]
cd "$(dirname "$0")/.."
L=$PWD/gridmap_slam_robot_amd/lib
NAMES=${1:-prod}; ROUNDS=${2:-2}; CFG=${3:-C5}
for r in $(seq 1 $ROUNDS); do
  for spec in $NAMES; do
    n=${spec%%:*}; e=""; [ "$spec" != "$n" ] && e=${spec#*:}
    lib=$L/exp_$n.so; [ "$n" = "prod" ] && lib=$L/libgridmapslam.so
    b=$(env $e GMS_LIBRARY=$lib python bench.py --config $CFG --steps 60 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step']*1e3,2), 'us/step', {k:v.get('avg_launch_us') for k,v in d['kernels'].items()})")
    echo "$CFG $spec | $b"
  done
done
