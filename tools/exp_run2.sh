#!/bin/bash
# like exp_run.sh, bench.py only, with optional environment per run: names may be "name" or "name:ENV=VAL"
cd "$(dirname "$0")/.."
L=$PWD/gridmap_slam_robot_amd/lib
NAMES=${1:-base}; ROUNDS=${2:-2}
for r in $(seq 1 $ROUNDS); do
  for spec in $NAMES; do
    n=${spec%%:*}; e=""; [ "$spec" != "$n" ] && e=${spec#*:}
    lib=$L/exp_$n.so; [ "$n" = "prod" ] && lib=$L/libgridmapslam.so
    b=$(env $e GMS_LIBRARY=$lib python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step']*1e3,2), 'us/step', {k:v.get('avg_launch_us') for k,v in d['kernels'].items()}, 'map_update', round(d.get('map_update_ms_per_scan',0)*1e3,2))")
    echo "$spec | $b"
  done
done
