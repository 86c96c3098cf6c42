#!/bin/bash
# k_score_c + k_partials + the step at 16 (45-beam) and 8 (90-beam) scoring segments of a 720-beam scan, at 16 384 particles (C3)
# and at config 4's 8192-particle shard: lib/seg16.so = the product build, lib/seg8.so = GMS_EXTRA_FLAGS=-DGMS_SCORE_SEGLEN=90.
cd "$(dirname "$0")/.."
L=$PWD/gridmap_slam_robot_amd/lib
for r in 1 2 3; do
  for v in seg16 seg8; do
    for cfg in "C3:--config C3" "C4_8192:--config C4 --particles 8192"; do
      name=${cfg%%:*}; args=${cfg#*:}
      GMS_LIBRARY=$L/$v.so python bench.py $args --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --report /tmp/nseg_rep.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernel_us']
print('$v $name step %.2f us  k_score_c %.2f  k_partials %.2f  k_norm_raycast %.2f  k_lik_resample %.2f' % (d['ms_per_step']*1e3, k.get('k_score_c',0), k.get('k_partials',0), k.get('k_norm_raycast',0), k.get('k_lik_resample',0)), flush=True)"
    done
  done
done
