#!/bin/bash
# Per-particle-map mode (SLAM.java's own shape) on the GPU box: bench.py --particle-maps at the two operating points, optionally
# under alternative settings.  usage: tools/pm_bench.sh <tag> [ENV=VALUE ...]   -> gpurun_out/pm/<tag>_{500,500b180,4096}.json
tag=$1; shift
mkdir -p gpurun_out/pm
for spec in "500,6,90:500:50" "500,6,180:500b180:50" "4096,12.8,180:4096:20"; do
  IFS=: read cfg name steps <<< "$spec"
  env "$@" python bench.py --particle-maps $cfg --steps $steps --no-cpu-baseline --report gpurun_out/pm/${tag}_${name}_report.json > gpurun_out/pm/${tag}_${name}.json 2> gpurun_out/pm/${tag}_${name}.err || tail -n 5 gpurun_out/pm/${tag}_${name}.err
done
