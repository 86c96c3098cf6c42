#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel-trace statistics and the separate FETCH_SIZE / WRITE_SIZE counter passes
# of bench.py (C3 default, C5) and of the dense-map likelihood rebuild; raw output under gpurun_out/prof/, the summaries
# that are kept go to profiles/<round>/ afterwards (tools/kstats.py, tools/pmc_summary.py).
# usage: collect_profiles.sh <round dir name, e.g. r02>
R=${1:-r06}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_$R"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
stats() {   # name, command...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- "$@" > "$OUT/$name.stdout" 2> "$OUT/$name.stderr"
  local f; f=$(find "$OUT/$name" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${name}_kernel_stats.csv"
}
pmc() {     # name, counter, command...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --kernel-trace --pmc "$ctr" --output-format csv -d "$OUT/${name}_$ctr" -- "$@" > /dev/null 2> "$OUT/${name}_$ctr.stderr"
}
B="python3 $ROOT/bench.py --no-cpu-baseline --no-secondary"
stats c3_bench $B --steps 200 --warmup 20
pmc c3 FETCH_SIZE $B --steps 50 --warmup 5
pmc c3 WRITE_SIZE $B --steps 50 --warmup 5
stats c5_bench $B --config C5 --steps 20 --warmup 3
pmc c5 FETCH_SIZE $B --config C5 --steps 10 --warmup 2
pmc c5 WRITE_SIZE $B --config C5 --steps 10 --warmup 2
K="python3 $ROOT/tools/kbench.py --only likelihood --dense --iters 50"
stats dense_likelihood $K
pmc dense FETCH_SIZE $K
pmc dense WRITE_SIZE $K
stats c3_full_rebuild $B --full-rebuild --steps 100 --warmup 10
# the reference's own filter shape (one GridMapData per particle): kernel trace at both sizes, fabric traffic at the size that leaves the caches
PM="python3 $ROOT/bench.py --no-cpu-baseline"
stats pm500 $PM --particle-maps 500,6,90 --steps 50 --report "$OUT/pm500_report.json"
stats pm4096 $PM --particle-maps 4096,12.8,180 --steps 10 --report "$OUT/pm4096_report.json"
pmc pm4096 FETCH_SIZE $PM --particle-maps 4096,12.8,180 --steps 6 --report "$OUT/pm4096_pmc_report.json"
pmc pm4096 WRITE_SIZE $PM --particle-maps 4096,12.8,180 --steps 6 --report "$OUT/pm4096_pmc_report.json"
# round 6: the small size's fabric traffic with likelihoodData on demand and with every cell rebuilt (the like-for-like form), the pose
# refinement, the reference-shape filter through the sharded route on one rank
pmc pm500 FETCH_SIZE $PM --particle-maps 500,6,90 --steps 30 --report "$OUT/pm500_pmc_report.json"
pmc pm500 WRITE_SIZE $PM --particle-maps 500,6,90 --steps 30 --report "$OUT/pm500_pmc_report.json"
GMS_SLAM_EAGER_LIK=1 pmc pm500eager FETCH_SIZE $PM --particle-maps 500,6,90 --steps 30 --report "$OUT/pm500e_pmc_report.json"
GMS_SLAM_EAGER_LIK=1 pmc pm500eager WRITE_SIZE $PM --particle-maps 500,6,90 --steps 30 --report "$OUT/pm500e_pmc_report.json"
GMS_SLAM_EAGER_LIK=1 pmc pm4096eager FETCH_SIZE $PM --particle-maps 4096,12.8,180 --steps 6 --report "$OUT/pm4096e_pmc_report.json"
GMS_SLAM_EAGER_LIK=1 pmc pm4096eager WRITE_SIZE $PM --particle-maps 4096,12.8,180 --steps 6 --report "$OUT/pm4096e_pmc_report.json"
stats pm500_refine $PM --particle-maps 500,6,90 --refine --steps 50 --report "$OUT/pm500_refine_report.json"
stats trace_replay python3 $ROOT/bench.py --trace $ROOT/tests/golden/recording_360.bin --steps 300 --warmup 20
cd "$ROOT"
mkdir -p "$OUT/keep"
python3 $ROOT/tools/pmc_kernels.py "$OUT/keep/per_particle_maps_kernels.json" k_slam "$OUT/pm4096_FETCH_SIZE" "$OUT/pm4096_WRITE_SIZE" "$OUT/pm4096" > "$OUT/keep/per_particle_maps_kernels.txt" 2>&1
python3 $ROOT/tools/pmc_kernels.py "$OUT/keep/per_particle_maps_500_traffic.json" k_slam "$OUT/pm500_FETCH_SIZE" "$OUT/pm500_WRITE_SIZE" "$OUT/pm500" > "$OUT/keep/per_particle_maps_500_traffic.txt" 2>&1
python3 $ROOT/tools/pmc_kernels.py "$OUT/keep/per_particle_maps_500_every_cell_traffic.json" k_slam "$OUT/pm500eager_FETCH_SIZE" "$OUT/pm500eager_WRITE_SIZE" > "$OUT/keep/per_particle_maps_500_every_cell_traffic.txt" 2>&1
python3 $ROOT/tools/pmc_kernels.py "$OUT/keep/per_particle_maps_4096_every_cell_traffic.json" k_slam "$OUT/pm4096eager_FETCH_SIZE" "$OUT/pm4096eager_WRITE_SIZE" > "$OUT/keep/per_particle_maps_4096_every_cell_traffic.txt" 2>&1
for n in c3_bench c5_bench dense_likelihood c3_full_rebuild trace_replay pm500 pm4096 pm500_refine; do
  cp "$OUT/${n}_kernel_stats.csv" "$OUT/keep/" 2>/dev/null
  cp "$OUT/$n.stdout" "$OUT/keep/${n}_under_rocprof.json" 2>/dev/null
done
python3 tools/pmc_summary.py "$OUT/c3_FETCH_SIZE" "$OUT/c3_WRITE_SIZE" "$OUT/keep" c3_pmc C3
python3 tools/pmc_summary.py "$OUT/c5_FETCH_SIZE" "$OUT/c5_WRITE_SIZE" "$OUT/keep" c5_pmc C5
python3 tools/pmc_summary.py "$OUT/dense_FETCH_SIZE" "$OUT/dense_WRITE_SIZE" "$OUT/keep" dense_pmc dense_likelihood
# the microbenchmarks behind the ceilings bench.py quotes and behind DESIGN.md's launch-structure decision
python3 tools/microbench/run_all.py > "$OUT/microbench.stdout" 2> "$OUT/microbench.stderr"
cp gpurun_out/microbench/microbench.json "$OUT/keep/" 2>/dev/null
cp gpurun_out/microbench/persistent_step.txt "$OUT/keep/persistent_step_model_run.txt" 2>/dev/null
# the bench lines below quote this collection's own PMC traffic and look-up ceilings: install them where bench.py reads them
mkdir -p "profiles/$R"
cp "$OUT/keep/pmc_traffic.json" "$OUT/keep/microbench.json" "profiles/$R/" 2>/dev/null
# un-profiled bench lines (stdout = the one compact line; --report = the full report)
python3 bench.py --report "$OUT/keep/bench_report.json" > "$OUT/keep/bench.json" 2> "$OUT/bench.stderr"
python3 bench.py --steps 20 --warmup 5 --report "$OUT/keep/bench_steps20_report.json" > "$OUT/keep/bench_steps20.json" 2>> "$OUT/bench.stderr"
Q="--no-cpu-baseline --no-secondary"
python3 bench.py --force-sharded $Q --report "$OUT/keep/bench_sharded_one_rank_report.json" > "$OUT/keep/bench_sharded_one_rank.json" 2>> "$OUT/bench.stderr"
python3 bench.py --config C4 --force-sharded $Q --steps 50 --warmup 5 --report "$OUT/keep/bench_c4_sharded_one_rank_report.json" > "$OUT/keep/bench_c4_sharded_one_rank.json" 2>> "$OUT/bench.stderr"
python3 bench.py --host-inputs $Q --report "$OUT/keep/bench_host_inputs_report.json" > "$OUT/keep/bench_host_inputs.json" 2>> "$OUT/bench.stderr"
python3 bench.py --full-rebuild $Q --report "$OUT/keep/bench_full_rebuild_report.json" > "$OUT/keep/bench_full_rebuild.json" 2>> "$OUT/bench.stderr"
python3 bench.py --config C5 --steps 50 --warmup 5 --report "$OUT/keep/bench_c5_report.json" > "$OUT/keep/bench_c5.json" 2>> "$OUT/bench.stderr"
python3 bench.py --trace tests/golden/recording_360.bin --steps 300 --warmup 20 --report "$OUT/keep/bench_trace_replay_report.json" > "$OUT/keep/bench_trace_replay.json" 2>> "$OUT/bench.stderr"
python3 bench.py --particle-maps 500,6,90 --steps 50 --report "$OUT/keep/bench_particle_maps_500_report.json" > "$OUT/keep/bench_particle_maps_500.json" 2>> "$OUT/bench.stderr"
python3 bench.py --particle-maps 500,6,180 --steps 50 --no-cpu-baseline --report "$OUT/keep/bench_particle_maps_500_b180_report.json" > "$OUT/keep/bench_particle_maps_500_b180.json" 2>> "$OUT/bench.stderr"
python3 bench.py --particle-maps 4096,12.8,180 --steps 20 --no-cpu-baseline --report "$OUT/keep/bench_particle_maps_4096_report.json" > "$OUT/keep/bench_particle_maps_4096.json" 2>> "$OUT/bench.stderr"
python3 bench.py --particle-maps 500,6,90 --refine --steps 50 --report "$OUT/keep/bench_particle_maps_500_refine_report.json" > "$OUT/keep/bench_particle_maps_500_refine.json" 2>> "$OUT/bench.stderr"
python3 bench.py --particle-maps 500,6,180 --refine --steps 50 --report "$OUT/keep/bench_particle_maps_500_b180_refine_report.json" > "$OUT/keep/bench_particle_maps_500_b180_refine.json" 2>> "$OUT/bench.stderr"
python3 bench.py --particle-maps 1024,6,90 --force-sharded --steps 200 --report "$OUT/keep/bench_particle_maps_sharded_one_rank_report.json" > "$OUT/keep/bench_particle_maps_sharded_one_rank.json" 2>> "$OUT/bench.stderr"
python3 bench.py --particle-maps 4096,12.8,180 --force-sharded --steps 20 --report "$OUT/keep/bench_particle_maps_4096_sharded_one_rank_report.json" > "$OUT/keep/bench_particle_maps_4096_sharded_one_rank.json" 2>> "$OUT/bench.stderr"
# one rank's SLAM.update at the shard sizes of 2 / 4 / 8 GPUs for 4096 x 256^2 (DESIGN.md section 7's table), and the refinement with the field in memory
for n in 2048 1024 512; do
  python3 bench.py --particle-maps $n,12.8,180 --steps 20 --no-cpu-baseline --report "$OUT/keep/pm_shard_${n}_report.json" > "$OUT/keep/pm_shard_$n.json" 2>> "$OUT/bench.stderr"
done
python3 tools/pm_refine_big.py > "$OUT/keep/pm_refine_4096.txt" 2>> "$OUT/bench.stderr"
python3 tools/ab_update.py gridmap_slam_robot_amd/lib/libgridmapslam.so gridmap_slam_robot_amd/lib/libgridmapslam.so 500 6.0 90 2 > "$OUT/keep/pm500_update_unbracketed.txt" 2>> "$OUT/bench.stderr"
python3 tools/ab_update.py gridmap_slam_robot_amd/lib/libgridmapslam.so gridmap_slam_robot_amd/lib/libgridmapslam.so 4096 12.8 180 2 > "$OUT/keep/pm4096_update_unbracketed.txt" 2>> "$OUT/bench.stderr"
PM_CASES=3 python3 tools/pm_refine_probe.py > "$OUT/keep/pm_refine_probe.txt" 2>> "$OUT/bench.stderr"
bash tools/pmc_refine.sh > "$OUT/keep/refine_counters.txt" 2>&1
cp gpurun_out/pmc_refine/summary.json "$OUT/keep/refine_counters.json" 2>/dev/null
bash tools/pmc_likelihood.sh > "$OUT/keep/dense_likelihood_counters.txt" 2>&1
cp gpurun_out/pmc_lik/summary.json "$OUT/keep/dense_likelihood_counters.json" 2>/dev/null
bash tools/nseg_table.sh > "$OUT/keep/nseg_8_vs_16.txt" 2>&1
# the per-GPU step of the fixed-population series (config 4) at 1 / 2 / 4 / 8 GPUs' shard sizes, on one GPU (no exchange): DESIGN.md section 7's curve
for n in 65536 32768 16384 8192; do
  python3 bench.py --config C4 --particles $n $Q --steps 100 --warmup 10 --report "$OUT/keep/bench_c4_shard_${n}_report.json" > "$OUT/keep/bench_c4_shard_${n}.json" 2>> "$OUT/bench.stderr"
done
# the float domain of the rounded primitives, exhausted, with the record kept
GMS_EXHAUSTIVE=1 python3 -m pytest tests/test_gpu_exhaustive_float.py -m gpu -q > "$OUT/exhaustive.stdout" 2>&1
cp gpurun_out/exhaustive_float.json "$OUT/keep/" 2>/dev/null
# stage timeline of a C3 step from the instrumented build, when it was shipped along (lib/exp_stamps.so)
if [ -f gridmap_slam_robot_amd/lib/exp_stamps.so ]; then
  GMS_LIBRARY=$ROOT/gridmap_slam_robot_amd/lib/exp_stamps.so python3 tools/stamps.py > "$OUT/keep/c3_step_timeline.txt" 2>> "$OUT/bench.stderr"
  GMS_LIBRARY=$ROOT/gridmap_slam_robot_amd/lib/exp_stamps.so python3 tools/stamps.py --config C2 2>> "$OUT/bench.stderr" | head -24 > "$OUT/keep/c2_step_timeline.txt"
fi
if [ -f build/exp/lib_stamps.so ]; then cp build/exp/lib_stamps.so gridmap_slam_robot_amd/lib/exp_stamps.so; fi
if [ -f gridmap_slam_robot_amd/lib/exp_stamps.so ]; then
  GMS_LIBRARY=$ROOT/gridmap_slam_robot_amd/lib/exp_stamps.so python3 tools/refine_stamps.py > "$OUT/keep/refine500_timeline.txt" 2>> "$OUT/bench.stderr"
  GMS_LIBRARY=$ROOT/gridmap_slam_robot_amd/lib/exp_stamps.so python3 tools/pm_stamps.py 500 6 90 > "$OUT/keep/pm500_timeline.txt" 2>> "$OUT/bench.stderr"
  GMS_LIBRARY=$ROOT/gridmap_slam_robot_amd/lib/exp_stamps.so python3 tools/pm_stamps.py 1024 12.8 180 > "$OUT/keep/pm1024x256_timeline.txt" 2>> "$OUT/bench.stderr"
fi
# where the dense rebuild's time goes (lib/lik_exp{1,2,3}.so, when shipped along) and the log-normalising loop beside the plain one
bash tools/lik_phases.sh > "$OUT/keep/dense_likelihood_phases.txt" 2>> "$OUT/bench.stderr"
python3 tools/lognorm_probe.py > "$OUT/keep/lognorm_probe.txt" 2>> "$OUT/bench.stderr"
# config 5 split by map: one GPU's share at 1 / 2 / 4 / 8 GPUs
for mm in 64 32 16 8; do
  python3 bench.py --config C5 --maps $mm --steps 30 --warmup 4 $Q --report "$OUT/keep/bench_c5_maps_${mm}_report.json" > "$OUT/keep/bench_c5_maps_${mm}.json" 2>> "$OUT/bench.stderr"
done
ls -la "$OUT/keep"
for f in "$OUT"/keep/*_kernel_stats.csv; do echo "== $f"; python3 tools/kstats.py "$f" | head -8; done
