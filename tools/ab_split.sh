# dirty-tile likelihood rebuilds with two workgroups per tile (product) against one (GMS_LIK_SPLIT=0), same library, one box; parity tests first
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_lik_skip.py tests/test_gpu_lazy_likelihood.py tests/test_gpu_configs.py tests/test_gpu_tile_census_and_streams.py tests/test_gpu_trace_replay.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror"
run() { python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 20 "$@" --report /tmp/rep.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ms/step', d['ms_per_step'], d['kernel_us'], d.get('map_update_ms_per_scan'))"; }
for r in 1 2 3; do
echo "one workgroup per tile"; GMS_LIK_SPLIT=0 run
echo "two"; run
done
echo "C2 one"; GMS_LIK_SPLIT=0 run --config C2
echo "C2 two"; run --config C2
echo "C5 (batched: never split)"; run --config C5 --steps 20 --warmup 3
