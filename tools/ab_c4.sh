run() { python bench.py --no-secondary --no-cpu-baseline --steps 100 --warmup 10 "$@" --report /tmp/rep.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ms/step', d['ms_per_step'], 'kernels', d['kernel_us'])"; }
for n in 65536 32768 24576; do
echo "population $n spread 0"; GMS_SCORE_SPREAD=0 run --config C4 --particles $n
echo "population $n spread 1"; GMS_SCORE_SPREAD=1 run --config C4 --particles $n
done
