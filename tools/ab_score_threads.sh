run() { python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 20 "$@" --report /tmp/rep.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ms/step', d['ms_per_step'], 'kernels', d['kernel_us'])"; }
for n in 8192 4096; do
 echo "particles $n: forced 1024 lanes"; GMS_SCORE_THREADS=1024 run --config C3 --particles $n
 echo "particles $n: auto"; run --config C3 --particles $n
done
echo "C2 forced 1024"; GMS_SCORE_THREADS=1024 run --config C2
echo "C2 auto"; run --config C2
echo "C3 auto"; run
