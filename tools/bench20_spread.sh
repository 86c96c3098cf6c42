# the driver's form of the bench (--steps 20 --warmup 5), ten processes with the untimed pre-roll and ten without: how a 1 ms timed region spreads
for p in 300 0 300 0 300 0 300 0 300 0 300 0 300 0 300 0; do
echo "preroll $p: $(GMS_BENCH_PREROLL=$p python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --report /tmp/r.json 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
done
