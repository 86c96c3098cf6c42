# the driver's form of the bench (--steps 20 --warmup 5), N processes with the untimed pre-roll and N without: how a 1 ms timed region spreads
N=${1:-8}
for i in $(seq $N); do for p in 300 0; do
echo "preroll $p: $(GMS_BENCH_PREROLL=$p python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --report /tmp/r.json 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('host_issue_ms_per_step'))")"
done; done
