# A/B of an environment setting on one box: bash tools/ab_env.sh VAR=VALUE [bench args...]
SET=$1; shift
run() { python bench.py --no-secondary --no-cpu-baseline "$@" --report /tmp/rep.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ms/step', d['ms_per_step'], d['kernel_us'])"; }
for r in 1 2 3; do
echo "default"; run "$@"
echo "$SET"; env $SET python bench.py --no-secondary --no-cpu-baseline "$@" --report /tmp/rep.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ms/step', d['ms_per_step'], d['kernel_us'])"
done
