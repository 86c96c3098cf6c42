# A/B of two builds of the library on one box: bash tools/ab_lib.sh <other .so> [bench args...]
OTHER=$1; shift
run() { python bench.py --no-secondary --no-cpu-baseline --steps 200 --warmup 20 "$@" --report /tmp/rep.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ms/step', d['ms_per_step'], d['kernel_us'])"; }
for r in 1 2 3; do
echo "other"; GMS_LIBRARY=$OTHER run "$@"
echo "this tree"; run "$@"
done
