for m in 8 16; do for o in "" 1; do for t in "" 512 256; do
  r=$(GMS_SCORE_ORDER=$o GMS_SCORE_THREADS=$t python bench.py --config C5 --maps $m --steps 30 --warmup 4 --no-cpu-baseline --no-secondary --report /tmp/r.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f'%d['ms_per_step'], d['kernel_us'])")
  echo "maps=$m order=${o:-auto} threads=${t:-auto}: $r"
done; done; done
