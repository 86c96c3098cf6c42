# likelihoodData stores written through (product) against plain stores (lib/lik_plain.so = GMS_EXTRA_FLAGS=-DGMS_LIK_STORE_PLAIN), one box
L=$PWD/gridmap_slam_robot_amd/lib
pm() { python3 bench.py --particle-maps $1 --steps $2 --no-cpu-baseline --report /tmp/r.json 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for r in 1 2; do
for v in lik_plain libgridmapslam; do
echo "$v dense: $(GMS_LIBRARY=$L/$v.so python3 tools/kbench.py --only likelihood --dense --iters 100 2>/dev/null | tail -1)"
echo "$v room: $(GMS_LIBRARY=$L/$v.so python3 tools/kbench.py --only likelihood --iters 100 2>/dev/null | tail -1)"
echo "$v pm500: $(GMS_LIBRARY=$L/$v.so pm 500,6,90 50)"
echo "$v pm4096: $(GMS_LIBRARY=$L/$v.so pm 4096,12.8,180 10)"
echo "$v full-rebuild step: $(GMS_LIBRARY=$L/$v.so python3 bench.py --full-rebuild --no-cpu-baseline --no-secondary --report /tmp/r.json 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
done; done
