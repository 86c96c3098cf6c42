# per-particle-map update: builds lib/pm_*.so against the product, alternated on one box; then the parity tests of this tree
L=$PWD/gridmap_slam_robot_amd/lib
pm() { python3 bench.py --particle-maps $1 --steps $2 --no-cpu-baseline --report /tmp/r.json 2>/dev/null >/dev/null; python3 -c "
import json; d=json.load(open('/tmp/r.json')); k=d.get('kernels') or d.get('per_kernel') or {}
print(d.get('ms_per_step'), {a: (b.get('us') if isinstance(b, dict) else b) for a, b in k.items()} if k else list(d.keys())[:12])"; }
for r in 1 2; do
for v in $(ls $L | grep '^pm_' | sed 's/.so//') libgridmapslam; do
echo "$v pm500: $(GMS_LIBRARY=$L/$v.so pm 500,6,90 50)"
echo "$v pm500b180: $(GMS_LIBRARY=$L/$v.so pm 500,6,180 50)"
echo "$v pm4096: $(GMS_LIBRARY=$L/$v.so pm 4096,12.8,180 10)"
done; done
python -m pytest tests/test_gpu_slam_particle_maps.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error"
