# A/B of the likelihood kernels of two builds on one box: parity tests of this tree first, then dense rebuild, per-particle maps, C3 step
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_lik_skip.py tests/test_gpu_slam_particle_maps.py tests/test_gpu_lazy_likelihood.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" > gpurun_out/ab_tests.txt
O=$PWD/gridmap_slam_robot_amd/lib/old.so
for r in 1 2 3; do
echo old; GMS_LIBRARY=$O python3 tools/kbench.py --only likelihood --dense --iters 100 2>/dev/null | tail -1
echo new; python3 tools/kbench.py --only likelihood --dense --iters 100 2>/dev/null | tail -1
done
for r in 1 2; do
echo old; GMS_LIBRARY=$O python3 bench.py --particle-maps 500,6,90 --steps 50 --no-cpu-baseline --report /tmp/r.json 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
echo new; python3 bench.py --particle-maps 500,6,90 --steps 50 --no-cpu-baseline --report /tmp/r.json 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done
bash tools/ab_lib.sh $O
