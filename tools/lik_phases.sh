# Where the dense-map likelihood rebuild spends its time: builds with GMS_EXTRA_FLAGS=-DGMS_LIK_EXP=1 (staging only: loads, classes, LDS),
# =2 (+ horizontal sums), =3 (+ vertical sums, no stores) as lib/lik_exp{1,2,3}.so beside the product, alternated on one box.
L=$PWD/gridmap_slam_robot_amd/lib
for r in 1 2 3; do
for v in lik_exp1 lik_exp2 lik_exp3 libgridmapslam; do
[ -f $L/$v.so ] || continue
echo -n "$v "; GMS_LIBRARY=$L/$v.so python3 tools/kbench.py --only likelihood --dense --iters 100 2>/dev/null | tail -1
done; done
