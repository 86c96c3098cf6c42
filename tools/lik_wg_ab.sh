# persistent likelihood workgroups per CU (GMS_EXTRA_FLAGS=-DGMS_LIK_WG_PER_CU=n builds as lib/lik_wg<n>.so) against the product's 5
L=$PWD/gridmap_slam_robot_amd/lib
for r in 1 2 3; do
for v in lik_wg4 lik_wg4w4 libgridmapslam lik_wg6 lik_wg8; do
[ -f $L/$v.so ] || continue
echo "$v dense: $(GMS_LIBRARY=$L/$v.so python3 tools/kbench.py --only likelihood --dense --iters 100 2>/dev/null | tail -1) room: $(GMS_LIBRARY=$L/$v.so python3 tools/kbench.py --only likelihood --iters 100 2>/dev/null | tail -1)"
done; done
