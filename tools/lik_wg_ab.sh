# persistent likelihood workgroups per CU (GMS_EXTRA_FLAGS="-DGMS_LIK_WG_PER_CU=n [-DGMS_LIK_WAVES_PER_EU=n]" builds as lib/lik_wg*.so) against the product's 5
L=$PWD/gridmap_slam_robot_amd/lib
for r in 1 2 3; do
for v in $(ls $L | grep '^lik_wg' | sed 's/.so//') libgridmapslam; do
echo "$v dense: $(GMS_LIBRARY=$L/$v.so python3 tools/kbench.py --only likelihood --dense --iters 100 2>/dev/null | tail -1) room: $(GMS_LIBRARY=$L/$v.so python3 tools/kbench.py --only likelihood --iters 100 2>/dev/null | tail -1)"
done; done
