# the full likelihood rebuild as a burst (5 + 50 launches: PREROLL=0, the form of every earlier round) or under SUSTAINED load (PREROLL=300: 300
# rebuilds back to back in front of the measured ones); usage: PREROLL=300 bash tools/collect_dense_likelihood.sh <out dir under gpurun_out/>
ROOT=$PWD; OUT=$ROOT/gpurun_out/${1:-dense2}; PREROLL=${PREROLL:-0}; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
K="python3 $ROOT/tools/kbench.py --only likelihood --dense --iters 50 --preroll $PREROLL"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $K > $OUT/stats.stdout 2> $OUT/stats.stderr
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/dense_likelihood_kernel_stats.csv
K2="python3 $ROOT/tools/kbench.py --only likelihood --iters 50 --preroll $PREROLL"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_room -- $K2 > $OUT/stats_room.stdout 2> $OUT/stats_room.stderr
cp $(find $OUT/stats_room -name "*kernel_stats.csv" | head -1) $OUT/room_likelihood_kernel_stats.csv
cd $ROOT
bash tools/pmc_likelihood.sh > $OUT/dense_likelihood_counters.txt 2>&1
cp gpurun_out/pmc_lik/summary.json $OUT/dense_likelihood_counters.json
bash tools/lik_phases.sh > $OUT/dense_likelihood_phases.txt 2>&1
python3 tools/kstats.py $OUT/dense_likelihood_kernel_stats.csv | head -3
python3 tools/kstats.py $OUT/room_likelihood_kernel_stats.csv | head -3
