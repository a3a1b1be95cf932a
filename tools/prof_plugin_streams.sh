set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_prof_plugin
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for S in 1 2; do
  GTSFM_PAIR_STREAMS=$S rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$S -o p -- python $GRAFT_REPO_ROOT/tools/bench_plugin.py --keypoints 5000 > $OUT/s$S.log 2>&1
  find $OUT/s$S -name "*kernel_trace.csv" -delete
  f=$(find $OUT/s$S -name "*kernel_stats.csv" | head -1); echo "== GTSFM_PAIR_STREAMS=$S"; head -8 $f | cut -c1-90,180-300
done
