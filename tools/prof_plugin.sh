#!/bin/bash
# Kernel-trace stats of the per-call plugin path (one pair at a time, N = 5000 and 2048): where a single pair's time goes.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_plugin
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for K in 5000 2048; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k$K -o p -- python $GRAFT_REPO_ROOT/tools/bench_plugin.py --keypoints $K > $OUT/k$K.log 2>&1
  find $OUT/k$K -name "*kernel_trace.csv" -delete
  f=$(find $OUT/k$K -name "*kernel_stats.csv" | head -1)
  echo "== N=$K"; head -16 $f | cut -c1-110,200-320
done
