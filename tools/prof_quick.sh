#!/bin/bash
# Quick per-kernel averages of the headline workload's launch shapes (250 of its 1000 pairs: 23 views at the cap; one stream, eager launches, the
# workload's launches only): gpurun_out/prof_quick/<tag>_kernel_stats.csv. Usage: tools/prof_quick.sh <tag> [extra bench.py flags]
set -u
TAG=${1:-quick}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_quick
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$TAG -o $TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --streams 1 --graphs 0 \
    --no-cpu-baseline --no-secondary --no-roofline --pairs 250 --details-file "" "$@" > $OUT/$TAG.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
cp $OUT/$TAG/*/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv 2>/dev/null || cp $OUT/$TAG/*kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
head -16 $OUT/${TAG}_kernel_stats.csv | cut -c1-170
grep '"metric"' $OUT/$TAG.log | cut -c1-160
