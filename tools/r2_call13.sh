#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2c13
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cap5000 -o c5 -- python $GRAFT_REPO_ROOT/bench.py --keypoints 5000 --images 21 --pairs 200 --pair-chunk 8 --steps 1 --warmup 1 --streams 1 --graphs 0 --no-secondary --no-cpu-baseline > $OUT/c5.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
tail -1 $OUT/c5.log | head -c 200; echo
head -16 $OUT/cap5000/c5_kernel_stats.csv | cut -c1-170
