#!/bin/bash
# Kernel stats of the headline workload with ONE stream (kernels do not overlap, so the per-kernel average durations are
# the isolated ones that bench.py's roofline microbenchmarks measure). Output: gpurun_out/prof_streams1/.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_streams1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/full -o full -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline > $OUT/full.log 2>&1
rm -f $OUT/full/*kernel_trace.csv
grep '"metric"' $OUT/full.log | cut -c1-200
head -6 $OUT/full/full_kernel_stats.csv | cut -c1-140
