#!/bin/bash
# Round 5: every sweep kernel (Sinkhorn, LightGlue double log-softmax, extraction; narrow and wide tiers) with unconditional buffer-resource
# loads -- the whole matcher test file, then the sweeps' micro-benchmarks.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05y
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_matchers_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > $OUT/tests.txt
timeout 600 python tools/bench_assign.py > $OUT/bench_assign.txt 2>&1
timeout 600 python tools/bench_sweeps.py 1024 2048 5000:16 5000 5000:1 > $OUT/bench_sweeps.txt 2>&1
cat $OUT/tests.txt; grep -v "waves=8" $OUT/bench_assign.txt; cut -c1-250 $OUT/bench_sweeps.txt
