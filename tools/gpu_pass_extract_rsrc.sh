#!/bin/bash
# Round 5: the wide extraction kernel with unconditional buffer-resource loads (exact wait counts) -- the matcher tests that cover every sweep tier,
# then the assignment-stage micro-benchmark.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05x
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_matchers_gpu.py -m gpu -q -p no:cacheprovider -x -k "tier or sweep or assignment or ragged or oracle or golden" 2>&1 | tail -8 > $OUT/tests.txt
timeout 600 python tools/bench_assign.py > $OUT/bench_assign.txt 2>&1
cat $OUT/tests.txt; cat $OUT/bench_assign.txt
