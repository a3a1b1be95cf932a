#!/bin/bash
# Round 5: the double log-softmax sweep with one exponential per element for the online column statistics -- LightGlue tests, then the
# assignment-stage micro-benchmark.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05z
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_matchers_gpu.py tests/test_lightglue_hf_golden_gpu.py -m gpu -q -p no:cacheprovider -k "lightglue or lg or tier or sweep or assignment" 2>&1 | tail -8 > $OUT/tests.txt
GTSFM_ASSIGN_BENCH_SHORT=1 timeout 600 python tools/bench_assign.py > $OUT/bench_assign.txt 2>&1
cat $OUT/tests.txt; grep -v "waves=8" $OUT/bench_assign.txt
