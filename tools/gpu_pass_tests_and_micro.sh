#!/bin/bash
# Round 5, first GPU pass: the whole -m gpu suite (all failures, not the first), the printed tables of the LightGlue pins, the assignment-stage
# micro-benchmark (extraction tiers after the branch-free rewrite) and the per-call plugin path.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05a
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -40 > $OUT/gpu_tests.txt
python -m pytest tests/test_lightglue_hf_golden_gpu.py tests/test_lightglue_fp64_arbiter_gpu.py -m gpu -q -s -k "cap or float64" 2>&1 | grep "HFCAP\|ARBITER\|passed\|failed" > $OUT/lightglue_pins.txt
python tools/bench_assign.py > $OUT/bench_assign.txt 2>&1
python tools/bench_plugin.py > $OUT/bench_plugin.txt 2>&1
tail -15 $OUT/gpu_tests.txt; cat $OUT/lightglue_pins.txt; cat $OUT/bench_assign.txt; tail -12 $OUT/bench_plugin.txt
cat gpurun_out/config1_observed.json 2>/dev/null | head -60
