#!/bin/bash
# Round 5: the driver's bench command once more with profiles/r05_pmc_traffic.json in the tree (so that the line's `traffic` figures cite this round's
# counters), and the matcher / golden / config-1 / contract / arbiter / SuperPoint files with BOTH opt-in arithmetic switches in the environment.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_r05b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.log 2>&1
grep "^{" $OUT/bench_default.log | cut -c1-300; grep real $OUT/bench_default.log
GTSFM_ATTENTION_MATH=bf16x3 GTSFM_GEMM_MATH=bf16x3 python -m pytest tests/test_matchers_gpu.py tests/test_lightglue_hf_golden_gpu.py tests/test_bench_shapes_gpu.py tests/test_config1_lund_door_gpu.py \
  tests/test_reference_contract_gpu.py tests/test_lightglue_fp64_arbiter_gpu.py tests/test_superpoint_gpu.py tests/test_attention_bf16x3_gpu.py -m gpu -q -p no:cacheprovider \
  -k "not differs_from_exact and not bf16x3_arithmetic and not two_stream_pipeline" 2>&1 | tail -12 > $OUT/gpu_tests_bf16x3_both_switches.txt
tail -6 $OUT/gpu_tests_bf16x3_both_switches.txt
