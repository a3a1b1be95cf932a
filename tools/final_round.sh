#!/bin/bash
# Round-6 final evidence (GPU box), everything on the tree as it is: the whole -m gpu suite, the matcher / golden / config files again under both opt-in
# switches, smoke, the driver's bench command (BASELINE config 3 as written; every other config and the opt-in arithmetic as secondary legs), BASELINE
# config 4 as ONE scene through ShardedDetDescCorrespondenceGenerator plain and under a one-rank RCCL group, tools/scale_selfcheck.sh 1, the per-call
# path's clean profile and the micro-benchmarks. Everything lands in gpurun_out/final_r06 and is copied to profiles/ by hand.
# (tools/prof_r06.sh -- kernel trace of the headline, PMC traffic, SQ counters -- is its own call: bench.py reads the traffic file it produces.)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_r06
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 > $OUT/gpu_tests.txt
GTSFM_ATTENTION_MATH=bf16x3 GTSFM_GEMM_MATH=bf16x3 python -m pytest tests/test_matchers_gpu.py tests/test_lightglue_hf_golden_gpu.py tests/test_lightglue_fp64_arbiter_gpu.py \
    tests/test_config1_lund_door_gpu.py tests/test_reference_contract_gpu.py tests/test_superpoint_gpu.py tests/test_attention_bf16x3_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > $OUT/gpu_tests_bf16x3_both_switches.txt
GTSFM_ATTENTION_MATH=f16x2 GTSFM_GEMM_MATH=f16x2 python -m pytest tests/test_matchers_gpu.py tests/test_lightglue_hf_golden_gpu.py tests/test_lightglue_fp64_arbiter_gpu.py \
    tests/test_config1_lund_door_gpu.py tests/test_reference_contract_gpu.py tests/test_superpoint_gpu.py tests/test_attention_bf16x3_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -6 > $OUT/gpu_tests_f16x2_both_switches.txt
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.log 2>&1
cp gpurun_out/bench_details.json $OUT/bench_details.json
python bench.py --mode scene --steps 1 --warmup 1 --no-secondary --no-cpu-baseline --dump-matches 1 --details-file $OUT/scene_details.json > $OUT/bench_scene_config4_cap5000.log 2>&1
GTSFM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python bench.py --mode scene --steps 1 --warmup 1 --no-secondary --no-cpu-baseline --no-roofline --dump-matches 1 \
    --details-file $OUT/scene_rccl_details.json > $OUT/bench_scene_config4_cap5000_rccl_one_rank.log 2>&1
SCALE_SELFCHECK_OUT=$OUT/scale_selfcheck bash tools/scale_selfcheck.sh 1 > $OUT/scale_selfcheck_n1.txt 2>&1
bash tools/prof_plugin_resident.sh 5000 > $OUT/plugin_resident.txt 2>&1
cp gpurun_out/prof_plugin_resident/plugin_resident_kernel_stats_k5000.csv $OUT/ 2>/dev/null
python tools/bench_attention.py --quick > $OUT/bench_attention.txt 2>&1
python tools/bench_sweeps.py 2048 5000:16 5000:1 > $OUT/bench_sweeps.txt 2>&1
python tools/bench_score_gemm.py 5000 > $OUT/bench_score_gemm.txt 2>&1
for f in gpu_tests.txt gpu_tests_bf16x3_both_switches.txt gpu_tests_f16x2_both_switches.txt smoke.txt scale_selfcheck_n1.txt; do echo "== $f"; tail -3 $OUT/$f; done
for f in bench_default bench_scene_config4_cap5000 bench_scene_config4_cap5000_rccl_one_rank; do echo "== $f"; grep "^{" $OUT/$f.log | cut -c1-300; done
grep real $OUT/bench_default.log; grep "^{" $OUT/bench_default.log | wc -c
grep -h "keypoints\|matcher kernels" $OUT/plugin_resident.txt | cut -c1-250
