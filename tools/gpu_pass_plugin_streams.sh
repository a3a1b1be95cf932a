#!/bin/bash
# Round 5, second GPU pass: the matcher suite after the two-stream single-pair form and the four-wave extraction tier, the per-call plugin path
# with one / two launch sequences per pair, the assignment-stage micro-benchmark, and a kernel trace of the two-stream plugin path.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05b
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_matchers_gpu.py tests/test_attention_bf16x3_gpu.py tests/test_config1_lund_door_gpu.py tests/test_lightglue_hf_golden_gpu.py tests/test_bench_shapes_gpu.py tests/test_abi_from_c.py -m gpu -q -p no:cacheprovider 2>&1 | tail -30 > $OUT/gpu_tests.txt
tail -6 $OUT/gpu_tests.txt
GTSFM_PAIR_STREAMS=1 python tools/bench_plugin.py --keypoints 5000 2048 > $OUT/bench_plugin_one_stream.txt 2>&1
python tools/bench_plugin.py --keypoints 5000 2048 > $OUT/bench_plugin_two_streams.txt 2>&1
for f in one_stream two_streams; do echo "== $f"; grep -o '"match_ms_each_call": [^]]*]\|"pairs_per_s_match_only_by_worker_threads": {[^}]*}\|"match_ms_per_pair_resident": [0-9.]*\|"pairs_per_s_match_only_resident": [0-9.]*\|"synchronous_ms_per_pair": [0-9.]*' $OUT/bench_plugin_$f.txt; done
python tools/bench_assign.py 2>&1 | grep -v amdgpu.ids > $OUT/bench_assign.txt; head -4 $OUT/bench_assign.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_plugin -o p -- python $GRAFT_REPO_ROOT/tools/bench_plugin.py --keypoints 5000 > $OUT/prof_plugin.log 2>&1
find $OUT/prof_plugin -name "*kernel_trace.csv" -delete
f=$(find $OUT/prof_plugin -name "*kernel_stats.csv" | head -1); head -14 $f | cut -c1-110,200-330
