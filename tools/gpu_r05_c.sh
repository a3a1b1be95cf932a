#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05c
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_matchers_gpu.py -m gpu -q -p no:cacheprovider -k "attention or two_stream or single_pair or first_layer or threads" 2>&1 | tail -8 > $OUT/gpu_tests.txt
tail -4 $OUT/gpu_tests.txt
GTSFM_PAIR_STREAMS=1 python tools/bench_plugin.py --keypoints 5000 2048 > $OUT/bench_plugin_one_stream.txt 2>&1
python tools/bench_plugin.py --keypoints 5000 2048 > $OUT/bench_plugin_two_streams.txt 2>&1
for f in one_stream two_streams; do echo "== $f"; grep -o '"match_ms_each_call": [^]]*]\|"pairs_per_s_match_only_by_worker_threads": {[^}]*}\|"match_ms_per_pair_resident": [0-9.]*\|"pairs_per_s_match_only_resident": [0-9.]*\|"synchronous_ms_per_pair": [0-9.]*' $OUT/bench_plugin_$f.txt; done
python tools/bench_attention.py --quick > $OUT/bench_attention.txt 2>&1; tail -20 $OUT/bench_attention.txt
