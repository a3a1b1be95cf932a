#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_matchers_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -p no:cacheprovider -k "extraction or tier or assignment or golden or oracle or max_keypoints or full_depth" 2>&1 | tail -8 > $OUT/gpu_tests.txt
tail -4 $OUT/gpu_tests.txt
python tools/bench_assign.py 2>&1 | grep -v amdgpu.ids > $OUT/bench_assign.txt; grep "extract_waves=4\|layernorm" $OUT/bench_assign.txt
echo "== nontemporal off"; GTSFM_SWEEP_NT_MB=1000000 python tools/bench_assign.py 2>&1 | grep "N=5000 pairs=16 extract_waves=4"
echo "== narrow tier up to 2048 (round 4)"; GTSFM_EXTRACT_NARROW_COLS=2048 python tools/bench_assign.py 2>&1 | grep "N=2048 pairs=32 extract_waves=4"
