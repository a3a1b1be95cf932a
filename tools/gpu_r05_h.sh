#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_matchers_gpu.py tests/test_bench_shapes_gpu.py tests/test_config1_lund_door_gpu.py -m gpu -q -p no:cacheprovider -k "sinkhorn or superglue or golden or all_66 or tier" 2>&1 | tail -8 > $OUT/gpu_tests.txt
tail -4 $OUT/gpu_tests.txt
python tools/bench_sweeps.py 5000:16 5000:21 5000:8 3000:32 4096:24 8000:8 2048:32 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        r = json.loads(line); print(r['launch_shape'], r['avg_iteration_ms'], 'ms', r['achieved'], 'GB/s', r['frac'])
" | tee $OUT/bench_sweeps.txt
