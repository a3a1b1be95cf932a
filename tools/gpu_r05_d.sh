#!/bin/bash
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05d
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_rccl_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -15 > $OUT/gpu_tests.txt
tail -5 $OUT/gpu_tests.txt
bash tools/prof_r05.sh > $OUT/prof.log 2>&1
tail -60 $OUT/prof.log
