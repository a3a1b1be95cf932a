#!/bin/bash
# Round-2 GPU call 1: full GPU parity suite on the merged state (LDS-DMA GEMM, one-exp Sinkhorn), GEMM shapes, benches.
set -u
OUT=gpurun_out/r2c1
mkdir -p $OUT
( time timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider ) > $OUT/pytest.txt 2>&1
tail -30 $OUT/pytest.txt
DBG_LIST=0 timeout 200 python tools/gpu_gemm_ablation.py 2>&1 | tee $OUT/gemm.txt
timeout 600 python bench.py --steps 5 --warmup 2 2> $OUT/bench_lg.err | tail -1 > $OUT/bench_lg.json; cut -c1-400 $OUT/bench_lg.json
timeout 600 python bench.py --matcher superglue --sinkhorn 100 --steps 2 --warmup 1 --no-secondary 2> $OUT/bench_sg100.err | tail -1 > $OUT/bench_sg100.json; cut -c1-400 $OUT/bench_sg100.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_lg -o lg -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$OUT/prof_lg.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_sg -o sg -- python $GRAFT_REPO_ROOT/bench.py --matcher superglue --sinkhorn 100 --pairs 256 --steps 1 --warmup 1 --streams 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/$OUT/prof_sg.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*_kernel_stats.csv" | head
for f in $(find $OUT -name "*_kernel_stats.csv"); do echo "== $f"; head -14 $f | cut -c1-160; done
