#!/bin/bash
# Round-2 GPU call 6: LDS-DMA pieces interleaved with the MFMAs
set -u
OUT=gpurun_out/r2c11
mkdir -p $OUT
for shape in "131072 256 768" "131072 512 512" "131072 512 256 1" "131072 256 512" "131072 256 256" "32768 256 768" "5000 256 4800"; do
  timeout 60 tools/bin/gemm_dma_walk $shape | tr '\n' ' ' | sed "s/^/walk2: /"; echo
done 2>&1 | tee $OUT/gemm_walk2.txt
for shape in "131072 256 768" "131072 512 512"; do timeout 60 tools/bin/gemm_dma_walk_trace $shape; done 2>&1 | tee $OUT/gemm_walk2_trace.txt
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x ) > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2> $OUT/bench_lg.err | tail -1 > $OUT/bench_lg.json; cut -c1-200 $OUT/bench_lg.json
timeout 600 python bench.py --matcher superglue --sinkhorn 100 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2> $OUT/bench_sg100.err | tail -1 > $OUT/bench_sg100.json; cut -c1-200 $OUT/bench_sg100.json
