#!/bin/bash
# Round-2 GPU call 3: cycle budget and variants of the LDS-DMA GEMM; parity + bench with the projection folding
set -u
OUT=gpurun_out/r2c3
mkdir -p $OUT
for v in v0 v1 v2 v3; do
  for shape in "131072 256 768" "131072 512 512" "131072 512 256 1"; do
    timeout 60 tools/bin/gemm_dma_$v $shape | sed "s/^/$v: /"
  done
done 2>&1 | tee $OUT/gemm_variants.txt
for v in v0 v3; do
  for shape in "131072 256 768" "131072 512 512"; do
    timeout 60 tools/bin/gemm_dma_${v}_trace $shape | sed "s/^/$v: /"
  done
done 2>&1 | tee $OUT/gemm_trace.txt
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x ) > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2> $OUT/bench_lg.err | tail -1 > $OUT/bench_lg.json; cut -c1-200 $OUT/bench_lg.json
timeout 600 python bench.py --matcher superglue --sinkhorn 100 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2> $OUT/bench_sg100.err | tail -1 > $OUT/bench_sg100.json; cut -c1-200 $OUT/bench_sg100.json
