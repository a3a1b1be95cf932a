#!/bin/bash
# Round-2 GPU call 7: ragged score GEMM launch, hipGraph replay of full chunks
set -u
OUT=gpurun_out/r2c7
mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x ) > $OUT/pytest.txt 2>&1
tail -25 $OUT/pytest.txt
for cfg in "--graphs 1 --streams 2" "--graphs 0 --streams 2" "--graphs 1 --streams 1" "--graphs 0 --streams 1"; do
  timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline $cfg 2> $OUT/bench_lg.err | tail -1 > $OUT/bench_lg.json; echo "$cfg: $(cut -c1-110 $OUT/bench_lg.json)"
done
timeout 600 python bench.py --matcher superglue --sinkhorn 100 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2> $OUT/bench_sg100.err | tail -1 > $OUT/bench_sg100.json; cut -c1-200 $OUT/bench_sg100.json
tail -3 $OUT/bench_lg.err $OUT/bench_sg100.err
