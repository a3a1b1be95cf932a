#!/bin/bash
# Round-2 GPU call 4: column-walking LDS-DMA GEMM (+ rotary epilogue)
set -u
OUT=gpurun_out/r2c4
mkdir -p $OUT
for nb in 0 1 2 3; do
  for shape in "131072 256 768" "131072 512 512" "131072 512 256 1" "131072 256 512"; do
    GTSFM_GEMM_NB=$nb timeout 60 tools/bin/gemm_dma_walk $shape | tr '\n' ' ' | sed "s/^/walk nb=$nb: /"; echo
  done
done 2>&1 | tee $OUT/gemm_walk.txt
GTSFM_GEMM_DMA=tile timeout 60 tools/bin/gemm_dma_walk 131072 256 768 | tr '\n' ' ' | sed "s/^/tile: /" | tee -a $OUT/gemm_walk.txt; echo
for shape in "131072 256 768" "131072 512 512"; do timeout 60 tools/bin/gemm_dma_walk_trace $shape; done 2>&1 | tee $OUT/gemm_walk_trace.txt
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x ) > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2> $OUT/bench_lg.err | tail -1 > $OUT/bench_lg.json; cut -c1-200 $OUT/bench_lg.json
timeout 600 python bench.py --matcher superglue --sinkhorn 100 --steps 2 --warmup 1 --no-secondary --no-cpu-baseline 2> $OUT/bench_sg100.err | tail -1 > $OUT/bench_sg100.json; cut -c1-200 $OUT/bench_sg100.json
python - <<PY
import json
for f in ("bench_lg", "bench_sg100"):
    d = json.load(open("$OUT/%s.json" % f))
    print(f, d["value"], [(r["kernel"][:16], r.get("launch_shape"), r["frac"]) for r in d["roofline_other"]])
PY
