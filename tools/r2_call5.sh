#!/bin/bash
# Round-2 GPU call 5: wave-specialised LDS-DMA GEMM
set -u
OUT=gpurun_out/r2c5
mkdir -p $OUT
for form in ws walk; do
  for shape in "131072 256 768" "131072 512 512" "131072 256 512" "131072 256 256" "32768 256 768" "5000 256 4800"; do
    GTSFM_GEMM_DMA=$form timeout 60 tools/bin/gemm_dma_walk $shape | tr '\n' ' ' | sed "s/^/$form: /"; echo
  done
done 2>&1 | tee $OUT/gemm_ws.txt
for nb in 1 2 3; do GTSFM_GEMM_NB=$nb timeout 60 tools/bin/gemm_dma_walk 131072 256 768 | tr '\n' ' ' | sed "s/^/ws nb=$nb: /"; echo; done | tee -a $OUT/gemm_ws.txt
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider -x ) > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2> $OUT/bench_lg.err | tail -1 > $OUT/bench_lg.json; cut -c1-200 $OUT/bench_lg.json
GTSFM_GEMM_DMA=walk timeout 600 python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-120
