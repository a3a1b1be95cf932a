#!/bin/bash
set -u
for v in base step0 base step0; do
  for shape in "131072 256 768" "131072 512 512" "131072 512 256 1"; do
    timeout 60 tools/bin/gemm_dma_$v $shape | tail -1 | sed "s/^/$v: /"
  done
done
for v in base step0; do timeout 60 tools/bin/gemm_dma_${v}_trace 131072 256 768 | sed "s/^/$v: /"; done
