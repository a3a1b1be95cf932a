#!/bin/bash
# Round-2 opening moves on the GPU box (≈ 6 GPU-minutes): measure the two prototypes that ended round 1 green but untuned,
# and gate the library integration of the LDS-DMA GEMM (git branch r2-gemm-dma) on the full parity suite.
#   gpurun --timeout 900 -- 'bash tools/round2_first_steps.sh'
# Run it from a checkout of r2-gemm-dma with the library built (python -m gtsfm_amd.csrc.build) to test the integration;
# from main it only measures the prototypes.
set -u
OUT=gpurun_out/round2_first
mkdir -p $OUT
echo "== prototypes (tools/experimental)"
for shape in "131072 256 768" "131072 512 512" "131072 512 256" "131072 256 256"; do timeout 60 tools/experimental/gemm_dma $shape | tail -1; done | tee $OUT/gemm_dma.txt
timeout 60 tools/experimental/attention_dma | tee $OUT/attention_dma.txt
echo "== library"
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $OUT/pytest_gpu.txt
DBG_LIST=0 timeout 100 python tools/gpu_gemm_ablation.py | tee $OUT/gemm_library.txt
GTSFM_GEMM=mfma DBG_LIST=0 timeout 100 python tools/gpu_gemm_ablation.py | sed 's/^/register-staged kernel: /' | tee -a $OUT/gemm_library.txt
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | tee $OUT/bench_lightglue.json | cut -c1-200
timeout 300 python bench.py --matcher superglue --sinkhorn 100 --no-cpu-baseline 2>/dev/null | tail -1 | tee $OUT/bench_superglue100.json | cut -c1-200
