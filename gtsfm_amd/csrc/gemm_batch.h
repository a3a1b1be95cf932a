// Ragged batches of independent products through ONE launch of the LDS-DMA GEMM (the matchers' per-pair score GEMMs,
// superglue.py:257-258 / LightGlue's sim = mdesc0 mdesc1^T): problem b = rows [a_row, a_row + counts[m_idx]) of A times rows
// [w_row, w_row + counts[n_idx]) of W, written at C + c_off with row stride ldc; every other GemmParams field is shared, M / N
// are the batch maxima (grid sizing). Round 1 looped over the pairs on the host: 32 launches per chunk through one shared
// pack buffer, with the host's n0 / n1 arrays in the loop.
#pragma once

#include "gemm_kernels.h"

struct GemmProblem {
    long long c_off;  // floats
    int a_row, w_row, m_idx, n_idx, ldc, pad;
};

struct GemmBatch {
    const GemmProblem* problems;  // device, [nproblems]
    const int* counts;            // device
    int nproblems;
    // optional LayerNorm + GELU over the N = 512 output columns (LightGlue's ffn.1 / ffn.2 behind ffn.0): the workgroup walks
    // all four column blocks of its row tile, then normalises and activates its own 128 rows in place (the rows are L2-hot and
    // the pass runs beside the other workgroups' MFMAs) -- the separate layernorm_gelu_kernel and its HBM round trip go away
    const float* ln_gamma;
    const float* ln_beta;
};

int launch_gemm_dma_batched(const GemmParams& p, const GemmBatch& bt, hipStream_t stream);
