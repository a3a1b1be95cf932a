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
};

int launch_gemm_dma_batched(const GemmParams& p, const GemmBatch& bt, hipStream_t stream);
