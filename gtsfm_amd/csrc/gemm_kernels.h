// GEMM launchers (gemm_mfma_kernels.hip: register-staged, packed weights; gemm_dma_kernels.hip: LDS-DMA, row-major weights).
#pragma once

#include "common.h"

struct GemmParams {
    const float* A;  // [M][lda], first K columns are read
    int lda, M, K;
    const int* m_dev;  // optional: row count read from device memory (<= M)
    const float* wpack;  // packed W[N][K] (pack_linear_weights / pack_rows); used when wraw is null or K % 32 != 0
    const float* wraw;   // optional row-major W[N][ldw] (nn.Linear layout, or an activation matrix): LDS-DMA kernel
    int ldw;
    const int* n_dev;    // optional: column count read from device memory (<= N); LDS-DMA kernel only
    const float* bias;   // [ceil(N/64)*64] or null
    int N;
    float* C;  // [M][ldc], columns c_coff .. c_coff+N-1 are written
    int ldc, c_coff;
    const float* res;  // optional residual [M][ldres]: C = res + act(alpha * (A W^T + bias))
    int ldres;
    float alpha;
    int relu;
    // optional per-M-tile masking for ragged batches whose sequences start at multiples of 128 rows
    const int* tile_cnt_idx;  // [M tiles] index into live_counts
    const int* tile_row0;     // [M tiles] first row of the tile within its sequence
    const int* live_counts;
    // optional rotary epilogue (LDS-DMA kernel only): columns [0, rot_cols) are rotated pairwise with the per-row (cos, sin)
    // pairs rot_enc[row][f][2], f = (column % 64) / 2 (LightGlue apply_cached_rotary_emb on the q and k parts of Wqkv)
    const float* rot_enc;
    int rot_cols;
    int math;       // 0 = exact fp32 (default; SuperPoint's 1x1 convolutions always), 1 = bf16x3, 2 = f16x2 (opt-in: set by the matchers' forward and the
                    // stand-alone linear entry points from GTSFM_GEMM_MATH via gemm_math_from_env(); LDS-DMA kernel only)
    int nb_per_wg;  // filled by the launcher: 128-column blocks one workgroup walks
    int super_rows; // filled by the launcher (LDS-DMA kernel): 0 = a row tile's column blocks run side by side on one XCD; r > 0 = wide products
                    // (more than 8 column groups: the matchers' score matrices) walk super-tiles of r row tiles x 8 column groups per XCD
    int debug;      // developer ablation switches (GTSFM_GEMM_DEBUG): 1 = skip epilogue, 2 = skip A loads
};

int launch_gemm(const GemmParams& p, hipStream_t stream);
int gemm_math_from_env();  // GTSFM_GEMM_MATH=bf16x3 -> 1, f16x2 -> 2, else 0; read per call by the callers that honour the switch
bool gemm_uses_dma(int K, int ldw);  // whether launch_gemm picks the LDS-DMA kernel for row-major weights of this shape
int launch_gemm_dma(const GemmParams& p, hipStream_t stream);  // gemm_dma_kernels.hip; launch_gemm dispatches to it
int launch_pack_rows(const float* B, int ldb, int N, const int* n_dev, int K, float* out, hipStream_t stream);
