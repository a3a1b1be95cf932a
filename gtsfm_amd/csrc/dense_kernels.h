// Launchers and weight packers for the fp32-MFMA dense kernels (see dense_kernels.hip, mfma_tiles.h).
#pragma once

#include "common.h"

struct ConvParams {
    const float* in;   // NHWC activations [B][H][W][in_stride], channels in_coff .. in_coff+Cin-1 are read
    int in_stride, in_coff;
    float* out;        // NHWC [B][Ho][Wo][out_stride], channels out_coff .. out_coff+Cout-1 are written
    int out_stride, out_coff;
    const float* wpack;  // packed weights (pack_conv3x3_weights)
    const float* bias;   // [ceil(Cout/64)*64]
    int B, H, W, Cin, Cout;
    int relu, pool;      // pool: fused 2x2/stride-2 max-pool, Ho = H/2, Wo = W/2 (floor)
    int tiles_x, tiles_y;  // filled by the launcher
    // optional fused first layer: when img != null the input activation is relu(conv1a(img)) computed on the fly
    const void* img;   // [B][H][W] gray image, fp32 or uint8
    int img_is_u8;
    const float* w1a;  // conv1a weights [9 taps][64]
    const float* b1a;  // conv1a bias [64]
};

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
    int nb_per_wg;  // filled by the launcher: 128-column blocks one workgroup walks
    int debug;      // developer ablation switches (GTSFM_GEMM_DEBUG): 1 = skip epilogue, 2 = skip A loads
};

int launch_conv3x3(const ConvParams& p, hipStream_t stream);
int launch_gemm(const GemmParams& p, hipStream_t stream);
bool gemm_uses_dma(int K, int ldw);  // whether launch_gemm picks the LDS-DMA kernel for row-major weights of this shape
int launch_gemm_dma(const GemmParams& p, hipStream_t stream);  // gemm_dma_kernels.hip; launch_gemm dispatches to it
int launch_pack_rows(const float* B, int ldb, int N, const int* n_dev, int K, float* out, hipStream_t stream);

size_t packed_conv3x3_floats(int cin, int cout);
size_t packed_linear_floats(int k, int n);
void pack_conv3x3_weights(const float* w, int cin, int cout, float* out);
void pack_linear_weights(const float* w, int k_real, int k, int n, float* out);
