// fp32-input MFMA tile machinery shared by the 3x3 convolution and the dense (1x1 conv / linear) kernels.
//
// Every dense op on the hot path is C[M,N] = A[M,K] * W[N,K]^T (+bias, activation...) with fp32 operands:
//   * SuperPoint conv3x3  (superpoint.py:119-134): M = pixels, K = 9*Cin (im2col on the fly from an LDS halo tile)
//   * SuperPoint conv1x1, SuperGlue Conv1d(k=1) (superglue.py:49-60,92-119), LightGlue nn.Linear: plain GEMM
// and is executed on v_mfma_f32_32x32x2_f32, which is bit-for-bit a k-ordered fmaf chain at the fp32 vector rate
// (157 TFLOP/s peak on MI355X) -- exact fp32, as the 1e-4 / bit-exact-keypoint contract requires.
//
// Workgroup = 256 threads = 4 waves arranged 2 (M) x 2 (N); workgroup tile = 128 rows x 64 columns;
// wave tile = 64 rows x 32 columns = two 32x32 MFMA accumulators (32 VGPRs).
//   A operand: from LDS, one ds_read_b128 per 32-row tile per 8-deep k-step (lane l: row l&31, k-half l>>5,
//              4 consecutive k) -> feeds 4 MFMAs.
//   B operand: straight from global/L2 in a pre-packed, lane-linear layout: one 1 KiB wave-load (16 B per lane,
//              fully coalesced) per 8-deep k-step -> feeds 8 MFMAs. All workgroups stream the same packed weights,
//              so they stay L2-resident; no LDS staging and no barrier in the k-loop.
// The order in which k-values meet inside one MFMA is (k0,k4),(k1,k5),(k2,k6),(k3,k7) per 8-deep step -- a fixed,
// deterministic summation order.
#pragma once

#include "common.h"

#define MT_TILE_M 128
#define MT_TILE_N 64
#define MT_LDS_ROW 68  // floats per staged row: 64 + 4 pad -> 272 B stride, conflict-free ds_read_b128

// Packed weight layout (floats): [n_block][k_step][wave_n (2)][lane (64)][4]
//   lane = k_half*32 + j ; element e -> W[n = n_block*64 + wave_n*32 + j][k(k_step, k_half, e)]
#define MT_PACK_STEP_FLOATS 512  // one k-step of one n-block (both wave_n halves)

__device__ __forceinline__ f32x4 mt_load_b(const float* __restrict__ wpack_nblk, int kstep, int wn, int lane) {
    return *reinterpret_cast<const f32x4*>(wpack_nblk + (size_t)kstep * MT_PACK_STEP_FLOATS + wn * 256 + lane * 4);
}

// 8 MFMAs: two 32x32 accumulators x 4 k-pairs.
__device__ __forceinline__ void mt_step(f32x16& acc0, f32x16& acc1, const f32x4 a0, const f32x4 a1, const f32x4 b) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
}

// Accumulator element r of a 32x32 tile lives at row (r&3) + 8*(r>>2) + 4*(lane>>5), column lane&31.
__device__ __forceinline__ int mt_acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
