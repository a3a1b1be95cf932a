// fp32-MFMA dense kernels for gfx950: 3x3 convolution (NHWC, LDS halo tile, on-the-fly im2col) and GEMM
// (1x1 convolution / Conv1d(k=1) / nn.Linear) with fused bias / ReLU / 2x2 max-pool / residual epilogues.
// Replaces the ATen conv2d / conv1d / linear / max_pool2d calls of
//   thirdparty/SuperGluePretrainedNetwork/models/superpoint.py:148-162,190-191
//   thirdparty/SuperGluePretrainedNetwork/models/superglue.py:49-60,98-107,110-119,254
// See mfma_tiles.h for the tiling and the packed weight layout.

#include <stdlib.h>

#include "dense_kernels.h"
#include "mfma_tiles.h"

// ---------------------------------------------------------------------------------------------------------------
// conv3x3, stride 1, zero pad 1, NHWC. Workgroup tile: 8 x 16 output pixels x 64 output channels.
// LDS: (8+2) x (16+2) halo pixels x 64 input channels of the current 64-channel chunk (row stride 68 floats).
// ---------------------------------------------------------------------------------------------------------------

#define CV_TH 8
#define CV_TW 16
#define CV_HW (CV_TW + 2)
#define CV_HALO_PIX ((CV_TH + 2) * CV_HW)
// LDS pitch of one halo row in floats: 18 pixels x 68 + 56 = 1280 = 0 (mod 64 banks). The 32 pixels a wave reads per
// ds_read_b128 span two halo rows (16 + 16); with the natural pitch 18 x 68 = 8 (mod 64) the second row's lanes land on
// banks the first row's lanes of the same 16-lane service group already use (46 % of the LDS-active cycles were
// conflict cycles, SQ_LDS_BANK_CONFLICT); with pitch = 0 (mod 64) lane j of either row sits at bank 4 j like in the
// uniformly strided GEMM / attention tiles.
#define CV_ROW_PITCH (CV_HW * MT_LDS_ROW + 56)
#define CV_HALO_FLOATS ((CV_TH + 2) * CV_ROW_PITCH)

// FUSE_C1A: the input activation is not read from HBM but recomputed on the fly from the gray image: the halo tile of
// conv1b's input IS relu(conv1a(image)) (superpoint.py:148-149), 9 fma per value in conv1a_kernel's tap order (bit-identical
// to the stand-alone kernel). Removes conv1a's 256 B/pixel write and conv1b's 360 B/pixel read.
template <bool FUSE_C1A>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    int bid = blockIdx.x;
    const int tx = bid % p.tiles_x;
    bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int x0 = tx * CV_TW, y0 = ty * CV_TH;
    const int nb = blockIdx.y;
    const int nchunks = p.Cin >> 6;
    const int total_steps = nchunks * 72;
    const float* __restrict__ wp = p.wpack + (size_t)nb * total_steps * MT_PACK_STEP_FLOATS;

    const int j = lane & 31, kh = lane >> 5;
    const int cout = nb * 64 + wn * 32 + j;
    const float bias = p.bias[cout];  // bias array is padded to a multiple of 64
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc0[r] = bias;
        acc1[r] = bias;
    }

    const int a_base0 = (4 * wm + (j >> 4)) * CV_ROW_PITCH + (j & 15) * MT_LDS_ROW + kh * 4;
    const int a_base1 = a_base0 + 2 * CV_ROW_PITCH;
    const float* __restrict__ in_b = p.in + (size_t)b * p.H * p.W * p.in_stride + p.in_coff;

    int kstep = 0;
    f32x4 bcur = mt_load_b(wp, 0, wn, lane);
    // Weights run THREE k-steps ahead of their use and the A fragments one, with the issue order of every k-step pinned
    // (1 weight load, 2 LDS reads, 8 MFMAs). Measured on the SuperPoint conv stack, % of the fp32 MFMA peak: weights one
    // step ahead, reads at use, free schedule 76.7 (162 VGPRs: the compiler hoists loads at will); weights two ahead
    // 81.6; + A one ahead, pinned 84.8 (74 VGPRs); weights three ahead 86.0; four ahead 85.0.
    f32x4 bnxt = mt_load_b(wp, 1, wn, lane);  // total_steps >= 72
    f32x4 bnx2 = mt_load_b(wp, 2, wn, lane);
    for (int cc = 0; cc < nchunks; ++cc) {
        // (raising the wave priority for staging / epilogue as in the GEMM and attention kernels was measured here and
        // costs 2.5 points: 74.1 % vs 76.7 %)
        if (cc > 0) __syncthreads();
        // stage the halo tile of this 64-channel chunk: 180 pixels x 16 float4
        if (FUSE_C1A) {
            float* w1 = lds + CV_HALO_FLOATS;  // [9 taps][64] + [64] bias
            for (int idx = tid; idx < 10 * 64 / 4; idx += 256)
                *reinterpret_cast<f32x4*>(&w1[idx * 4]) = *reinterpret_cast<const f32x4*>((idx < 144 ? p.w1a : p.b1a - 576) + idx * 4);
            __syncthreads();
            const size_t img_b = (size_t)b * p.H * p.W;
            for (int idx = tid; idx < CV_HALO_PIX * 16; idx += 256) {
                const int pix = idx >> 4, q = idx & 15;
                const int gy = y0 - 1 + pix / CV_HW, gx = x0 - 1 + pix % CV_HW;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                    v = *reinterpret_cast<const f32x4*>(&w1[576 + q * 4]);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int iy = gy + ky - 1, ix = gx + kx - 1;
                            float a = 0.f;
                            if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                                a = p.img_is_u8 ? (float)reinterpret_cast<const uint8_t*>(p.img)[img_b + (size_t)iy * p.W + ix] / 255.0f
                                                : reinterpret_cast<const float*>(p.img)[img_b + (size_t)iy * p.W + ix];
                            const f32x4 wt = *reinterpret_cast<const f32x4*>(&w1[(ky * 3 + kx) * 64 + q * 4]);
                            v.x = fmaf(a, wt.x, v.x);
                            v.y = fmaf(a, wt.y, v.y);
                            v.z = fmaf(a, wt.z, v.z);
                            v.w = fmaf(a, wt.w, v.w);
                        }
                    }
                    v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
                }
                *reinterpret_cast<f32x4*>(&lds[(pix / CV_HW) * CV_ROW_PITCH + (pix % CV_HW) * MT_LDS_ROW + q * 4]) = v;
            }
        } else {
            for (int idx = tid; idx < CV_HALO_PIX * 16; idx += 256) {
                const int pix = idx >> 4, q = idx & 15;
                const int gy = y0 - 1 + pix / CV_HW, gx = x0 - 1 + pix % CV_HW;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
                    v = *reinterpret_cast<const f32x4*>(in_b + ((size_t)gy * p.W + gx) * p.in_stride + cc * 64 + q * 4);
                *reinterpret_cast<f32x4*>(&lds[(pix / CV_HW) * CV_ROW_PITCH + (pix % CV_HW) * MT_LDS_ROW + q * 4]) = v;
            }
        }
        __syncthreads();
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = (tap / 3) * CV_ROW_PITCH + (tap % 3) * MT_LDS_ROW;
            f32x4 a0 = *reinterpret_cast<const f32x4*>(&lds[a_base0 + toff]);
            f32x4 a1 = *reinterpret_cast<const f32x4*>(&lds[a_base1 + toff]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                const int nxt = (kstep + 3 < total_steps) ? kstep + 3 : total_steps - 1;
                const f32x4 bnext = mt_load_b(wp, nxt, wn, lane);
                f32x4 a0n = a0, a1n = a1;
                if (c8 < 7) {
                    a0n = *reinterpret_cast<const f32x4*>(&lds[a_base0 + toff + (c8 + 1) * 8]);
                    a1n = *reinterpret_cast<const f32x4*>(&lds[a_base1 + toff + (c8 + 1) * 8]);
                }
                mt_step(acc0, acc1, a0, a1, bcur);
                bcur = bnxt;
                bnxt = bnx2;
                bnx2 = bnext;
                a0 = a0n, a1 = a1n;
                ++kstep;
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (c8 < 7) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
        }
    }

    const bool cvalid = cout < p.Cout;
    if (!p.pool) {
        float* __restrict__ out_b = p.out + (size_t)b * p.H * p.W * p.out_stride + p.out_coff + cout;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mt_acc_row(r, lane);
                const int y = y0 + 4 * wm + 2 * t + (row >> 4), x = x0 + (row & 15);
                float v = t ? acc1[r] : acc0[r];
                if (p.relu) v = fmaxf(v, 0.f);
                if (cvalid && y < p.H && x < p.W) out_b[((size_t)y * p.W + x) * p.out_stride] = v;
            }
        }
    } else {
        // fused 2x2/stride-2 max-pool (floor): the 2x2 window of an output pixel is lane-local in the 32x32
        // accumulator layout (columns pair up in regs r, r+1; the two tile rows in regs r, r+8).
        const int Ho = p.H >> 1, Wo = p.W >> 1;
        float* __restrict__ out_b = p.out + (size_t)b * Ho * Wo * p.out_stride + p.out_coff + cout;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int py = (y0 + 4 * wm + 2 * t) >> 1;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int rr = 0; rr < 4; rr += 2) {
                    const int r = rr + 4 * g;
                    float v;
                    if (t == 0)
                        v = fmaxf(fmaxf(acc0[r], acc0[r + 1]), fmaxf(acc0[r + 8], acc0[r + 9]));
                    else
                        v = fmaxf(fmaxf(acc1[r], acc1[r + 1]), fmaxf(acc1[r + 8], acc1[r + 9]));
                    if (p.relu) v = fmaxf(v, 0.f);
                    const int px = (x0 + rr + 8 * g + 4 * kh) >> 1;
                    if (cvalid && py < Ho && px < Wo) out_b[((size_t)py * Wo + px) * p.out_stride] = v;
                }
            }
        }
    }
}

int launch_conv3x3(const ConvParams& pin, hipStream_t stream) {
    ConvParams p = pin;
    GTSFM_CHECK_ARG(p.Cin % 64 == 0 && p.Cin >= 64, "conv3x3: Cin must be a multiple of 64 (got %d)", p.Cin);
    GTSFM_CHECK_ARG(p.in_stride % 4 == 0 && p.in_coff % 4 == 0, "conv3x3: input stride/offset must be 16-byte aligned");
    p.tiles_x = ceil_div(p.W, CV_TW);
    p.tiles_y = ceil_div(p.H, CV_TH);
    dim3 grid(p.B * p.tiles_x * p.tiles_y, ceil_div(p.Cout, 64));
    size_t lds_bytes = (size_t)CV_HALO_FLOATS * sizeof(float);
    if (p.img) {
        GTSFM_CHECK_ARG(p.Cin == 64 && p.w1a && p.b1a, "conv3x3: the fused first layer needs Cin == 64 and conv1a weights");
        lds_bytes += 640 * sizeof(float);
        hipLaunchKernelGGL(conv3x3_mfma_kernel<true>, grid, dim3(256), lds_bytes, stream, p);
    } else {
        hipLaunchKernelGGL(conv3x3_mfma_kernel<false>, grid, dim3(256), lds_bytes, stream, p);
    }
    GTSFM_CHECK_LAUNCH("conv3x3_mfma_kernel");
    return GTSFM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// GEMM: C[M, N] = epilogue(A[M, K] * W[N, K]^T) with packed W. Workgroup tile 128 rows x 128 columns (4 waves as
// 2 (M) x 2 (N), wave tile 64 x 64 = four 32x32 accumulators), K staged through LDS in 64-deep chunks.
// Block order: blockIdx.x walks the column blocks of one row tile first, so the A tile staged by neighbouring
// workgroups is the same and stays L2-resident.
// ---------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ void gemm_step(f32x16& c00, f32x16& c01, f32x16& c10, f32x16& c11, const f32x4 a0, const f32x4 a1,
                                          const f32x4 b0, const f32x4 b1) {
    // The weight fragment is passed as the MFMA's A operand and the activation fragment as its B operand: the
    // accumulator then holds, per lane, ONE output row and 16 output columns in groups of 4 consecutive ones, so the
    // epilogue moves 16 bytes per lane per instruction (4x fewer store instructions than the row-per-register form;
    // the store tail is issue-bound, cdna_hip_programming.md T21).
#define GS(e)                                                                 \
    c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.e, a0.e, c00, 0, 0, 0);     \
    c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1.e, a0.e, c01, 0, 0, 0);     \
    c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.e, a1.e, c10, 0, 0, 0);     \
    c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1.e, a1.e, c11, 0, 0, 0);
    GS(x) GS(y) GS(z) GS(w)
#undef GS
}


#include "trace.h"

template <bool HAS_RES>
__global__ __launch_bounds__(256, 2) void gemm_mfma_kernel(GemmParams p) {
    // A workgroup owns a 128-row tile and walks p.nb_per_wg 128-column blocks of the output with ONE software
    // pipeline: the loop runs over (column block, 64-deep K chunk) pairs; while the waves run the MFMAs of one chunk
    // out of one LDS buffer, the A rows of the next chunk travel global/L2 -> registers (issued before the MFMAs) ->
    // the other buffer (written after them). One barrier per chunk; the prologue is paid once per workgroup and the
    // epilogue stores of a column block drain under the next block's MFMAs.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int M = p.m_dev ? *p.m_dev : p.M;
    const int mt = blockIdx.y;
    const int m0 = mt * MT_TILE_M;
    if (p.tile_cnt_idx) {
        // ragged batch with 128-row-aligned sequences: this tile belongs to one sequence whose live row count sits in
        // device memory (LightGlue early stop / point pruning shrink it without host synchronisation)
        const int c = p.live_counts[p.tile_cnt_idx[mt]];
        const int r0 = p.tile_row0[mt];
        if (r0 >= c) return;
        M = min(M, m0 + c - r0);
    }
    if (m0 >= M) return;
    const int nblocks = (p.N + 63) >> 6;                     // 64-column blocks in the output
    const int cb0 = blockIdx.x * p.nb_per_wg;                // first 128-column block of this workgroup
    const int ncb = min(p.nb_per_wg, ((p.N + 127) >> 7) - cb0);
    const int total_steps = p.K >> 3;
    const int nchunks = (p.K + 63) >> 6;
    const int j = lane & 31, kh = lane >> 5;
    const int a_base0 = (64 * wm + j) * MT_LDS_ROW + kh * 4;
    const int a_base1 = a_base0 + 32 * MT_LDS_ROW;
    constexpr int BUF = MT_TILE_M * MT_LDS_ROW;

    // staging: thread t moves float4 #(t + 256 i), i = 0..7: row = idx / 16, 16-byte column = idx % 16
    f32x4 st[8];
    const int srow = tid >> 4, sq = tid & 15;
    auto stage_load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int gr = m0 + srow + 16 * i, gk = k0 + sq * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gr < M && gk < p.K && !(p.debug & 2)) v = *reinterpret_cast<const f32x4*>(p.A + (size_t)gr * p.lda + gk);
            st[i] = v;
        }
    };
    auto stage_store = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(&buf[(srow + 16 * i) * MT_LDS_ROW + sq * 4]) = st[i];
    };
    // B operands: global k-step index g = column-block index * total_steps + k-step; they run two steps ahead so that
    // their (in-order) wait never has to cover the younger A-row loads
    const int gsteps = ncb * total_steps;
    auto ldb = [&](int g, int half) {
        g = g < gsteps ? g : gsteps - 1;
        const int cbi = g / total_steps, s = g - cbi * total_steps;
        int nb = (cb0 + cbi) * 2 + wn;
        nb = nb < nblocks ? nb : nblocks - 1;
        return mt_load_b(p.wpack + (size_t)nb * total_steps * MT_PACK_STEP_FLOATS, s, half, lane);
    };

    __builtin_amdgcn_s_setprio(3);
    // bias of this workgroup's columns -> LDS (zeros without a bias / beyond N; the array is padded to a multiple of 64)
    float* bias_lds = lds + 2 * BUF;
    for (int i = tid; i < ncb * 128; i += 256) {
        const int col = cb0 * 128 + i;
        bias_lds[i] = (p.bias && col < nblocks * 64) ? p.bias[col] : 0.f;
    }
    stage_load(0);
    f32x4 b0c = ldb(0, 0), b1c = ldb(0, 1), b0n = ldb(1, 0), b1n = ldb(1, 1);
    stage_store(lds);
    __syncthreads();
    f32x16 c00, c01, c10, c11;
    int g = 0;
    GT_DECL
    const int iters = ncb * nchunks;
    int it = 0;
    for (int cbi = 0; cbi < ncb; ++cbi) {
        const int nb = (cb0 + cbi) * 2 + wn;  // 64-column block of this wave
        const bool active = nb < nblocks;     // waves beyond N still help staging
        const int colb = nb * 64 + 4 * kh;  // + 32 * (column half) + 8 * (r >> 2) + (r & 3)
        // accumulators start at the bias, read from the LDS copy made in the prologue: a global load here would queue
        // behind the previous block's 16 stores (vmcnt is in order) and wait ~5 k cycles for their acknowledgement
        {
            const float* bl = bias_lds + cbi * 128 + wn * 64 + 4 * kh;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b0v = *reinterpret_cast<const f32x4*>(bl + 8 * q);
                const f32x4 b1v = *reinterpret_cast<const f32x4*>(bl + 32 + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    c00[4 * q + e] = c10[4 * q + e] = b0v[e];
                    c01[4 * q + e] = c11[4 * q + e] = b1v[e];
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        GT_SEG(4)
        for (int c = 0; c < nchunks; ++c, ++it) {
            const float* buf = lds + (it & 1) * BUF;
            if (it + 1 < iters) stage_load(((c + 1 < nchunks) ? c + 1 : 0) * 64);
            const int nsteps = min(8, total_steps - c * 8);
            if (nsteps == 8) {
                f32x4 a0 = *reinterpret_cast<const f32x4*>(&buf[a_base0]);
                f32x4 a1 = *reinterpret_cast<const f32x4*>(&buf[a_base1]);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // own group for the chunk's first two A reads: the per-step groups below stay aligned
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8) {
                    const f32x4 b0f = ldb(g + 2, 0), b1f = ldb(g + 2, 1);
                    f32x4 a0n = a0, a1n = a1;
                    if (c8 < 7) {  // A fragments run one k-step ahead (within the chunk)
                        a0n = *reinterpret_cast<const f32x4*>(&buf[a_base0 + (c8 + 1) * 8]);
                        a1n = *reinterpret_cast<const f32x4*>(&buf[a_base1 + (c8 + 1) * 8]);
                    }
                    gemm_step(c00, c01, c10, c11, a0, a1, b0c, b1c);
                    b0c = b0n, b1c = b1n, b0n = b0f, b1n = b1f;
                    a0 = a0n, a1 = a1n;
                    ++g;
                    // pin the issue order per k-step: the two B loads (consumed two steps later) and the two A reads (next
                    // step) go out FIRST, then the 16 MFMAs -- left alone, the scheduler sinks the loads next to their use
                    // and every step stalls on vmcnt / lgkmcnt
                    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                    if (c8 < 7) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
                }
            } else {
                for (int c8 = 0; c8 < nsteps; ++c8) {
                    const f32x4 b0f = ldb(g + 2, 0), b1f = ldb(g + 2, 1);
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(&buf[a_base0 + c8 * 8]);
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(&buf[a_base1 + c8 * 8]);
                    gemm_step(c00, c01, c10, c11, a0, a1, b0c, b1c);
                    b0c = b0n, b1c = b1n, b0n = b0f, b1n = b1f;
                    ++g;
                }
            }
            // Everything up to the next chunk's first MFMA (epilogue, LDS staging, barrier, accumulator re-initialisation)
            // is VALU / memory-issue work that loses the issue arbitration against the MFMAs of the SIMD's other wave
            // and crawls at ~40 cycles per instruction unless it runs at raised priority; while it lasts this wave
            // feeds the matrix pipe nothing.
            GT_SEG(c == 0 ? 1 : 0)
            __builtin_amdgcn_s_setprio(3);
            if (c == nchunks - 1 && active && !(p.debug & 1)) {
                // the row offsets are loop-invariant; keep the compiler from hoisting 64 addresses out of the column-block
                // loop (they would live across the whole pipeline and spill): make the base opaque per block
                int row_base = m0 + 64 * wm + j;
                asm volatile("" : "+v"(row_base));
                const bool vec_ok = ((p.N & 3) == 0) && ((p.ldc & 3) == 0) && ((p.c_coff & 3) == 0) && (!HAS_RES || (p.ldres & 3) == 0);
                // scale / ReLU as whole-tile passes under uniform branches (no per-element selects)
                if (p.alpha != 1.0f) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) c00[r] *= p.alpha, c01[r] *= p.alpha, c10[r] *= p.alpha, c11[r] *= p.alpha;
                }
                if (p.relu) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        c00[r] = fmaxf(c00[r], 0.f), c01[r] = fmaxf(c01[r], 0.f), c10[r] = fmaxf(c10[r], 0.f), c11[r] = fmaxf(c11[r], 0.f);
                }
                if (vec_ok) {
                    // 16 stores of 16 bytes per lane: group i = 4 t + q, t = 2 (row half) + (column half), q = 8-column step.
                    // The plain and the residual variant are SEPARATE code: merged, the compiler guards every store with
                    // s_waitcnt vmcnt(0) for the residual load that might precede it, and since stores count in vmcnt too,
                    // each store then waits for the previous store's acknowledgement (22 k cycles per block instead of 2 k).
                    auto group = [&](int i) -> f32x4 {
                        const int t = i >> 2, q = i & 3;
                        const f32x16& ct = (t == 0) ? c00 : (t == 1) ? c01 : (t == 2) ? c10 : c11;
                        return f32x4{ct[4 * q], ct[4 * q + 1], ct[4 * q + 2], ct[4 * q + 3]};
                    };
                    auto gcol = [&](int i) { return colb + 32 * ((i >> 2) & 1) + 8 * (i & 3); };
                    // one divergent region per row half (a lane = a row), straight-line loads / stores inside: exec-masked
                    // branches around single accesses would again make the compiler wait for vmcnt(0) at every join
#pragma unroll
                    for (int hrow = 0; hrow < 2; ++hrow) {
                        const int row = row_base + 32 * hrow;
                        if (row < M) {
                            float* crow = p.C + (size_t)row * p.ldc + p.c_coff;
                            if (!HAS_RES) {
#pragma unroll
                                for (int i = 8 * hrow; i < 8 * hrow + 8; ++i)
                                    if (gcol(i) < p.N) *reinterpret_cast<f32x4*>(crow + gcol(i)) = group(i);
                            } else {
                                // residual loads run one group ahead of the stores: each wait covers one load that is older
                                // than every store still in flight
                                const float* rrow = p.res + (size_t)row * p.ldres;
                                const int last_col = p.N - 4;  // clamp instead of predicating the load (uniform, always valid)
                                f32x4 cur = *reinterpret_cast<const f32x4*>(rrow + min(gcol(8 * hrow), last_col));
#pragma unroll
                                for (int i = 8 * hrow; i < 8 * hrow + 8; ++i) {
                                    f32x4 nxt = cur;
                                    if (i + 1 < 8 * hrow + 8) nxt = *reinterpret_cast<const f32x4*>(rrow + min(gcol(i + 1), last_col));
                                    const f32x4 v = cur + group(i);
                                    if (gcol(i) < p.N) *reinterpret_cast<f32x4*>(crow + gcol(i)) = v;
                                    cur = nxt;
                                }
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int row = row_base + 32 * (t >> 1);
                        if (row >= M) continue;
                        const f32x16& ct = (t == 0) ? c00 : (t == 1) ? c01 : (t == 2) ? c10 : c11;
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const int col = colb + 32 * (t & 1) + 8 * gq;
                            float* cp = p.C + (size_t)row * p.ldc + p.c_coff + col;
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (col + e < p.N) cp[e] = HAS_RES ? p.res[(size_t)row * p.ldres + col + e] + ct[4 * gq + e] : ct[4 * gq + e];
                        }
                    }
                }
            }
            GT_SEG(2)
            if (it + 1 < iters) {
                stage_store(lds + ((it + 1) & 1) * BUF);
                __syncthreads();
            }
            GT_SEG(3)
            if (c + 1 < nchunks) __builtin_amdgcn_s_setprio(0);  // (a new column block first re-initialises the accumulators)
        }
    }
#ifdef GTSFM_TRACE
    if (lane == 0 && g_gemm_trace) {
        unsigned long long* o = g_gemm_trace + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
        for (int k = 0; k < 5; ++k) o[k] = gseg[k];
        o[5] = (unsigned)__builtin_amdgcn_s_memtime() - gt_begin;
        o[6] = ncb;
        o[7] = nchunks;
    }
#endif
}


int launch_gemm(const GemmParams& p, hipStream_t stream) {
    GTSFM_CHECK_ARG(p.K % 8 == 0 && p.K >= 8, "gemm: K must be a multiple of 8 (got %d)", p.K);
    GTSFM_CHECK_ARG(p.lda % 4 == 0, "gemm: lda must be a multiple of 4 (got %d)", p.lda);
    if (p.M <= 0) return GTSFM_OK;
    // column blocks per workgroup: as many as possible (prologue paid once, A tile L2-hot) while the grid still holds
    // >= 2 workgroups per CU
    // Both operands by LDS-DMA when the caller has row-major weights and K is a multiple of the 32-deep stage
    // (GTSFM_GEMM=mfma forces the register-staged kernel below for A/B measurements)
    if (p.wraw && gemm_uses_dma(p.K, p.ldw)) return launch_gemm_dma(p, stream);
    GTSFM_CHECK_ARG(p.wpack, "gemm: no packed weights for the register-staged kernel");
    GemmParams q = p;
    const int ncb_total = ceil_div(p.N, 128), mtiles = ceil_div(p.M, MT_TILE_M);
    int nbw = ncb_total;
    while (nbw > 1 && (long long)mtiles * ceil_div(ncb_total, nbw) < 512) --nbw;
    static const char* env = getenv("GTSFM_GEMM_NB");
    if (env && atoi(env) > 0) nbw = atoi(env);
    q.nb_per_wg = nbw;
    static const char* dbg = getenv("GTSFM_GEMM_DEBUG");
    q.debug = dbg ? atoi(dbg) : 0;
    dim3 grid(ceil_div(ncb_total, nbw), mtiles);
    size_t lds_bytes = ((size_t)2 * MT_TILE_M * MT_LDS_ROW + (size_t)nbw * 128) * sizeof(float);  // two A chunks + the bias
    static const char* pad = getenv("GTSFM_GEMM_LDS_PAD");  // developer switch: extra LDS bytes to force 1 workgroup per CU
    if (pad) lds_bytes += (size_t)atoi(pad);
    if (q.res)
        hipLaunchKernelGGL(gemm_mfma_kernel<true>, grid, dim3(256), lds_bytes, stream, q);
    else
        hipLaunchKernelGGL(gemm_mfma_kernel<false>, grid, dim3(256), lds_bytes, stream, q);
    GTSFM_CHECK_LAUNCH("gemm_mfma_kernel");
    return GTSFM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Pack a row-major activation matrix B[N, K] (device) into the packed weight layout, so that products of two
// activation matrices (score GEMMs: superglue.py:257) run through the same GEMM kernel.
// ---------------------------------------------------------------------------------------------------------------

__global__ void pack_rows_kernel(const float* __restrict__ B, int ldb, int N, const int* n_dev, int K, float* __restrict__ out) {
    const int Nr = n_dev ? *n_dev : N;
    const int total_steps = K >> 3;
    const size_t total = (size_t)ceil_div(N, 64) * total_steps * 128;  // float4 elements
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = idx & 63;
        const int wn = (idx >> 6) & 1;
        const size_t rest = idx >> 7;
        const int kstep = rest % total_steps;
        const int nb = rest / total_steps;
        const int n = nb * 64 + wn * 32 + (lane & 31);
        const int k = kstep * 8 + (lane >> 5) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < Nr) v = *reinterpret_cast<const f32x4*>(B + (size_t)n * ldb + k);
        reinterpret_cast<f32x4*>(out)[idx] = v;
    }
}

int launch_pack_rows(const float* B, int ldb, int N, const int* n_dev, int K, float* out, hipStream_t stream) {
    GTSFM_CHECK_ARG(K % 8 == 0 && ldb % 4 == 0, "pack_rows: K %% 8 and ldb %% 4 must be 0");
    if (N <= 0) return GTSFM_OK;
    const size_t total = (size_t)ceil_div(N, 64) * (K >> 3) * 128;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_rows_kernel, dim3(blocks), dim3(256), 0, stream, B, ldb, N, n_dev, K, out);
    GTSFM_CHECK_LAUNCH("pack_rows_kernel");
    return GTSFM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Host-side weight packing (done once at load time).
// ---------------------------------------------------------------------------------------------------------------

size_t packed_conv3x3_floats(int cin, int cout) { return (size_t)ceil_div(cout, 64) * (cin / 64) * 72 * MT_PACK_STEP_FLOATS; }
size_t packed_linear_floats(int k, int n) { return (size_t)ceil_div(n, 64) * (k / 8) * MT_PACK_STEP_FLOATS; }

// w: [cout][cin][3][3] (torch Conv2d layout). Zero-fills padded output channels.
void pack_conv3x3_weights(const float* w, int cin, int cout, float* out) {
    const int nblocks = ceil_div(cout, 64), nchunks = cin / 64;
    size_t o = 0;
    for (int nb = 0; nb < nblocks; ++nb)
        for (int cc = 0; cc < nchunks; ++cc)
            for (int tap = 0; tap < 9; ++tap)
                for (int c8 = 0; c8 < 8; ++c8)
                    for (int wn = 0; wn < 2; ++wn)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 4; ++e) {
                                const int n = nb * 64 + wn * 32 + (lane & 31);
                                const int c = cc * 64 + c8 * 8 + (lane >> 5) * 4 + e;
                                out[o++] = (n < cout) ? w[((size_t)n * cin + c) * 9 + tap] : 0.f;
                            }
}

// w: [n][k_real] row-major (torch Linear / Conv1d(k=1) layout); k is the padded depth (multiple of 8).
void pack_linear_weights(const float* w, int k_real, int k, int n, float* out) {
    const int nblocks = ceil_div(n, 64), steps = k / 8;
    size_t o = 0;
    for (int nb = 0; nb < nblocks; ++nb)
        for (int s = 0; s < steps; ++s)
            for (int wn = 0; wn < 2; ++wn)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int nn = nb * 64 + wn * 32 + (lane & 31);
                        const int kk = s * 8 + (lane >> 5) * 4 + e;
                        out[o++] = (nn < n && kk < k_real) ? w[(size_t)nn * k_real + kk] : 0.f;
                    }
}
