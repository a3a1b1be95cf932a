// fp32-MFMA 3x3 convolution for gfx950 (NHWC, LDS halo tile, on-the-fly im2col) with fused bias / ReLU / 2x2 max-pool
// epilogues, and the host-side weight packers. The GEMMs live in gemm_mfma_kernels.hip (register-staged, packed weights) and
// gemm_dma_kernels.hip (LDS-DMA, row-major weights).
// Replaces the ATen conv2d / conv1d / linear / max_pool2d calls of
//   thirdparty/SuperGluePretrainedNetwork/models/superpoint.py:148-162,190-191
//   thirdparty/SuperGluePretrainedNetwork/models/superglue.py:49-60,98-107,110-119,254
// See mfma_tiles.h for the tiling and the packed weight layout.

#include <stdlib.h>

#include "conv_kernels.h"
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
#define CV_PATCH_W (CV_TW + 4)  // gray-image patch under the halo tile of the fused first layer: (CV_TH + 4) x (CV_TW + 4) floats

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
            // the gray image under the halo tile and one pixel around it, as conv1a sees it: u8 / 255 (or the float image), zero
            // outside the image (conv1a's padding). One load, one conversion and one division per PIXEL here; round 1-2 did all
            // three per (halo pixel, tap, 4-channel group) -- 9 x 16 times as often, with global-load latency in the inner loop --
            // and the fused layer, 43 % of SuperPoint's FLOPs, ran at 0.63 of the MFMA peak where the plain 64 -> 64 layers reach 0.8
            float* patch = w1 + 640;           // [CV_TH + 4][CV_PATCH_W]
            for (int idx = tid; idx < 10 * 64 / 4; idx += 256)
                *reinterpret_cast<f32x4*>(&w1[idx * 4]) = *reinterpret_cast<const f32x4*>((idx < 144 ? p.w1a : p.b1a - 576) + idx * 4);
            const size_t img_b = (size_t)b * p.H * p.W;
            if (tid < (CV_TH + 4) * CV_PATCH_W) {
                const int iy = y0 - 2 + tid / CV_PATCH_W, ix = x0 - 2 + tid % CV_PATCH_W;
                float a = 0.f;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                    a = p.img_is_u8 ? (float)reinterpret_cast<const uint8_t*>(p.img)[img_b + (size_t)iy * p.W + ix] / 255.0f
                                    : reinterpret_cast<const float*>(p.img)[img_b + (size_t)iy * p.W + ix];
                patch[tid] = a;
            }
            __syncthreads();
            // relu(conv1a) of the 180 halo pixels ON THE MATRIX PIPE (round 4): per pixel and channel v = bias, then v = fmaf(patch[tap], w[tap], v)
            // for the nine taps in order -- exactly what v_mfma_f32_32x32x2_f32 computes as a k-ordered fmaf chain from the accumulator's
            // start value (k = tap; a tenth, zero tap fills the fifth instruction: fmaf(0, 0, v) = v). 6 pixel blocks x 2 channel blocks x 5
            // MFMAs = 60 per workgroup replace 2880 x 36 VALU fmas (rounds 1-3: the fused layer ran at 0.78 - 0.80 of the MFMA peak where the
            // plain pooled 64 -> 64 layer reaches 0.85; the difference was this staging loop). Bit-identical to the VALU form and to
            // conv1a_kernel. Weights = A operand (rows = channels), patch values = B operand (columns = pixels): a lane owns a pixel and
            // receives 4 x 4 consecutive channels, written as four 16-byte stores into the halo tile.
            for (int chain = wave; chain < 12; chain += 4) {
                const int pb = chain >> 1, cb = chain & 1;
                const int pix = 32 * pb + j;
                const int cpix = pix < CV_HALO_PIX ? pix : CV_HALO_PIX - 1;
                const int hy = cpix / CV_HW, hx = cpix % CV_HW;
                const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
                const float* pa = patch + hy * CV_PATCH_W + hx;  // image pixel (gy - 1, gx - 1)
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = w1[576 + 32 * cb + (r & 3) + 8 * (r >> 2) + 4 * kh];
#pragma unroll
                for (int s5 = 0; s5 < 5; ++s5) {
                    const int tap = 2 * s5 + kh;  // k index of this lane half
                    const bool live = tap < 9;
                    const int tt = live ? tap : 0;
                    const float bval = live ? pa[(tt / 3) * CV_PATCH_W + tt % 3] : 0.f;
                    const float aval = live ? w1[tt * 64 + 32 * cb + j] : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aval, bval, acc, 0, 0, 0);
                }
                const bool inside = pix < CV_HALO_PIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                if (pix < CV_HALO_PIX) {
                    float* dst = &lds[hy * CV_ROW_PITCH + hx * MT_LDS_ROW + 32 * cb + 4 * kh];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        f32x4 v = {fmaxf(acc[4 * g4], 0.f), fmaxf(acc[4 * g4 + 1], 0.f), fmaxf(acc[4 * g4 + 2], 0.f), fmaxf(acc[4 * g4 + 3], 0.f)};
                        if (!inside) v = f32x4{0.f, 0.f, 0.f, 0.f};  // conv1b's zero padding, not relu(conv1a) of an outside pixel
                        *reinterpret_cast<f32x4*>(dst + 8 * g4) = v;
                    }
                }
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
    if (!p.pool && ((p.Cout | p.out_stride | p.out_coff) & 3) == 0) {
        // Unpooled output, transposed through LDS (round 4): in the accumulator layout a lane owns ONE output channel of 32 pixels, so a
        // store instruction moves 64 x 4 B and a wave needs 32 of them (the unpooled 64 -> 64 layer ran 5 points below the pooled one:
        // 0.80 against 0.85, tools/bench_conv.py). Through a 32 x 32 scratch tile per wave -- its slice of the halo tile, which nobody
        // reads any more -- 8 lanes cover the 128 contiguous bytes of a pixel's 32 channels: 8 store instructions of 64 x 16 B. Same values.
        __syncthreads();  // every wave is done with the halo tile
        float* scr = lds + wave * 1024;
        float* __restrict__ out_t = p.out + (size_t)b * p.H * p.W * p.out_stride + p.out_coff + nb * 64 + wn * 32;
        const int tp = lane >> 3, tc = lane & 7;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = t ? acc1[r] : acc0[r];
                if (p.relu) v = fmaxf(v, 0.f);
                scr[mt_acc_row(r, lane) * 32 + j] = v;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = tp + 8 * i;  // pixel of the wave's 32: tile row 4 wm + 2 t + (row >> 4), column row & 15
                const f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * 32 + 4 * tc);
                const int y = y0 + 4 * wm + 2 * t + (row >> 4), x = x0 + (row & 15);
                if (y < p.H && x < p.W && nb * 64 + wn * 32 + 4 * tc < p.Cout)
                    *reinterpret_cast<f32x4*>(out_t + ((size_t)y * p.W + x) * p.out_stride + 4 * tc) = v;
            }
        }
    } else if (!p.pool) {
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
        lds_bytes += (640 + (CV_TH + 4) * CV_PATCH_W) * sizeof(float);
        hipLaunchKernelGGL(conv3x3_mfma_kernel<true>, grid, dim3(256), lds_bytes, stream, p);
    } else {
        hipLaunchKernelGGL(conv3x3_mfma_kernel<false>, grid, dim3(256), lds_bytes, stream, p);
    }
    GTSFM_CHECK_LAUNCH("conv3x3_mfma_kernel");
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
