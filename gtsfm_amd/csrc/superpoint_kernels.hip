// SuperPoint-specific gfx950 kernels: first-layer convolution, detector softmax + depth-to-space, simple-NMS,
// row-major keypoint extraction, bilinear descriptor sampling + L2 normalisation.
// Replaces the ATen op sequences of thirdparty/SuperGluePretrainedNetwork/models/superpoint.py:47-92,148,163-196.
// All of these are HBM/L2-bound comparison or elementwise work: coalesced NHWC access, wave-shuffle reductions,
// LDS-tiled separable max-pools. Compiled with -ffp-contract=off so the scalar fp32 expressions below keep the
// reference's rounding sequence.

#include "superpoint_kernels.h"

// ---------------------------------------------------------------------------------------------------------------
// conv1a: 3x3, 1 -> 64 channels, + bias + ReLU (superpoint.py:119,148). HBM-write-bound (256 B per pixel).
// Thread = (pixel, 4 output channels); 16 threads cover the 64 channels of a pixel -> 1 KiB coalesced per wave.
// wpack: [9 taps][64], bias: [64]. Input is fp32 [B][H][W] or uint8 [B][H][W] (converted as
// astype(float32) / 255.0, gtsfm/frontend/detector_descriptor/superpoint.py:73-75).
// ---------------------------------------------------------------------------------------------------------------

template <bool U8>
__global__ __launch_bounds__(256) void conv1a_kernel(const void* __restrict__ img, int H, int W, const float* __restrict__ wpack,
                                                     const float* __restrict__ bias, float* __restrict__ out) {
    const int q = threadIdx.x & 15;
    const int pl = threadIdx.x >> 4;  // 16 pixels per pass
    const int y = blockIdx.y, b = blockIdx.z;
    f32x4 w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const f32x4*>(wpack + t * 64 + q * 4);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + q * 4);
    const size_t img_b = (size_t)b * H * W;
    for (int x = blockIdx.x * 64 + pl; x < min(W, blockIdx.x * 64 + 64); x += 16) {
        f32x4 acc = bv;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int gy = y + ky - 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int gx = x + kx - 1;
                float v = 0.f;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                    if (U8)
                        v = (float)reinterpret_cast<const uint8_t*>(img)[img_b + (size_t)gy * W + gx] / 255.0f;
                    else
                        v = reinterpret_cast<const float*>(img)[img_b + (size_t)gy * W + gx];
                }
                const f32x4 wt = w[ky * 3 + kx];
                acc.x = fmaf(v, wt.x, acc.x);
                acc.y = fmaf(v, wt.y, acc.y);
                acc.z = fmaf(v, wt.z, acc.z);
                acc.w = fmaf(v, wt.w, acc.w);
            }
        }
        acc.x = fmaxf(acc.x, 0.f);
        acc.y = fmaxf(acc.y, 0.f);
        acc.z = fmaxf(acc.z, 0.f);
        acc.w = fmaxf(acc.w, 0.f);
        *reinterpret_cast<f32x4*>(out + ((img_b + (size_t)y * W + x) * 64 + q * 4)) = acc;
    }
}

int launch_conv1a(const void* img, int is_u8, int B, int H, int W, const float* wpack, const float* bias, float* out,
                  hipStream_t stream) {
    dim3 grid(ceil_div(W, 64), H, B);
    if (is_u8)
        hipLaunchKernelGGL(conv1a_kernel<true>, grid, dim3(256), 0, stream, img, H, W, wpack, bias, out);
    else
        hipLaunchKernelGGL(conv1a_kernel<false>, grid, dim3(256), 0, stream, img, H, W, wpack, bias, out);
    GTSFM_CHECK_LAUNCH("conv1a_kernel");
    return GTSFM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Detector head: softmax over the 65 logits of a cell, drop the dustbin, depth-to-space 8x8
// (superpoint.py:163-166). One wave per cell: lane c owns channel c -> pixel (8*cy + c/8, 8*cx + c%8).
// ---------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void softmax_d2s_kernel(const float* __restrict__ logits, int ld, int ncells_total, int Hc,
                                                          int Wc, float* __restrict__ scores) {
    const int lane = threadIdx.x & 63;
    const int cell = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (cell >= ncells_total) return;
    const float* row = logits + (size_t)cell * ld;
    const float v = row[lane];
    const float d = row[64];
    const float m = fmaxf(wave_max(v), d);
    const float e = expf(v - m);
    const float ed = expf(d - m);
    const float s = wave_sum(e) + ed;
    const int cx = cell % Wc;
    const int cy = (cell / Wc) % Hc;
    const int b = cell / (Wc * Hc);
    const int W8 = Wc * 8;
    scores[((size_t)b * Hc * 8 + cy * 8 + (lane >> 3)) * W8 + cx * 8 + (lane & 7)] = e / s;
}

int launch_softmax_d2s(const float* logits, int ld, int B, int Hc, int Wc, float* scores, hipStream_t stream) {
    const int ncells = B * Hc * Wc;
    if (ncells == 0) return GTSFM_OK;
    hipLaunchKernelGGL(softmax_d2s_kernel, dim3(ceil_div(ncells, 4)), dim3(256), 0, stream, logits, ld, ncells, Hc, Wc, scores);
    GTSFM_CHECK_LAUNCH("softmax_d2s_kernel");
    return GTSFM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// simple_nms (superpoint.py:47-62): comparison-only, bit-exact. Three LDS-tiled kernels, each a separable
// (2r+1)^2 max-pool with implicit -inf padding fused with the surrounding elementwise logic:
//   MODE 0: mask  = (S == pool(S))
//   MODE 1: supp  = pool(mask) > 0 ;  SS = supp ? 0 : S
//   MODE 2: mask |= (SS == pool(SS)) & !supp ;  if out: out = mask ? S : 0
// Tile: 32 x 32 outputs per 256-thread workgroup.
// ---------------------------------------------------------------------------------------------------------------

#define NMS_T 32
#define NMS_MAXR 8

template <int MODE>
__global__ __launch_bounds__(256) void nms_pool_kernel(int H, int W, int r, const float* __restrict__ S, uint8_t* __restrict__ mask,
                                                       uint8_t* __restrict__ supp, float* __restrict__ SS, float* __restrict__ out) {
    __shared__ float tile[(NMS_T + 2 * NMS_MAXR) * (NMS_T + 2 * NMS_MAXR)];
    __shared__ float hmax[(NMS_T + 2 * NMS_MAXR) * NMS_T];
    const int b = blockIdx.z;
    const size_t base = (size_t)b * H * W;
    const int x0 = blockIdx.x * NMS_T, y0 = blockIdx.y * NMS_T;
    const int TW = NMS_T + 2 * r;
    const float NEG_INF = -__builtin_inff();
    for (int idx = threadIdx.x; idx < TW * TW; idx += 256) {
        const int ly = idx / TW, lx = idx % TW;
        const int gy = y0 - r + ly, gx = x0 - r + lx;
        float v = NEG_INF;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const size_t g = base + (size_t)gy * W + gx;
            if (MODE == 0) v = S[g];
            if (MODE == 1) v = (float)mask[g];
            if (MODE == 2) v = SS[g];
        }
        tile[idx] = v;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < TW * NMS_T; idx += 256) {
        const int ly = idx / NMS_T, lx = idx % NMS_T;
        float m = NEG_INF;
        for (int k = 0; k <= 2 * r; ++k) m = fmaxf(m, tile[ly * TW + lx + k]);
        hmax[idx] = m;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NMS_T * NMS_T; idx += 256) {
        const int ly = idx / NMS_T, lx = idx % NMS_T;
        const int gy = y0 + ly, gx = x0 + lx;
        if (gy >= H || gx >= W) continue;
        float m = NEG_INF;
        for (int k = 0; k <= 2 * r; ++k) m = fmaxf(m, hmax[(ly + k) * NMS_T + lx]);
        const size_t g = base + (size_t)gy * W + gx;
        const float center = tile[(ly + r) * TW + lx + r];
        if (MODE == 0) {
            mask[g] = (center == m) ? 1 : 0;
        } else if (MODE == 1) {
            const bool sp = m > 0.f;
            supp[g] = sp ? 1 : 0;
            SS[g] = sp ? 0.f : S[g];
        } else {
            const bool sp = supp[g] != 0;
            const bool mk = (mask[g] != 0) || ((center == m) && !sp);
            mask[g] = mk ? 1 : 0;
            if (out) out[g] = mk ? S[g] : 0.f;
        }
    }
}

int launch_simple_nms(const float* S, int B, int H, int W, int radius, uint8_t* mask, uint8_t* supp, float* SS, float* out,
                      hipStream_t stream) {
    GTSFM_CHECK_ARG(radius >= 0 && radius <= NMS_MAXR, "simple_nms: radius %d not in [0, %d]", radius, NMS_MAXR);
    if (B * H * W == 0) return GTSFM_OK;
    dim3 grid(ceil_div(W, NMS_T), ceil_div(H, NMS_T), B);
    hipLaunchKernelGGL(nms_pool_kernel<0>, grid, dim3(256), 0, stream, H, W, radius, S, mask, supp, SS, (float*)nullptr);
    for (int it = 0; it < 2; ++it) {
        hipLaunchKernelGGL(nms_pool_kernel<1>, grid, dim3(256), 0, stream, H, W, radius, S, mask, supp, SS, (float*)nullptr);
        hipLaunchKernelGGL(nms_pool_kernel<2>, grid, dim3(256), 0, stream, H, W, radius, S, mask, supp, SS,
                           it == 1 ? out : (float*)nullptr);
    }
    GTSFM_CHECK_LAUNCH("nms_pool_kernel");
    return GTSFM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Keypoint extraction: nonzero(score > thr) in row-major order, gather scores, remove_borders, flip to (x, y)
// float (superpoint.py:170-178,187). Row counts -> per-image exclusive scan -> ordered per-row compaction.
// ---------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ bool kp_valid(float s, int y, int x, int H, int W, float thr, int border) {
    return (s > thr) && (y >= border) && (y < H - border) && (x >= border) && (x < W - border);
}

__global__ __launch_bounds__(256) void kp_apply_mask_kernel(float* __restrict__ nms, int H, int W, const uint8_t* __restrict__ valid, int img_h, int img_w) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= W) return;
    if (valid[((size_t)b * img_h + y) * img_w + x] != 1) nms[((size_t)b * H + y) * W + x] = 0.f;  // below any positive threshold
}

int launch_apply_keypoint_mask(float* nms, int B, int H, int W, const uint8_t* valid_mask, int img_h, int img_w, hipStream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0) return GTSFM_OK;
    hipLaunchKernelGGL(kp_apply_mask_kernel, dim3(ceil_div(W, 256), H, B), dim3(256), 0, stream, nms, H, W, valid_mask, img_h, img_w);
    GTSFM_CHECK_LAUNCH("kp_apply_mask_kernel");
    return GTSFM_OK;
}

__global__ __launch_bounds__(64) void kp_count_rows_kernel(const float* __restrict__ nms, int H, int W, float thr, int border,
                                                           int* __restrict__ rowcnt) {
    const int y = blockIdx.x, b = blockIdx.y;
    const float* row = nms + ((size_t)b * H + y) * W;
    int cnt = 0;
    for (int x = threadIdx.x; x < W; x += 64) cnt += kp_valid(row[x], y, x, H, W, thr, border) ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (threadIdx.x == 0) rowcnt[(size_t)b * H + y] = cnt;
}

__global__ __launch_bounds__(1024) void kp_scan_rows_kernel(const int* __restrict__ rowcnt, int H, int capacity, int* __restrict__ rowoff,
                                                            int* __restrict__ count, int* __restrict__ count_raw) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int y0 = 0; y0 < H; y0 += 1024) {
        const int y = y0 + threadIdx.x;
        const int v = (y < H) ? rowcnt[(size_t)b * H + y] : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        const int carry = carry_s;
        if (y < H) rowoff[(size_t)b * H + y] = carry + wbase + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wbase + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int total = carry_s;
        count_raw[b] = total;
        count[b] = total < capacity ? total : capacity;
    }
}

__global__ __launch_bounds__(64) void kp_write_rows_kernel(const float* __restrict__ nms, int H, int W, float thr, int border,
                                                           const int* __restrict__ rowoff, int capacity, float* __restrict__ kp_xy,
                                                           float* __restrict__ kp_score) {
    const int y = blockIdx.x, b = blockIdx.y;
    const float* row = nms + ((size_t)b * H + y) * W;
    int off = rowoff[(size_t)b * H + y];
    const int lane = threadIdx.x;
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        const float s = (x < W) ? row[x] : 0.f;
        const bool v = (x < W) && kp_valid(s, y, x, H, W, thr, border);
        const unsigned long long bal = __ballot(v);
        const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
        if (v && pos < capacity) {
            kp_xy[((size_t)b * capacity + pos) * 2 + 0] = (float)x;
            kp_xy[((size_t)b * capacity + pos) * 2 + 1] = (float)y;
            kp_score[(size_t)b * capacity + pos] = s;
        }
        off += __popcll(bal);
    }
}

int launch_extract_keypoints(const float* nms, int B, int H, int W, float thr, int border, int capacity, int* rowcnt, int* rowoff,
                             int* count, int* count_raw, float* kp_xy, float* kp_score, hipStream_t stream) {
    if (B == 0) return GTSFM_OK;
    if (H * W == 0) {
        if (hipMemsetAsync(count, 0, sizeof(int) * B, stream) != hipSuccess || hipMemsetAsync(count_raw, 0, sizeof(int) * B, stream) != hipSuccess)
            return GTSFM_ERR_HIP;
        return GTSFM_OK;
    }
    hipLaunchKernelGGL(kp_count_rows_kernel, dim3(H, B), dim3(64), 0, stream, nms, H, W, thr, border, rowcnt);
    hipLaunchKernelGGL(kp_scan_rows_kernel, dim3(B), dim3(1024), 0, stream, rowcnt, H, capacity, rowoff, count, count_raw);
    hipLaunchKernelGGL(kp_write_rows_kernel, dim3(H, B), dim3(64), 0, stream, nms, H, W, thr, border, rowoff, capacity, kp_xy,
                       kp_score);
    GTSFM_CHECK_LAUNCH("kp_extract kernels");
    return GTSFM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Descriptor sampling (superpoint.py:80-92,192,195-196): channel-L2-normalise the dense descriptors, bilinear
// grid_sample with align_corners=True and zero padding at the keypoint, L2-normalise again. The dense
// normalisation is folded into the sampler: the norm of each of the 4 corner cells is recomputed per keypoint
// (4 wave reductions) instead of a separate 16.8 MB read+write pass. One wave per keypoint, lane = 4 channels.
// dense: [B][Hc*Wc][ld] (NHWC, 256 channels at dense_coff). Output: [B][capacity][256], row i <-> keypoint i.
// ---------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void sample_descriptors_kernel(const float* __restrict__ dense, int ld, int Hc, int Wc,
                                                                 const float* __restrict__ kp_xy, const int* __restrict__ count,
                                                                 int capacity, float* __restrict__ desc) {
    const int b = blockIdx.y;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= count[b]) return;
    const int lane = threadIdx.x & 63;
    const float s = 8.0f;
    float x = kp_xy[((size_t)b * capacity + k) * 2 + 0];
    float y = kp_xy[((size_t)b * capacity + k) * 2 + 1];
    // keypoints - s/2 + 0.5 ; /= (w*s - s/2 - 0.5) ; *2 - 1
    x = (x - s / 2.0f) + 0.5f;
    y = (y - s / 2.0f) + 0.5f;
    x = x / (float)((double)Wc * 8.0 - 4.0 - 0.5);
    y = y / (float)((double)Hc * 8.0 - 4.0 - 0.5);
    x = x * 2.0f - 1.0f;
    y = y * 2.0f - 1.0f;
    // grid_sample, align_corners=True: ((coord + 1) / 2) * (size - 1)
    const float ix = ((x + 1.0f) / 2.0f) * (float)(Wc - 1);
    const float iy = ((y + 1.0f) / 2.0f) * (float)(Hc - 1);
    const float ix_nw = floorf(ix), iy_nw = floorf(iy);
    const float ix_se = ix_nw + 1.0f, iy_se = iy_nw + 1.0f;
    const float w_nw = (ix_se - ix) * (iy_se - iy);
    const float w_ne = (ix - ix_nw) * (iy_se - iy);
    const float w_sw = (ix_se - ix) * (iy - iy_nw);
    const float w_se = (ix - ix_nw) * (iy - iy_nw);
    const int x0 = (int)ix_nw, y0 = (int)iy_nw;
    const float* dense_b = dense + (size_t)b * Hc * Wc * ld;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int cx = x0 + (c & 1), cy = y0 + (c >> 1);
        const float w = (c == 0) ? w_nw : (c == 1) ? w_ne : (c == 2) ? w_sw : w_se;
        if (cx >= 0 && cx < Wc && cy >= 0 && cy < Hc) {  // wave-uniform
            const f32x4 d = *reinterpret_cast<const f32x4*>(dense_b + ((size_t)cy * Wc + cx) * ld + lane * 4);
            const float ss = wave_sum(d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w);
            const float denom = fmaxf(sqrtf(ss), 1e-12f);
            // ATen's vectorised grid_sample accumulates the four corners as an fma chain (verified bit-for-bit)
            acc.x = fmaf(d.x / denom, w, acc.x);
            acc.y = fmaf(d.y / denom, w, acc.y);
            acc.z = fmaf(d.z / denom, w, acc.z);
            acc.w = fmaf(d.w / denom, w, acc.w);
        }
    }
    const float ss = wave_sum(acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w);
    const float denom = fmaxf(sqrtf(ss), 1e-12f);
    f32x4 o = {acc.x / denom, acc.y / denom, acc.z / denom, acc.w / denom};
    *reinterpret_cast<f32x4*>(desc + ((size_t)b * capacity + k) * 256 + lane * 4) = o;
}

int launch_sample_descriptors(const float* dense, int ld, int B, int Hc, int Wc, const float* kp_xy, const int* count, int capacity,
                              float* desc, hipStream_t stream) {
    if (B == 0 || capacity == 0) return GTSFM_OK;
    GTSFM_CHECK_ARG(ld % 4 == 0, "sample_descriptors: ld must be a multiple of 4");
    hipLaunchKernelGGL(sample_descriptors_kernel, dim3(ceil_div(capacity, 4), B), dim3(256), 0, stream, dense, ld, Hc, Wc, kp_xy, count,
                       capacity, desc);
    GTSFM_CHECK_LAUNCH("sample_descriptors_kernel");
    return GTSFM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Device-side top-k by response (the selection gtsfm/common/keypoints.py:89-110 performs on the host with
// np.argpartition), keeping the survivors in row-major detection order. One workgroup per image: 4-pass radix select
// on the fp32 bit patterns (scores are positive, so the unsigned order is the numeric order), then an ordered
// compaction. Ties at the k-th value are broken by detection order (argpartition's choice is implementation-defined).
// ---------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(1024) void kp_select_topk_kernel(const float* __restrict__ scores, const int* __restrict__ count, int cap,
                                                              int k, const float* __restrict__ xy, float* __restrict__ out_xy,
                                                              float* __restrict__ out_score, int* __restrict__ out_count) {
    __shared__ int hist[256];
    __shared__ int wsum[16];
    __shared__ unsigned sel_prefix;
    __shared__ int sel_remaining, carry_s, tie_carry_s;
    const int b = blockIdx.x;
    const int n = min(count[b], cap);
    const unsigned* bits = reinterpret_cast<const unsigned*>(scores) + (size_t)b * cap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned tau = 0;
    int ties_wanted = 0;
    const bool select = n > k;
    if (select) {
        if (tid == 0) {
            sel_prefix = 0;
            sel_remaining = k;
        }
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned prefix = sel_prefix;
            const unsigned himask = (pass == 0) ? 0u : (0xffffffffu << (shift + 8));
            for (int i = tid; i < n; i += 1024) {
                const unsigned v = bits[i];
                if ((v & himask) == prefix) atomicAdd(&hist[(v >> shift) & 255], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int remaining = sel_remaining, bin = 255;
                for (; bin > 0; --bin) {
                    if (hist[bin] >= remaining) break;
                    remaining -= hist[bin];
                }
                sel_prefix = prefix | ((unsigned)bin << shift);
                sel_remaining = remaining;
            }
            __syncthreads();
        }
        tau = sel_prefix;
        ties_wanted = sel_remaining;
    }
    if (tid == 0) carry_s = tie_carry_s = 0;
    __syncthreads();
    float* oxy = out_xy + (size_t)b * k * 2;
    float* osc = out_score + (size_t)b * k;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const unsigned v = (i < n) ? bits[i] : 0u;
        const int is_tie = (select && i < n && v == tau) ? 1 : 0;
        // exclusive rank among ties
        int incl = is_tie;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        const int tie_rank = tie_carry_s + wbase + incl - is_tie;
        __syncthreads();
        if (tid == 1023) tie_carry_s += wbase + incl;
        const int keep = (i < n) && (!select || v > tau || (is_tie && tie_rank < ties_wanted)) ? 1 : 0;
        incl = keep;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        const int pos = carry_s + wbase + incl - keep;
        if (keep && pos < k) {
            oxy[pos * 2 + 0] = xy[((size_t)b * cap + i) * 2 + 0];
            oxy[pos * 2 + 1] = xy[((size_t)b * cap + i) * 2 + 1];
            osc[pos] = scores[(size_t)b * cap + i];
        }
        __syncthreads();
        if (tid == 1023) carry_s += wbase + incl;
        __syncthreads();
    }
    if (tid == 0) out_count[b] = min(n, k);
}

int launch_select_topk(const float* scores, const int* count, int B, int cap, int k, const float* xy, float* out_xy, float* out_score,
                       int* out_count, hipStream_t stream) {
    GTSFM_CHECK_ARG(k > 0 && cap > 0, "select_topk: k and capacity must be positive");
    if (B <= 0) return GTSFM_OK;
    hipLaunchKernelGGL(kp_select_topk_kernel, dim3(B), dim3(1024), 0, stream, scores, count, cap, k, xy, out_xy, out_score, out_count);
    GTSFM_CHECK_LAUNCH("kp_select_topk_kernel");
    return GTSFM_OK;
}
