// fp32-MFMA GEMM with register-staged A rows and packed weights (round 1's kernel): serves the shapes the LDS-DMA GEMM
// (gemm_dma_kernels.hip) does not take -- K not a multiple of 32 (SuperGlue's keypoint encoder: K = 8, 32, 64, ...), callers
// without row-major weights (gtsfm_linear_f32) -- and launch_gemm's dispatch between the two. Built with -ffp-contract=off.
// Replaces the ATen conv2d(k=1) / conv1d / linear calls of
//   thirdparty/SuperGluePretrainedNetwork/models/superpoint.py:162,191
//   thirdparty/SuperGluePretrainedNetwork/models/superglue.py:49-60,98-107,110-119,254
// See mfma_tiles.h for the tiling and the packed weight layout.

#include <stdlib.h>

#include "conv_kernels.h"  // packed_linear_floats
#include "gemm_kernels.h"
#include "mfma_tiles.h"

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

