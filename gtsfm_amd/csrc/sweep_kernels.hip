// Score-matrix sweeps with register-resident rows: Sinkhorn iterations (superglue.py:141-147), LightGlue's double
// log-softmax, and the match extraction (superglue.py:266-276, LightGlue filter_matches) fused into ONE pass over Z.
//
// Round 1 staged 6-16 rows of Z in a 49-64 KiB LDS slab per workgroup and went through three barrier-separated phases
// (stage, row reductions, column partials) with two library expf per element: 303 + 38 us per iteration for 537 MB of
// couplings matrices = 1.8 TB/s, VALU / latency-bound. Here a row never touches LDS:
//   * a WAVE owns a row: 64 lanes x NCH float4 (NCH = 8 covers 2048 columns) = 32 values per lane, loaded with one
//     fully coalesced 1 KiB request per float4 and reduced with two wave shuffles trees (max, sum);
//   * the next row of the wave is already in flight (second register buffer) while the current one is reduced: 12 waves
//     per CU x 8 KiB outstanding, no barrier anywhere in the row loop;
//   * ONE hardware exponential per element (v_exp_f32 on a pre-scaled argument). SuperGlue's column sums reuse the row
//     exponentials: exp(z_ij + u_i + v_j) = e_ij * mu_i / s_i with e_ij = exp(z_ij + v_j - max_i), s_i = sum_j e_ij,
//     every term <= mu_i < 1 (no running maximum needed) and the dustbin row gives every column a term of healthy
//     size; v_j' = log nu_j - (log sum_i(...) - v_j). The dustbin row and column are the constant bin_score
//     (superglue.py:156-160) and are never read from memory;
//   * a wave accumulates the column partials of its 8 rows in registers; the workgroup's 4 waves are combined through LDS
//     once, after the loop; a small second kernel sums the row-block partials in a fixed order (deterministic).
// LightGlue's double softmax has no dustbin bounding the column sums and similarities of +-100, so its column statistics
// are kept as online (max, sum) pairs per lane-column with their own exponentials (it runs once per forward pass).
// The extraction sweep reads Z once more and produces the row arg-maxima (wave reduction) and the column arg-maxima
// (per-lane running best over the wave's rows, combined across waves / row blocks), replacing round 1's two separate
// sweeps (best_rows + best_cols = 0.93 ms per 32-pair chunk at N = 2048).
// Score matrices wider than 2048 columns (GTSfM's 5000-keypoint cap, deep_front_end.yaml:29) do not fit one wave's
// registers (20 float4 per lane x 4 buffers). They take the *_wide_kernel forms below: the 4 (or 8) waves of a
// workgroup SHARE a row -- wave w owns the 256-column chunks w, w + NW, w + 2 NW, ... of every one of the block's 32 rows,
// so its column statistics are complete without any cross-wave step, and a Sinkhorn row costs ONE barrier: every wave publishes
// the (max, sum) of its slice relative to its OWN maximum, and all waves combine the NW pairs in a fixed order
// (exp(t - max) = exp(t - max_w) exp(max_w - max)); the double log-softmax and the extraction need no barrier per row at all (the slices'
// row statistics / candidates meet once per block). 5120 columns with 4 waves (round 3), 10240 with 8; beyond that the
// LDS-staged kernels of matcher_kernels.hip. Which kernel a pair takes depends only on ITS OWN column count -- a batch wider
// than 2048 launches every tier and the blocks of the other tiers' pairs return at once -- so results do not depend on the
// batch a pair travels in. Built with -ffp-contract=off.
// Every row kernel reads the matrix through a buffer resource with NO branch around a load (sw_load_slice_rsrc, round 5): with
// `if (col < n) load` the compiler's wait-count pass cannot count loads in flight and every wait becomes s_waitcnt vmcnt(0) -- the row
// prefetched for the next iteration is waited for together with the current one and the double buffering overlaps nothing.

#include <stdlib.h>

#include "matcher_kernels.h"

#define SW_ROWS 32        // rows per workgroup: 4 waves x 8 rows (wave w owns rows r0 + w, r0 + w + 4, ...)
#define SW_MAX_COLS 2048  // widest row one wave holds in registers
#define SW_WIDE4_COLS 5120   // widest row the 4 waves of a workgroup hold together (5 chunks of 256 columns per wave)
#define SW_WIDE8_COLS 10240  // ... 8 waves

constexpr float SW_LOG2E = 1.44269504088896340736f;
constexpr float SW_FLT_MIN = 1.17549435e-38f;

__device__ __forceinline__ float sw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// fmaxf as the one instruction it is: for operands that come from memory the compiler puts a canonicalising v_max_f32 x, x in front of every
// fmaxf (IEEE mode: a signalling NaN has to be quieted first) -- 40 of 320 vector instructions per row in the double log-softmax sweep.
// v_max_f32 itself returns the other operand for a NaN, which is all the sweeps rely on.
__device__ __forceinline__ float sw_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Score-matrix reads are nontemporal (NT: aux bit 1 of the buffer load) when the matrices of one launch together exceed the 256 MiB Infinity
// Cache: every sweep streams them from HBM anyway and reads that do not allocate in the caches are 4-8 % faster (4.63 -> 4.92 TB/s, 21 pairs at
// 5000 columns; 4.87 -> 5.18 at 2048, 32 pairs); when they fit (1-2 pairs at the cap, <= 15 at 2048) the plain reads hit the cache on the next
// sweep and NT costs 11 %. The launcher chooses (speed only: the same values are loaded).

// A row (or a wave's slice of it) through a buffer resource, with NO branch around any load and ONE 32-bit register of address per chunk: the pair's matrix is
// the resource, the row's byte offset rides in a scalar register, the lane's column offset (loop-invariant) in voff[c]. A chunk at or beyond
// column n re-reads the row's first 16 bytes (voff = 0; the caller makes those lanes' values harmless).
//   * a load inside a branch makes the compiler's wait-count pass give up counting: every later wait becomes s_waitcnt vmcnt(0) -- "wait for
//     every load in flight", the rows prefetched for LATER iterations included -- and the prefetch hides nothing;
//   * flat 64-bit addresses cost two registers per chunk and row in flight and a 64-bit add per load.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int NCH, bool NT = false>
__device__ __forceinline__ void sw_load_slice_rsrc(__amdgpu_buffer_rsrc_t z, unsigned row_bytes, const int (&voff)[NCH], f32x4 (&dst)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const u32x4 raw = __builtin_amdgcn_raw_buffer_load_b128(z, voff[c], (int)row_bytes, NT ? 2 : 0);  // aux bit 1: nontemporal
        dst[c] = __builtin_bit_cast(f32x4, raw);
    }
}

// The pair's score matrix as a buffer resource + the row pitch in bytes. Every piece goes through readfirstlane: a descriptor the compiler
// cannot prove wave-uniform is read in a loop over the lanes' values around each load. res_empty is the same resource with ZERO records: a load
// through it is out of range, returns zeros and moves no data -- what the unconditional prefetch past a wave's last row (and of Sinkhorn's
// synthetic dustbin row) reads; a scalar select between the two descriptors is not a branch around the load.
#define SW_MAKE_ROW_RESOURCE(res, pitch, Zptr, mrows_, ld_)                                                                                  \
    const unsigned long long res##_addr = reinterpret_cast<unsigned long long>(Zptr);                                                        \
    float* const res##_uni = reinterpret_cast<float*>((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(res##_addr >> 32)) << 32 | \
                                                      (unsigned)__builtin_amdgcn_readfirstlane((int)res##_addr));                            \
    const unsigned pitch = (unsigned)__builtin_amdgcn_readfirstlane(ld_) * 4u;                                                               \
    const __amdgpu_buffer_rsrc_t res =                                                                                                       \
        __builtin_amdgcn_make_buffer_rsrc(res##_uni, 0, (int)((unsigned)__builtin_amdgcn_readfirstlane(mrows_) * pitch), 0x00020000);        \
    const __amdgpu_buffer_rsrc_t res##_empty = __builtin_amdgcn_make_buffer_rsrc(res##_uni, 0, 0, 0x00020000)

// ---------------------------------------------------------------------------------------------------------------
// SuperGlue: one Sinkhorn iteration = sinkhorn_rows_kernel + sinkhorn_cols_kernel
// ---------------------------------------------------------------------------------------------------------------

// 168 registers (3 waves per SIMD) spill 33 of them at NCH = 8; 2 waves per SIMD still keep 64 KiB of rows in flight per CU
template <int NCH, bool NT>
__global__ __launch_bounds__(256, NCH >= 8 ? 2 : 3) void sinkhorn_rows_kernel(const float* __restrict__ zbuf, const PairDesc* __restrict__ pairs,
                                                               const SeqDesc* __restrict__ seqs, const int* __restrict__ counts,
                                                               float* __restrict__ rowvec, const float* __restrict__ colvec,
                                                               float* __restrict__ partials, float alpha, int pair0) {
    __shared__ __attribute__((aligned(16))) float red[4][NCH * 256];
    __shared__ float red_bin[4];
    const int p = pair0 + blockIdx.y;
    const PairDesc pd = pairs[p];
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    const int rows = m + 1;
    const int r0 = blockIdx.x * SW_ROWS;
    if (r0 >= rows || n > SW_MAX_COLS) return;  // wider pairs: sinkhorn_rows_wide_kernel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ld = pd.ld;
    const float* Z = zbuf + pd.z_off;
    const int vec0 = vec_off(s0, 2 * p), vec1 = vec_off(s1, 2 * p + 1);
    const float NEG = neg_inf();
    const float mn = (float)m + (float)n;
    const float norm = -logf(mn);
    const float inv_mn = 1.0f / mn;
    f32x4 v[NCH], acc[NCH], za[NCH], zb[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 4 * (lane + 64 * c);
        v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (col < n) v[c] = *reinterpret_cast<const f32x4*>(colvec + vec1 + col);
        acc[c] = za[c] = zb[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float t_bin = alpha + colvec[vec1 + n];  // dustbin column: Z[i][n] = bin_score for every row
    float acc_bin = 0.f;

    auto process = [&](int i, f32x4(&zz)[NCH]) {
        if (__builtin_amdgcn_readfirstlane(i) >= m) {  // dustbin row: Z[m][j] = bin_score. A real (scalar) branch, taken by one row per pair: as
            asm volatile("");                          // selects it costs every row a v_cndmask per element AND is hoisted above the row's
#pragma unroll                                         // first use, where it waits for the loads of the NEXT row as well
            for (int c = 0; c < NCH; ++c) zz[c] = f32x4{alpha, alpha, alpha, alpha};
        }
        float mx = t_bin;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = 4 * (lane + 64 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = (col + e < n) ? zz[c][e] + v[c][e] : NEG;
                zz[c][e] = t;
                mx = fmaxf(mx, t);
            }
        }
        mx = wave_max(mx);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ev = sw_exp2((zz[c][e] - mx) * SW_LOG2E);  // exp2(-inf) = 0 for the masked columns
                zz[c][e] = ev;
                s += ev;
            }
        }
        s = wave_sum(s);
        const float e_bin = sw_exp2((t_bin - mx) * SW_LOG2E);
        s += e_bin;
        const float lse = logf(s) + mx;
        const float log_mu = (i < m) ? norm : logf((float)n) + norm;
        if (lane == 0) rowvec[vec0 + i] = log_mu - lse;  // u_i (superglue.py:145)
        // exp(u_i + max_i) = mu_i / s_i: weight of this row's exponentials in the column sums
        const float w = ((i < m) ? inv_mn : (float)n * inv_mn) / s;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[c][e] = fmaf(zz[c][e], w, acc[c][e]);
        }
        acc_bin = fmaf(e_bin, w, acc_bin);
    };

    SW_MAKE_ROW_RESOURCE(zres, ldb, Z, m, ld);
    int voff[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 4 * (lane + 64 * c);
        voff[c] = col < n ? 4 * col : 0;
    }
    // this wave's rows: r0 + wave, + 4, ... One row in flight behind the one being processed; every load unconditional (past the wave's last row
    // the last row is read again: an L2 hit) so that the waits are exact counts -- see sw_load_slice_rsrc
    const int iend = (r0 + SW_ROWS < rows) ? r0 + SW_ROWS : rows;
    int i = r0 + __builtin_amdgcn_readfirstlane(wave);
    if (i < iend) {
        const int lastw = i + ((iend - 1 - i) & ~3);
        // the dustbin row (r = m) and the prefetch past the wave's last row read nothing (zres_empty); process() puts bin_score in the former
        auto load = [&](int r, f32x4(&dst)[NCH]) {
            const bool real = r <= lastw && r < m;  // wave-uniform by construction: a scalar compare
            sw_load_slice_rsrc<NCH, NT>(real ? zres : zres_empty, real ? (unsigned)r * ldb : 0u, voff, dst);
        };
        load(i, za);
#pragma unroll 1
        for (;;) {
            load(i + 4, zb);
            __builtin_amdgcn_sched_barrier(0);  // the next row's loads are issued BEFORE the wait for this row's
            process(i, za);
            i += 4;
            if (i >= iend) break;
            load(i + 4, za);
            __builtin_amdgcn_sched_barrier(0);
            process(i, zb);
            i += 4;
            if (i >= iend) break;
        }
    }

    // column partials of the block's 32 rows: sum of the 4 waves, in a fixed order
#pragma unroll
    for (int c = 0; c < NCH; ++c) *reinterpret_cast<f32x4*>(&red[wave][4 * (lane + 64 * c)]) = acc[c];
    if (lane == 0) red_bin[wave] = acc_bin;
    __syncthreads();
    float* part = partials + pd.part_off + (size_t)blockIdx.x * ld;
    const float bin_sum = ((red_bin[0] + red_bin[1]) + red_bin[2]) + red_bin[3];
    for (int f = threadIdx.x; f <= NCH * 64; f += 256) {  // one float4 more than a row holds: the dustbin column when n = 256 NCH
        const int col = 4 * f;
        if (col > n) continue;
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        if (f < NCH * 64) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(&red[0][col]), a1 = *reinterpret_cast<const f32x4*>(&red[1][col]);
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(&red[2][col]), a3 = *reinterpret_cast<const f32x4*>(&red[3][col]);
            sum = ((a0 + a1) + a2) + a3;
        }
        const int d = n - col;  // the dustbin column sits inside this float4 when 0 <= d < 4
        if (d == 0) sum.x = bin_sum;
        if (d == 1) sum.y = bin_sum;
        if (d == 2) sum.z = bin_sum;
        if (d == 3) sum.w = bin_sum;
        *reinterpret_cast<f32x4*>(part + col) = sum;  // col <= n, col % 4 == 0 -> col + 3 < ld
    }
}

// v_j = log nu_j - logsumexp_i(Z_ij + u_i) from the row-block partials sum_i exp(z_ij + u_i + v_j_old) (superglue.py:146)
__global__ __launch_bounds__(256) void sinkhorn_cols_kernel(const PairDesc* __restrict__ pairs, const SeqDesc* __restrict__ seqs,
                                                            const int* __restrict__ counts, const float* __restrict__ partials,
                                                            float* __restrict__ colvec, int pair0) {
    const int p = pair0 + blockIdx.y;
    const PairDesc pd = pairs[p];
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j > n) return;
    const int nblk = ceil_div(m + 1, SW_ROWS);
    const float* part = partials + pd.part_off + j;
    const size_t bs = (size_t)pd.ld;
    float s0a = 0.f, s1a = 0.f, s2a = 0.f, s3a = 0.f;
    int b = 0;
    for (; b + 4 <= nblk; b += 4) {
        s0a += part[(b + 0) * bs];
        s1a += part[(b + 1) * bs];
        s2a += part[(b + 2) * bs];
        s3a += part[(b + 3) * bs];
    }
    for (; b < nblk; ++b) s0a += part[b * bs];
    const float sum = (s0a + s1a) + (s2a + s3a);
    const int vec1 = vec_off(s1, 2 * p + 1);
    const float norm = -logf((float)m + (float)n);
    const float log_nu = (j < n) ? norm : logf((float)m) + norm;
    // a column whose terms all underflowed gets the smallest normal number instead of log(0)
    colvec[vec1 + j] = log_nu - (logf(fmaxf(sum, SW_FLT_MIN)) - colvec[vec1 + j]);
}

// ---------------------------------------------------------------------------------------------------------------
// LightGlue: row log-sum-exp + online column (max, sum) statistics in one pass (sigmoid_log_double_softmax)
// ---------------------------------------------------------------------------------------------------------------

template <int NCH, bool NT = false>
__global__ __launch_bounds__(256, 2) void lg_rows_kernel(const float* __restrict__ zbuf, const PairDesc* __restrict__ pairs,
                                                         const SeqDesc* __restrict__ seqs, const int* __restrict__ counts,
                                                         float* __restrict__ rowvec, float* __restrict__ partials) {
    __shared__ __attribute__((aligned(16))) float red_m[4][NCH * 256];
    __shared__ __attribute__((aligned(16))) float red_s[4][NCH * 256];
    const int p = blockIdx.y;
    const PairDesc pd = pairs[p];
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    const int r0 = blockIdx.x * SW_ROWS;
    if (r0 >= m || n > SW_MAX_COLS) return;  // wider pairs: lg_rows_wide_kernel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ld = pd.ld;
    const float* Z = zbuf + pd.z_off;
    const int vec0 = vec_off(s0, 2 * p);
    const float NEG = neg_inf();
    f32x4 cm[NCH], cs[NCH], za[NCH], zb[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        cm[c] = f32x4{NEG, NEG, NEG, NEG};
        cs[c] = za[c] = zb[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto process = [&](int i, f32x4(&zz)[NCH]) {
        float mx = NEG;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = 4 * (lane + 64 * c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = (col + e < n) ? zz[c][e] : NEG;
                zz[c][e] = t;
                mx = sw_max(mx, t);
            }
        }
        mx = wave_max(mx);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = zz[c][e];
                s += sw_exp2((t - mx) * SW_LOG2E);
                // online column statistics with ONE exponential: of (old maximum - new maximum) and (t - new maximum) one is 0 and the other
                // -|t - old maximum| (exp2(0) = 1 exactly: the same bits as two exponentials). The kernel is bound by the vector ALU, not by
                // HBM -- three quarter-rate v_exp_f32 per element were 12 of its 29 issue slots per element. A masked column (t = -inf, maximum
                // -inf) gets NaN here; columns >= n are never read.
                const float d = t - cm[c][e];
                const float ex = sw_exp2(-fabsf(d) * SW_LOG2E);
                const bool up = d > 0.f;  // the column maximum moves to this row
                cs[c][e] = cs[c][e] * (up ? ex : 1.f) + (up ? 1.f : ex);
                cm[c][e] = sw_max(cm[c][e], t);
            }
        }
        s = wave_sum(s);
        if (lane == 0) rowvec[vec0 + i] = logf(s) + mx;
    };
    SW_MAKE_ROW_RESOURCE(zres, ldb, Z, m, ld);
    int voff[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 4 * (lane + 64 * c);
        voff[c] = col < n ? 4 * col : 0;
    }
    // this wave's rows: r0 + wave, + 4, ...; unconditional loads with exact wait counts (see sinkhorn_rows_kernel)
    const int iend = (r0 + SW_ROWS < m) ? r0 + SW_ROWS : m;
    int i = r0 + __builtin_amdgcn_readfirstlane(wave);
    if (i < iend) {
        const int lastw = i + ((iend - 1 - i) & ~3);
        // past the last row: nothing is read (zres_empty)
        auto load = [&](int r, f32x4(&dst)[NCH]) {
            const bool real = r <= lastw;  // wave-uniform by construction: a scalar compare
            sw_load_slice_rsrc<NCH, NT>(real ? zres : zres_empty, real ? (unsigned)r * ldb : 0u, voff, dst);
        };
        load(i, za);
#pragma unroll 1
        for (;;) {
            load(i + 4, zb);
            __builtin_amdgcn_sched_barrier(0);
            process(i, za);
            i += 4;
            if (i >= iend) break;
            load(i + 4, za);
            __builtin_amdgcn_sched_barrier(0);
            process(i, zb);
            i += 4;
            if (i >= iend) break;
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        *reinterpret_cast<f32x4*>(&red_m[wave][4 * (lane + 64 * c)]) = cm[c];
        *reinterpret_cast<f32x4*>(&red_s[wave][4 * (lane + 64 * c)]) = cs[c];
    }
    __syncthreads();
    // block partial per column: plane 0 = max, plane 1 = sum of exp(. - max), at part_off + block * 2 ld
    float* part = partials + pd.part_off + (size_t)blockIdx.x * 2 * ld;
    for (int j = threadIdx.x; j < n; j += 256) {
        float M = red_m[0][j];
#pragma unroll
        for (int w = 1; w < 4; ++w) M = fmaxf(M, red_m[w][j]);
        float S = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = red_m[w][j];
            if (mw != NEG) S += red_s[w][j] * sw_exp2((mw - M) * SW_LOG2E);
        }
        part[j] = M;
        part[ld + j] = S;
    }
}

__global__ __launch_bounds__(256) void lg_cols_kernel(const PairDesc* __restrict__ pairs, const SeqDesc* __restrict__ seqs,
                                                      const int* __restrict__ counts, const float* __restrict__ partials,
                                                      float* __restrict__ colvec) {
    const int p = blockIdx.y;
    const PairDesc pd = pairs[p];
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int nblk = ceil_div(m, SW_ROWS);
    const float* part = partials + pd.part_off + j;
    const size_t bs = (size_t)pd.ld * 2;
    // the row-block partials of a column are 2 ld floats apart: eight loads in flight per thread (one pair at the 5000-keypoint cap is 20
    // workgroups walking 157 blocks -- with one dependent load per step the kernel took 69 us, longer than the sweep over the matrix itself);
    // the sum keeps its order, block 0 first
    float M = neg_inf();
    int b = 0;
    for (; b + 8 <= nblk; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(b + u) * bs];
#pragma unroll
        for (int u = 0; u < 8; ++u) M = fmaxf(M, v[u]);
    }
    for (; b < nblk; ++b) M = fmaxf(M, part[b * bs]);
    float S = 0.f;
    for (b = 0; b + 8 <= nblk; b += 8) {
        float mv[8], sv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) mv[u] = part[(b + u) * bs], sv[u] = part[(b + u) * bs + pd.ld];
#pragma unroll
        for (int u = 0; u < 8; ++u) sv[u] = sv[u] * sw_exp2((mv[u] - M) * SW_LOG2E);
#pragma unroll
        for (int u = 0; u < 8; ++u) S += sv[u];
    }
    for (; b < nblk; ++b) S += part[b * bs + pd.ld] * sw_exp2((part[b * bs] - M) * SW_LOG2E);
    colvec[vec_off(s1, 2 * p + 1) + j] = logf(S) + M;
}

// ---------------------------------------------------------------------------------------------------------------
// Match extraction: ONE pass over the inner m x n block gives the row arg-maxima and the column arg-maxima of the final
// assignment matrix (assign_value), first index on ties as torch.max does.
// ---------------------------------------------------------------------------------------------------------------

#define SW_NO_INDEX 0x7fffffff

template <bool SG, int NCH>
__global__ __launch_bounds__(256, NCH >= 8 ? 1 : 3) void extract_rows_kernel(const float* __restrict__ zbuf, const PairDesc* __restrict__ pairs,
                                                              const SeqDesc* __restrict__ seqs, const int* __restrict__ counts,
                                                              const float* __restrict__ rowvec, const float* __restrict__ colvec,
                                                              const float* __restrict__ zlogit, float* __restrict__ max0,
                                                              int* __restrict__ idx0, float* __restrict__ partials, int n_max) {
    __shared__ __attribute__((aligned(16))) float red_v[4][NCH * 256];
    __shared__ __attribute__((aligned(16))) int red_i[4][NCH * 256];
    const int p = blockIdx.y;
    const PairDesc pd = pairs[p];
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    const int r0 = blockIdx.x * SW_ROWS;
    if (r0 >= m || n > n_max) return;  // wider pairs: extract_rows_wide_kernel (arg-maxima do not depend on the slicing: the launcher picks the bound)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ld = pd.ld;
    const float* Z = zbuf + pd.z_off;
    const int vec0 = vec_off(s0, 2 * p), vec1 = vec_off(s1, 2 * p + 1);
    const float NEG = neg_inf();
    const float norm = SG ? -logf((float)m + (float)n) : 0.f;
    f32x4 bj[NCH], cj[NCH], cbv[NCH], za[NCH], zb[NCH];
    int cbi[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 4 * (lane + 64 * c);
        cj[c] = za[c] = zb[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        // columns beyond n (the row's padding to 4 floats, the chunks past the matrix) carry a column term that turns every value into
        // -inf or NaN -- SuperGlue adds b_j (-inf), LightGlue subtracts it (+inf) -- so that no comparison below can select them whatever the
        // padding holds: the row loop needs no per-element bounds test (compiled as a branch per element until round 5: 0.26 of the roof)
        bj[c] = SG ? f32x4{NEG, NEG, NEG, NEG} : f32x4{-NEG, -NEG, -NEG, -NEG};
        cbv[c] = f32x4{NEG, NEG, NEG, NEG};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cbi[c][e] = SW_NO_INDEX;
            if (col + e < n) {
                bj[c][e] = colvec[vec1 + col + e];
                if (!SG) cj[c][e] = logsigmoid(zlogit[s1.row_off + col + e]);
            }
        }
    }
    // LightGlue's row term logsigmoid(z0_i) of the block's 32 rows, one per lane, computed ahead of the sweep (log1pf / expf stay out
    // of the row loop)
    float ci_lane = 0.f;
    if (!SG && r0 + (lane & 31) < m) ci_lane = logsigmoid(zlogit[s0.row_off + r0 + (lane & 31)]);
    auto process = [&](int i, const f32x4(&zz)[NCH]) {
        const float a_i = rowvec[vec0 + i];
        const float c_i = SG ? 0.f : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ci_lane), __builtin_amdgcn_readfirstlane(i - r0)));
        float best = NEG;
        int bcode = -1;  // 4 * chunk + element of the lane's best value: inline constants instead of a register per column index
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float val = assign_value<SG>(zz[c][e], a_i, bj[c][e], norm, c_i, cj[c][e]);  // -inf / NaN beyond column n: never selected
                const bool row_better = val > best;  // columns ascend with the code within a lane: the first maximum wins
                best = row_better ? val : best;
                bcode = row_better ? 4 * c + e : bcode;
                const bool col_better = val > cbv[c][e];  // rows ascend within a wave
                cbv[c][e] = col_better ? val : cbv[c][e];
                cbi[c][e] = col_better ? i : cbi[c][e];
            }
        }
        int bidx = bcode < 0 ? SW_NO_INDEX : 4 * (lane + 64 * (bcode >> 2)) + (bcode & 3);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int oj = __shfl_xor(bidx, off, 64);
            const bool take_ob = (ob > best) | ((ob == best) & (oj < bidx));  // bitwise: || and && compile to a branch per step
            best = take_ob ? ob : best;
            bidx = take_ob ? oj : bidx;
        }
        if (lane == 0) {
            max0[s0.row_off + i] = best;
            idx0[s0.row_off + i] = (bidx == SW_NO_INDEX) ? 0 : bidx;  // all-NaN row: stay in range
        }
    };
    SW_MAKE_ROW_RESOURCE(zres, ldb, Z, m, ld);
    int voff[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 4 * (lane + 64 * c);
        voff[c] = col < n ? 4 * col : 0;
    }
    // this wave's rows: r0 + wave, + 4, ...; unconditional loads with exact wait counts (see sinkhorn_rows_kernel)
    const int iend = (r0 + SW_ROWS < m) ? r0 + SW_ROWS : m;
    int i = r0 + __builtin_amdgcn_readfirstlane(wave);
    if (i < iend) {
        const int lastw = i + ((iend - 1 - i) & ~3);
        // past the last row: nothing is read (zres_empty)
        auto load = [&](int r, f32x4(&dst)[NCH]) {
            const bool real = r <= lastw;  // wave-uniform by construction: a scalar compare
            sw_load_slice_rsrc<NCH>(real ? zres : zres_empty, real ? (unsigned)r * ldb : 0u, voff, dst);
        };
        load(i, za);
#pragma unroll 1
        for (;;) {
            load(i + 4, zb);
            __builtin_amdgcn_sched_barrier(0);
            process(i, za);
            i += 4;
            if (i >= iend) break;
            load(i + 4, za);
            __builtin_amdgcn_sched_barrier(0);
            process(i, zb);
            i += 4;
            if (i >= iend) break;
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        *reinterpret_cast<f32x4*>(&red_v[wave][4 * (lane + 64 * c)]) = cbv[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) red_i[wave][4 * (lane + 64 * c) + e] = cbi[c][e];
    }
    __syncthreads();
    // block partial per column: plane 0 = best value, plane 1 = its row (as int bits), at part_off + block * 2 ld
    float* part = partials + pd.part_off + (size_t)blockIdx.x * 2 * ld;
    for (int j = threadIdx.x; j < n; j += 256) {
        float bv = red_v[0][j];
        int bi = red_i[0][j];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ov = red_v[w][j];
            const int oi = red_i[w][j];
            const bool take_ov = ov > bv || (ov == bv && oi < bi);  // selects, not a branch
            bv = take_ov ? ov : bv;
            bi = take_ov ? oi : bi;
        }
        part[j] = bv;
        reinterpret_cast<int*>(part)[ld + j] = bi;
    }
}

__global__ __launch_bounds__(256) void extract_cols_kernel(const PairDesc* __restrict__ pairs, const SeqDesc* __restrict__ seqs,
                                                           const int* __restrict__ counts, const float* __restrict__ partials,
                                                           int* __restrict__ idx1) {
    const int p = blockIdx.y;
    const PairDesc pd = pairs[p];
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int nblk = ceil_div(m, SW_ROWS);
    const float* part = partials + pd.part_off + j;
    const size_t bs = (size_t)pd.ld * 2;
    float bv = neg_inf();
    int bi = SW_NO_INDEX;
    int b = 0;
    for (; b + 8 <= nblk; b += 8) {  // eight blocks' loads in flight (see lg_cols_kernel), compared in block order
        float ov[8];
        int oi[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ov[u] = part[(b + u) * bs], oi[u] = reinterpret_cast<const int*>(part)[(b + u) * bs + pd.ld];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (ov[u] > bv || (ov[u] == bv && oi[u] < bi)) bv = ov[u], bi = oi[u];
    }
    for (; b < nblk; ++b) {
        const float ov = part[b * bs];
        const int oi = reinterpret_cast<const int*>(part)[b * bs + pd.ld];
        const bool take_ov = ov > bv || (ov == bv && oi < bi);  // selects, not a branch
        bv = take_ov ? ov : bv;
        bi = take_ov ? oi : bi;
    }
    idx1[s1.row_off + j] = (bi == SW_NO_INDEX) ? 0 : bi;  // all-NaN column: stay in range
}

// ---------------------------------------------------------------------------------------------------------------
// Rows wider than 2048 columns: the NW waves of a workgroup share a row (see the header). Wave w, chunk c of its registers
// <-> columns 256 (w + NW c) + 4 lane .. + 3: the mapping does not depend on NCH (unused chunks are masked), so a pair's
// result does not depend on the widest pair of its batch. A pair belongs to the tier NW = 4 when 2048 < n <= 5120 and to
// NW = 8 when 5120 < n <= 10240.
// ---------------------------------------------------------------------------------------------------------------

template <int NW>
__device__ __forceinline__ bool sw_wide_tier(int n) {
    return NW == 4 ? (n > SW_MAX_COLS && n <= SW_WIDE4_COLS) : (n > SW_WIDE4_COLS && n <= SW_WIDE8_COLS);
}

template <int NW, int NCH, bool NT>
__global__ __launch_bounds__(64 * NW) void sinkhorn_rows_wide_kernel(const float* __restrict__ zbuf, const PairDesc* __restrict__ pairs,
                                                                     const SeqDesc* __restrict__ seqs, const int* __restrict__ counts,
                                                                     float* __restrict__ rowvec, const float* __restrict__ colvec,
                                                                     float* __restrict__ partials, float alpha, int pair0) {
    __shared__ float xm[2][NW], xs[2][NW];  // per-wave (max, sum) of the row in flight, double-buffered by row parity
    const int p = pair0 + blockIdx.y;
    const PairDesc pd = pairs[p];
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    const int rows = m + 1;
    const int r0 = blockIdx.x * SW_ROWS;
    if (r0 >= rows || !sw_wide_tier<NW>(n)) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ld = pd.ld;
    const float* Z = zbuf + pd.z_off;
    const int vec0 = vec_off(s0, 2 * p), vec1 = vec_off(s1, 2 * p + 1);
    const float NEG = neg_inf();
    const float mn = (float)m + (float)n;
    const float norm = -logf(mn);
    const float inv_mn = 1.0f / mn;
    f32x4 v[NCH], acc[NCH], za[NCH], zb[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 256 * (wave + NW * c) + 4 * lane;
        v[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (col < n) v[c] = *reinterpret_cast<const f32x4*>(colvec + vec1 + col);
        acc[c] = za[c] = zb[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float t_bin = alpha + colvec[vec1 + n];  // dustbin column: Z[i][n] = bin_score for every row
    float acc_bin = 0.f;                           // kept by every wave (same value)

    auto process = [&](int i, f32x4(&zz)[NCH], int slot) {
        if (__builtin_amdgcn_readfirstlane(i) >= m) {  // dustbin row: Z[m][j] = bin_score. A real (scalar) branch, taken by one row per pair: as
            asm volatile("");                          // selects it costs every row a v_cndmask per element AND is hoisted above the row's
#pragma unroll                                         // first use, where it waits for the loads of the NEXT row as well
            for (int c = 0; c < NCH; ++c) zz[c] = f32x4{alpha, alpha, alpha, alpha};
        }
        float mw = NEG;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = 256 * (wave + NW * c) + 4 * lane;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = (col + e < n) ? zz[c][e] + v[c][e] : NEG;
                zz[c][e] = t;
                mw = fmaxf(mw, t);
            }
        }
        mw = wave_max(mw);
        const bool has = mw != NEG;         // false: none of this wave's columns exist (n is far below the tier's width)
        const float mref = has ? mw : 0.f;  // the slice's exponentials are taken relative to its own maximum
        float sw = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ev = sw_exp2((zz[c][e] - mref) * SW_LOG2E);  // exp2(-inf) = 0 for the masked columns
                zz[c][e] = ev;
                sw += ev;
            }
        }
        sw = wave_sum(sw);
        if (lane == 0) xm[slot][wave] = mw, xs[slot][wave] = sw;
        __syncthreads();
        float mx = t_bin;
#pragma unroll
        for (int w = 0; w < NW; ++w) mx = fmaxf(mx, xm[slot][w]);
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += xs[slot][w] * sw_exp2((xm[slot][w] - mx) * SW_LOG2E);  // empty slice: 0 * exp2(-inf) = 0
        const float e_bin = sw_exp2((t_bin - mx) * SW_LOG2E);
        s += e_bin;
        const float lse = logf(s) + mx;
        const float log_mu = (i < m) ? norm : logf((float)n) + norm;
        if (threadIdx.x == 0) rowvec[vec0 + i] = log_mu - lse;  // u_i (superglue.py:145)
        // exp(u_i + max_i) = mu_i / s_i; this wave's exponentials additionally carry exp(max_w - max_i)
        const float wrow = ((i < m) ? inv_mn : (float)n * inv_mn) / s;
        const float wgt = has ? sw_exp2((mw - mx) * SW_LOG2E) * wrow : 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[c][e] = fmaf(zz[c][e], wgt, acc[c][e]);
        }
        acc_bin = fmaf(e_bin, wrow, acc_bin);
    };

    const int rend = (r0 + SW_ROWS < rows) ? r0 + SW_ROWS : rows, last = rend - 1;
    SW_MAKE_ROW_RESOURCE(zres, ldb, Z, m, ld);
    int voff[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 256 * (wave + NW * c) + 4 * lane;
        voff[c] = col < n ? 4 * col : 0;
    }
    // every load unconditional (past the block's last row the last row is read again: an L2 hit) so that the waits are exact counts -- see
    // sw_load_slice_rsrc
    // the dustbin row (r = m) and the prefetch past the block's last row read nothing (zres_empty); process() puts bin_score in the former
    auto load = [&](int r, f32x4(&dst)[NCH]) {
        const bool real = r <= last && r < m;  // wave-uniform by construction: a scalar compare
        sw_load_slice_rsrc<NCH, NT>(real ? zres : zres_empty, real ? (unsigned)r * ldb : 0u, voff, dst);
    };
    int i = r0;
    load(i, za);
#pragma unroll 1
    for (;; i += 2) {  // uniform for the workgroup: every wave walks the same rows
        load(i + 1, zb);
        __builtin_amdgcn_sched_barrier(0);  // the next row's loads are issued BEFORE the wait for this row's
        process(i, za, 0);
        if (i + 1 >= rend) break;
        load(i + 2, za);
        __builtin_amdgcn_sched_barrier(0);
        process(i + 1, zb, 1);
        if (i + 2 >= rend) break;
    }

    // column partials of the block's 32 rows: a wave's columns are its own, no combination step
    float* part = partials + pd.part_off + (size_t)blockIdx.x * ld;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 256 * (wave + NW * c) + 4 * lane;
        if (col >= n) continue;
        f32x4 sum = acc[c];
        const int d = n - col;  // the dustbin column sits inside this float4 when d < 4
        if (d == 1) sum.y = acc_bin;
        if (d == 2) sum.z = acc_bin;
        if (d == 3) sum.w = acc_bin;
        *reinterpret_cast<f32x4*>(part + col) = sum;  // col < n, col % 4 == 0 -> col + 3 < ld
    }
    if (threadIdx.x == 0 && (n & 3) == 0) part[n] = acc_bin;  // ... or opens a float4 of its own
}

template <int NW, int NCH, bool NT = false>
__global__ __launch_bounds__(64 * NW) void lg_rows_wide_kernel(const float* __restrict__ zbuf, const PairDesc* __restrict__ pairs,
                                                               const SeqDesc* __restrict__ seqs, const int* __restrict__ counts,
                                                               float* __restrict__ rowvec, float* __restrict__ partials) {
    // per row of the block and wave: (max, sum of exp(. - max)) of the wave's slice. No barrier inside the row loop (a barrier per row makes
    // every wave wait for the slowest wave's loads of every row): the slices' statistics meet once, after the loop -- see extract_rows_wide_kernel
    __shared__ float xm[SW_ROWS][NW], xs[SW_ROWS][NW];
    const int p = blockIdx.y;
    const PairDesc pd = pairs[p];
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    const int r0 = blockIdx.x * SW_ROWS;
    if (r0 >= m || !sw_wide_tier<NW>(n)) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ld = pd.ld;
    const float* Z = zbuf + pd.z_off;
    const int vec0 = vec_off(s0, 2 * p);
    const float NEG = neg_inf();
    f32x4 cm[NCH], cs[NCH], za[NCH], zb[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        cm[c] = f32x4{NEG, NEG, NEG, NEG};
        cs[c] = za[c] = zb[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto process = [&](int i, f32x4(&zz)[NCH]) {
        float mw = NEG;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int col = 256 * (wave + NW * c) + 4 * lane;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = (col + e < n) ? zz[c][e] : NEG;  // also what makes the clamped loads of sw_load_slice_rsrc harmless
                zz[c][e] = t;
                mw = sw_max(mw, t);
            }
        }
        mw = wave_max(mw);
        const float mref = (mw != NEG) ? mw : 0.f;
        float sw = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = zz[c][e];
                sw += sw_exp2((t - mref) * SW_LOG2E);
                // online column statistics with ONE exponential: of (old maximum - new maximum) and (t - new maximum) one is 0 and the other
                // -|t - old maximum| (exp2(0) = 1 exactly: the same bits as two exponentials). The kernel is bound by the vector ALU, not by
                // HBM -- three quarter-rate v_exp_f32 per element were 12 of its 29 issue slots per element. A masked column (t = -inf, maximum
                // -inf) gets NaN here; columns >= n are never read.
                const float d = t - cm[c][e];
                const float ex = sw_exp2(-fabsf(d) * SW_LOG2E);
                const bool up = d > 0.f;  // the column maximum moves to this row
                cs[c][e] = cs[c][e] * (up ? ex : 1.f) + (up ? 1.f : ex);
                cm[c][e] = sw_max(cm[c][e], t);
            }
        }
        sw = wave_sum(sw);
        if (lane == 0) xm[i - r0][wave] = mw, xs[i - r0][wave] = sw;
    };
    const int rend = (r0 + SW_ROWS < m) ? r0 + SW_ROWS : m, last = rend - 1;
    // one row in flight behind the one being processed, every load unconditional and through the pair's buffer resource (sw_load_slice_rsrc:
    // exact wait counts -- with a branch around the loads every wait was vmcnt(0) and the prefetched row was waited for as well)
    SW_MAKE_ROW_RESOURCE(zres, ldb, Z, m, ld);
    int voff[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 256 * (wave + NW * c) + 4 * lane;
        voff[c] = col < n ? 4 * col : 0;
    }
    // past the last row: nothing is read (zres_empty)
    auto load = [&](int r, f32x4(&dst)[NCH]) {
        const bool real = r <= last;  // wave-uniform by construction: a scalar compare
        sw_load_slice_rsrc<NCH, NT>(real ? zres : zres_empty, real ? (unsigned)r * ldb : 0u, voff, dst);
    };
    int i = r0;
    load(i, za);
#pragma unroll 1
    for (;; i += 2) {
        load(i + 1, zb);
        __builtin_amdgcn_sched_barrier(0);  // the next row's loads are issued BEFORE the wait for this row's (the scheduler puts them after)
        process(i, za);
        if (i + 1 >= rend) break;
        load(i + 2, za);
        __builtin_amdgcn_sched_barrier(0);
        process(i + 1, zb);
        if (i + 2 >= rend) break;
    }
    // block partial per column: plane 0 = max, plane 1 = sum of exp(. - max), at part_off + block * 2 ld
    float* part = partials + pd.part_off + (size_t)blockIdx.x * 2 * ld;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 256 * (wave + NW * c) + 4 * lane;
        if (col >= n) continue;
        *reinterpret_cast<f32x4*>(part + col) = cm[c];  // col < n, col % 4 == 0 -> col + 3 < ld (columns >= n are never read)
        *reinterpret_cast<f32x4*>(part + ld + col) = cs[c];
    }
    // the one barrier of the kernel; thread r merges row r0 + r over the waves, wave 0 first (the order the per-row form summed in)
    __syncthreads();
    if ((int)threadIdx.x < rend - r0) {
        const int r = threadIdx.x;
        float mx = xm[r][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) mx = fmaxf(mx, xm[r][w]);
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += xs[r][w] * sw_exp2((xm[r][w] - mx) * SW_LOG2E);
        rowvec[vec0 + r0 + r] = logf(sum) + mx;
    }
}

template <bool SG, int NW, int NCH, bool NT = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 3 : 2) void extract_rows_wide_kernel(const float* __restrict__ zbuf, const PairDesc* __restrict__ pairs,
                                                                    const SeqDesc* __restrict__ seqs, const int* __restrict__ counts,
                                                                    const float* __restrict__ rowvec, const float* __restrict__ colvec,
                                                                    const float* __restrict__ zlogit, float* __restrict__ max0,
                                                                    int* __restrict__ idx0, float* __restrict__ partials, int n_above) {
    // per row of the block and wave: the best value / column of the wave's slice. The waves of a workgroup never wait for each other inside
    // the row loop (round 5: a barrier per row made every wave wait for the slowest wave's loads of EVERY row -- 70 % of the wave cycles parked,
    // profiles/r05_sq_counters.csv); the slices' candidates meet once, after the loop.
    __shared__ float xv[SW_ROWS][NW];
    __shared__ int xi[SW_ROWS][NW];
    // The column TERMS of a wave's slice (b_j; LightGlue: logsigmoid(z1_j) too) live in LDS, not in registers: they are read-only, a wave reads
    // back only what it wrote itself (no barrier), one ds_read_b128 per chunk and term and row (20 KiB per term at four waves x five chunks; 10 KB
    // of LDS reads per wave and row against 5 KB of HBM reads: far below the LDS rate). That frees 8 registers per chunk, which pay for a THIRD row
    // buffer (two rows in flight behind the one being processed) at three workgroups per CU instead of two: 120 KB in flight per CU against 40.
    __shared__ __attribute__((aligned(16))) float col_b[NW][NCH][256];
    __shared__ __attribute__((aligned(16))) float col_c[SG ? 1 : NW][SG ? 1 : NCH][256];
    const int p = blockIdx.y;
    const PairDesc pd = pairs[p];
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    const int r0 = blockIdx.x * SW_ROWS;
    // this launch's pairs: n_above < n <= the widest row NW waves hold. Arg-maxima do not depend on how a row is cut into slices, so --
    // unlike the Sinkhorn / double-softmax sweeps, whose sums do -- the launcher is free to pick the tiering (launch_extract_matches)
    if (r0 >= m || n <= n_above || n > (NW == 4 ? SW_WIDE4_COLS : SW_WIDE8_COLS)) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ld = pd.ld;
    const float* Z = zbuf + pd.z_off;
    const int vec0 = vec_off(s0, 2 * p), vec1 = vec_off(s1, 2 * p + 1);
    const float NEG = neg_inf();
    const float norm = SG ? -logf((float)m + (float)n) : 0.f;
    f32x4 cbv[NCH], za[NCH], zb[NCH];
    int cbi[NCH][4];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 256 * (wave + NW * c) + 4 * lane;
        za[c] = zb[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        // columns beyond n (the row's padding to 4 floats, the chunks past the matrix) carry a column term that turns every value into
        // -inf or NaN -- SuperGlue adds b_j (-inf), LightGlue subtracts it (+inf) -- so that no comparison below can select them whatever the
        // padding holds: the row loop needs no per-element bounds test (compiled as a branch per element until round 5: 0.26 of the roof)
        f32x4 bj = SG ? f32x4{NEG, NEG, NEG, NEG} : f32x4{-NEG, -NEG, -NEG, -NEG};
        f32x4 cj = f32x4{0.f, 0.f, 0.f, 0.f};
        cbv[c] = f32x4{NEG, NEG, NEG, NEG};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cbi[c][e] = SW_NO_INDEX;
            if (col + e < n) {
                bj[e] = colvec[vec1 + col + e];
                if (!SG) cj[e] = logsigmoid(zlogit[s1.row_off + col + e]);
            }
        }
        *reinterpret_cast<f32x4*>(&col_b[wave][c][4 * lane]) = bj;
        if (!SG) *reinterpret_cast<f32x4*>(&col_c[wave][c][4 * lane]) = cj;
    }
    float ci_lane = 0.f;  // logsigmoid(z0_i) of the block's rows, one per lane (see extract_rows_kernel)
    if (!SG && r0 + (lane & 31) < m) ci_lane = logsigmoid(zlogit[s0.row_off + r0 + (lane & 31)]);
    auto process = [&](int i, const f32x4(&zz)[NCH]) {
        const float a_i = rowvec[vec0 + i];
        const float c_i = SG ? 0.f : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ci_lane), __builtin_amdgcn_readfirstlane(i - r0)));
        float best = NEG;
        int bcode = -1;  // 4 * chunk + element of the lane's best value: wave-uniform constants (columns ascend with the code within a lane)
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const f32x4 bj = *reinterpret_cast<const f32x4*>(&col_b[wave][c][4 * lane]);
            const f32x4 cj = SG ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(&col_c[SG ? 0 : wave][SG ? 0 : c][4 * lane]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float val = assign_value<SG>(zz[c][e], a_i, bj[e], norm, c_i, cj[e]);  // -inf / NaN beyond column n: never selected
                const bool row_better = val > best;  // columns ascend within a lane: the first maximum wins
                best = row_better ? val : best;
                bcode = row_better ? 4 * c + e : bcode;
                const bool col_better = val > cbv[c][e];  // rows ascend
                cbv[c][e] = col_better ? val : cbv[c][e];
                cbi[c][e] = col_better ? i : cbi[c][e];
            }
            // the chunk's column state is final HERE (unpinned, the compiler computes every chunk's values first and updates the column state
            // after the wave reduction: 20 more live registers), and one chunk's column terms are in registers at a time (unfenced: all 40 are
            // read up front)
            asm volatile("" : "+v"(cbv[c]), "+v"(cbi[c][0]), "+v"(cbi[c][1]), "+v"(cbi[c][2]), "+v"(cbi[c][3]));
            __builtin_amdgcn_sched_barrier(0);
        }
        int bidx = bcode < 0 ? SW_NO_INDEX : 256 * (wave + NW * (bcode >> 2)) + 4 * lane + (bcode & 3);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int oj = __shfl_xor(bidx, off, 64);
            const bool take_ob = (ob > best) | ((ob == best) & (oj < bidx));  // bitwise: || and && compile to a branch per step
            best = take_ob ? ob : best;
            bidx = take_ob ? oj : bidx;
        }
        if (lane == 0) xv[i - r0][wave] = best, xi[i - r0][wave] = bidx;
    };
    const int rend = (r0 + SW_ROWS < m) ? r0 + SW_ROWS : m, last = rend - 1;
    // Two rows in flight behind the one being processed, every load unconditional (sw_load_slice_rsrc; past the block's last row the
    // last row is read again: <= 2 of 32 rows, L2 hits) so that the waits are exact counts: s_waitcnt vmcnt(2 NCH) before row i leaves rows
    // i + 1 and i + 2 in flight. Until this form every wait was vmcnt(0) -- each row paid the full memory latency (3 us per row at the cap, 70 %
    // of the wave cycles waiting, VALU issue 18 %: profiles/r05_sq_counters.csv) and only the occupancy overlapped anything.
    SW_MAKE_ROW_RESOURCE(zres, ldb, Z, m, ld);
    int voff[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 256 * (wave + NW * c) + 4 * lane;
        voff[c] = col < n ? 4 * col : 0;
    }
    // past the last row: nothing is read (zres_empty)
    auto load = [&](int r, f32x4(&dst)[NCH]) {
        const bool real = r <= last;  // wave-uniform by construction: a scalar compare
        sw_load_slice_rsrc<NCH, NT>(real ? zres : zres_empty, real ? (unsigned)r * ldb : 0u, voff, dst);
    };
    int i = r0;
    load(i, za);
#pragma unroll 1
    for (;; i += 2) {
        load(i + 1, zb);
        process(i, za);
        if (i + 1 >= rend) break;
        load(i + 2, za);
        process(i + 1, zb);
        if (i + 2 >= rend) break;
    }
    // block partial per column: plane 0 = best value, plane 1 = its row (as int bits), at part_off + block * 2 ld
    float* part = partials + pd.part_off + (size_t)blockIdx.x * 2 * ld;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int col = 256 * (wave + NW * c) + 4 * lane;
        if (col >= n) continue;
        *reinterpret_cast<f32x4*>(part + col) = cbv[c];
        *reinterpret_cast<int4*>(reinterpret_cast<int*>(part) + ld + col) = int4{cbi[c][0], cbi[c][1], cbi[c][2], cbi[c][3]};
    }
    // the one barrier of the kernel: every wave's candidates of every row of the block are in LDS; thread r merges row r0 + r over the
    // waves in ascending order (slices ascend with the wave index within a chunk but interleave across chunks: the tie rule compares indices)
    __syncthreads();
    if ((int)threadIdx.x < rend - r0) {
        const int r = threadIdx.x;
        float bv = xv[r][0];
        int bi = xi[r][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const float ov = xv[r][w];
            const int oi = xi[r][w];
            const bool take_ov = ov > bv || (ov == bv && oi < bi);  // selects, not a branch
            bv = take_ov ? ov : bv;
            bi = take_ov ? oi : bi;
        }
        max0[s0.row_off + r0 + r] = bv;
        idx0[s0.row_off + r0 + r] = (bi == SW_NO_INDEX) ? 0 : bi;  // all-NaN row: stay in range
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------------------------------------------

// GTSFM_SWEEP=lds forces the LDS-staged kernels for every width (A/B measurements)
static bool use_register_rows(int max_n) {
    static const char* which = getenv("GTSFM_SWEEP");
    return max_n <= SW_WIDE8_COLS && !(which && which[0] == 'l');
}

// Rows of the score matrix behind one block of column partials (sizes the partials buffer: matcher_api.hip)
int sweep_partial_rows(int max_n, int ext) { return use_register_rows(max_n) ? SW_ROWS : sweep_rows_per_block(max_n + ext); }

// widest row ONE wave holds in the match extraction (4 chunks); GTSFM_EXTRACT_NARROW_COLS = 2048 restores the round-4 bound (A/B measurements)
static int max_cols_narrow_extract() {
    const char* env = getenv("GTSFM_EXTRACT_NARROW_COLS");
    return (env && atoi(env) >= 256 && atoi(env) <= SW_MAX_COLS) ? atoi(env) : 1024;
}
static int chunks_for(int max_n) { return max_n <= 256 ? 1 : max_n <= 512 ? 2 : max_n <= 1024 ? 4 : 8; }
// register chunks per wave of the wide tiers: the batch's widest pair of the tier decides (3 .. 5)
static int wide_chunks_for(int max_n, int nw) {
    const int cap = nw == 4 ? SW_WIDE4_COLS : SW_WIDE8_COLS;
    const int nch = ceil_div(max_n < cap ? max_n : cap, 256 * nw);
    return nch < 3 ? 3 : nch;
}

#define SW_DISPATCH(nch, LAUNCH)    \
    switch (nch) {                  \
        case 1: { LAUNCH(1); break; } \
        case 2: { LAUNCH(2); break; } \
        case 4: { LAUNCH(4); break; } \
        default: { LAUNCH(8); break; } \
    }
#define SW_DISPATCH_WIDE(nw, nch, LAUNCH)          \
    {                                              \
        if ((nw) == 4) {                           \
            switch (nch) {                         \
                case 3: { LAUNCH(4, 3); break; }   \
                case 4: { LAUNCH(4, 4); break; }   \
                default: { LAUNCH(4, 5); break; }  \
            }                                      \
        } else {                                   \
            switch (nch) {                         \
                case 3: { LAUNCH(8, 3); break; }   \
                case 4: { LAUNCH(8, 4); break; }   \
                default: { LAUNCH(8, 5); break; }  \
            }                                      \
        }                                          \
    }

int launch_sinkhorn(const SweepArgs& a, float bin_score, int iters, hipStream_t stream) {
    if (a.npairs <= 0) return GTSFM_OK;
    if (!use_register_rows(a.max_n)) return launch_sinkhorn_lds(a, bin_score, iters, stream);
    int rc = launch_sg_fill_bins(a, bin_score, stream);  // also sets v = 0 (superglue.py:143); the `ot` output reads the bins
    if (rc != GTSFM_OK) return rc;
    // All pairs of the batch go through one launch per iteration. Iterating groups of pairs whose couplings matrices fit
    // the 256 MiB Infinity Cache together was measured (GTSFM_SINKHORN_GROUP_MB = 64 / 100 / 160 / 220 MiB at N = 2048,
    // 32 pairs): 0.288 / 0.174 / 0.161 / 0.129 ms per iteration against 0.123 ms ungrouped -- the smaller grids leave CUs
    // idle and pay the launch gaps more often than the cache saves. The switch stays for experiments.
    static const char* env = getenv("GTSFM_SINKHORN_GROUP_MB");
    const double budget_mb = env ? atof(env) : 0.0;
    const double z_mb = (double)(a.max_m + 1) * ((a.max_n + 1 + 3) / 4 * 4) * 4.0 / (1024.0 * 1024.0);
    int group = a.npairs;
    if (budget_mb > 0.0) {
        group = (int)(budget_mb / z_mb);
        group = group < 1 ? 1 : (group > a.npairs ? a.npairs : group);
    }
    const int nch = chunks_for(a.max_n);
    // nontemporal score-matrix reads once the launch's matrices cannot stay in the Infinity Cache between sweeps (see sw_load_slice_rsrc)
    const char* nt_env = getenv("GTSFM_SWEEP_NT_MB");  // read per call (tests switch it)
    const double nt_mb = nt_env ? atof(nt_env) : 256.0;
    for (int pair0 = 0; pair0 < a.npairs; pair0 += group) {
        const int g = (a.npairs - pair0 < group) ? a.npairs - pair0 : group;
        const bool nt = g * z_mb > nt_mb;
        const dim3 grid_rows(ceil_div(a.max_m + 1, SW_ROWS), g), grid_cols(ceil_div(a.max_n + 1, 256), g);
        for (int it = 0; it < iters; ++it) {
#define SW_SINKHORN_ARGS stream, a.zbuf, a.pairs, a.seqs, a.counts, a.rowvec, a.colvec, a.partials, bin_score, pair0
#define SW_LAUNCH_SINKHORN(N)                                                                           \
    if (nt) hipLaunchKernelGGL((sinkhorn_rows_kernel<N, true>), grid_rows, dim3(256), 0, SW_SINKHORN_ARGS); \
    else hipLaunchKernelGGL((sinkhorn_rows_kernel<N, false>), grid_rows, dim3(256), 0, SW_SINKHORN_ARGS)
#define SW_LAUNCH_SINKHORN_WIDE(NW, N)                                                                                 \
    if (nt) hipLaunchKernelGGL((sinkhorn_rows_wide_kernel<NW, N, true>), grid_rows, dim3(64 * NW), 0, SW_SINKHORN_ARGS); \
    else hipLaunchKernelGGL((sinkhorn_rows_wide_kernel<NW, N, false>), grid_rows, dim3(64 * NW), 0, SW_SINKHORN_ARGS)
            // every tier the batch can hold; the blocks of pairs of another tier return at once
            if (a.max_n > SW_WIDE4_COLS) SW_DISPATCH_WIDE(8, wide_chunks_for(a.max_n, 8), SW_LAUNCH_SINKHORN_WIDE)
            if (a.max_n > SW_MAX_COLS) SW_DISPATCH_WIDE(4, wide_chunks_for(a.max_n, 4), SW_LAUNCH_SINKHORN_WIDE)
            SW_DISPATCH(nch, SW_LAUNCH_SINKHORN)
            hipLaunchKernelGGL(sinkhorn_cols_kernel, grid_cols, dim3(256), 0, stream, a.pairs, a.seqs, a.counts, a.partials, a.colvec, pair0);
        }
    }
    GTSFM_CHECK_LAUNCH("sinkhorn_rows/cols_kernel");
    return GTSFM_OK;
}

int launch_double_softmax_lse(const SweepArgs& a, hipStream_t stream) {
    if (a.npairs <= 0 || a.max_m <= 0 || a.max_n <= 0) return GTSFM_OK;
    if (!use_register_rows(a.max_n)) return launch_double_softmax_lse_lds(a, stream);
    const dim3 grid_rows(ceil_div(a.max_m, SW_ROWS), a.npairs), grid_cols(ceil_div(a.max_n, 256), a.npairs);
    // nontemporal score-matrix reads once the launch's matrices exceed the Infinity Cache: nothing of them survives until the extraction sweep
    // anyway (see sw_load_slice_rsrc; GTSFM_SWEEP_NT_MB as for the other sweeps; the same values are loaded)
    const char* nt_env = getenv("GTSFM_SWEEP_NT_MB");
    const bool nt = (double)a.npairs * a.max_m * a.max_n * 4.0 / (1024.0 * 1024.0) > (nt_env ? atof(nt_env) : 256.0);
#define SW_LG_ARGS stream, a.zbuf, a.pairs, a.seqs, a.counts, a.rowvec, a.partials
#define SW_LAUNCH_LG(N)                                                                       \
    if (nt) hipLaunchKernelGGL((lg_rows_kernel<N, true>), grid_rows, dim3(256), 0, SW_LG_ARGS); \
    else hipLaunchKernelGGL((lg_rows_kernel<N, false>), grid_rows, dim3(256), 0, SW_LG_ARGS)
#define SW_LAUNCH_LG_WIDE(NW, N)                                                                           \
    if (nt) hipLaunchKernelGGL((lg_rows_wide_kernel<NW, N, true>), grid_rows, dim3(64 * NW), 0, SW_LG_ARGS); \
    else hipLaunchKernelGGL((lg_rows_wide_kernel<NW, N, false>), grid_rows, dim3(64 * NW), 0, SW_LG_ARGS)
    if (a.max_n > SW_WIDE4_COLS) SW_DISPATCH_WIDE(8, wide_chunks_for(a.max_n, 8), SW_LAUNCH_LG_WIDE)
    if (a.max_n > SW_MAX_COLS) SW_DISPATCH_WIDE(4, wide_chunks_for(a.max_n, 4), SW_LAUNCH_LG_WIDE)
    SW_DISPATCH(chunks_for(a.max_n), SW_LAUNCH_LG)
    hipLaunchKernelGGL(lg_cols_kernel, grid_cols, dim3(256), 0, stream, a.pairs, a.seqs, a.counts, a.partials, a.colvec);
    GTSFM_CHECK_LAUNCH("lg_rows/cols_kernel");
    return GTSFM_OK;
}

int launch_extract_matches(const SweepArgs& a, int superglue, const float* zlogit, float threshold, float* max0, int* idx0, int* idx1,
                           int* matches, float* mscores, hipStream_t stream) {
    if (a.npairs <= 0 || a.max_m <= 0 || a.max_n <= 0) return GTSFM_OK;
    if (!use_register_rows(a.max_n)) return launch_extract_matches_lds(a, superglue, zlogit, threshold, max0, idx0, idx1, matches, mscores, stream);
    const dim3 grid_rows(ceil_div(a.max_m, SW_ROWS), a.npairs), grid_cols(ceil_div(a.max_n, 256), a.npairs);
    // One wave per row up to SW_EXTRACT_NARROW_COLS = 1024 columns (4 register chunks), the waves of a workgroup share a row above: four up to
    // 5120 columns, eight up to 10240. Until round 5 one wave held rows of up to 2048 columns (8 chunks: 256 VGPRs + 81 .. 150 spilled to AGPRs,
    // 1.7 TB/s at N = 2048) and LightGlue took eight waves for every wider row (with a bounds BRANCH per element its four-wave form needed 256
    // VGPRs + 44 spilled at GTSfM's cap). Round 5, in this order: a branch-free row loop (poisoned column terms, selects) and no barrier per row
    // (0.767 -> 0.49 ms for LightGlue, 16 pairs at the cap); then no branch around any LOAD (sw_load_slice_rsrc: exact wait counts where every
    // wait had been vmcnt(0)) with the column terms in LDS and three workgroups per CU (153 VGPRs): 0.49 -> 0.33 ms. Arg-maxima do not depend on
    // how a row is cut into slices, so -- unlike the Sinkhorn / double-softmax sweeps, whose sums do -- this launcher is free to choose;
    // GTSFM_EXTRACT_WAVES=8 sends every row above 1024 columns to the eight-wave tier, =4 is the default.
    const int narrow_max = max_cols_narrow_extract();
#define SW_LAUNCH_EXTRACT_SG(N)                                                                                                         \
    hipLaunchKernelGGL((extract_rows_kernel<true, N>), grid_rows, dim3(256), 0, stream, a.zbuf, a.pairs, a.seqs, a.counts, a.rowvec, a.colvec, \
                       zlogit, max0, idx0, a.partials, narrow_max)
#define SW_LAUNCH_EXTRACT_LG(N)                                                                                                          \
    hipLaunchKernelGGL((extract_rows_kernel<false, N>), grid_rows, dim3(256), 0, stream, a.zbuf, a.pairs, a.seqs, a.counts, a.rowvec, a.colvec, \
                       zlogit, max0, idx0, a.partials, narrow_max)
#define SW_EXTRACT_WIDE_ARGS(NW) grid_rows, dim3(64 * NW), 0, stream, a.zbuf, a.pairs, a.seqs, a.counts, a.rowvec, a.colvec, zlogit, max0, idx0, a.partials, (NW) == 8 ? above8 : narrow_max
#define SW_LAUNCH_EXTRACT_SG_WIDE(NW, N)                                                                \
    if (nt) hipLaunchKernelGGL((extract_rows_wide_kernel<true, NW, N, true>), SW_EXTRACT_WIDE_ARGS(NW)); \
    else hipLaunchKernelGGL((extract_rows_wide_kernel<true, NW, N, false>), SW_EXTRACT_WIDE_ARGS(NW))
#define SW_LAUNCH_EXTRACT_LG_WIDE(NW, N)                                                                 \
    if (nt) hipLaunchKernelGGL((extract_rows_wide_kernel<false, NW, N, true>), SW_EXTRACT_WIDE_ARGS(NW)); \
    else hipLaunchKernelGGL((extract_rows_wide_kernel<false, NW, N, false>), SW_EXTRACT_WIDE_ARGS(NW))
    // the extraction is the LAST reader of the matrices: once the launch's matrices exceed the Infinity Cache, nontemporal reads (same values)
    // leave it to data somebody will read again (see sw_load_slice_rsrc; GTSFM_SWEEP_NT_MB as for the Sinkhorn sweeps)
    const char* nt_env = getenv("GTSFM_SWEEP_NT_MB");
    const bool nt = (double)a.npairs * a.max_m * a.max_n * 4.0 / (1024.0 * 1024.0) > (nt_env ? atof(nt_env) : 256.0);
    const char* ew_env = getenv("GTSFM_EXTRACT_WAVES");
    const bool four_up_to_5120 = !(ew_env && ew_env[0] == '8');
    const int above8 = four_up_to_5120 ? SW_WIDE4_COLS : narrow_max;
    const int narrow_chunks = chunks_for(a.max_n < narrow_max ? a.max_n : narrow_max);
    if (superglue) {
        if (a.max_n > above8) SW_DISPATCH_WIDE(8, wide_chunks_for(a.max_n, 8), SW_LAUNCH_EXTRACT_SG_WIDE)
        if (four_up_to_5120 && a.max_n > narrow_max) SW_DISPATCH_WIDE(4, wide_chunks_for(a.max_n, 4), SW_LAUNCH_EXTRACT_SG_WIDE)
        SW_DISPATCH(narrow_chunks, SW_LAUNCH_EXTRACT_SG)
    } else {
        if (a.max_n > above8) SW_DISPATCH_WIDE(8, wide_chunks_for(a.max_n, 8), SW_LAUNCH_EXTRACT_LG_WIDE)
        if (four_up_to_5120 && a.max_n > narrow_max) SW_DISPATCH_WIDE(4, wide_chunks_for(a.max_n, 4), SW_LAUNCH_EXTRACT_LG_WIDE)
        SW_DISPATCH(narrow_chunks, SW_LAUNCH_EXTRACT_LG)
    }
    hipLaunchKernelGGL(extract_cols_kernel, grid_cols, dim3(256), 0, stream, a.pairs, a.seqs, a.counts, a.partials, idx1);
    GTSFM_CHECK_LAUNCH("extract_rows/cols_kernel");
    return launch_mutual_matches(a, threshold, max0, idx0, idx1, matches, mscores, stream);
}
