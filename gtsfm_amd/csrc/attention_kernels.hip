// Multi-head softmax attention on exact-fp32 MFMA for gfx950 (flash-style: the N x M probability tensor is never
// materialised). One kernel serves
//   SuperGlue self/cross attention    thirdparty/SuperGluePretrainedNetwork/models/superglue.py:85-89,98-106
//   LightGlue self attention (after rotary) and both directions of its bidirectional cross attention
// Inputs are token-major [tokens][channels] with head h occupying channels [64h, 64h+64) (SuperGlue's
// "head = fast axis" layout is removed by permuting the projection weights at load time).
//
// Tiling (head_dim = 64): workgroup = 4 waves = 128 queries of one head of one problem; each wave owns 32 queries and
// walks the keys in tiles of 64 staged through LDS (K as [key][d], V transposed as [d][key], row stride 68 floats).
// The score tile is computed TRANSPOSED, S^T[key][q] = K Q^T, so that a query is a lane: the softmax row reductions
// are in-register (+ one cross-half shuffle), and the exponentiated accumulator registers are directly the B operand
// of the second product O^T[d][q] += V^T[d][key] P^T[key][q] -- no LDS round trip for P, no layout shuffles.
// Per 64-key tile and wave: 64 + 64 v_mfma_f32_32x32x2_f32.

#include <stdlib.h>

#include "attention_kernels.h"
#include "mfma_tiles.h"

#define AT_KT 64       // keys per tile
#define AT_QW 32       // queries per wave
#define AT_QB 128      // queries per workgroup
#define AT_ROW 68      // LDS row stride (floats)
#define AT_TILE (AT_KT * AT_ROW)
#define AT_REBASE 8.0f  // rebase the softmax reference when the running maximum moved by more than this (base-2 units)

// Developer timeline (tools/trace_attention.hip builds this file with -DGTSFM_TRACE; the product build has none of it):
// per wave, shader-clock cycles summed over all key tiles for each segment of the tile loop. The stamps sit where the
// LDS queue is empty anyway (s_memtime returns through lgkmcnt) and are fenced against instruction motion.
#ifdef GTSFM_TRACE
__device__ unsigned long long* g_attn_trace;  // [workgroup][wave][8]: S issue, softmax, PV issue, barrier 1, store + barrier 2, total, hw id, tiles
#define TRACE_DECL unsigned t_prev = (unsigned)__builtin_amdgcn_s_memtime(); const unsigned t_begin = t_prev; unsigned seg[5] = {0, 0, 0, 0, 0};
#define TRACE_SEG(k)                                                  \
    {                                                                 \
        __builtin_amdgcn_sched_barrier(0);                            \
        const unsigned t_now = (unsigned)__builtin_amdgcn_s_memtime(); \
        seg[k] += t_now - t_prev;                                     \
        t_prev = t_now;                                               \
        __builtin_amdgcn_sched_barrier(0);                            \
    }
#else
#define TRACE_DECL
#define TRACE_SEG(k)
#endif

__global__ __launch_bounds__(256, 3) void attention_mfma_kernel(AttnParams p) {
    // LDS: one K tile and one V tile ([key][d], row stride 68; 34 KiB -> three workgroups per CU, which keeps the matrix
    // pipe fed better than two workgroups with double-buffered tiles: 83 % vs 80 % of peak). While the waves work on
    // tile t, tile t+1 travels global -> registers (issued before the MFMAs) and is written to LDS after them, between
    // two barriers: no exposed global latency.
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ks = lds;
    float* Vs = lds + AT_TILE;
    // XCD-aware block order (speed only): workgroups are dispatched round-robin over the 8 XCDs, each with a private
    // L2. All query tiles of one (problem, head) share the same K / V, so they are given linear ids that are congruent
    // mod 8 -> same XCD -> K / V are fetched into ONE L2 instead of eight.
    const int b = blockIdx.x;
    const int groups = p.heads * p.nproblems;
    const int k_in_xcd = b >> 3;
    const int g = (k_in_xcd / p.qtiles) * 8 + (b & 7);
    if (g >= groups) return;
    const int h = g % p.heads;
    const AttnProblem pr = p.problems[g / p.heads];
    const int nq = p.counts[pr.q_cnt_idx], nk = p.counts[pr.k_cnt_idx];
    const int q0 = (k_in_xcd % p.qtiles) * AT_QB;
    if (q0 >= nq) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int qrow = q0 + wave * AT_QW + j;
    const bool qvalid = qrow < nq;

    // Q fragment (B operand of S^T = K Q^T): lane (q = j, kh) holds Q[q][8t + 4kh .. +3], t = 0..7
    f32x4 qreg[8];
    {
        const float* qp = p.q + (size_t)(pr.q_off + (qvalid ? qrow : 0)) * p.ldq + h * 64 + kh * 4;
        // softmax runs in the base-2 domain (exp(x) = exp2(x log2 e), one v_exp_f32 per element); the factor
        // scale * log2(e) is folded into Q once instead of into every score
        const float scale2 = p.scale * 1.44269504088896340736f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            f32x4 v = *reinterpret_cast<const f32x4*>(qp + t * 8);
            if (!qvalid) v = f32x4{0.f, 0.f, 0.f, 0.f};
            qreg[t] = v * scale2;
        }
    }
    f32x16 o0, o1;  // O^T: rows d 0..31 / 32..63, column q
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float m = 0.f, l = 0.f;  // m: lazy reference maximum (set by the first tile)

    const float* kbase = p.k + (size_t)pr.k_off * p.ldk + h * 64;
    const float* vbase = p.v + (size_t)pr.k_off * p.ldv + h * 64;

    // staging: thread t moves float4 #(t + 256 i), i = 0..3, of the K tile and of the V tile (64 rows x 16 float4 each)
    const int srow = tid >> 4, sq = tid & 15;
    f32x4 kst[4], vst[4];
    auto stage_load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = k0 + srow + 16 * i;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (key < nk) {
                kv = *reinterpret_cast<const f32x4*>(kbase + (size_t)key * p.ldk + sq * 4);
                vv = *reinterpret_cast<const f32x4*>(vbase + (size_t)key * p.ldv + sq * 4);
            }
            kst[i] = kv, vst[i] = vv;
        }
    };
    auto stage_store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(&Ks[(srow + 16 * i) * AT_ROW + sq * 4]) = kst[i];
            *reinterpret_cast<f32x4*>(&Vs[(srow + 16 * i) * AT_ROW + sq * 4]) = vst[i];
        }
    };

    const int ntiles = (nk + AT_KT - 1) / AT_KT;
    stage_load(0);
    stage_store(0);
    __syncthreads();
    TRACE_DECL
    for (int t = 0; t < ntiles; ++t) {
        const int k0 = t * AT_KT;
        const float* Kt = Ks;
        const float* Vt = Vs;
        if (t + 1 < ntiles) stage_load(k0 + AT_KT);

        // S^T = K Q^T  (two 32-key tiles). The accumulators start at -m (m = reference maximum of this query, see
        // below), so the MFMA chain delivers s - m directly and the softmax needs no subtraction pass.
        f32x16 s0, s1;
        const float neg_m = -m;
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = s1[r] = neg_m;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(&Kt[j * AT_ROW + u * 8 + kh * 4]);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(&Kt[(32 + j) * AT_ROW + u * 8 + kh * 4]);
            mt_step(s0, s1, a0, a1, qreg[u]);
        }
        TRACE_SEG(0)
        // Everything between the two MFMA phases is latency-bound VALU / LDS / barrier work that crawls when the other
        // two waves of the SIMD win the issue arbitration with their MFMAs; while it lasts this wave offers the matrix
        // pipe nothing. Run it at raised priority so the wave is back to feeding the pipe as soon as possible.
        __builtin_amdgcn_s_setprio(3);
        // mask (last tile only)
        if (k0 + AT_KT > nk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (key >= nk) s0[r] = -__builtin_inff();
                if (key + 32 >= nk) s1[r] = -__builtin_inff();
            }
        }
        // Online softmax with a LAZY reference maximum (per query = per lane; the two lane halves hold different keys of
        // the same query): m follows the running maximum only when that has moved by more than AT_REBASE (base-2 units),
        // so exp2(s - m) <= 2^AT_REBASE stays far from overflow while most tiles skip the rebase (subtract + rescale of O
        // and l) entirely. The VALU work of a tile -- which crawls while the other waves of the SIMD keep the matrix
        // pipe busy -- drops from ~180 to ~90 instructions. out = O / l does not depend on the choice of m.
        float mloc = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, fmaxf(s0[r], s1[r]));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));  // finite on every tile: key k0 is always valid
        const bool rebase = (t == 0) || (mloc > AT_REBASE);
        if (__any(rebase)) {  // wave-uniform
            const float d = rebase ? mloc : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s0[r] -= d;
                s1[r] -= d;
            }
            if (t > 0) {  // (first tile: O = l = 0 and m = 0)
                const float alpha = __builtin_amdgcn_exp2f(-d);
                l *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o0[r] *= alpha;
                    o1[r] *= alpha;
                }
            }
            m += d;
        }
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(s0[r]);
            s1[r] = __builtin_amdgcn_exp2f(s1[r]);
            lsum += s0[r] + s1[r];
        }
        lsum += __shfl_xor(lsum, 32, 64);
        l += lsum;
        __builtin_amdgcn_s_setprio(0);
        TRACE_SEG(1)
        // O^T += V^T P^T. Accumulator register r of S^T tile T holds key 32T + (r&3) + 8(r>>2) + 4kh, so it IS the B
        // operand of k-step r; the A operand V^T[d = lane][key] is a conflict-free row read of the row-major V tile.
#pragma unroll
        for (int T = 0; T < 2; ++T) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int key = 32 * T + 8 * g + 4 * kh;
                f32x4 a0, a1, b;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a0[e] = Vt[(key + e) * AT_ROW + j];
                    a1[e] = Vt[(key + e) * AT_ROW + 32 + j];
                    b[e] = T ? s1[4 * g + e] : s0[4 * g + e];
                }
                mt_step(o0, o1, a0, a1, b);
            }
        }
        TRACE_SEG(2)
        if (t + 1 < ntiles) {
            __builtin_amdgcn_s_setprio(3);
            __syncthreads();
            TRACE_SEG(3)
            stage_store(0);
            __syncthreads();
            __builtin_amdgcn_s_setprio(0);
            TRACE_SEG(4)
        }
    }
#ifdef GTSFM_TRACE
    if (lane == 0 && g_attn_trace) {
        unsigned long long* o = g_attn_trace + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int k = 0; k < 5; ++k) o[k] = seg[k];
        o[5] = (unsigned)__builtin_amdgcn_s_memtime() - t_begin;
        o[6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID
        o[7] = ntiles;
    }
#endif

    if (!qvalid) return;
    const float inv = (l > 0.f) ? 1.0f / l : 0.f;
    float* op = p.out + (size_t)(pr.q_off + qrow) * p.ldo + h * 64 + kh * 4;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 v0 = {o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
        const f32x4 v1 = {o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
        *reinterpret_cast<f32x4*>(op + 8 * g) = v0;
        *reinterpret_cast<f32x4*>(op + 32 + 8 * g) = v1;
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Default kernel since round 2 (GTSFM_ATTENTION=mfma selects the one above): K / V tiles by LDS-DMA into an XOR-swizzled unpadded image and a second
// score tile in the freed registers, so that the softmax of tile t runs in the shadow of the S MFMAs of tile t + 1
// inside one wave. Derivation, LDS-bank argument and round-1 status: tools/experimental/attention_dma.hip.
// ---------------------------------------------------------------------------------------------------------------
#define ATD_TILE_FLOATS (AT_KT * 64)
#ifndef ATD_WGS_PER_CU
#define ATD_WGS_PER_CU 2
#endif

__device__ __forceinline__ void mfma8(f32x16& acc0, f32x16& acc1, const f32x4 a0, const f32x4 a1, const f32x4 b) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b.x, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b.y, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b.z, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b.w, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b.w, acc1, 0, 0, 0);
}

__global__ __launch_bounds__(256, ATD_WGS_PER_CU) void attention_dma_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ks = lds;                // [64 keys][64 floats], swizzled
    float* Vs = lds + ATD_TILE_FLOATS;  // same
    // XCD-aware block order as in attention_mfma_kernel
    const int b = blockIdx.x;
    const int groups = p.heads * p.nproblems;
    const int k_in_xcd = b >> 3;
    const int g = (k_in_xcd / p.qtiles) * 8 + (b & 7);
    if (g >= groups) return;
    const int h = g % p.heads;
    const AttnProblem pr = p.problems[g / p.heads];
    const int nq = p.counts[pr.q_cnt_idx], nk = p.counts[pr.k_cnt_idx];
    const int q0 = (k_in_xcd % p.qtiles) * AT_QB;
    if (q0 >= nq) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int qrow = q0 + wave * 32 + j;
    const bool qvalid = qrow < nq;

    f32x4 qreg[8];  // Q fragment, pre-scaled by scale * log2(e) (base-2 softmax)
    {
        const float* qp = p.q + (size_t)(pr.q_off + (qvalid ? qrow : 0)) * p.ldq + h * 64 + kh * 4;
        const float scale2 = p.scale * 1.44269504088896340736f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            f32x4 v = *reinterpret_cast<const f32x4*>(qp + t * 8);
            if (!qvalid) v = f32x4{0.f, 0.f, 0.f, 0.f};
            qreg[t] = v * scale2;
        }
    }
    const float* kbase = p.k + (size_t)pr.k_off * p.ldk + h * 64;
    const float* vbase = p.v + (size_t)pr.k_off * p.ldv + h * 64;
    const int ntiles = (nk + AT_KT - 1) / AT_KT;
    if (nk <= 0) {  // no keys: the output rows are zero (as attention_mfma_kernel); uniform for the workgroup
        if (qvalid) {
            float* op = p.out + (size_t)(pr.q_off + qrow) * p.ldo + h * 64 + kh * 32;
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(op + 4 * i) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }

    // DMA of one 64 x 64 tile: 16 instructions per workgroup, 4 per wave; instruction i of wave w covers rows
    // 16 w + 4 i .. + 3; lane l writes position l % 16 of row (l / 16) and fetches chunk (l % 16) ^ (row & 15)
    const int drow = lane >> 4, dpos = lane & 15;
    auto tile_dma = [&](const float* base, int ld, int k0, float* dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rb = 16 * wave + 4 * i, r = rb + drow;
            int key = k0 + r;
            key = key < nk ? key : nk - 1;  // clamp: keys beyond nk are masked to -inf (their V rows meet P = 0)
            __builtin_amdgcn_global_load_lds(base + (size_t)key * ld + ((dpos ^ (r & 15)) << 2), dst + rb * 64, 16, 0, 0);
        }
    };
    auto kfrag = [&](int row, int u) {  // floats 8 u + 4 kh .. + 3 of key row `row`
        return *reinterpret_cast<const f32x4*>(Ks + row * 64 + (((2 * u + kh) ^ (row & 15)) << 2));
    };
    auto velem = [&](int key, int d) { return Vs[key * 64 + ((((d >> 2) ^ (key & 15)) << 2) | (d & 3))]; };
    auto s_phase = [&](f32x16& s0, f32x16& s1, float neg_m) {  // S^T tile = K Q^T - m (accumulators start at -m)
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = s1[r] = neg_m;
#pragma unroll
        for (int u = 0; u < 8; ++u) mfma8(s0, s1, kfrag(j, u), kfrag(32 + j, u), qreg[u]);
    };

    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float m = 0.f, l = 0.f;  // lazy reference maximum and running denominator, as in attention_mfma_kernel
    f32x16 sc0, sc1;         // score tile being soft-maxed
    f32x16 sn0, sn1;         // score tile being accumulated

    // prologue: K(0) -> S(0); then K(1) and V(0) in flight
    tile_dma(kbase, p.ldk, 0, Ks);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    __syncthreads();
    s_phase(sc0, sc1, 0.f);
    __syncthreads();  // every wave is done reading K(0)
    if (ntiles > 1) tile_dma(kbase, p.ldk, AT_KT, Ks);
    tile_dma(vbase, p.ldv, 0, Vs);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();

    // (A variant with the S MFMAs and the softmax VALU arranged in shared straight-line blocks -- rebase branches moved
    // between two halves of the S phase -- was measured in round 2: 256 registers, 495 vs 499 image-pairs/s in the
    // workload, not kept.)
    for (int t = 0; t < ntiles; ++t) {
        const int k0 = t * AT_KT;
        const bool more = t + 1 < ntiles;
        // ---- phase 1: S(t+1) MFMAs in whose shadow the softmax of tile t runs
        // (the reference maximum used for S(t+1)'s accumulator start is the one BEFORE tile t's possible rebase; the
        // difference is applied below when that tile is soft-maxed: its own rebase test sees scores relative to the old m)
        const float m_start = m;
        if (more) s_phase(sn0, sn1, -m_start);
        if (k0 + AT_KT > nk) {  // mask (last tile only)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (key >= nk) sc0[r] = -__builtin_inff();
                if (key + 32 >= nk) sc1[r] = -__builtin_inff();
            }
        }
        float mloc = fmaxf(sc0[0], sc1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, fmaxf(sc0[r], sc1[r]));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const bool rebase = (t == 0) || (mloc > AT_REBASE);
        float d = 0.f;
        if (__any(rebase)) {  // wave-uniform
            d = rebase ? mloc : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc0[r] -= d;
                sc1[r] -= d;
            }
            if (t > 0) {
                const float alpha = __builtin_amdgcn_exp2f(-d);
                l *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o0[r] *= alpha;
                    o1[r] *= alpha;
                }
            }
            m += d;
        }
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc0[r] = __builtin_amdgcn_exp2f(sc0[r]);
            sc1[r] = __builtin_amdgcn_exp2f(sc1[r]);
            lsum += sc0[r] + sc1[r];
        }
        lsum += __shfl_xor(lsum, 32, 64);
        l += lsum;
        // the next tile was accumulated relative to m_start; bring it to the (possibly rebased) reference
        if (more && __any(d != 0.f)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sn0[r] -= d;
                sn1[r] -= d;
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);  // own V(t) DMA landed (issued one phase ago)
        __syncthreads();                     // B1: K buffer free, V(t) visible
        if (t + 2 < ntiles) tile_dma(kbase, p.ldk, k0 + 2 * AT_KT, Ks);
        // ---- phase 2: O^T += V^T P^T
#pragma unroll
        for (int T = 0; T < 2; ++T) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int key = 32 * T + 8 * gq + 4 * kh;
                f32x4 a0, a1, bb;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a0[e] = velem(key + e, j);
                    a1[e] = velem(key + e, 32 + j);
                    bb[e] = T ? sc1[4 * gq + e] : sc0[4 * gq + e];
                }
                mfma8(o0, o1, a0, a1, bb);
            }
        }
        if (more) {
            __builtin_amdgcn_s_waitcnt(0x0f70);  // own K(t+2) DMA landed
            __syncthreads();                     // B2: V buffer free, K(t+2) visible
            tile_dma(vbase, p.ldv, k0 + AT_KT, Vs);
            sc0 = sn0, sc1 = sn1;
        }
    }
    if (!qvalid) return;
    const float inv = (l > 0.f) ? 1.0f / l : 0.f;
    float* op = p.out + (size_t)(pr.q_off + qrow) * p.ldo + h * 64 + kh * 4;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        *reinterpret_cast<f32x4*>(op + 8 * gq) = f32x4{o0[4 * gq] * inv, o0[4 * gq + 1] * inv, o0[4 * gq + 2] * inv, o0[4 * gq + 3] * inv};
        *reinterpret_cast<f32x4*>(op + 32 + 8 * gq) = f32x4{o1[4 * gq] * inv, o1[4 * gq + 1] * inv, o1[4 * gq + 2] * inv, o1[4 * gq + 3] * inv};
    }
}


int launch_attention(const AttnParams& p, int nproblems, int max_q, hipStream_t stream) {
    GTSFM_CHECK_ARG(p.ldq % 4 == 0 && p.ldk % 4 == 0 && p.ldv % 4 == 0 && p.ldo % 4 == 0, "attention: leading dimensions must be multiples of 4");
    GTSFM_CHECK_ARG(p.heads > 0, "attention: heads must be positive");
    if (nproblems <= 0 || max_q <= 0) return GTSFM_OK;
    AttnParams q = p;
    q.qtiles = ceil_div(max_q, AT_QB);
    q.nproblems = nproblems;
    const int groups = p.heads * nproblems;
    dim3 grid(ceil_div(groups, 8) * 8 * q.qtiles);
    // Default since round 2: the LDS-DMA kernel. In isolation both reach 84-85 % of the fp32 MFMA peak at N = 2048; inside
    // the detect+match workload (two streams) the DMA kernel is 1.2 % faster end to end (501 vs 495 image-pairs/s, A/B run twice,
    // profiles/r02_attention_ab.txt). GTSFM_ATTENTION=mfma selects the register-staged kernel of round 1.
    static const char* which = getenv("GTSFM_ATTENTION");
    if (which && which[0] == 'm')
        hipLaunchKernelGGL(attention_mfma_kernel, grid, dim3(256), (size_t)2 * AT_TILE * sizeof(float), stream, q);
    else
        hipLaunchKernelGGL(attention_dma_kernel, grid, dim3(256), (size_t)2 * ATD_TILE_FLOATS * sizeof(float), stream, q);
    GTSFM_CHECK_LAUNCH("attention kernel");
    return GTSFM_OK;
}
