// Multi-head softmax attention on exact-fp32 MFMA for gfx950 (flash-style: the N x M probability tensor is never
// materialised). One kernel serves
//   SuperGlue self/cross attention    thirdparty/SuperGluePretrainedNetwork/models/superglue.py:85-89,98-106
//   LightGlue self attention (after rotary) and both directions of its bidirectional cross attention
// Inputs are token-major [tokens][channels] with head h occupying channels [64h, 64h+64) (SuperGlue's
// "head = fast axis" layout is removed by permuting the projection weights at load time).
//
// Tiling (head_dim = 64): workgroup = 4 waves = 128 queries of one head of one problem; each wave owns 32 queries and
// walks the keys in tiles of 64 that travel global -> LDS by DMA (global_load_lds) into an XOR-swizzled unpadded image.
// The score tile is computed TRANSPOSED, S^T[key][q] = K Q^T, so that a query is a lane: the softmax row reductions
// are in-register (+ one cross-half shuffle), and the exponentiated accumulator registers are directly the B operand
// of the second product O^T[d][q] += V^T[d][key] P^T[key][q] -- no LDS round trip for P, no layout shuffles. A second
// score tile lives in registers so that the softmax of tile t runs in the shadow of the S MFMAs of tile t + 1 inside
// one wave. Per 64-key tile and wave: 64 + 64 v_mfma_f32_32x32x2_f32. Derivation, LDS-bank argument and the round-1 /
// round-2 measurements (register-staged predecessor, double-buffered K/V, ...): docs/experimental/attention_dma.hip, docs/HISTORY.md.
//
// Key SEGMENTS (round 3; their length a function of the problem's size since round 6: at_seg_tiles below). The keys of a problem are cut into
// segments of 8 .. 16 tiles (512 .. 1024 keys). Every segment runs
// the online softmax from a FRESH state (O = 0, l = 0, reference maximum from its first tile) and the segments' (O, m, l) are
// merged in ascending order with one fixed formula (at_merge). Two schedules execute exactly this arithmetic:
//   fused  (attention_dma_kernel<false>): a workgroup walks all segments of its 128 queries; the running merged O sits in a
//          32 KiB LDS slab between segments (thread-private slots: no barrier) -- the batched workloads, where problems x
//          heads x query tiles already fill the chip;
//   split  (attention_dma_kernel<true> + attention_combine_kernel): one workgroup per (query tile, segment) writes its
//          unnormalised (O, m, l) to a workspace, a small second kernel merges them -- single pairs (the per-call plugin API:
//          one pair at N = 2048 is 128 workgroups unsplit on a chip that holds 512).
// Same operations in the same order on the same values: the two schedules are BIT-IDENTICAL (tests/test_matchers_gpu.py),
// so which one runs is a pure launch-geometry decision and batched == single-pair results stay bit-exact.

#include <stdlib.h>

#include <type_traits>

#include "attention_kernels.h"
#include "bf16x3.h"
#include "f16x2.h"
#include "mfma_tiles.h"

#define AT_KT 64       // keys per tile
#define AT_QB 128      // queries per workgroup of 4 waves (the 8-wave form of the fused schedule owns 256)
#define AT_REBASE 8.0f  // rebase the softmax reference when the running maximum moved by more than this (base-2 units)
#ifndef AT_SEG_TILES
#define AT_SEG_TILES 16  // key tiles per segment, at most
#endif

// Developer timeline (tools/trace_attention.hip builds this file with -DGTSFM_TRACE; the product build has none of it): per wave,
// shader-clock cycles summed over all key tiles for each segment of the tile loop, stamps fenced against instruction motion.
#ifdef GTSFM_TRACE
__device__ unsigned long long* g_attn_trace;  // [workgroup][wave][10]: S issue, softmax, B1, PV issue, B2 + DMA issue, merge, total, tiles, start stamp, end stamp
#define TRACE_DECL const unsigned long long t_abs0 = __builtin_amdgcn_s_memtime(); unsigned t_prev = (unsigned)t_abs0; const unsigned t_begin_clk = t_prev; unsigned seg[6] = {0, 0, 0, 0, 0, 0};
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

#define ATD_TILE_FLOATS (AT_KT * 64)   // K tile: [64 keys][64 floats], 16-byte chunks XOR-swizzled by key & 15
#define ATD_VBLOCK_FLOATS 288         // V tile: 16 blocks of 4 key rows (what one DMA instruction writes: 1 KiB) + 128 B of padding each
#define ATD_VTILE_FLOATS (16 * ATD_VBLOCK_FLOATS)
#define ATD_STAGE_FLOATS (ATD_TILE_FLOATS + ATD_VTILE_FLOATS)
#ifndef ATD_DBUF
// 1: K / V tiles double-buffered, ONE barrier per key tile, DMA issued a whole tile ahead (256 VGPRs, 68 KiB of LDS per workgroup).
// Measured round 3 (tools/bench_attention.py, same box, A/B/A): 3.202 / 3.186 / 3.206 ms per launch of 16 sequences at N = 5000 and
// 2.079 / 2.069 / 2.084 ms for 64 at N = 2048 (double / single / double): no difference -- the cycle budget (tools/trace_attention.hip)
// shows the one barrier absorbing exactly the wait of the former two (3.4 k cycles per tile) while the tile period stays at
// 18.1 k cycles against 16.4 k of MFMA work: the waves are not held up by each other's jitter but queue for the matrix pipe, and the
// rest of the gap to the nominal peak is clock (2.16 GHz under the profiler's counters, profiles/r03_effective_clock.csv).
// The single-buffered form (198 VGPRs, 34 KiB) therefore stays the default.
#define ATD_DBUF 0
#endif
#ifndef ATD_FUSED_WAVES
#define ATD_FUSED_WAVES 4  // waves per workgroup of the fused schedule
#endif
#ifndef ATD_INTERLEAVE_SOFTMAX
// 1: the S(t+1) MFMAs and the softmax VALU of tile t as ONE fenced, interleaved instruction stream (8 MFMAs, then a quarter of the row
// maximum / of the exponentials, ...). Built on the hypothesis that the two workgroups of a CU fall into step and idle the matrix pipe
// together during their softmax phases (the ~1.5 k idle cycles per tile of the cycle budget); measured A/B/A on one box: 2.047 / 2.031 /
// 2.047 ms (64 sequences, N = 2048) and 3.146 / 3.130 / 3.145 ms (16, N = 5000), interleaved / plain / interleaved: 0.6 % SLOWER. Off.
#define ATD_INTERLEAVE_SOFTMAX 0
#endif
#ifndef ATD_UNROLL2
// Single-buffered loop unrolled twice (the two score tiles swap roles instead of being copied): 166 -> 207 VGPRs, still two waves per
// SIMD; A/B/A on one box (tools/bench_attention.py): 2.106 / 2.071 / 2.107 ms for 64 sequences at N = 2048, 3.227 / 3.163 / 3.238 ms
// for 16 at N = 5000 (rolled / unrolled / rolled): +1.7 % / +2.0 %. The softmax-phase VALU work is what the ablations price highest.
#define ATD_UNROLL2 1
#endif
#ifndef ATD_PARK_GLOBAL
#define ATD_PARK_GLOBAL ATD_DBUF  // the fused schedule parks its merged state in the caller's workspace instead of LDS
#endif
#define ATD_LDS_TILE_FLOATS ((ATD_DBUF ? 2 : 1) * ATD_STAGE_FLOATS)
#define ATD_OC_FLOATS (34 * 256)  // merged state between segments: 32 accumulator registers + (m, l), x 256 threads
#ifndef ATD_WGS_PER_CU
#define ATD_WGS_PER_CU 2
#endif

// Key tiles per segment of a problem with nq queries and nk keys (round 6): a PURE FUNCTION OF THE PROBLEM'S OWN SIZE -- never of the launch, the
// batch or the device -- so the fused and the split schedule, a pair alone and a pair inside a batch all cut the same keys into the same segments
// and merge them in the same order: bit-identical by construction, as with the fixed 16-tile segments of rounds 3 - 5.
// Why not a constant: the split schedule's work units are (128-query tile, segment) pairs and ONE keypoint-set pair offers 8 (problem, head)
// groups of them. At the 5000-keypoint cap that was 8 x 40 x 5 = 1600 units on the 512 workgroup slots of a 256-CU chip: 3.1 rounds, the fourth
// nearly empty (0.62 of peak for the launch the per-call plugin API lives on). The function picks the even segment length in 8 .. 16 tiles that
// minimises rounds x segment length for that single-pair geometry on 512 slots (ties: the longer segment): 10 tiles at the cap = 8 x 40 x 8 =
// 2560 units = 5.0 rounds. The constants 8 and 512 are part of the DEFINITION (they are not queried from the device): results must not depend
// on the chip the code runs on. Batches run the fused schedule, which only pays one more LDS round trip of its state per extra segment.
#define AT_SEG_MIN_TILES 8
__host__ __device__ inline int at_seg_tiles(int nq, int nk) {
    const int qt = (nq + AT_QB - 1) / AT_QB, nt = (nk + AT_KT - 1) / AT_KT;
    int best = AT_SEG_TILES, best_cost = 0x7fffffff;
    for (int seg = AT_SEG_TILES; seg >= AT_SEG_MIN_TILES; seg -= 2) {
        const int nseg = (nt + seg - 1) / seg;
        const int rounds = (8 * qt * nseg + 511) / 512;
        const int cost = rounds * (nt < seg ? nt : seg);
        if (cost < best_cost) best = seg, best_cost = cost;
    }
    return best;
}

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

// A query's keys live in the two 32-lane halves of the wave: combine a per-lane value with the other half's. v_permlane32_swap (gfx950)
// exchanges the upper half of one register with the lower half of another in the VALU -- `__shfl_xor(v, 32)` is a ds_bpermute, an LDS
// round trip on the softmax's critical path twice per tile. Same values, same (commutative) operation: bit-identical to the shuffle form.
__device__ __forceinline__ float at_halves_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float at_halves_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// THE merge of two softmax partial states over disjoint key sets, (O, m, l) <- (O, m, l) (+) (Os, ms, ls): m are reference
// exponents (base 2), O = sum 2^(s - m) v, l = sum 2^(s - m). Both the fused kernel and the combine kernel of the split
// schedule call exactly this (explicitly rounded products and sum: no contraction can tell them apart).
struct AtMergeWeights {
    float a, b, m;
};
__device__ __forceinline__ AtMergeWeights at_merge_weights(float m, float ms) {
    AtMergeWeights w;
    w.m = fmaxf(m, ms);
    w.a = __builtin_amdgcn_exp2f(m - w.m);   // one of the two is exp2(0) = 1 exactly
    w.b = __builtin_amdgcn_exp2f(ms - w.m);
    return w;
}
__device__ __forceinline__ float at_merge(float acc, float seg, const AtMergeWeights& w) {
    return __fadd_rn(__fmul_rn(acc, w.a), __fmul_rn(seg, w.b));
}

// NWV = waves per workgroup: 4 (128 queries, two workgroups per CU) or 8 (256 queries, one workgroup per CU: every K / V tile that
// travels into LDS then serves twice as many queries). A query's arithmetic does not depend on NWV.
template <bool SPLIT, int NWV = 4>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? ATD_WGS_PER_CU : 1) void attention_dma_kernel(AttnParams p) {
    constexpr int NT = 64 * NWV;  // threads
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // LDS: K tile(s) [64 keys][64 floats] swizzled, then V tile(s) in padded 4-row blocks (see velem); buffer b of K at
    // Ks + b * ATD_TILE_FLOATS, of V at Vs + b * ATD_VTILE_FLOATS
    float* Ks = lds;
    float* Vs = lds + (ATD_DBUF ? 2 : 1) * ATD_TILE_FLOATS;
    // fused schedule only: this thread's merged (O, m, l) of the segments done so far is parked between segments in a 34 KiB LDS
    // slab behind the tiles (thread-private slots: no barrier). The double-buffered build has no LDS to spare for it and parks
    // in a workgroup-private slab of the caller's workspace instead -- which the PMC counters showed as 1.4 GB of extra HBM-side
    // traffic per 32-sequence launch at N = 5000 (5120 workgroups x 34 KiB x 4 segment boundaries, written and read back):
    // harmless for the time of an MFMA-bound kernel, but one more reason the single-buffered form is the default.
    float* Oc = (ATD_PARK_GLOBAL && !SPLIT && p.park) ? p.park + (size_t)blockIdx.x * (34 * NT) : lds + ATD_LDS_TILE_FLOATS;
    // XCD-aware block order (speed only): workgroups are dispatched round-robin over the 8 XCDs, each with a private L2. All
    // query tiles (and segments) of one (problem, head) share the same K / V, so they get linear ids that are congruent
    // mod 8 -> same XCD -> K / V are fetched into ONE L2 instead of eight.
    // Fewer than 8 (problem, head) groups in the launch (one keypoint set of a single pair: 4): a group's workgroups are dealt over
    // xcd_rep = 8 / groups XCDs instead of leaving the others idle (round 5; until then such a launch ran on half the chip).
    const int b = blockIdx.x;
    const int groups = p.heads * p.nproblems;
    const int k_in_xcd = b >> 3;
    const int per_group = SPLIT ? p.qtiles * p.nseg : p.qtiles;
    const int rep = p.xcd_rep > 1 ? p.xcd_rep : 1;
    const int g = rep > 1 ? (b & 7) / rep : (k_in_xcd / per_group) * 8 + (b & 7);
    if (g >= groups) return;
    const int within = rep > 1 ? k_in_xcd * rep + (b & 7) % rep : k_in_xcd % per_group;  // split: segment-major, the query tiles of one segment run side by side
    if (within >= per_group) return;
    const int seg_of_wg = SPLIT ? within / p.qtiles : 0;
    const int h = g % p.heads;
    const AttnProblem pr = p.problems[g / p.heads];
    const int nq = p.counts[pr.q_cnt_idx], nk = p.counts[pr.k_cnt_idx];
    const int q0 = (SPLIT ? within % p.qtiles : within) * (32 * NWV);
    if (q0 >= nq) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int qrow = q0 + wave * 32 + j;
    const bool qvalid = qrow < nq;
    const int ntiles = (nk + AT_KT - 1) / AT_KT;
    const int seg_tiles = at_seg_tiles(nq, nk);  // wave-uniform: scalar arithmetic
    const int t_begin = SPLIT ? seg_of_wg * seg_tiles : 0;
    const int t_end = SPLIT ? (ntiles < t_begin + seg_tiles ? ntiles : t_begin + seg_tiles) : ntiles;
    if (SPLIT && t_begin >= ntiles) return;  // this problem has fewer segments than the launch provides for (or no keys: combine writes zeros)
    if (!SPLIT && !p.park && !p.lds_has_oc && ntiles > seg_tiles) __builtin_trap();  // the host promised one segment and reserved no parking space

    f32x4 qreg[8];  // Q fragment (B operand of S^T = K Q^T): lane (q = j, kh) holds Q[q][8t + 4kh .. +3], pre-scaled by scale * log2(e)
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
    const float* kbase = p.k + (size_t)pr.k_off * p.ldk + h * 64;
    const float* vbase = p.v + (size_t)pr.k_off * p.ldv + h * 64;
    if (!SPLIT && nk <= 0) {  // no keys: the output rows are zero; uniform for the workgroup
        if (qvalid) {
            float* op = p.out + (size_t)(pr.q_off + qrow) * p.ldo + h * 64 + kh * 32;
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(op + 4 * i) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }

    // DMA of one 64 x 64 tile: 16 instructions per workgroup, 4 per wave; instruction i of wave w covers rows
    // 16 i + 4 w .. + 3; lane l writes position l % 16 of row (l / 16) and fetches chunk (l % 16) ^ (row & 15). With this row
    // assignment row & 15 = 4 w + l / 16 is the same for a lane's four instructions: ONE swizzled column offset per lane
    // (round 2 had rows 16 w + 4 i: four 64-bit address registers pairs per tensor, the kernel sits at the register ceiling)
    //
    // K: the 16-byte chunks are XOR-swizzled (a lane reads ITS key row at a compile-time column, ds_read_b128: row stride 256 B
    // would put all lanes on the same banks). V: rows are stored as they come, but every 4-row block (= one DMA instruction)
    // is followed by 128 bytes of padding. The P V product reads V[key][d = lane] with compile-time keys: lanes 0-31 read 32
    // consecutive floats of one row, lanes 32-63 the same columns 4 rows (one block + 128 B = 32 banks) further -- conflict-free,
    // and every address is ONE per-lane base plus an immediate. (Round 2 swizzled V like K: the XOR with key & 15 gave the
    // compiler 64 distinct per-lane addresses, a quarter of the register file, in a kernel at the 256-register ceiling.)
    const int drow = lane >> 4, dpos = lane & 15;
    const int dma_col_k = (dpos ^ ((4 * wave + drow) & 15)) << 2, dma_col_v = dpos << 2;
    auto tile_dma = [&](const float* base, int ld, int k0, float* dst, bool is_v) {
#pragma unroll
        for (int i = 0; i < 16 / NWV; ++i) {  // 16 pieces of 4 rows per tile, 16 / NWV per wave: rows 4 NWV i + 4 wave .. + 3
            const int rb = 4 * NWV * i + 4 * wave;
            int key = k0 + rb + drow;
            key = key < nk ? key : nk - 1;  // clamp: keys beyond nk are masked to -inf (their V rows meet P = 0)
            __builtin_amdgcn_global_load_lds(base + (size_t)key * ld + (is_v ? dma_col_v : dma_col_k), dst + (is_v ? (rb >> 2) * ATD_VBLOCK_FLOATS : rb * 64), 16, 0, 0);
        }
    };
    auto kfrag = [&](int buf, int row, int u) {  // floats 8 u + 4 kh .. + 3 of key row `row` of K buffer `buf` (compile-time buf and u: immediates)
        return *reinterpret_cast<const f32x4*>(Ks + buf * ATD_TILE_FLOATS + row * 64 + (((2 * u + kh) ^ (row & 15)) << 2));
    };
    const float* vlane = Vs + kh * ATD_VBLOCK_FLOATS + j;  // this lane's V column; key4 = key - 4 kh is a compile-time constant at every use
    auto velem = [&](int buf, int key4, int dhalf) { return vlane[buf * ATD_VTILE_FLOATS + (key4 >> 2) * ATD_VBLOCK_FLOATS + (key4 & 3) * 64 + dhalf * 32]; };
    auto s_phase = [&](int buf, f32x16& s0, f32x16& s1, float neg_m) {  // S^T tile = K Q^T - m (accumulators start at -m)
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = s1[r] = neg_m;
#pragma unroll
        for (int u = 0; u < 8; ++u) mfma8(s0, s1, kfrag(buf, j, u), kfrag(buf, 32 + j, u), qreg[u]);
    };

    f32x16 o0, o1;  // O^T of the CURRENT segment: rows d 0..31 / 32..63, column q
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float m = 0.f, l = 0.f;  // current segment: lazy reference maximum (set by the segment's first tile) and running denominator
    f32x16 sa0, sa1, sb0, sb1;  // two score tiles: one being soft-maxed and multiplied with V, one being accumulated (roles alternate per tile)

    // prologue: K(t_begin) -> S; then the next K tile and V(t_begin) in flight
    tile_dma(kbase, p.ldk, t_begin * AT_KT, Ks, false);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    __syncthreads();
    s_phase(0, sa0, sa1, 0.f);
    if (!ATD_DBUF) __syncthreads();  // single buffer: every wave is done reading the first K tile
    if (t_begin + 1 < t_end) tile_dma(kbase, p.ldk, (t_begin + 1) * AT_KT, Ks + (ATD_DBUF ? ATD_TILE_FLOATS : 0), false);
    tile_dma(vbase, p.ldv, t_begin * AT_KT, Vs, true);
    if (!ATD_DBUF) {
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
    }

    // One key tile. PAR (compile time) = parity of the tile inside its segment: with double buffers tile t lives in K / V buffer
    // PAR, and sc / sn swap roles from tile to tile without a copy (the loop below is unrolled twice).
    // Double-buffered schedule, ONE barrier per tile, at the top: behind it K(t+1) and V(t) have landed and are visible, and every
    // wave has finished S(t) [K buffer PAR] and P V(t-1) [V buffer PAR ^ 1] -- exactly the two buffers the DMA of K(t+2) and
    // V(t+1), issued right behind the barrier, overwrites; they have a whole tile of MFMAs to land. (Round 2's attempt at this
    // spilled 13 registers inside the loop and lost 15 %; the padded V layout freed 54.)
    // (A variant with the S MFMAs and the softmax VALU arranged in shared straight-line blocks was measured in round 2: not kept;
    // raised wave priority over the softmax, round 3: the softmax phase shrinks from 2.9 k to 2.0 k cycles per tile and the
    // barrier waits grow by as much, tools/trace_attention.hip.)
    TRACE_DECL
    int ts_run = 0;  // tile index inside its segment (both schedules start at a segment boundary)
    auto tile_step = [&](auto par_c, auto last_c, f32x16& sc0, f32x16& sc1, f32x16& sn0, f32x16& sn1, const int t) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool more = !decltype(last_c)::value;  // compile time: the last tile of the walk has its own body (no S phase to interleave with)
        constexpr int KB = ATD_DBUF ? PAR : 0, KB_NEXT = ATD_DBUF ? PAR ^ 1 : 0, VB = ATD_DBUF ? PAR : 0, VB_NEXT = ATD_DBUF ? PAR ^ 1 : 0;
        const int k0 = t * AT_KT;
        const int ts = ts_run;  // tile index inside its segment
        ts_run = (ts + 1 == seg_tiles) ? 0 : ts + 1;
        const bool next_fresh = !SPLIT && more && ts == seg_tiles - 1;  // fused: the next tile opens a new segment
        if (ATD_DBUF) {
            __builtin_amdgcn_s_waitcnt(0x0f70);  // own pieces of K(t+1) and V(t) landed (issued one tile ago)
            __syncthreads();
            if (t + 2 < t_end) tile_dma(kbase, p.ldk, k0 + 2 * AT_KT, Ks + KB * ATD_TILE_FLOATS, false);
            if (more) tile_dma(vbase, p.ldv, k0 + AT_KT, Vs + VB_NEXT * ATD_VTILE_FLOATS, true);
            TRACE_SEG(2)
        }
        // ---- phase 1: S(t+1) MFMAs in whose shadow the softmax of tile t runs
        // (the reference maximum used for S(t+1)'s accumulator start is the one BEFORE tile t's possible rebase; the
        // difference is applied below when that tile is soft-maxed: its own rebase test sees scores relative to the old m.
        // A tile that opens a segment starts from reference 0, like the very first one.)
        const float m_start = m;
        if (!more && k0 + AT_KT > nk) {  // mask: only the last tile of the keys can be partial (compiled into the last-tile body only)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (key >= nk) sc0[r] = -__builtin_inff();
                if (key + 32 >= nk) sc1[r] = -__builtin_inff();
            }
        }
        // Online softmax with a LAZY reference maximum (per query = per lane; the two lane halves hold different keys of
        // the same query): m follows the running maximum only when that has moved by more than AT_REBASE (base-2 units),
        // so exp2(s - m) <= 2^AT_REBASE stays far from overflow while most tiles skip the rebase (subtract + rescale of O
        // and l) entirely. out = O / l does not depend on the choice of m.
        auto row_max = [&]() {
#if defined(AT_ABLATE) && (AT_ABLATE & 1)  // developer ablation builds (tools/build_variant.sh; results are garbage): bit 1 = no softmax arithmetic
            return sc0[0];
#else
            float mx = fmaxf(sc0[0], sc1[0]);
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(sc0[r], sc1[r]));
            return mx;
#endif
        };
        auto exponentiate = [&]() {
            float sum = 0.f;
#if defined(AT_ABLATE) && (AT_ABLATE & 1)
            sum = sc0[3] + sc1[5];
#else
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc0[r] = __builtin_amdgcn_exp2f(sc0[r]);
                sc1[r] = __builtin_amdgcn_exp2f(sc1[r]);
                sum += sc0[r] + sc1[r];
            }
#endif
            return sum;
        };
        float mloc;
#if ATD_INTERLEAVE_SOFTMAX
        // The S(t+1) MFMAs and the softmax VALU of tile t as ONE instruction stream (two straight-line blocks around the rare
        // rebase branch): a wave that issues its 64 S MFMAs first and its ~90 softmax instructions afterwards offers the matrix
        // pipe nothing for the length of the softmax, and the two workgroups of a CU fall into step (the one behind has the pipe to
        // itself while the other is in its softmax, and catches up), so both idle the pipe together: ~1.5 k cycles per tile in the
        // cycle budget. Interleaved, every gap between two MFMAs carries two or three VALU instructions and no such phase exists.
        if (more) {
            const float neg_m = next_fresh ? 0.f : -m_start;
#pragma unroll
            for (int r = 0; r < 16; ++r) sn0[r] = sn1[r] = neg_m;
            // four k-steps, each: 8 MFMAs + the maximum over 8 of the 32 scores, fenced so that the scheduler keeps the interleave
            mloc = -__builtin_inff();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                mfma8(sn0, sn1, kfrag(KB_NEXT, j, u), kfrag(KB_NEXT, 32 + j, u), qreg[u]);
#if !(defined(AT_ABLATE) && (AT_ABLATE & 1))
#pragma unroll
                for (int r = 4 * u; r < 4 * u + 4; ++r) mloc = fmaxf(mloc, fmaxf(sc0[r], sc1[r]));
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#if defined(AT_ABLATE) && (AT_ABLATE & 1)
            mloc = sc0[0];
#endif
        } else {
            mloc = row_max();
        }
#else
        if (more) s_phase(KB_NEXT, sn0, sn1, next_fresh ? 0.f : -m_start);
        TRACE_SEG(0)
        mloc = row_max();
#endif
        mloc = at_halves_max(mloc);  // finite on every tile: key k0 is always valid
        const bool rebase = (ts == 0) || (mloc > AT_REBASE);
        float d = 0.f;
        if (__any(rebase)) {  // wave-uniform
            d = rebase ? mloc : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc0[r] -= d;
                sc1[r] -= d;
            }
            if (ts > 0) {  // (a segment's first tile: O = l = 0 and m = 0)
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
        float lsum;
#if ATD_INTERLEAVE_SOFTMAX
        if (more) {
            lsum = 0.f;
#pragma unroll
            for (int u = 4; u < 8; ++u) {  // each k-step: 8 MFMAs + 8 of the 32 exponentials and their sum
                mfma8(sn0, sn1, kfrag(KB_NEXT, j, u), kfrag(KB_NEXT, 32 + j, u), qreg[u]);
#if !(defined(AT_ABLATE) && (AT_ABLATE & 1))
#pragma unroll
                for (int r = 4 * (u - 4); r < 4 * (u - 4) + 4; ++r) {
                    sc0[r] = __builtin_amdgcn_exp2f(sc0[r]);
                    sc1[r] = __builtin_amdgcn_exp2f(sc1[r]);
                    lsum += sc0[r] + sc1[r];
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
#if defined(AT_ABLATE) && (AT_ABLATE & 1)
            lsum = sc0[3] + sc1[5];
#endif
        } else {
            lsum = exponentiate();
        }
#else
        lsum = exponentiate();
#endif
        lsum = at_halves_sum(lsum);
        l += lsum;
        // the next tile was accumulated relative to m_start; bring it to the (possibly rebased) reference
        if (more && !next_fresh && __any(d != 0.f)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sn0[r] -= d;
                sn1[r] -= d;
            }
        }
        TRACE_SEG(1)
        if (!ATD_DBUF) {
#if !(defined(AT_ABLATE) && (AT_ABLATE & 2))  // bit 2 = no waits, no barriers in the tile loop
            __builtin_amdgcn_s_waitcnt(0x0f70);  // own V(t) DMA landed (issued one phase ago)
            __syncthreads();                     // B1: K buffer free, V(t) visible
#endif
#if !(defined(AT_ABLATE) && (AT_ABLATE & 4))  // bit 4 = no tile DMA in the loop
            if (t + 2 < t_end) tile_dma(kbase, p.ldk, k0 + 2 * AT_KT, Ks, false);
#endif
            TRACE_SEG(2)
        }
        // ---- phase 2: O^T += V^T P^T. Accumulator register r of S^T tile T holds key 32T + (r&3) + 8(r>>2) + 4kh, so it IS
        // the B operand of k-step r; the A operand V^T[d = lane][key] is a conflict-free read of the padded V tile.
#pragma unroll
        for (int T = 0; T < 2; ++T) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int key4 = 32 * T + 8 * gq;  // this lane's keys: key4 + 4 kh + e
                f32x4 a0, a1, bb;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a0[e] = velem(VB, key4 + e, 0);
                    a1[e] = velem(VB, key4 + e, 1);
                    bb[e] = T ? sc1[4 * gq + e] : sc0[4 * gq + e];
                }
                mfma8(o0, o1, a0, a1, bb);
            }
        }
        TRACE_SEG(3)
        if (!ATD_DBUF && more) {
#if !(defined(AT_ABLATE) && (AT_ABLATE & 2))
            __builtin_amdgcn_s_waitcnt(0x0f70);  // own K(t+2) DMA landed
            __syncthreads();                     // B2: V buffer free, K(t+2) visible
#endif
#if !(defined(AT_ABLATE) && (AT_ABLATE & 4))
            tile_dma(vbase, p.ldv, k0 + AT_KT, Vs, true);
#endif
        }
        TRACE_SEG(4)
        // ---- fused schedule, end of a segment: fold (O, m, l) into the merged state, which waits in Oc (each thread reads and
        // writes only its own 34 slots, so no barrier is involved) while the registers serve the next segment.
        if (!SPLIT && (next_fresh || (!more && t >= seg_tiles))) {
            if (t >= seg_tiles) {  // not the first segment: merged <- merged (+) this segment
                const AtMergeWeights w = at_merge_weights(Oc[32 * NT + tid], m);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o0[r] = at_merge(Oc[r * NT + tid], o0[r], w);
                    o1[r] = at_merge(Oc[(16 + r) * NT + tid], o1[r], w);
                }
                l = at_merge(Oc[33 * NT + tid], l, w);
                m = w.m;
            }
            if (more) {  // park the merged state (O, m, l) and start the next segment from a fresh one
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    Oc[r * NT + tid] = o0[r];
                    Oc[(16 + r) * NT + tid] = o1[r];
                    o0[r] = o1[r] = 0.f;
                }
                Oc[32 * NT + tid] = m;
                Oc[33 * NT + tid] = l;
                m = 0.f, l = 0.f;
            }
            TRACE_SEG(5)
        }
    };
    using Par0 = std::integral_constant<int, 0>;
    using Par1 = std::integral_constant<int, 1>;
    using NotLast = std::false_type;
    using Last = std::true_type;
    // t_begin is even (segment lengths are even): parity of t = parity inside the segment. Two tiles per iteration, the two score
    // tiles swapping roles (no copies); the last tile of the walk runs a body without an S phase.
    int t = t_begin;
#if ATD_DBUF || ATD_UNROLL2
    for (; t + 2 < t_end; t += 2) {
        tile_step(Par0{}, NotLast{}, sa0, sa1, sb0, sb1, t);
        tile_step(Par1{}, NotLast{}, sb0, sb1, sa0, sa1, t + 1);
    }
    if (t + 1 < t_end) {
        tile_step(Par0{}, NotLast{}, sa0, sa1, sb0, sb1, t);
        tile_step(Par1{}, Last{}, sb0, sb1, sa0, sa1, t + 1);
    } else {
        tile_step(Par0{}, Last{}, sa0, sa1, sb0, sb1, t);
    }
#else
    for (; t + 1 < t_end; ++t) {
        tile_step(Par0{}, NotLast{}, sa0, sa1, sb0, sb1, t);
        sa0 = sb0, sa1 = sb1;
    }
    tile_step(Par0{}, Last{}, sa0, sa1, sb0, sb1, t);
#endif
#ifdef GTSFM_TRACE
    if (lane == 0 && g_attn_trace) {
        unsigned long long* o = g_attn_trace + ((size_t)blockIdx.x * 4 + wave) * 10;
        for (int k = 0; k < 6; ++k) o[k] = seg[k];
        const unsigned long long t_abs1 = __builtin_amdgcn_s_memtime();
        o[6] = (unsigned)t_abs1 - t_begin_clk;
        o[7] = t_end - t_begin;
        o[8] = t_abs0, o[9] = t_abs1;
    }
#endif
    if (!qvalid) return;
    if (SPLIT) {  // this segment's unnormalised state; attention_combine_kernel merges the segments
        const size_t row = (size_t)seg_of_wg * p.part_rows + pr.q_off + qrow;
        float* po = p.part_o + row * (p.heads * 64) + h * 64 + kh * 4;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            *reinterpret_cast<f32x4*>(po + 8 * gq) = f32x4{o0[4 * gq], o0[4 * gq + 1], o0[4 * gq + 2], o0[4 * gq + 3]};
            *reinterpret_cast<f32x4*>(po + 32 + 8 * gq) = f32x4{o1[4 * gq], o1[4 * gq + 1], o1[4 * gq + 2], o1[4 * gq + 3]};
        }
        if (kh == 0) *reinterpret_cast<float2*>(p.part_ml + (row * p.heads + h) * 2) = float2{m, l};
        return;
    }
    const float inv = (l > 0.f) ? 1.0f / l : 0.f;
    float* op = p.out + (size_t)(pr.q_off + qrow) * p.ldo + h * 64 + kh * 4;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        *reinterpret_cast<f32x4*>(op + 8 * gq) = f32x4{o0[4 * gq] * inv, o0[4 * gq + 1] * inv, o0[4 * gq + 2] * inv, o0[4 * gq + 3] * inv};
        *reinterpret_cast<f32x4*>(op + 32 + 8 * gq) = f32x4{o1[4 * gq] * inv, o1[4 * gq + 1] * inv, o1[4 * gq + 2] * inv, o1[4 * gq + 3] * inv};
    }
}

// Split schedule, second kernel: out[q][h] = merge of the segments' (O, m, l) in ascending order / l. One thread per
// (query, head, float4 of the 64 channels): 4 queries per workgroup.
__global__ __launch_bounds__(256) void attention_combine_kernel(AttnParams p) {
    const AttnProblem pr = p.problems[blockIdx.y];
    const int nq = p.counts[pr.q_cnt_idx], nk = p.counts[pr.k_cnt_idx];
    const int qrow = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qrow >= nq) return;
    const int lane = threadIdx.x & 63, h = lane >> 4, c = lane & 15;
    if (h >= p.heads) return;
    const int ntiles = (nk + AT_KT - 1) / AT_KT;
    const int seg_tiles = at_seg_tiles(nq, nk);
    const int nseg = (ntiles + seg_tiles - 1) / seg_tiles;
    const size_t row = (size_t)pr.q_off + qrow;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    float m = 0.f, l = 0.f;
    for (int s = 0; s < nseg; ++s) {
        const size_t prow = (size_t)s * p.part_rows + row;
        const f32x4 os = *reinterpret_cast<const f32x4*>(p.part_o + prow * (p.heads * 64) + h * 64 + c * 4);
        const float2 ml = *reinterpret_cast<const float2*>(p.part_ml + (prow * p.heads + h) * 2);
        if (s == 0) {
            o = os, m = ml.x, l = ml.y;
        } else {
            const AtMergeWeights w = at_merge_weights(m, ml.x);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = at_merge(o[e], os[e], w);
            l = at_merge(l, ml.y, w);
            m = w.m;
        }
    }
    const float inv = (l > 0.f) ? 1.0f / l : 0.f;
    *reinterpret_cast<f32x4*>(p.out + row * p.ldo + h * 64 + c * 4) = f32x4{o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
}

// ---------------------------------------------------------------------------------------------------------------------
// Opt-in arithmetic (GTSFM_ATTENTION_MATH=bf16x3, AttnParams::math = 1): the two products of attention, S^T = K Q^T and
// O^T += V^T P^T, on v_mfma_f32_32x32x16_bf16 with every fp32 operand split into THREE bf16 pieces and fp32 accumulation.
//
//   x = hi + mid + lo EXACTLY: hi = x truncated to its top 8 significant bits (a bf16), mid = (x - hi) truncated, lo = the rest
//   (8 + 8 + 8 = the 24 bits of an fp32 significand; bf16 shares fp32's exponent range). A product x y is then the sum of nine
//   bf16 x bf16 products, each EXACT in fp32; six of them are executed -- hi hi, hi mid, mid hi, mid mid, hi lo, lo hi -- and the
//   three dropped ones (mid lo, lo mid: ~2^-24 |x y| each, < 2^-22 |x y| worst case because the split truncates; lo lo: 2^-30) are of the
//   order of one fp32 rounding of the product (tests/test_bf16x3_host.py). The sums
//   accumulate in fp32 inside the MFMA. So every score and every output element carries fp32-class error (the same 2^-24-per-term
//   class as the fmaf chain of v_mfma_f32_32x32x2_f32), but NOT the same bits: this mode is not bit-identical to the exact-fp32
//   kernel above, which stays the default and the one every parity statement is made with.
//   Cost: 6 bf16 MFMAs of 32 cycles (32x32x16) replace 8 fp32 MFMAs of 64 cycles (32x32x2 x 8 k) per 32 x 32 x 16 block: 3/8.
//
// Layout. K and V are split ONCE per launch by attention_x3_split_kernel into tiles of 64 keys that the main kernel moves into
// LDS by DMA exactly as they lie: per (tensor, piece, head, problem, key tile) 64 rows x 64 bf16 = 8 KiB.
//   K tile:   row = key, columns = channels                      (A operand of S^T = K Q^T: lane (key, kh) reads 8 channels = 16 B)
//   V^T tile: row = channel d, columns = key SLOTS               (A operand of O^T += V^T P^T: lane (d, kh) reads 8 slots = 16 B)
// A slot is the position a key has in the B operand P^T that the S^T accumulators form WITHOUT any data movement: accumulator
// register r of key block T holds key 32 T + (r & 3) + 8 (r >> 2) + 4 kh; registers 8 c .. 8 c + 7 of block T, converted and
// packed in pairs, are the B operand of k-step s = 2 T + c, whose lane half kh supplies k = 8 kh .. 8 kh + 7. Hence
//   slot 16 s + 8 kh + i  <->  key 32 (s >> 1) + 16 (s & 1) + 4 kh + (i & 3) + 8 (i >> 2),
// the permutation the split kernel applies when it transposes V. In LDS the 16-byte chunks of a row (128 B) are XOR-swizzled by
// (row >> 1) & 7: the 16 lanes of a ds_read_b128 group read 16 consecutive rows at one logical chunk -> 16 distinct (bank group,
// chunk) positions, conflict-free; the DMA applies the swizzle on the global side (lane-linear LDS image).
// Keys beyond a problem's count are written as zeros by the split kernel (their scores are masked to -inf, P = 0 meets V = 0).
//
// Second opt-in arithmetic (round 6, GTSFM_ATTENTION_MATH=f16x2, AttnParams::math = 2; f16x2.h): the same kernels with NP = 2 fp16 pieces per
// operand and THREE v_mfma_f32_32x32x16_f16 per block (lo hi, hi lo, hi hi) -- half the matrix-pipe work of bf16x3 and a third less LDS
// traffic (two 8 KiB pieces per tile and tensor), at the same per-term error class (<= 2^-22 |x y| + an absolute floor of 2^-25 (|x| + |y|)).
// fp16 has no headroom above 65504, so the softmax weights are kept in range by the kernel itself: in this mode the reference exponent sits
// P_SHIFT = 7 BELOW the running maximum after a rebase (weights <= 2^7 then, and <= 2^15 before the next rebase at AT_REBASE + 7), which also
// lifts the small weights of a row out of fp16's subnormal range: a weight 2^-21 of the row's largest is still a normal fp16. O / l does not
// depend on the reference. K, V and the pre-scaled Q are split as they are; a value beyond +-65504 gives inf -> NaN outputs (never a clamp).
// ---------------------------------------------------------------------------------------------------------------------

#define X3_PIECE_BYTES 8192               // one 16-bit piece of a 64 x 64 tile
#define X3_TILE_BYTES(NP) ((NP) * X3_PIECE_BYTES)  // hi | mid | lo (bf16x3), hi | lo (f16x2)
#define X3_LDS_BYTES(NP) (2 * X3_TILE_BYTES(NP))   // K tile + V^T tile, single-buffered
#define X3_P_SHIFT 7.0f                   // f16x2: the softmax reference sits this far (base-2 units) below the running maximum after a rebase

// Byte offset of tile (tensor ten = 0: K, 1: V^T; piece of np; head h; problem g; key tile t) in the split buffer.
__device__ __forceinline__ size_t x3_tile_offset(const AttnParams& p, int np, int ten, int piece, int h, int g, int t) {
    return ((((size_t)(ten * np + piece) * p.heads + h) * p.nproblems + g) * p.x3_tiles + t) * X3_PIECE_BYTES;
}

// grid (key tiles, problems, heads), 256 threads: one 64-key tile of K and of V of one head -> NP 16-bit pieces each (3: bf16, 2: fp16).
template <int NP>
__global__ __launch_bounds__(256) void attention_x3_split_kernel(AttnParams p) {
    using SM = SplitMath<NP>;
    __shared__ float vt[64 * 65];
    const int t = blockIdx.x, g = blockIdx.y, h = blockIdx.z;
    const AttnProblem pr = p.problems[g];
    const int nk = p.counts[pr.k_cnt_idx];
    if (t * AT_KT >= nk) return;
    const int valid = nk - t * AT_KT;  // rows of this tile that are keys (>= 64: all)
    const int tid = threadIdx.x;
    unsigned char* xb = reinterpret_cast<unsigned char*>(p.x3);
    {   // K: thread = (key row, 16 channels)
        const int r = tid >> 2, c0 = (tid & 3) * 16;
        const float* src = p.k + (size_t)(pr.k_off + t * AT_KT + (r < valid ? r : 0)) * p.ldk + h * 64 + c0;
        u32x4 pc[2][NP];
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            f32x4 v0 = *reinterpret_cast<const f32x4*>(src + 8 * o), v1 = *reinterpret_cast<const f32x4*>(src + 8 * o + 4);
            if (r >= valid) v0 = v1 = f32x4{0.f, 0.f, 0.f, 0.f};
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            SM::split8(v, pc[o]);
        }
#pragma unroll
        for (int piece = 0; piece < NP; ++piece) {
            unsigned char* dst = xb + x3_tile_offset(p, NP, 0, piece, h, g, t) + r * 128 + c0 * 2;
            *reinterpret_cast<u32x4*>(dst) = pc[0][piece];
            *reinterpret_cast<u32x4*>(dst + 16) = pc[1][piece];
        }
    }
    {   // V: through LDS (rows = keys, padded), then thread = (channel d, two slot octets): the transposed, slot-permuted tile
        const int r = tid >> 2, c0 = (tid & 3) * 16;
        const float* src = p.v + (size_t)(pr.k_off + t * AT_KT + (r < valid ? r : 0)) * p.ldv + h * 64 + c0;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * q4);
            if (r >= valid) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) vt[r * 65 + c0 + 4 * q4 + e] = v[e];
        }
        __syncthreads();
        const int d = tid & 63;
#pragma unroll
        for (int oo = 0; oo < 2; ++oo) {
            const int oct = (tid >> 6) * 2 + oo;  // slots 8 oct .. 8 oct + 7 = k-step s = oct >> 1, lane half kh = oct & 1
            const int s = oct >> 1, kh = oct & 1;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = vt[(32 * (s >> 1) + 16 * (s & 1) + 4 * kh + (i & 3) + 8 * (i >> 2)) * 65 + d];
            u32x4 pc[NP];
            SM::split8(v, pc);
#pragma unroll
            for (int piece = 0; piece < NP; ++piece)
                *reinterpret_cast<u32x4*>(xb + x3_tile_offset(p, NP, 1, piece, h, g, t) + d * 128 + oct * 16) = pc[piece];
        }
    }
}

// The attention kernel on the split tiles: same problem / segment / schedule structure as attention_dma_kernel (128 queries of one
// head per workgroup of 4 waves, a wave owns 32 queries, 64-key tiles, key segments of at_seg_tiles(nq, nk) tiles merged by at_merge, fused or split
// schedule), ONE transposed score tile per wave. Per tile and wave: 48 + 48 v_mfma_f32_32x32x16_bf16 (NP = 3) or 24 + 24 v_mfma_f32_32x32x16_f16 (NP = 2). The fused schedule keeps its
// merged state in registers (round 6; rounds 4 - 5 parked it in the caller's workspace: 48 KiB of LDS tiles + a 34 KiB LDS slab would
// leave one workgroup per CU).
// Measured on MI355X, round 4 (tools/bench_attention.py, 32 sequences x 4 heads at N = 5000; exact fp32: 6.17 - 6.22 ms): 3.99 - 4.02 ms
// including the 0.17 ms split pass = 1.55 x (N = 2048, 64 sequences: 2.03 -> 1.39 ms). Four more elaborate loops were built and measured
// within +-3 % of this one (operand reads and the P split software-pipelined under the MFMAs with sched_group_barrier; three workgroups
// per CU at <= 168 VGPRs; two score tiles with the softmax under the next tile's S MFMAs, 242 VGPRs -- that one also differed from run to
// run in the sixth digit when two streams shared the chip, cause not found, and was dropped): the SQ counters show the matrix pipe busy
// 67 - 71 % of the cycles but the clock at 1.6 GHz under the counters (1.8 GHz free-running) where the exact-fp32 kernel holds
// 2.2 - 2.3 GHz. At this duty cycle the bf16 pipe is power-limited: 3 / 8 of the matrix-pipe cycles buy 1.55 x, not 2.67 x.
// NP = 2 (f16x2, round 6): 24 + 24 v_mfma_f32_32x32x16_f16 per tile and wave, 32 KiB of LDS tiles, the softmax reference X3_P_SHIFT below the maximum.
#ifndef X2_WGS_PER_CU
#define X2_WGS_PER_CU 3  // f16x2: 163 VGPRs and 32 KiB of LDS tiles leave room for a third workgroup per CU (three waves per SIMD share the matrix pipe)
#endif
template <bool SPLIT, int NP>
__global__ __launch_bounds__(256, NP == 2 ? X2_WGS_PER_CU : 2) void attention_x3_kernel(AttnParams p) {
    using SM = SplitMath<NP>;
    constexpr float P_SHIFT = NP == 2 ? X3_P_SHIFT : 0.f;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_x3[];
    unsigned char* Kl = lds_x3;
    unsigned char* Vl = lds_x3 + X3_TILE_BYTES(NP);
    const int b = blockIdx.x;
    const int groups = p.heads * p.nproblems;
    const int k_in_xcd = b >> 3;
    const int per_group = SPLIT ? p.qtiles * p.nseg : p.qtiles;
    const int g = (k_in_xcd / per_group) * 8 + (b & 7);
    if (g >= groups) return;
    const int within = k_in_xcd % per_group;
    const int seg_of_wg = SPLIT ? within / p.qtiles : 0;
    const int h = g % p.heads, prob = g / p.heads;
    const AttnProblem pr = p.problems[prob];
    const int nq = p.counts[pr.q_cnt_idx], nk = p.counts[pr.k_cnt_idx];
    const int q0 = (SPLIT ? within % p.qtiles : within) * AT_QB;
    if (q0 >= nq) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int qrow = q0 + wave * 32 + j;
    const bool qvalid = qrow < nq;
    const int ntiles = (nk + AT_KT - 1) / AT_KT;
    const int seg_tiles = at_seg_tiles(nq, nk);
    const int t_begin = SPLIT ? seg_of_wg * seg_tiles : 0;
    const int t_end = SPLIT ? (ntiles < t_begin + seg_tiles ? ntiles : t_begin + seg_tiles) : ntiles;
    if (SPLIT && t_begin >= ntiles) return;

    if (!SPLIT && nk <= 0) {  // no keys: the output rows are zero; uniform for the workgroup
        if (qvalid) {
            float* op = p.out + (size_t)(pr.q_off + qrow) * p.ldo + h * 64 + kh * 32;
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(op + 4 * i) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }

    // Q^T fragment (B operand of S^T = K Q^T): lane (q = j, kh), k-step u: channels 16 u + 8 kh .. + 7, pre-scaled by scale * log2(e)
    u32x4 qf[4][NP];
    {
        const float* qp = p.q + (size_t)(pr.q_off + (qvalid ? qrow : 0)) * p.ldq + h * 64 + kh * 8;
        const float scale2 = p.scale * 1.44269504088896340736f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4 v0 = *reinterpret_cast<const f32x4*>(qp + 16 * u), v1 = *reinterpret_cast<const f32x4*>(qp + 16 * u + 4);
            if (!qvalid) v0 = v1 = f32x4{0.f, 0.f, 0.f, 0.f};
            const float v[8] = {v0[0] * scale2, v0[1] * scale2, v0[2] * scale2, v0[3] * scale2, v1[0] * scale2, v1[1] * scale2, v1[2] * scale2, v1[3] * scale2};
            SM::split8(v, qf[u]);
        }
    }

    // DMA of one tile of one tensor: 8 NP pieces of 1 KiB (NP 16-bit pieces x 8 x [8 rows x 128 B]); wave w moves pieces w, w + 4, ..:
    // their first row is 8 (w + 4 i) -> ((row >> 1) & 7) = ((lane >> 4) + 4 w) & 7 for all of them: one swizzled source offset per lane.
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x3);
    const int dma_src = ((lane >> 3) * 128) + ((((lane & 7) ^ (((lane >> 4) + 4 * wave) & 7))) << 4);
    auto tile_dma = [&](int ten, int t, unsigned char* dst) {
#pragma unroll
        for (int i = 0; i < 2 * NP; ++i) {
            const int pc = wave + 4 * i, piece = pc >> 3, pp = pc & 7;
            const unsigned char* src = xb + x3_tile_offset(p, NP, ten, piece, h, prob, t) + pp * 1024 + dma_src;
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const unsigned*>(src), reinterpret_cast<unsigned*>(dst + piece * X3_PIECE_BYTES + pp * 1024), 16, 0, 0);
        }
    };
    // A-operand reads: lane (row = j [+ 32], kh), k-step u -> logical chunk 2 u + kh of its row, swizzled by (row >> 1) & 7 (equal for j and j + 32)
    int frag_off[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) frag_off[u] = j * 128 + (((2 * u + kh) ^ ((j >> 1) & 7)) << 4);
    auto frag = [&](const unsigned char* tile, int piece, int blk, int u) {
        return *reinterpret_cast<const u32x4*>(tile + piece * X3_PIECE_BYTES + blk * 4096 + frag_off[u]);
    };

    f32x16 o0, o1;  // O^T of the current segment: rows d 0..31 / 32..63, column q
#pragma unroll
    for (int r = 0; r < 16; ++r) o0[r] = o1[r] = 0.f;
    float m = 0.f, l = 0.f;
    // fused schedule: the merged (O, m, l) of the segments done so far stays in REGISTERS between segments (round 6; 34 more VGPRs in a kernel that
    // had 45 to spare). Rounds 4 - 5 parked it in a workgroup-private slab of the workspace -- 34 floats per thread written and read back at every
    // segment boundary: 1.35 GB written + ~1.2 GB re-fetched per 32-sequence launch at N = 5000 with round 6's 8 segments (0.86 GB with five),
    // half of the launch's HBM-side traffic (profiles/r06_pmc_traffic.json, collected before this change) -- because this kernel's 48 KiB of LDS
    // tiles leave no room for the 34 KiB slab the exact-fp32 kernel parks in. Same values, same merge: bit-identical.
    f32x16 pk0, pk1;
    float pkm = 0.f, pkl = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) pk0[r] = pk1[r] = 0.f;

    tile_dma(0, t_begin, Kl);
    tile_dma(1, t_begin, Vl);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
    __syncthreads();

    // One key tile; `last_c` (compile time): the last tile of the walk has its own body -- the only one that can be partial (the mask of keys beyond nk:
    // 64 compare / select instructions per lane) and the only one without a successor to fetch. ts: tile index inside its segment.
    auto tile_step = [&](auto last_c, const int t, const int ts) {
        constexpr bool more = !decltype(last_c)::value;
        const int k0 = t * AT_KT;
        // ---- S^T = K Q^T - m
        f32x16 s0, s1;
        {
            const float neg_m = -m;  // 0 at the start of a segment
#pragma unroll
            for (int r = 0; r < 16; ++r) s0[r] = s1[r] = neg_m;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            u32x4 a0[NP], a1[NP];
#pragma unroll
            for (int piece = 0; piece < NP; ++piece) a0[piece] = frag(Kl, piece, 0, u), a1[piece] = frag(Kl, piece, 1, u);
            SM::product(s0, s1, a0, a1, qf[u]);
        }
        if (!more && k0 + AT_KT > nk) {  // only the last tile of the keys can be partial
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (key >= nk) s0[r] = -__builtin_inff();
                if (key + 32 >= nk) s1[r] = -__builtin_inff();
            }
        }
        // ---- online softmax with a lazy reference maximum (as in attention_dma_kernel)
        float mloc = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, fmaxf(s0[r], s1[r]));
        mloc = at_halves_max(mloc);
        // (f16x2: the reference is kept P_SHIFT below the maximum, so the weights fill fp16's range from the top: <= 2^(AT_REBASE + P_SHIFT) = 2^15)
        const bool rebase = (ts == 0) || (mloc > AT_REBASE + P_SHIFT);
        if (__any(rebase)) {
            const float d = rebase ? mloc - P_SHIFT : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s0[r] -= d, s1[r] -= d;
            if (ts > 0) {
                const float alpha = __builtin_amdgcn_exp2f(-d);
                l *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) o0[r] *= alpha, o1[r] *= alpha;
            }
            m += d;
        }
        float lsum = 0.f;
        if constexpr (NP == 2) {  // two running sums, one per key block (v_pk_add_f32: half the add instructions; an order of its own, like everything in this mode)
            f32x2v sum2 = {0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s0[r] = __builtin_amdgcn_exp2f(s0[r]);
                s1[r] = __builtin_amdgcn_exp2f(s1[r]);
                sum2 += f32x2v{s0[r], s1[r]};
            }
            lsum = sum2.x + sum2.y;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s0[r] = __builtin_amdgcn_exp2f(s0[r]);
                s1[r] = __builtin_amdgcn_exp2f(s1[r]);
                lsum += s0[r] + s1[r];
            }
        }
        l += at_halves_sum(lsum);
        // B1: every wave is done with K(t) and V(t) has landed (own pieces: vmcnt, the others': the barrier) -> the K buffer takes
        // tile t + 1, which has the whole P V phase to land
        // (The sched_barrier fences keep [wait, barrier, DMA issue] a block of its own. With `more` a compile-time constant the DMA issue is
        // straight-line code and the scheduler interleaved it with the softmax tail and the last P V MFMAs across the barrier; that build returned
        // scores that differed from run to run (1e-5 .. 8e-4, 6 - 35 % of the repetitions) WHEN A SECOND STREAM SHARED THE CHIP -- never alone, never
        // with the attention launch on its own on two streams. Any one of the fences in front of the DMA issue removed it (0 of 1200 repetitions, both
        // arithmetics; tools/x3_two_stream_diag.py, profiles/r06_x3_two_stream_bisect.txt); which reordering was the harmful one was not established.
        // tests/test_attention_bf16x3_gpu.py::test_bf16x3_two_stream_pipeline_is_deterministic holds the property.)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        if (more) tile_dma(0, t + 1, Kl);
        __builtin_amdgcn_sched_barrier(0);
        // ---- O^T += V^T P^T: k-step s = 2 T + c takes registers 8 c .. 8 c + 7 of score block T, split into NP pieces on the fly
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 pb[NP];
            {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (s >> 1) ? s1[8 * (s & 1) + i] : s0[8 * (s & 1) + i];
                SM::split8(v, pb);
            }
            u32x4 a0[NP], a1[NP];
#pragma unroll
            for (int piece = 0; piece < NP; ++piece) a0[piece] = frag(Vl, piece, 0, s), a1[piece] = frag(Vl, piece, 1, s);
            SM::product(o0, o1, a0, a1, pb);
        }
        // B2: every wave is done with V(t), K(t + 1) has landed -> the V buffer takes tile t + 1 (it lands under the next S phase + softmax)
        if (more) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            tile_dma(1, t + 1, Vl);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- fused schedule, end of a segment: fold (O, m, l) into the merged state (registers pk0 / pk1 / pkm / pkl)
        const bool seg_end = more && ts == seg_tiles - 1;
        if (!SPLIT && (seg_end || (!more && t >= seg_tiles))) {
            if (t >= seg_tiles) {
                const AtMergeWeights w = at_merge_weights(pkm, m);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o0[r] = at_merge(pk0[r], o0[r], w);
                    o1[r] = at_merge(pk1[r], o1[r], w);
                }
                l = at_merge(pkl, l, w);
                m = w.m;
            }
            if (more) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pk0[r] = o0[r];
                    pk1[r] = o1[r];
                    o0[r] = o1[r] = 0.f;
                }
                pkm = m, pkl = l;
                m = 0.f, l = 0.f;
            }
        }
    };
    {
        int t = t_begin, ts = 0;
        for (; t + 1 < t_end; ++t, ts = (ts + 1 == seg_tiles) ? 0 : ts + 1) tile_step(std::false_type{}, t, ts);
        tile_step(std::true_type{}, t, ts);
    }
    if (!qvalid) return;
    if (SPLIT) {
        const size_t row = (size_t)seg_of_wg * p.part_rows + pr.q_off + qrow;
        float* po = p.part_o + row * (p.heads * 64) + h * 64 + kh * 4;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            *reinterpret_cast<f32x4*>(po + 8 * gq) = f32x4{o0[4 * gq], o0[4 * gq + 1], o0[4 * gq + 2], o0[4 * gq + 3]};
            *reinterpret_cast<f32x4*>(po + 32 + 8 * gq) = f32x4{o1[4 * gq], o1[4 * gq + 1], o1[4 * gq + 2], o1[4 * gq + 3]};
        }
        if (kh == 0) *reinterpret_cast<float2*>(p.part_ml + (row * p.heads + h) * 2) = float2{m, l};
        return;
    }
    const float inv = (l > 0.f) ? 1.0f / l : 0.f;
    float* op = p.out + (size_t)(pr.q_off + qrow) * p.ldo + h * 64 + kh * 4;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        *reinterpret_cast<f32x4*>(op + 8 * gq) = f32x4{o0[4 * gq] * inv, o0[4 * gq + 1] * inv, o0[4 * gq + 2] * inv, o0[4 * gq + 3] * inv};
        *reinterpret_cast<f32x4*>(op + 32 + 8 * gq) = f32x4{o1[4 * gq] * inv, o1[4 * gq + 1] * inv, o1[4 * gq + 2] * inv, o1[4 * gq + 3] * inv};
    }
}

// Segments the launch must provide for: an upper bound over every problem with at most max_k keys (a segment has at least AT_SEG_MIN_TILES tiles;
// which length a problem takes depends on its own query count too, which the host does not know for device-resident counts). Workgroups of
// segments a problem does not have exit at once; 1 = no problem of the launch can have more than one segment.
static int at_segments(int max_k) { return ceil_div(ceil_div(max_k < 1 ? 1 : max_k, AT_KT), AT_SEG_MIN_TILES); }
int attention_segments(int max_k) { return at_segments(max_k); }
// true in the double-buffered build (-DATD_DBUF=1): the fused schedule parks its merged state in ONE slab of the caller's workspace indexed by
// blockIdx, so two fused launches must not share that workspace concurrently (the two-stream form of a pair checks this)
bool attention_parks_in_workspace() { return ATD_PARK_GLOBAL != 0; }

// The split schedule pays one extra round trip of O through the workspace; it is chosen when the unsplit launch would not fill
// the chip's workgroup slots (256 CUs x ATD_WGS_PER_CU) twice and the keys span more than one segment. Measured (MI355X,
// tools/bench_attention.py, fraction of the fp32 MFMA peak, fused -> split): one pair at N = 2048 (128 workgroups) 0.36 -> 0.63,
// one pair at N = 5000 (320) 0.51 -> 0.70, two pairs at N = 5000 (640) 0.67 -> 0.76, two pairs at N = 2048 (256) 0.71 -> 0.72;
// 32 pairs at N = 2048 (4096) 0.84 -> 0.81 and 8 pairs at N = 5000 (2560) 0.82 -> 0.79: batches stay fused. The segment
// structure itself costs the fused schedule 0.2 % (A/B against a build with one unbounded segment).
static bool at_geometry_wants_split(int nproblems, int heads, int max_q, int max_k) {  // launch geometry alone (no environment)
    return at_segments(max_k) >= 2 && (long long)nproblems * heads * ceil_div(max_q, AT_QB) < 2LL * gtsfm_cu_count() * ATD_WGS_PER_CU;
}
static bool at_wants_split(int nproblems, int heads, int max_q, int max_k) {
    const char* env = getenv("GTSFM_ATTENTION_SPLIT");  // "0": never, "1": whenever there is more than one segment (read per launch: tests toggle it)
    if (at_segments(max_k) < 2 || (env && env[0] == '0')) return false;
    if (env && env[0] == '1') return true;
    return at_geometry_wants_split(nproblems, heads, max_q, max_k);
}

static int at_fused_waves() {  // GTSFM_ATTENTION_WAVES = 4 | 8 (read per launch: A/B measurements; results are identical)
    const char* env = getenv("GTSFM_ATTENTION_WAVES");
    return env && env[0] == '8' ? 8 : (env && env[0] == '4' ? 4 : ATD_FUSED_WAVES);
}
// XCDs a (problem, head) group's workgroups are dealt over when the launch has fewer than 8 groups (1, 2 or 4: the divisors of 8); 1 otherwise
static int at_xcd_rep(int groups) { return (groups > 0 && groups < 8 && 8 % groups == 0) ? 8 / groups : 1; }
static int at_fused_grid(int nproblems, int heads, int max_q) { return ceil_div(heads * nproblems, 8) * 8 * ceil_div(max_q, 32 * at_fused_waves()); }

int attention_math_from_env() {  // read per call: GTSFM_ATTENTION_MATH = "bf16x3" / "f16x2" select the split products, anything else exact fp32
    const char* env = getenv("GTSFM_ATTENTION_MATH");
    if (env && env[0] == 'b') return ATTN_MATH_BF16X3;
    if (env && env[0] == 'f' && env[1] == '1') return ATTN_MATH_F16X2;  // (not "f32")
    return ATTN_MATH_F32;
}

static int x3_pieces(int math) { return math == ATTN_MATH_F16X2 ? 2 : 3; }
static size_t x3_split_floats(int nproblems, int heads, int max_k, int math) {  // K and V^T, NP 16-bit pieces each, whole 64-key tiles
    return (size_t)2 * x3_pieces(math) * heads * nproblems * ceil_div(max_k < 1 ? 1 : max_k, AT_KT) * (X3_PIECE_BYTES / sizeof(float));
}

// Sized from the launch GEOMETRY, never from GTSFM_ATTENTION_SPLIT: a workspace is held across calls (the pipeline's per-stream
// workspaces, captured graphs, callers' own) while the switch is read per launch. A launch that wants the split schedule and finds
// the workspace too small (the switch was turned on after sizing) runs the fused schedule instead -- the two are bit-identical.
// The arithmetic mode is the caller's statement (it changes results): bf16x3 adds the split K / V^T tiles.
size_t attention_workspace_floats(int nproblems, int heads, int max_q, int max_k, size_t rows, int math) {
    if (nproblems <= 0) return 0;
    const bool x3 = math == ATTN_MATH_BF16X3 || math == ATTN_MATH_F16X2;
    const size_t base = x3 ? x3_split_floats(nproblems, heads, max_k, math) : 0;
    if (at_segments(max_k) < 2) return base;  // one segment: neither schedule needs memory
    const size_t split = (size_t)at_segments(max_k) * rows * ((size_t)heads * 64 + (size_t)heads * 2);
    const size_t park = x3 ? 0  // bf16x3 / f16x2: the fused schedule keeps its merged state in registers (round 6)
                           : (ATD_PARK_GLOBAL ? (size_t)at_fused_grid(nproblems, heads, max_q) * (ATD_OC_FLOATS / 4 * at_fused_waves()) : 0);  // exact fp32, single buffers: parked in LDS
    return base + (at_geometry_wants_split(nproblems, heads, max_q, max_k) ? (split > park ? split : park) : park);
}

static int launch_attention_x3(const AttnParams& p, int nproblems, int max_q, hipStream_t stream) {
    AttnParams q = p;
    const int max_k = p.max_k > 0 ? p.max_k : max_q;
    GTSFM_CHECK_ARG(p.max_k > 0, "attention (bf16x3 / f16x2): the caller must state the largest key count");
    const bool f16 = p.math == ATTN_MATH_F16X2;
    q.qtiles = ceil_div(max_q, AT_QB);
    q.nproblems = nproblems;
    q.x3_tiles = ceil_div(max_k, AT_KT);
    const size_t x3_floats = x3_split_floats(nproblems, p.heads, max_k, p.math);
    GTSFM_CHECK_ARG(p.workspace && p.workspace_floats >= x3_floats, "attention (bf16x3 / f16x2): workspace too small for the split K / V tiles (%zu < %zu floats)",
                    p.workspace_floats, x3_floats);
    q.x3 = p.workspace;
    float* rest = p.workspace + x3_floats;
    const size_t rest_floats = p.workspace_floats - x3_floats;
    if (f16)
        hipLaunchKernelGGL(attention_x3_split_kernel<2>, dim3(q.x3_tiles, nproblems, p.heads), dim3(256), 0, stream, q);
    else
        hipLaunchKernelGGL(attention_x3_split_kernel<3>, dim3(q.x3_tiles, nproblems, p.heads), dim3(256), 0, stream, q);
    const int groups = p.heads * nproblems;
    const int nseg = at_segments(max_k);
    const size_t need_split = (size_t)nseg * p.part_rows * ((size_t)p.heads * 64 + (size_t)p.heads * 2);
    bool split = p.force_split > 0 || (p.force_split == 0 && at_wants_split(nproblems, p.heads, max_q, max_k));
    if (nseg < 2) split = false;
    if (split && p.force_split <= 0 && (p.part_rows == 0 || rest_floats < need_split)) split = false;
    q.park = nullptr, q.lds_has_oc = 0, q.xcd_rep = 1;
    if (split) {
        q.nseg = nseg;
        GTSFM_CHECK_ARG(p.part_rows > 0 && rest_floats >= need_split, "attention (bf16x3): workspace too small for the split schedule (%zu < %zu floats)", rest_floats, need_split);
        q.part_o = rest;
        q.part_ml = rest + (size_t)nseg * p.part_rows * p.heads * 64;
        dim3 grid(ceil_div(groups, 8) * 8 * q.qtiles * q.nseg);
        if (f16)
            hipLaunchKernelGGL((attention_x3_kernel<true, 2>), grid, dim3(256), X3_LDS_BYTES(2), stream, q);
        else
            hipLaunchKernelGGL((attention_x3_kernel<true, 3>), grid, dim3(256), X3_LDS_BYTES(3), stream, q);
        hipLaunchKernelGGL(attention_combine_kernel, dim3(ceil_div(max_q, 4), nproblems), dim3(256), 0, stream, q);
    } else {
        q.nseg = 1;  // (the merged state between segments lives in registers: no parking space)
        dim3 grid(ceil_div(groups, 8) * 8 * q.qtiles);
        if (f16)
            hipLaunchKernelGGL((attention_x3_kernel<false, 2>), grid, dim3(256), X3_LDS_BYTES(2), stream, q);
        else
            hipLaunchKernelGGL((attention_x3_kernel<false, 3>), grid, dim3(256), X3_LDS_BYTES(3), stream, q);
    }
    GTSFM_CHECK_LAUNCH("attention kernel (bf16x3)");
    return GTSFM_OK;
}

int launch_attention(const AttnParams& p, int nproblems, int max_q, hipStream_t stream) {
    GTSFM_CHECK_ARG(p.ldq % 4 == 0 && p.ldk % 4 == 0 && p.ldv % 4 == 0 && p.ldo % 4 == 0, "attention: leading dimensions must be multiples of 4");
    GTSFM_CHECK_ARG(p.heads > 0 && p.heads <= 4, "attention: 1 to 4 heads");
    GTSFM_CHECK_ARG(p.math == ATTN_MATH_F32 || p.math == ATTN_MATH_BF16X3 || p.math == ATTN_MATH_F16X2, "attention: math is 0 (exact fp32), 1 (bf16x3) or 2 (f16x2)");
    if (nproblems <= 0 || max_q <= 0) return GTSFM_OK;
    if (p.math != ATTN_MATH_F32) return launch_attention_x3(p, nproblems, max_q, stream);
    AttnParams q = p;
    q.qtiles = ceil_div(max_q, AT_QB);
    q.nproblems = nproblems;
    q.park = nullptr, q.lds_has_oc = 0, q.xcd_rep = 1;
    const int groups = p.heads * nproblems;
    const int max_k = p.max_k > 0 ? p.max_k : max_q;
    const size_t tile_bytes = (size_t)ATD_LDS_TILE_FLOATS * sizeof(float);
    bool split = p.workspace != nullptr && (p.force_split > 0 || (p.force_split == 0 && at_wants_split(nproblems, p.heads, max_q, max_k)));
    const size_t need = (size_t)at_segments(max_k) * p.part_rows * ((size_t)p.heads * 64 + (size_t)p.heads * 2);
    if (split && p.force_split <= 0 && (p.part_rows == 0 || p.workspace_floats < need)) split = false;  // not sized for it: the fused schedule gives the same bits
    if (split) {
        q.nseg = at_segments(max_k);
        GTSFM_CHECK_ARG(p.part_rows > 0 && p.workspace_floats >= need, "attention: workspace too small for the split schedule (%zu < %zu floats)", p.workspace_floats, need);
        q.part_o = p.workspace;
        q.part_ml = p.workspace + (size_t)q.nseg * p.part_rows * p.heads * 64;
        q.xcd_rep = at_xcd_rep(groups);
        dim3 grid(q.xcd_rep > 1 ? ceil_div(q.qtiles * q.nseg, q.xcd_rep) * 8 : ceil_div(groups, 8) * 8 * q.qtiles * q.nseg);
        hipLaunchKernelGGL((attention_dma_kernel<true, 4>), grid, dim3(256), tile_bytes, stream, q);
        hipLaunchKernelGGL(attention_combine_kernel, dim3(ceil_div(max_q, 4), nproblems), dim3(256), 0, stream, q);
    } else {
        q.nseg = 1;
        const int nwv = at_fused_waves();
        q.qtiles = ceil_div(max_q, 32 * nwv);
        q.xcd_rep = ATD_PARK_GLOBAL ? 1 : at_xcd_rep(groups);  // (the double-buffered build's parking slabs are sized for the plain block order)
        dim3 grid(q.xcd_rep > 1 ? ceil_div(q.qtiles, q.xcd_rep) * 8 : at_fused_grid(nproblems, p.heads, max_q));
        size_t lds_bytes = tile_bytes;
        if (ATD_PARK_GLOBAL && p.workspace && p.workspace_floats >= (size_t)grid.x * (ATD_OC_FLOATS / 4 * nwv)) {
            q.park = p.workspace;  // (double-buffered build) merged state between segments in the workspace: two workgroups per CU
        } else if (p.max_k <= 0 || at_segments(max_k) > 1) {
            q.lds_has_oc = 1;      // ... in LDS behind the tiles (callers without a workspace)
            lds_bytes += (size_t)ATD_OC_FLOATS / 4 * nwv * sizeof(float);
        }                          // else: the caller vouches for one segment (the kernel traps if a problem has more)
        if (nwv == 8)
            hipLaunchKernelGGL((attention_dma_kernel<false, 8>), grid, dim3(512), lds_bytes, stream, q);
        else
            hipLaunchKernelGGL((attention_dma_kernel<false, 4>), grid, dim3(256), lds_bytes, stream, q);
    }
    GTSFM_CHECK_LAUNCH("attention kernel");
    return GTSFM_OK;
}
