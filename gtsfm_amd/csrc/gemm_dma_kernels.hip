// fp32-MFMA GEMM with both operands staged by LDS-DMA: the matchers' projection / FFN / score GEMMs (K % 32 == 0, row-major
// weights or a second activation matrix). Same contract as gemm_mfma_kernel (dense_kernels.hip), which keeps the other shapes.
// Built with -ffp-contract=off.

#include <stdlib.h>

#include "bf16x3.h"
#include "f16x2.h"
#include "gemm_kernels.h"
#include "gemm_batch.h"
#include "trace.h"

// ---------------------------------------------------------------------------------------------------------------
// GEMM with both operands staged by LDS-DMA (K % 32 == 0 and row-major weights available). See
// docs/experimental/gemm_dma.hip for the derivation of the swizzle and the round-1 measurements; same contract as
// gemm_mfma_kernel.
// LDS image of one operand stage: [128 rows][32 floats]; 16-byte chunk c of row r sits at chunk position
// c ^ ((r >> 1) & 7), which puts the 16 lanes of every ds_read_b128 service group on distinct banks; the DMA writes
// lane-linear, so the swizzle is applied to the per-lane global source address.
// ---------------------------------------------------------------------------------------------------------------
#define DM_KC 32
#define DM_A_FLOATS (128 * DM_KC)
#define DM_W_FLOATS (128 * DM_KC)
#define DM_STAGE_FLOATS (DM_A_FLOATS + DM_W_FLOATS)

__device__ __forceinline__ int dm_swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// ---------------------------------------------------------------------------------------------------------------
// The kernel. History (round 2): the first LDS-DMA kernel computed one tile per workgroup with every wave issuing its 8 pieces
// of a stage as a burst in front of the MFMAs and a row-per-lane epilogue. Its cycle budget at 131072 x 256 -> 768
// (tools/trace_gemm_dma.hip, profiles/r02_gemm_dma_cycle_budget_before.txt): wave lifetime 81 k cycles per tile of which
// the prologue (kernel arguments, bias, first DMA round trip, barrier) takes 17.7 k and the epilogue 7.5 k -- a third of
// a wave's life without a single MFMA, so the SIMD's other wave runs alone (68 % of the matrix pipe) or both idle;
// issuing the 8 LDS-DMA pieces of a stage costs 1.7 k cycles of 64-bit address arithmetic. Here
//   * a workgroup can walk `nb_per_wg` column blocks of its row tile with ONE continuous stage pipeline (the DMA of the next
//     block's first stage is issued before the last MFMAs of the current block: prologue once per workgroup; 73 -> 76 %);
//     since the two steps below the default is nb_per_wg = 1 again -- as fast, and A is read once (see launch_gemm_dma_batched);
//   * the 8 pieces of the next stage are issued BETWEEN the MFMAs of the current stage's first two k-steps (77 %);
//   * the epilogue is transposed through LDS so that stores and residual / rotary loads cover full 128-byte lines (79-80 %);
//   * DMA sources are a uniform base (SGPR pair, advanced per stage) plus 32-bit per-lane offsets computed once;
//   * the bias joins in the epilogue (its loads fly during the block), the residual rows / rotary cos-sin pairs of a block
//     are requested before its last stage's MFMAs and are in registers when the epilogue starts;
//   * ROT: LightGlue's rotary embedding (apply_cached_rotary_emb) on the q and k column blocks in the epilogue,
//     enc = [token][f][cos, sin];
//   * ragged batches: per-tile live counts (LightGlue's 128-row-aligned sequences) or a problem table (GemmBatch);
// Measured and dropped: a fifth loader wave per workgroup issuing all 32 pieces of a stage (62 %: one stage of slack with two
// buffers); all 8 pieces inside the first k-step (+-0); bias preloaded into the accumulators / fragments one k-step ahead (+-0);
// LayerNorm + GELU of the workgroup's own rows after its last column block (rounds 2-4, GTSFM_FUSED_LN: the workgroup has to walk all four
// column blocks of its row tile -- batched workload 495.8-499.9 against 500.1-501.3 image-pairs/s with the separate kernel, one pair at the cap
// 13.5 against 11.5 ms (80 workgroups on 256 CUs): slower in both, removed in round 4).
// ---------------------------------------------------------------------------------------------------------------
//   * X3 (round 4, opt-in GTSFM_GEMM_MATH=bf16x3): the same stages, DMA, epilogues -- only the products change: a stage's 32 k are two bf16
//     k-steps of 16; each lane splits the eight fp32 values of its weight row and of its activation row EXACTLY into three bf16 pieces in
//     registers (bf16x3.h) and every 32 x 32 x 16 block is six v_mfma_f32_32x32x16_bf16 with fp32 accumulation: fp32-class error per
//     term, NOT the bits of the default. 24 + 24 MFMAs of 32 cycles per stage and wave instead of 64 of 64 cycles.
//   * MATH = 2 (round 6, opt-in GTSFM_GEMM_MATH=f16x2; f16x2.h): the same with TWO fp16 pieces per operand and three v_mfma_f32_32x32x16_f16 per
//     block: 12 + 12 MFMAs per stage and wave. Operands beyond +-65504 give inf / NaN outputs (fp16 has no exponent headroom). Cycle budget at
//     163840 x 512 -> 512 (tools/trace_gemm_dma.hip, math 2): 2850 cycles per stage and wave for 768 cycles of own MFMA time (DMA issue 330, LDS reads +
//     split + MFMAs 2015, wait + barrier 510): the stage is too short to hide its own fixed costs. Tried: k-step 1's fragments read and split in source order
//     under k-step 0's MFMAs -- the scheduler regrouped it, 0.343 -> 0.364 ms: removed.
// MATH: 0 = exact fp32, 1 = bf16x3, 2 = f16x2.
template <bool HAS_RES, bool ROT, int MATH = 0>
__global__ __launch_bounds__(256, 2) void gemm_dma_walk_kernel(GemmParams p, GemmBatch bt) {
    constexpr bool X3 = MATH != 0;       // one of the split arithmetics
    constexpr int NP = MATH == 2 ? 2 : 3;  // 16-bit pieces per operand
    using SM = SplitMath<NP>;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [2 stages][A 4096 | W 4096]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int M = p.m_dev ? *p.m_dev : p.M;
    int N = p.n_dev ? *p.n_dev : p.N;
    if (bt.problems) {  // ragged batch of independent products (blockIdx.y = problem); p.M / p.N are the batch maxima
        const GemmProblem pr = bt.problems[blockIdx.y];
        p.A += (size_t)pr.a_row * p.lda, p.wraw += (size_t)pr.w_row * p.ldw, p.C += pr.c_off, p.ldc = pr.ldc;
        M = min(bt.counts[pr.m_idx], p.M), N = min(bt.counts[pr.n_idx], p.N);
    }
    // XCD-aware order (speed only): workgroup b runs on XCD b % 8; the column groups of one row tile get consecutive slots of
    // ONE XCD, so the A tile is fetched into one L2 and re-read there
    const int ncb_total = (p.N + 127) / 128, mtiles = (p.M + 127) / 128;
    const int groups = (ncb_total + p.nb_per_wg - 1) / p.nb_per_wg;
    const int b = blockIdx.x, kx = b >> 3;
    int mt = (kx / groups) * 8 + (b & 7), cb0 = (kx % groups) * p.nb_per_wg;
    if (p.super_rows > 0) {
        // Wide products (the score matrix of a pair at the cap: 40 x 40 tiles, W = image 1's 5000 x 256 descriptors = 5 MB against 4 MiB of L2): with
        // a whole row of column blocks side by side, every row tile streamed ALL of W through its XCD's L2 -- 286 MB moved for 110 MB algorithmic
        // (profiles/r05_pmc_traffic.json). Here the ~64 workgroups an XCD runs at a time form a super-tile of super_rows row tiles x 8 column
        // groups: 8 + 8 operand panels of 128 KB serve 64 tiles (41 panels in the row order). Speed only: a tile's arithmetic does not change.
        const int per = p.super_rows * 8, nsc = (groups + 7) >> 3;
        const int sb = kx / per, w = kx - sb * per;
        const int cg = (sb % nsc) * 8 + (w & 7);
        if (cg >= groups) return;
        mt = ((sb / nsc) * p.super_rows + (w >> 3)) * 8 + (b & 7), cb0 = cg * p.nb_per_wg;
    }
    if (mt >= mtiles) return;
    const int m0 = mt * 128;
    if (p.tile_cnt_idx) {  // ragged batch with 128-row-aligned sequences
        const int c = p.live_counts[p.tile_cnt_idx[mt]];
        const int r0 = p.tile_row0[mt];
        if (r0 >= c) return;
        M = min(M, m0 + c - r0);
    }
    if (m0 >= M) return;
    const int nblk = min(p.nb_per_wg, (N + 127) / 128 - cb0);  // column blocks with at least one live column
    if (nblk <= 0) return;
    const int j = lane & 31, kh = lane >> 5;
    const int nstages = p.K / DM_KC;
    const int total = nblk * nstages;

    // DMA: a wave moves 4 pieces x 8 rows of A and of W per stage (LDS rows 32 wave + 8 i + lane / 8); rows beyond M / N
    // are clamped (computed, never stored)
    const int drow = lane >> 3, dpos = lane & 7;
    const char* baseA = reinterpret_cast<const char*>(p.A + (size_t)m0 * p.lda);
    const char* baseW = reinterpret_cast<const char*>(p.wraw);
    unsigned offA[4], offW[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 32 * wave + 8 * i + drow;
        offA[i] = (unsigned)(min(r, M - 1 - m0) * p.lda + dm_swz(r, dpos) * 4) * 4u;
    }
    auto set_w_offsets = [&](int cbi) {
        const int n0 = (cb0 + cbi) * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 32 * wave + 8 * i + drow;
            offW[i] = (unsigned)(min(n0 + r, N - 1) * p.ldw + dm_swz(r, dpos) * 4) * 4u;
        }
    };
    // The 8 pieces of the NEXT stage are issued between the MFMAs of the current stage's first two k-steps (one piece per 4
    // MFMAs): among MFMAs a piece costs ~50 cycles of the wave's issue time, hidden under the matrix pipe; issued as a burst
    // in front of the MFMAs (first version of this kernel) the 8 pieces took 3.1 k cycles per stage during which the wave fed
    // the pipe nothing (profiles/r02_gemm_dma_walk_cycle_budget.txt). The last two k-steps cover the DMA latency.
    int dcb = 0, dst = 0;  // the next stage to fetch: column block, stage within it
    auto piece = [&](int i, const char* a, const char* w, float* sA, float* sW) {
        if (i < 4)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(a + offA[i]), sA + (32 * wave + 8 * i) * DM_KC, 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(w + offW[i - 4]), sW + (32 * wave + 8 * (i - 4)) * DM_KC, 16, 0, 0);
    };
    auto frag = [&](const float* base, int row, int step) {  // 16-byte fragment: floats 8 step + 4 kh .. + 3 of `row`
        return *reinterpret_cast<const f32x4*>(base + row * DM_KC + dm_swz(row, 2 * step + kh) * 4);
    };

    f32x16 c00, c01, c10, c11;  // (row half, column half) of the wave's 64 x 64 tile; lane = row, registers = columns
    f32x4 bia[2];               // bias of the lane's columns in the epilogue's transposed mapping: 4 columns in either column half of the wave's tile
    f32x4 aux[16];              // residual values (HAS_RES) / rotary (cos, sin) pairs (ROT) of the block being finished
    const bool vec_ok = ((N & 3) == 0) && ((p.ldc & 3) == 0) && ((p.c_coff & 3) == 0) && (!HAS_RES || (p.ldres & 3) == 0);
    const int nbias = (p.N + 63) / 64 * 64;
    const int row_lo = m0 + 64 * wm + j;  // the lane's rows in the MFMA layout: row_lo and row_lo + 32
    const int tr = lane >> 3, tc = lane & 7;  // transposed epilogue: lane -> (row tr + 8 i, 16-byte chunk tc) of a 32 x 32 tile

    GT_DECL
    set_w_offsets(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) piece(i, baseA, baseW, lds, lds + DM_A_FLOATS);
    dst = 1;
    if (dst == nstages) dst = 0, dcb = 1;
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's DMA has landed ...
    __syncthreads();                     // ... and so has everybody else's
    GT_SEG(0)
    int cbi = 0, st = 0;
#pragma unroll 1
    for (int it = 0; it < total; ++it) {
        const float* sA = lds + (it & 1) * DM_STAGE_FLOATS;
        const float* sW = sA + DM_A_FLOATS;
        const int n0 = (cb0 + cbi) * 128;
        const int colb = n0 + 64 * wn + 4 * kh;
        const bool last = st == nstages - 1;
        const bool rot = ROT && n0 < p.rot_cols;
        __builtin_amdgcn_s_setprio(3);
        if (st == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c00[r] = c01[r] = c10[r] = c11[r] = 0.f;
        }
        // source of the next stage (the stage after the last one re-fetches it: nobody reads that buffer)
        if (dcb >= nblk) dcb = nblk - 1, dst = nstages - 1;
        if (dst == 0) set_w_offsets(dcb);
        float* nA = lds + ((it + 1) & 1) * DM_STAGE_FLOATS;  // the other buffer was last read one stage ago
        float* nW = nA + DM_A_FLOATS;
        const char* na = baseA + (size_t)dst * (DM_KC * 4);
        const char* nw = baseW + (size_t)dst * (DM_KC * 4);
        if (++dst == nstages) dst = 0, ++dcb;
        if (last && vec_ok && !X3) {
            // the block's bias in the epilogue's transposed lane mapping (a lane finishes 4 consecutive columns of a row there: 8 registers; in
            // the MFMA layout, where a lane owns 32 + 32 columns of its row, the bias took 32 registers for the whole block and the residual /
            // rotary variants spilled: round 5, 256 VGPRs + 20 / 28 bytes of scratch). Bias, alpha, ReLU, rotary, residual are elementwise
            // and keep their order per element, so applying them after the transpose gives the same bits.
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int col = n0 + 64 * wn + 32 * h + 4 * tc;
                bia[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.bias && col < nbias) bia[h] = *reinterpret_cast<const f32x4*>(p.bias + col);
            }
        }
        if (last && vec_ok && !X3) {  // (X3: the operand pieces need the registers; the epilogue loads them where it uses them)
            // residual values / rotary (cos, sin) pairs of this block, in the epilogue's transposed lane mapping (see below):
            // requested now, in registers when the last MFMAs are done
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = m0 + 64 * wm + 32 * (t >> 1) + tr + 8 * i;
                    const int col = n0 + 64 * wn + 32 * (t & 1) + 4 * tc;
                    if (HAS_RES) {
                        aux[4 * t + i] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (row < M && col < N) aux[4 * t + i] = *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.ldres + col);
                    }
                    if (rot) {  // pairs f = (col % 64) / 2 and f + 1 of the row's [f][cos, sin] table
                        aux[4 * t + i] = f32x4{1.f, 0.f, 1.f, 0.f};
                        if (row < M) aux[4 * t + i] = *reinterpret_cast<const f32x4*>(p.rot_enc + (size_t)row * 64 + 32 * (t & 1) + 4 * tc);
                    }
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
        GT_SEG(1)
        const int ra = 64 * wm + j, rw = 64 * wn + j;
        // weights are the MFMA's A operand, activations its B operand (a lane then owns one output row)
#define GS(e)                                                             \
    c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.e, a0.e, c00, 0, 0, 0); \
    c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1.e, a0.e, c01, 0, 0, 0); \
    c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0.e, a1.e, c10, 0, 0, 0); \
    c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1.e, a1.e, c11, 0, 0, 0);
        if constexpr (X3) {
            // k-step h of the stage: lane (row, kh) owns floats 16 h + 8 kh .. + 7 of its row = the 16-byte chunks 4 h + 2 kh and 4 h + 2 kh + 1
            auto frag8 = [&](const float* base, int row, int h, u32x4 (&dst)[NP]) {
                const f32x4 lo4 = *reinterpret_cast<const f32x4*>(base + row * DM_KC + dm_swz(row, 4 * h + 2 * kh) * 4);
                const f32x4 hi4 = *reinterpret_cast<const f32x4*>(base + row * DM_KC + dm_swz(row, 4 * h + 2 * kh + 1) * 4);
                const float v[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                SM::split8(v, dst);
            };
#pragma unroll
            for (int h = 0; h < DM_KC / 16; ++h) {
                u32x4 a0[NP], a1[NP], b0[NP], b1[NP];
                frag8(sA, ra, h, a0), frag8(sA, ra + 32, h, a1);
#ifdef GTSFM_X3_ABLATE_WSPLIT  // developer ablation (tools/build_variant.sh; results are garbage): what weights that arrive ALREADY split would save -- the
                {              // weight fragments are read (same LDS traffic as three bf16 planes would cost: 2 x 16 B here, 3 x 16 B then) but not split
                    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(sW + rw * DM_KC + dm_swz(rw, 4 * h + 2 * kh) * 4);
                    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(sW + rw * DM_KC + dm_swz(rw, 4 * h + 2 * kh + 1) * 4);
                    const f32x4 lo5 = *reinterpret_cast<const f32x4*>(sW + (rw + 32) * DM_KC + dm_swz(rw + 32, 4 * h + 2 * kh) * 4);
                    const f32x4 hi5 = *reinterpret_cast<const f32x4*>(sW + (rw + 32) * DM_KC + dm_swz(rw + 32, 4 * h + 2 * kh + 1) * 4);
                    b0[0] = __builtin_bit_cast(u32x4, lo4), b0[1] = __builtin_bit_cast(u32x4, hi4), b0[NP - 1] = b0[0];
                    b1[0] = __builtin_bit_cast(u32x4, lo5), b1[1] = __builtin_bit_cast(u32x4, hi5), b1[NP - 1] = b1[0];
                }
#else
                frag8(sW, rw, h, b0), frag8(sW, rw + 32, h, b1);
#endif
                SM::product(c00, c01, b0, b1, a0);  // weights = MFMA A operand (output columns), activations = B operand: a lane owns an output row
#pragma unroll
                for (int i = 0; i < 4; ++i) piece(4 * h + i, na, nw, nA, nW);
                SM::product(c10, c11, b0, b1, a1);
            }
        } else {
#pragma unroll
        for (int s = 0; s < DM_KC / 8; ++s) {
            const f32x4 a0 = frag(sA, ra, s), a1 = frag(sA, ra + 32, s);
            const f32x4 b0 = frag(sW, rw, s), b1 = frag(sW, rw + 32, s);
            if (s < 2) {
                GS(x) piece(4 * s + 0, na, nw, nA, nW);
                GS(y) piece(4 * s + 1, na, nw, nA, nW);
                GS(z) piece(4 * s + 2, na, nw, nA, nW);
                GS(w) piece(4 * s + 3, na, nw, nA, nW);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // the step's 4 fragment reads
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // 4 MFMAs
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);  // one LDS-DMA piece
                }
            } else {
                GS(x) GS(y) GS(z) GS(w)
            }
        }
        }
#undef GS
        GT_SEG(2)
        __builtin_amdgcn_s_setprio(3);
        __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): next stage's DMA, bias / residual / rotary loads; older stores are long done
        __syncthreads();  // (also after the last stage: the epilogue below re-uses the consumed buffer as scratch)
        GT_SEG(3)
#ifdef GTSFM_GEMM_ABLATE_EPILOGUE  // developer ablation (tools/build_variant.sh): what a fully hidden epilogue would buy -- results are garbage
        if (last && p.alpha == 12345.f) {
#else
        if (last) {
#endif
            // epilogue of column block cbi; its stores drain under the next block's MFMAs
            if (vec_ok) {
                // Transposed through LDS, one 32 x 32 accumulator tile at a time: in the MFMA layout a lane owns a ROW, so a
                // 16-byte store instruction touches 32 rows x 32 B (64 cache lines per wave-instruction; the epilogue was
                // store-issue-bound: 6.0 k cycles per tile). After the transpose 8 lanes cover 128 contiguous bytes of a row:
                // 8 full lines per instruction, and the residual / rotary operands arrive the same way. Scratch = this wave's
                // OWN 4 KiB slice of the A stage all waves have just finished reading (only this wave's later DMA writes it).
                float* scr = const_cast<float*>(sA) + 32 * wave * DM_KC;
                // the lane's epilogue coordinates from a lane id read HERE (mbcnt behind an asm the compiler cannot hoist): derived from threadIdx
                // at the top of the kernel they are loop invariants that occupy registers through every MFMA stage, and the rotary variant of the
                // bf16x3 build spilled four of them
                int elane;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(elane));
                const int tr = elane >> 3, tc = elane & 7, j = elane & 31, kh = elane >> 5;
                if (X3) {  // (the operand pieces needed the registers during the last stage)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int col = n0 + 64 * wn + 32 * h + 4 * tc;
                        bia[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (p.bias && col < nbias) bia[h] = *reinterpret_cast<const f32x4*>(p.bias + col);
                    }
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f32x16& ct = (t == 0) ? c00 : (t == 1) ? c01 : (t == 2) ? c10 : c11;
                    if (X3) __builtin_amdgcn_sched_barrier(0);  // one tile's residual / rotary loads at a time: hoisted together they cost 64 registers
#pragma unroll
                    for (int q = 0; q < 4; ++q)  // row j, 16-byte chunk 2 q + kh, XOR-swizzled by the row (conflict-free writes)
                        *reinterpret_cast<f32x4*>(scr + j * 32 + (((2 * q + kh) ^ (j & 7)) << 2)) = f32x4{ct[4 * q], ct[4 * q + 1], ct[4 * q + 2], ct[4 * q + 3]};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int rr = tr + 8 * i;
                        f32x4 v = *reinterpret_cast<const f32x4*>(scr + rr * 32 + ((tc ^ (rr & 7)) << 2));
                        const int row = m0 + 64 * wm + 32 * (t >> 1) + rr;
                        const int col = n0 + 64 * wn + 32 * (t & 1) + 4 * tc;
                        if (X3) {
                            if (HAS_RES) {
                                aux[4 * t + i] = f32x4{0.f, 0.f, 0.f, 0.f};
                                if (row < M && col < N) aux[4 * t + i] = *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.ldres + col);
                            }
                            if (rot) {
                                aux[4 * t + i] = f32x4{1.f, 0.f, 1.f, 0.f};
                                if (row < M) aux[4 * t + i] = *reinterpret_cast<const f32x4*>(p.rot_enc + (size_t)row * 64 + 32 * (t & 1) + 4 * tc);
                            }
                        }
                        v += bia[t & 1];
                        if (p.alpha != 1.0f) v *= p.alpha;
                        if (p.relu) v = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
                        if (rot) {  // (x0, x1) -> (x0 c - x1 s, x1 c + x0 s) per feature pair, as apply_cached_rotary_emb
                            const f32x4 e = aux[4 * t + i];
                            v = f32x4{(v.x * e.x) + ((-v.y) * e.y), (v.y * e.x) + (v.x * e.y), (v.z * e.z) + ((-v.w) * e.w), (v.w * e.z) + (v.z * e.w)};
                        }
                        if (HAS_RES) v = aux[4 * t + i] + v;
                        if (row < M && col < N) *reinterpret_cast<f32x4*>(p.C + (size_t)row * p.ldc + p.c_coff + col) = v;
                    }
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row = row_lo + 32 * (t >> 1);
                    const int col0 = colb + 32 * (t & 1);
                    const f32x16& ct = (t == 0) ? c00 : (t == 1) ? c01 : (t == 2) ? c10 : c11;
                    if (row < M) {
                        float* crow = p.C + (size_t)row * p.ldc + p.c_coff;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int col = col0 + 8 * (r >> 2) + (r & 3);
                            if (col >= N) continue;
                            float v = ct[r] + (p.bias ? p.bias[col] : 0.f);  // same order per element as the vector path: bias, alpha, ReLU, residual
                            if (p.alpha != 1.0f) v *= p.alpha;
                            if (p.relu) v = fmaxf(v, 0.f);
                            crow[col] = HAS_RES ? p.res[(size_t)row * p.ldres + col] + v : v;
                        }
                    }
                }
            }
        }
        GT_SEG(4)
        if (++st == nstages) st = 0, ++cbi;
    }
#ifdef GTSFM_TRACE
    if (lane == 0 && g_gemm_trace) {
        unsigned long long* o = g_gemm_trace + ((size_t)blockIdx.x * 4 + wave) * 8;
        for (int k = 0; k < 5; ++k) o[k] = gseg[k];
        o[5] = (unsigned)__builtin_amdgcn_s_memtime() - gt_begin;
        o[6] = nblk;
        o[7] = nstages;
    }
#endif
}


// ---------------------------------------------------------------------------------------------------------------
// Small launches (round 3): the same product on 64 x 64 workgroup tiles. One keypoint-set PAIR (the per-call plugin API) has
// M = 2 x 2048 .. 2 x 5000 token rows: with 128 x 128 tiles a projection is 64 .. 480 workgroups on a chip that holds 512, every
// one of them a full-length K loop at one workgroup per CU (profiles/r03_plugin_single_pair_kernel_stats_*.csv: 38 us per
// 512 -> 256 launch at M = 4096 AND at M = 10240). Four times as many workgroups of a quarter of the work each put every CU to
// work and shorten the critical path. Same arithmetic, bit for bit: an output element is the same k-ordered chain of
// v_mfma_f32_32x32x2_f32 steps from a zero accumulator (fragments, stage depth and step order as in gemm_dma_walk_kernel), then
// bias, alpha, ReLU, rotary, residual in the same order and form -- which tile shape computes it is a launch-geometry decision.
// Workgroup = 4 waves as 2 (M) x 2 (N), one 32 x 32 accumulator each; LDS image of a stage as above with 64 rows per operand, two
// stages; a wave moves two 8-row pieces of A and of W per stage; a lane owns an output row and stores 4 x 16 bytes.
// ---------------------------------------------------------------------------------------------------------------
#define DS_A_FLOATS (64 * DM_KC)
#define DS_STAGE_FLOATS (2 * DS_A_FLOATS)

// X3 (round 4): the bf16x3 arithmetic of gemm_dma_walk_kernel<..., X3> with the same six products in the same order per block and k-step, so that the
// two tilings stay bit-identical to each other under GTSFM_GEMM_MATH=bf16x3 as well (batched == single-pair results). MATH = 2: f16x2, likewise.
template <bool HAS_RES, bool ROT, int MATH = 0>
__global__ __launch_bounds__(256, 4) void gemm_dma_small_kernel(GemmParams p) {
    constexpr bool X3 = MATH != 0;
    constexpr int NP = MATH == 2 ? 2 : 3;
    using SM = SplitMath<NP>;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [2 stages][A 2048 | W 2048]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int M = p.m_dev ? *p.m_dev : p.M;
    const int N = p.n_dev ? *p.n_dev : p.N;
    // block order: the column blocks of one row tile are neighbours (A rows shared through the L2)
    const int ncb = (p.N + 63) / 64;
    const int mt = blockIdx.x / ncb, cb = blockIdx.x % ncb;
    const int m0 = mt * 64, n0 = cb * 64;
    if (p.tile_cnt_idx) {  // ragged batch with 128-row-aligned sequences: this 64-row tile is half of a 128-row tile of one sequence
        const int c = p.live_counts[p.tile_cnt_idx[mt >> 1]];
        const int r0 = p.tile_row0[mt >> 1] + 64 * (mt & 1);
        if (r0 >= c) return;
        M = min(M, m0 + c - r0);
    }
    if (m0 >= M || n0 >= N) return;
    const int j = lane & 31, kh = lane >> 5;
    const int nstages = p.K / DM_KC;
    const int drow = lane >> 3, dpos = lane & 7;
    const char* baseA = reinterpret_cast<const char*>(p.A + (size_t)m0 * p.lda);
    const char* baseW = reinterpret_cast<const char*>(p.wraw);
    unsigned offA[2], offW[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = 16 * wave + 8 * i + drow;
        offA[i] = (unsigned)(min(r, M - 1 - m0) * p.lda + dm_swz(r, dpos) * 4) * 4u;  // rows beyond M / N are clamped (computed, never stored)
        offW[i] = (unsigned)(min(n0 + r, N - 1) * p.ldw + dm_swz(r, dpos) * 4) * 4u;
    }
    auto stage_dma = [&](int st, float* sA) {
        const char* a = baseA + (size_t)st * (DM_KC * 4);
        const char* w = baseW + (size_t)st * (DM_KC * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(a + offA[i]), sA + (16 * wave + 8 * i) * DM_KC, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(w + offW[i]), sA + DS_A_FLOATS + (16 * wave + 8 * i) * DM_KC, 16, 0, 0);
        }
    };
    auto frag = [&](const float* base, int row, int step) {  // 16-byte fragment: floats 8 step + 4 kh .. + 3 of `row`
        return *reinterpret_cast<const f32x4*>(base + row * DM_KC + dm_swz(row, 2 * step + kh) * 4);
    };
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    const int row = m0 + 32 * wm + j;             // the lane's output row
    const int colb = n0 + 32 * wn + 4 * kh;       // its columns: colb + 8 q + 0..3, q = 0..3
    const int nbias = (p.N + 63) / 64 * 64;
    f32x4 bia[4], aux[4];
    const bool rot = ROT && n0 < p.rot_cols;     // rot_cols is a multiple of 128: the same columns as in the 128-column blocks
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int cc = colb + 8 * q;
        bia[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (p.bias && cc < nbias) bia[q] = *reinterpret_cast<const f32x4*>(p.bias + cc);
        if (HAS_RES) {
            aux[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < M && cc < N) aux[q] = *reinterpret_cast<const f32x4*>(p.res + (size_t)row * p.ldres + cc);
        }
        if (ROT) {  // pairs f = (col % 64) / 2 and f + 1 of the row's [f][cos, sin] table
            aux[q] = f32x4{1.f, 0.f, 1.f, 0.f};
            if (rot && row < M) aux[q] = *reinterpret_cast<const f32x4*>(p.rot_enc + (size_t)row * 64 + (cc & 63));
        }
    }
    stage_dma(0, lds);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    const int ra = 32 * wm + j, rw = 32 * wn + j;
#pragma unroll 1
    for (int st = 0; st < nstages; ++st) {
        const float* sA = lds + (st & 1) * DS_STAGE_FLOATS;
        const float* sW = sA + DS_A_FLOATS;
        if (st + 1 < nstages) stage_dma(st + 1, lds + ((st + 1) & 1) * DS_STAGE_FLOATS);  // the other buffer was last read one stage ago
        if constexpr (X3) {
#pragma unroll
            for (int h = 0; h < DM_KC / 16; ++h) {
                u32x4 a3[NP], b3[NP];
                auto frag8 = [&](const float* base, int row, u32x4 (&dst)[NP]) {
                    const f32x4 lo4 = *reinterpret_cast<const f32x4*>(base + row * DM_KC + dm_swz(row, 4 * h + 2 * kh) * 4);
                    const f32x4 hi4 = *reinterpret_cast<const f32x4*>(base + row * DM_KC + dm_swz(row, 4 * h + 2 * kh + 1) * 4);
                    const float v[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                    SM::split8(v, dst);
                };
                frag8(sA, ra, a3), frag8(sW, rw, b3);
                SM::product1(c, b3, a3);  // the products of the large tiling in its order (weights = the MFMA's A operand)
            }
        } else {
#pragma unroll
        for (int s = 0; s < DM_KC / 8; ++s) {
            const f32x4 a = frag(sA, ra, s), b = frag(sW, rw, s);
            // weights are the MFMA's A operand, activations its B operand (a lane then owns one output row)
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x, a.x, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.y, a.y, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.z, a.z, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(b.w, a.w, c, 0, 0, 0);
        }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
    }
    if (row >= M) return;
    float* crow = p.C + (size_t)row * p.ldc + p.c_coff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = colb + 8 * q;
        if (col >= N) continue;  // N % 4 == 0: a 16-byte group is inside or outside
        f32x4 v = {c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]};
        v = v + bia[q];
        if (p.alpha != 1.0f) v = v * p.alpha;
        if (p.relu) v = f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
        if (rot) {  // (x0, x1) -> (x0 c - x1 s, x1 c + x0 s) per feature pair, as apply_cached_rotary_emb
            const f32x4 e = aux[q];
            v = f32x4{(v.x * e.x) + ((-v.y) * e.y), (v.y * e.x) + (v.x * e.y), (v.z * e.z) + ((-v.w) * e.w), (v.w * e.z) + (v.z * e.w)};
        }
        if (HAS_RES) v = aux[q] + v;
        *reinterpret_cast<f32x4*>(crow + col) = v;
    }
}

bool gemm_uses_dma(int K, int ldw) {
    static const char* which = getenv("GTSFM_GEMM");  // "mfma" forces the register-staged kernel (A/B measurements)
    return K % DM_KC == 0 && ldw % 4 == 0 && !(which && which[0] == 'm');
}

int gemm_math_from_env() {  // GTSFM_GEMM_MATH = "bf16x3" -> 1, "f16x2" -> 2, anything else (unset, "f32") -> 0
    const char* math_env = getenv("GTSFM_GEMM_MATH");
    if (math_env && math_env[0] == 'b') return 1;
    if (math_env && math_env[0] == 'f' && math_env[1] == '1') return 2;
    return 0;
}

int launch_gemm_dma(const GemmParams& p, hipStream_t stream) {
    GemmBatch none = {nullptr, nullptr, 0};
    return launch_gemm_dma_batched(p, none, stream);
}

int launch_gemm_dma_batched(const GemmParams& p, const GemmBatch& bt, hipStream_t stream) {
    if (p.M <= 0 || (bt.problems && bt.nproblems <= 0)) return GTSFM_OK;
    const int ncb = ceil_div(p.N, 128), mtiles = ceil_div(p.M, 128);
    const int nprob = bt.problems ? bt.nproblems : 1;
    GTSFM_CHECK_ARG(!(p.rot_enc && p.res), "gemm: rotary epilogue and residual are exclusive");
    GTSFM_CHECK_ARG(!p.rot_enc || (p.rot_cols % 128 == 0 && p.N % 4 == 0 && p.ldc % 4 == 0 && p.c_coff % 4 == 0), "gemm: rotary epilogue needs 16-byte aligned rows");
    GTSFM_CHECK_ARG(!bt.problems || (!p.m_dev && !p.n_dev && !p.tile_cnt_idx && bt.counts), "gemm: a batch takes its sizes from the problem table");
    // Column blocks per workgroup: ONE. The kernel can walk several blocks of a row tile with one stage pipeline, which paid
    // while the prologue and a burst of DMA pieces per stage were expensive (76 vs 73 % at 256 -> 768). With the pieces issued
    // between the MFMAs and the transposed epilogue, one block per workgroup is as fast (79.5 / 84.2 / 77.2 % at 256->768 /
    // 512->512 / 512->256+res against 77.7 / 82.5 / 76.0 walking all blocks; 500 vs 501 image-pairs/s in the workload) and the
    // workgroups of one row tile then run side by side on ONE XCD (block order below), so its A rows are fetched once and
    // re-read from that L2: 179 MB fetched per 131072 x 256 -> 768 launch instead of 829 MB (A itself: 134 MB).
    // GTSFM_GEMM_NB = n walks n blocks (experiments).
    GemmParams q = p;
    int nbw = 1;
    static const char* env = getenv("GTSFM_GEMM_NB");
    if (env && atoi(env) > 0) nbw = atoi(env) < ncb ? atoi(env) : ncb;
    q.nb_per_wg = nbw;
    // Small launches take 64 x 64 tiles (gemm_dma_small_kernel: bit-identical results). Measured (tools/bench_gemm_small.py, us per
    // launch, 128 x 128 -> 64 x 64 tiles): 4096 rows (one pair at N = 2048) 512->256 36.8 -> 15.3, 512->512 39.1 -> 24.6, 256->768
    // 24.5 -> 21.1; 10240 rows (one pair at the 5000 cap) 512->512 71.5 -> 51.9, 512->256 41.2 -> 33.0, 256->768 41.4 -> 43.4; 20480 rows
    // 512->512 107.0 -> 93.9 (640 large tiles), 256->768 72.5 -> 75.6 (960); 40960 rows 512->512 177 -> 193 (1280): the small tiling
    // wins up to ~700 large tiles. GTSFM_GEMM_SMALL_BELOW overrides the threshold (0: never).
    const bool vec_ok = ((p.N & 3) == 0) && ((p.ldc & 3) == 0) && ((p.c_coff & 3) == 0) && (!p.res || (p.ldres & 3) == 0);
    // p.math = 1 (the matchers under GTSFM_GEMM_MATH=bf16x3; NOT bit-identical to the default): both tilings run the same six
    // products per block in the same order, so batched == single-pair results stay bit-identical under the switch too. The switch is
    // a field the CALLER sets (matcher_api.hip, the stand-alone linear entry points): SuperPoint's convPb / convDb come through this
    // launcher as well and leave it 0 -- keypoint scores and dense descriptors never change with the environment.
    const bool x3 = p.math == 1, h2 = p.math == 2;
    const char* small_env = getenv("GTSFM_GEMM_SMALL_BELOW");
    const long long small_below = small_env ? atoll(small_env) : 700LL * gtsfm_cu_count() / 256;  // measured on 256 CUs; scales with the chip
    if (!bt.problems && vec_ok && !p.n_dev && nbw == 1 && (long long)mtiles * ncb < small_below) {
        const dim3 sgrid(ceil_div(p.M, 64) * ceil_div(p.N, 64));
        const size_t slds = (size_t)2 * DS_STAGE_FLOATS * sizeof(float);
        if (h2 && q.rot_enc)
            hipLaunchKernelGGL((gemm_dma_small_kernel<false, true, 2>), sgrid, dim3(256), slds, stream, q);
        else if (h2 && q.res)
            hipLaunchKernelGGL((gemm_dma_small_kernel<true, false, 2>), sgrid, dim3(256), slds, stream, q);
        else if (h2)
            hipLaunchKernelGGL((gemm_dma_small_kernel<false, false, 2>), sgrid, dim3(256), slds, stream, q);
        else if (x3 && q.rot_enc)
            hipLaunchKernelGGL((gemm_dma_small_kernel<false, true, 1>), sgrid, dim3(256), slds, stream, q);
        else if (x3 && q.res)
            hipLaunchKernelGGL((gemm_dma_small_kernel<true, false, 1>), sgrid, dim3(256), slds, stream, q);
        else if (x3)
            hipLaunchKernelGGL((gemm_dma_small_kernel<false, false, 1>), sgrid, dim3(256), slds, stream, q);
        else if (q.rot_enc)
            hipLaunchKernelGGL((gemm_dma_small_kernel<false, true>), sgrid, dim3(256), slds, stream, q);
        else if (q.res)
            hipLaunchKernelGGL((gemm_dma_small_kernel<true, false>), sgrid, dim3(256), slds, stream, q);
        else
            hipLaunchKernelGGL((gemm_dma_small_kernel<false, false>), sgrid, dim3(256), slds, stream, q);
        GTSFM_CHECK_LAUNCH("gemm_dma_small_kernel");
        return GTSFM_OK;
    }
    dim3 grid(ceil_div(mtiles, 8) * 8 * ceil_div(ncb, nbw), nprob);
    const int groups = ceil_div(ncb, nbw), local_rows = ceil_div(mtiles, 8);  // column groups; row tiles per XCD
    static const char* super_env = getenv("GTSFM_GEMM_SUPERTILE");             // "0": row order everywhere (experiments)
    q.super_rows = 0;
    if (groups > 8 && !(super_env && super_env[0] == '0')) {
        q.super_rows = local_rows < 8 ? local_rows : 8;
        grid.x = 8 * ceil_div(local_rows, q.super_rows) * q.super_rows * ceil_div(groups, 8) * 8;
    }
    const size_t lds_bytes = (size_t)2 * DM_STAGE_FLOATS * sizeof(float);
    if (h2) {  // opt-in arithmetic (GTSFM_GEMM_MATH=f16x2): the same launch geometry, stages and epilogues
        if (q.rot_enc)
            hipLaunchKernelGGL((gemm_dma_walk_kernel<false, true, 2>), grid, dim3(256), lds_bytes, stream, q, bt);
        else if (q.res)
            hipLaunchKernelGGL((gemm_dma_walk_kernel<true, false, 2>), grid, dim3(256), lds_bytes, stream, q, bt);
        else
            hipLaunchKernelGGL((gemm_dma_walk_kernel<false, false, 2>), grid, dim3(256), lds_bytes, stream, q, bt);
        GTSFM_CHECK_LAUNCH("gemm_dma_walk_kernel (f16x2)");
        return GTSFM_OK;
    }
    if (x3) {  // opt-in arithmetic (GTSFM_GEMM_MATH=bf16x3): the same launch geometry, stages and epilogues
        if (q.rot_enc)
            hipLaunchKernelGGL((gemm_dma_walk_kernel<false, true, 1>), grid, dim3(256), lds_bytes, stream, q, bt);
        else if (q.res)
            hipLaunchKernelGGL((gemm_dma_walk_kernel<true, false, 1>), grid, dim3(256), lds_bytes, stream, q, bt);
        else
            hipLaunchKernelGGL((gemm_dma_walk_kernel<false, false, 1>), grid, dim3(256), lds_bytes, stream, q, bt);
        GTSFM_CHECK_LAUNCH("gemm_dma_walk_kernel (bf16x3)");
        return GTSFM_OK;
    }
    if (q.rot_enc)
        hipLaunchKernelGGL((gemm_dma_walk_kernel<false, true>), grid, dim3(256), lds_bytes, stream, q, bt);
    else if (q.res)
        hipLaunchKernelGGL((gemm_dma_walk_kernel<true, false>), grid, dim3(256), lds_bytes, stream, q, bt);
    else
        hipLaunchKernelGGL((gemm_dma_walk_kernel<false, false>), grid, dim3(256), lds_bytes, stream, q, bt);
    GTSFM_CHECK_LAUNCH("gemm_dma_walk_kernel");
    return GTSFM_OK;
}
