// PROTOTYPE for round 2 -- not part of libgtsfm_amd.so. Status at the end of round 1 (one run, in the round's last GPU
// seconds): ragged self-check green (max error 4.8e-6), 84.8 % of the fp32 MFMA peak at 64 sequences x N = 2048 with TWO
// workgroups per CU and nothing tuned (attention_mfma_kernel: 84.5 % with three).
// fp32-MFMA flash attention (same contract and tiling as attention_mfma_kernel: head_dim 64, workgroup = 4 waves x 32
// queries, 64-key tiles, transposed score tile S^T = K Q^T so that a query is a lane) with two structural changes that
// the round-1 timeline asked for (per tile a wave has 8192 cycles of MFMA and ~4200 cycles of
// everything else, and the matrix pipe idles when all three waves of a SIMD are in that part at once):
//
//  1. K / V tiles travel by LDS-DMA (global_load_lds_dwordx4) into an XOR-swizzled, unpadded LDS image: no staging
//     registers (32 VGPRs), no ds_write pass, no "tile store" segment between two barriers.
//  2. The freed registers hold a second score tile, so the softmax of tile t (VALU) is issued in the shadow of the
//     S = K Q^T MFMAs of tile t + 1 INSIDE the same wave (FlashAttention-3's intra-warp pipelining): a wave offers the
//     matrix pipe MFMAs during its own softmax instead of relying on the other two waves of the SIMD.
//
// Per iteration t (K buffer holds K(t+1), V buffer holds V(t), both single-buffered, 32 KiB of LDS per workgroup):
//     phase 1: S(t+1) MFMAs (read K buffer)  ||  softmax(t) on the previous score tile
//     wait own V(t) DMA, barrier B1            -> K buffer free, V(t) visible;  issue DMA K(t+2) -> K buffer
//     phase 2: O += V^T P^T MFMAs (read V buffer)
//     wait own K(t+2) DMA, barrier B2          -> V buffer free, K(t+2) visible; issue DMA V(t+1) -> V buffer
// Each DMA has a whole MFMA phase (4-9 k cycles) to land before the barrier that publishes it.
//
// LDS image of a tile: [64 rows][64 floats] = 256 B per row = sixteen 16-byte chunks, chunk c of row r stored at chunk
// position c ^ (r & 15). A row spans all 64 banks, so the bank of an access is set by the chunk position alone:
//   * S phase, ds_read_b128 of chunk (2u + kh) by rows j of one 16-lane service group ({0-3,12-15,20-27} /
//     {4-11,16-19,28-31}): r & 15 takes 16 distinct values in either group -> 16 distinct positions -> conflict-free;
//   * P V phase, ds_read_b32 of V[key][d = lane] by 32 lanes: chunks d / 4 = 0..7 (or 8..15) XOR a constant stay 8
//     distinct positions within one 32-bank half -> conflict-free.
// The DMA writes lane-linear (1 KiB = 4 rows per instruction), so the swizzle goes on the per-lane global address.
//
// Registers: two score tiles + O + Q = 128 accumulator-class registers; the compiler wants 233 VGPRs (two workgroups
// per CU, the default here) and spills 65 when held to 168 (-DAT_WGS_PER_CU=3). Both are to be measured; if two fat
// waves per SIMD with intra-wave overlap lose to three thin ones, the next step is pipelining at 32-key granularity.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans docs/experimental/attention_dma.hip -o docs/experimental/attention_dma
//   docs/experimental/attention_dma            self-check against an fp64 CPU reference (ragged, small), then timing
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define AT_KT 64
#define AT_QB 128
#define TILE_FLOATS (AT_KT * 64)
#define AT_REBASE 8.0f
#ifndef AT_WGS_PER_CU
#define AT_WGS_PER_CU 2  // 3 needs <= 168 VGPRs: with two score tiles live the kernel then spills (65 registers); 2 -> ~190, no spills
#endif

struct AttnProblem {
    int q_off, q_cnt_idx, k_off, k_cnt_idx;
};
struct AttnParams {  // as gtsfm_amd/csrc/attention_kernels.h
    const float* q; int ldq;
    const float* k; int ldk;
    const float* v; int ldv;
    float* out; int ldo;
    const AttnProblem* problems;
    const int* counts;
    float scale;
    int heads;
    int qtiles, nproblems;
};

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

__global__ __launch_bounds__(256, AT_WGS_PER_CU) void attention_dma_kernel(AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ks = lds;                // [64 keys][64 floats], swizzled
    float* Vs = lds + TILE_FLOATS;  // same
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

#ifndef AT_INTERLEAVE
    // ---- variant validated on the GPU at the end of round 1 (the compiler keeps the S MFMAs and the softmax in separate
    // basic blocks here: the uniform branches between them stop its scheduler from interleaving the two)
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
#else
    // ---- NOT YET RUN: the same pipeline arranged so that MFMAs and softmax VALU share straight-line blocks. All tiles but
    // the last take the `true` path (never masked: only the last tile can be partial); the rare rebase / shift branches
    // sit BETWEEN the two halves of the S phase: block A = accumulator start + 32 MFMAs (u = 0..3) + row maximum,
    // block B = 32 MFMAs (u = 4..7) + 32 exp2 + row sum.
    auto pv_phase = [&]() {
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
    };
    auto rebase_by = [&](float d, bool scale_o) {  // rare path: shift the current scores, rescale O and l
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc0[r] -= d;
            sc1[r] -= d;
        }
        if (scale_o) {
            const float alpha = __builtin_amdgcn_exp2f(-d);
            l *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o0[r] *= alpha;
                o1[r] *= alpha;
            }
        }
        m += d;
    };
    for (int t = 0; t + 1 < ntiles; ++t) {
        const int k0 = t * AT_KT;
        // block A
        const float neg_m = -m;
#pragma unroll
        for (int r = 0; r < 16; ++r) sn0[r] = sn1[r] = neg_m;
#pragma unroll
        for (int u = 0; u < 4; ++u) mfma8(sn0, sn1, kfrag(j, u), kfrag(32 + j, u), qreg[u]);
        float mloc = fmaxf(sc0[0], sc1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mloc = fmaxf(mloc, fmaxf(sc0[r], sc1[r]));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const bool rebase = (t == 0) || (mloc > AT_REBASE);
        float d = 0.f;
        if (__any(rebase)) {
            d = rebase ? mloc : 0.f;
            rebase_by(d, t > 0);
        }
        // block B
#pragma unroll
        for (int u = 4; u < 8; ++u) mfma8(sn0, sn1, kfrag(j, u), kfrag(32 + j, u), qreg[u]);
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc0[r] = __builtin_amdgcn_exp2f(sc0[r]);
            sc1[r] = __builtin_amdgcn_exp2f(sc1[r]);
            lsum += sc0[r] + sc1[r];
        }
        lsum += __shfl_xor(lsum, 32, 64);
        l += lsum;
        if (__any(d != 0.f)) {  // the next tile was accumulated relative to the reference before this rebase
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sn0[r] -= d;
                sn1[r] -= d;
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);  // own V(t) DMA landed
        __syncthreads();                     // B1: K buffer free, V(t) visible
        if (t + 2 < ntiles) tile_dma(kbase, p.ldk, k0 + 2 * AT_KT, Ks);
        pv_phase();
        __builtin_amdgcn_s_waitcnt(0x0f70);  // own K(t+2) DMA landed
        __syncthreads();                     // B2: V buffer free, K(t+2) visible
        tile_dma(vbase, p.ldv, k0 + AT_KT, Vs);
        sc0 = sn0, sc1 = sn1;
    }
    {  // last tile: mask, softmax, P V
        const int t = ntiles - 1, k0 = t * AT_KT;
        if (k0 + AT_KT > nk) {
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
        if (__any(rebase)) rebase_by(rebase ? mloc : 0.f, t > 0);
        float lsum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sc0[r] = __builtin_amdgcn_exp2f(sc0[r]);
            sc1[r] = __builtin_amdgcn_exp2f(sc1[r]);
            lsum += sc0[r] + sc1[r];
        }
        lsum += __shfl_xor(lsum, 32, 64);
        l += lsum;
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();  // V(t) visible
        pv_phase();
    }
#endif
    if (!qvalid) return;
    const float inv = (l > 0.f) ? 1.0f / l : 0.f;
    float* op = p.out + (size_t)(pr.q_off + qrow) * p.ldo + h * 64 + kh * 4;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        *reinterpret_cast<f32x4*>(op + 8 * gq) = f32x4{o0[4 * gq] * inv, o0[4 * gq + 1] * inv, o0[4 * gq + 2] * inv, o0[4 * gq + 3] * inv};
        *reinterpret_cast<f32x4*>(op + 32 + 8 * gq) = f32x4{o1[4 * gq] * inv, o1[4 * gq + 1] * inv, o1[4 * gq + 2] * inv, o1[4 * gq + 3] * inv};
    }
}

static void launch(AttnParams p, int nproblems, int max_q) {
    p.qtiles = (max_q + AT_QB - 1) / AT_QB;
    p.nproblems = nproblems;
    const int groups = p.heads * nproblems;
    const dim3 grid(((groups + 7) / 8) * 8 * p.qtiles);
    hipLaunchKernelGGL(attention_dma_kernel, grid, dim3(256), (size_t)2 * TILE_FLOATS * sizeof(float), 0, p);
}

// ragged self-check against an fp64 reference: problems with different query / key counts, including tails
static bool self_check() {
    const int heads = 4;
    const int nqs[4] = {1, 70, 129, 200}, nks[4] = {5, 64, 131, 257};
    int rows = 0;
    std::vector<AttnProblem> pr;
    std::vector<int> cnt;
    for (int s = 0; s < 4; ++s) {  // queries of problem s live in block 2 s, keys / values in block 2 s + 1
        pr.push_back({rows, 2 * s, rows + nqs[s], 2 * s + 1});
        cnt.push_back(nqs[s]);
        cnt.push_back(nks[s]);
        rows += nqs[s] + nks[s];
    }
    std::vector<float> h((size_t)rows * 768), ho((size_t)rows * 256, 0.f);
    unsigned st = 99;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = ((st >> 8) * (1.0f / 16777216.0f) - 0.5f) * 6.0f; }
    float *qkv, *out;
    AttnProblem* dpr;
    int* dcnt;
    hipMalloc(&qkv, h.size() * 4); hipMalloc(&out, ho.size() * 4); hipMalloc(&dpr, pr.size() * sizeof(AttnProblem)); hipMalloc(&dcnt, cnt.size() * 4);
    hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(out, 0, ho.size() * 4);
    hipMemcpy(dpr, pr.data(), pr.size() * sizeof(AttnProblem), hipMemcpyHostToDevice);
    hipMemcpy(dcnt, cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice);
    AttnParams p{qkv, 768, qkv + 256, 768, qkv + 512, 768, out, 256, dpr, dcnt, 0.125f, heads, 0, 0};
    launch(p, 4, 200);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return false; }
    hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int s = 0; s < 4; ++s)
        for (int hh = 0; hh < heads; ++hh)
            for (int qi = 0; qi < nqs[s]; ++qi) {
                const float* q = &h[(size_t)(pr[s].q_off + qi) * 768 + hh * 64];
                std::vector<double> sc(nks[s]);
                double mx = -1e300;
                for (int ki = 0; ki < nks[s]; ++ki) {
                    const float* k = &h[(size_t)(pr[s].k_off + ki) * 768 + 256 + hh * 64];
                    double acc = 0;
                    for (int dd = 0; dd < 64; ++dd) acc += (double)q[dd] * k[dd];
                    sc[ki] = acc * 0.125;
                    mx = fmax(mx, sc[ki]);
                }
                double den = 0;
                for (auto& v : sc) { v = exp(v - mx); den += v; }
                for (int dd = 0; dd < 64; ++dd) {
                    double acc = 0;
                    for (int ki = 0; ki < nks[s]; ++ki) acc += sc[ki] * h[(size_t)(pr[s].k_off + ki) * 768 + 512 + hh * 64 + dd];
                    worst = fmax(worst, fabs(acc / den - ho[(size_t)(pr[s].q_off + qi) * 256 + hh * 64 + dd]));
                }
            }
    printf("self-check (4 ragged problems x 4 heads): max |error| = %.3e (%s)\n", worst, worst < 2e-5 ? "OK" : "FAILED");
    hipFree(qkv); hipFree(out); hipFree(dpr); hipFree(dcnt);
    return worst < 2e-5;
}

int main(int argc, char** argv) {
    const bool ok = self_check();
    const int nseq = argc > 1 ? atoi(argv[1]) : 64, n = argc > 2 ? atoi(argv[2]) : 2048, heads = 4;
    const size_t rows = (size_t)nseq * n;
    float *qkv, *out;
    hipMalloc(&qkv, rows * 768 * 4);
    hipMalloc(&out, rows * 256 * 4);
    std::vector<float> h(rows * 768);
    unsigned st = 1;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = ((st >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4f; }
    hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<AttnProblem> pr(nseq);
    std::vector<int> cnt(nseq, n);
    for (int s = 0; s < nseq; ++s) pr[s] = {s * n, s, s * n, s};
    AttnProblem* dpr; int* dcnt;
    hipMalloc(&dpr, nseq * sizeof(AttnProblem)); hipMalloc(&dcnt, nseq * 4);
    hipMemcpy(dpr, pr.data(), nseq * sizeof(AttnProblem), hipMemcpyHostToDevice);
    hipMemcpy(dcnt, cnt.data(), nseq * 4, hipMemcpyHostToDevice);
    AttnParams p{qkv, 768, qkv + 256, 768, qkv + 512, 768, out, 256, dpr, dcnt, 0.125f, heads, 0, 0};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch(p, nseq, n);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) launch(p, nseq, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double tf = 1024.0 * n * n * nseq / (ms * 1e-3) / 1e12;
    printf("attention_dma %d seq x %d: %.3f ms per launch, %.1f TFLOP/s (%.1f %% of 157.3); attention_mfma_kernel: 84.5 %%\n", nseq, n, ms, tf, 100 * tf / 157.3);
    return ok ? 0 : 1;
}
