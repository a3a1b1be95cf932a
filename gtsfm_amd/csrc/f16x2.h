// The "f16x2" arithmetic (round 6, opt-in like bf16x3.h): an fp32 number is carried as TWO fp16 pieces, x ~ hi + lo with hi = RN16(x) and
// lo = RN16(x - hi) (x - hi is exact in fp32), and a product of two such numbers is executed as THREE fp16 x fp16 MFMA products with fp32
// accumulation: lo hi, hi lo, hi hi. Every piece product is exact in fp32 (11 x 11 significand bits).
//   * What the two pieces keep: 11 + 11 significand bits plus lo's sign; |x - hi - lo| <= 2^-22 |x| in the worst case and 2^-25 |x| on average while
//     lo is a NORMAL fp16, i.e. for |x| >= 2^-3, and <= 2^-25 ABSOLUTE below (lo is then a subnormal fp16 with spacing 2^-24; the matrix pipe
//     takes subnormal inputs unflushed and v_cvt_pk_f16_f32 produces them: tools/probe_f16_mfma.hip, profiles/r06_probe_f16_mfma.txt). The
//     dropped product lo lo is <= 2^-22 |x y|. Per product term: < 2^-20.9 |x y| worst case, 2^-24 |x y| on average (bf16x3: < 2^-21, 2^-24),
//     and never more than 2^-21 |x y| + 2^-24 (|x| + |y|) whatever the magnitudes inside fp16's range (tests/test_f16x2_host.py).
//   * What it does NOT keep: fp32's exponent range. |x| > 65504 becomes +-inf in hi (round to nearest overflows) and the result NaN -- loud, never
//     a silently clamped value. The matchers' operands (projected descriptors, LayerNorm'ed tokens, trained weights, softmax weights scaled to
//     <= 2^15 by the attention kernel itself) are O(1) - O(100).
//   * Cost: 3 fp16 MFMAs of 32 cycles (32x32x16) per 32 x 32 x 16 block where bf16x3 takes 6 and exact fp32 8 of 64 cycles: 3 / 16 of the
//     matrix-pipe cycles of the default. The fp16 pipe sustains 1.72 PFLOP/s on random operands on this part (power: 0.68 of its nominal
//     2.5; bf16 1.87, fp32 0.152 = 0.97 of nominal), i.e. 573 TFLOP/s of fp32-class products against 311 (bf16x3) and 152 (exact).
#pragma once

#include "bf16x3.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

// (a, b) -> one register of two hi pieces (a in the low half) and one of two lo pieces
__device__ __forceinline__ void h2_split_pair(float a, float b, unsigned& hi, unsigned& lo) {
    const f16x2v h = __builtin_convertvector(f32x2v{a, b}, f16x2v);  // v_cvt_pk_f16_f32: round to nearest even
    hi = __builtin_bit_cast(unsigned, h);
    // the residuals x - hi (exact in fp32) straight from the packed register: v_fma_mix_f32 reads one half of it as an fp16 source -- one
    // instruction per element where `a - (float)h.x` compiles to a conversion and a subtraction
    float ra, rb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hi), "v"(b));
    const f16x2v l = __builtin_convertvector(f32x2v{ra, rb}, f16x2v);
    lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ f32x16 h2_mfma(const u32x4 a, const u32x4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// The two split arithmetics behind one interface: NP pieces per operand (3: bf16x3, 2: f16x2); dst[0] is the leading piece.
template <int NP>
struct SplitMath;

template <>
struct SplitMath<3> {
    static constexpr int kPieces = 3;
    __device__ static __forceinline__ void split8(const float (&v)[8], u32x4 (&dst)[3]) {
        x3_split8(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]}, dst);
    }
    __device__ static __forceinline__ void product(f32x16& acc0, f32x16& acc1, const u32x4 (&a0)[3], const u32x4 (&a1)[3], const u32x4 (&b)[3]) {
        x3_product(acc0, acc1, a0, a1, b);
    }
    // one accumulator: the products of x3_product in its order (a = the MFMA's A operand)
    __device__ static __forceinline__ void product1(f32x16& c, const u32x4 (&a)[3], const u32x4 (&b)[3]) {
        c = x3_mfma(a[2], b[0], c), c = x3_mfma(a[0], b[2], c), c = x3_mfma(a[1], b[1], c);
        c = x3_mfma(a[1], b[0], c), c = x3_mfma(a[0], b[1], c), c = x3_mfma(a[0], b[0], c);
    }
};

template <>
struct SplitMath<2> {
    static constexpr int kPieces = 2;
    __device__ static __forceinline__ void split8(const float (&v)[8], u32x4 (&dst)[2]) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) h2_split_pair(v[2 * w], v[2 * w + 1], hi[w], lo[w]);
        dst[0] = u32x4{hi[0], hi[1], hi[2], hi[3]};
        dst[1] = u32x4{lo[0], lo[1], lo[2], lo[3]};
    }
    // acc += A B with A = ah + al, B = bh + bl: the three products, smallest first
    __device__ static __forceinline__ void product(f32x16& acc0, f32x16& acc1, const u32x4 (&a0)[2], const u32x4 (&a1)[2], const u32x4 (&b)[2]) {
        acc0 = h2_mfma(a0[1], b[0], acc0), acc1 = h2_mfma(a1[1], b[0], acc1);  // lo hi
        acc0 = h2_mfma(a0[0], b[1], acc0), acc1 = h2_mfma(a1[0], b[1], acc1);  // hi lo
        acc0 = h2_mfma(a0[0], b[0], acc0), acc1 = h2_mfma(a1[0], b[0], acc1);  // hi hi
    }
    __device__ static __forceinline__ void product1(f32x16& c, const u32x4 (&a)[2], const u32x4 (&b)[2]) {
        c = h2_mfma(a[1], b[0], c), c = h2_mfma(a[0], b[1], c), c = h2_mfma(a[0], b[0], c);
    }
};
