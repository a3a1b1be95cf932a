// The "bf16x3" arithmetic shared by the opt-in attention and GEMM kernels: an fp32 number is split EXACTLY into three bf16 pieces
// (8 + 8 + 8 significand bits), a product of two such numbers is executed as six bf16 x bf16 MFMA products with fp32 accumulation
// (hi hi, hi mid, mid hi, mid mid, hi lo, lo hi; the three dropped products are ~2^-24 of the term, below 2^-21 of it in the worst case). See attention_kernels.hip.
#pragma once

#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct X3Split {
    unsigned hi, mid, lo;  // fp32 bit patterns whose top 16 bits are the bf16 pieces
};
__device__ __forceinline__ X3Split x3_split(float x) {
    X3Split r;
    r.hi = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(r.hi);  // exact
    r.mid = __float_as_uint(r1) & 0xffff0000u;
    r.lo = __float_as_uint(r1 - __uint_as_float(r.mid));  // exact; truncated to 8 bits when packed
    return r;
}
// two bf16 (the top halves of a and b) in one register, a in the low half
__device__ __forceinline__ unsigned x3_pack(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
__device__ __forceinline__ bf16x8 x3_frag(const u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ f32x16 x3_mfma(const u32x4 a, const u32x4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x3_frag(a), x3_frag(b), c, 0, 0, 0);
}
// acc += A B with A = ah + am + al, B = bh + bm + bl: the six products, smallest first
__device__ __forceinline__ void x3_product(f32x16& acc0, f32x16& acc1, const u32x4 (&a0)[3], const u32x4 (&a1)[3], const u32x4 (&b)[3]) {
    acc0 = x3_mfma(a0[2], b[0], acc0), acc1 = x3_mfma(a1[2], b[0], acc1);  // lo hi
    acc0 = x3_mfma(a0[0], b[2], acc0), acc1 = x3_mfma(a1[0], b[2], acc1);  // hi lo
    acc0 = x3_mfma(a0[1], b[1], acc0), acc1 = x3_mfma(a1[1], b[1], acc1);  // mid mid
    acc0 = x3_mfma(a0[1], b[0], acc0), acc1 = x3_mfma(a1[1], b[0], acc1);  // mid hi
    acc0 = x3_mfma(a0[0], b[1], acc0), acc1 = x3_mfma(a1[0], b[1], acc1);  // hi mid
    acc0 = x3_mfma(a0[0], b[0], acc0), acc1 = x3_mfma(a1[0], b[0], acc1);  // hi hi
}


// eight fp32 values (two 16-byte fragments of one row: k .. k + 7) -> the three bf16 pieces of an MFMA operand
__device__ __forceinline__ void x3_split8(const f32x4 lo4, const f32x4 hi4, u32x4 (&dst)[3]) {
    unsigned hi[8], mid[8], lo[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const X3Split a = x3_split(lo4[e]), b = x3_split(hi4[e]);
        hi[e] = a.hi, mid[e] = a.mid, lo[e] = a.lo;
        hi[4 + e] = b.hi, mid[4 + e] = b.mid, lo[4 + e] = b.lo;
    }
    dst[0] = u32x4{x3_pack(hi[0], hi[1]), x3_pack(hi[2], hi[3]), x3_pack(hi[4], hi[5]), x3_pack(hi[6], hi[7])};
    dst[1] = u32x4{x3_pack(mid[0], mid[1]), x3_pack(mid[2], mid[3]), x3_pack(mid[4], mid[5]), x3_pack(mid[6], mid[7])};
    dst[2] = u32x4{x3_pack(lo[0], lo[1]), x3_pack(lo[2], lo[3]), x3_pack(lo[4], lo[5]), x3_pack(lo[6], lo[7])};
}
