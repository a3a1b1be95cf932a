// Does a VALU write to an MFMA's SrcA / SrcB registers, issued right behind the MFMA, reach the matrix core before the MFMA has read them
// (a write-after-read hazard the compiler does not pad: LLVM's recognizer covers SrcC only)? One "victim" wave per SIMD issues
//     v_mfma_f32_32x32x16_f16 acc, A, B, 0 ; v_mov A, garbage ; v_mov B, garbage
// while `hammer` other waves on the same SIMD keep the matrix pipe busy with dependent and independent MFMAs (so that the victim's MFMA has
// to queue). The result is compared with the same MFMA followed by 32 idle cycles before the overwrite.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_war.hip -o tools/bin/probe_mfma_war && tools/bin/probe_mfma_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int GAP>  // s_nop between the MFMA and the overwrite: 0 = none
__device__ __forceinline__ float victim_once(unsigned a0, unsigned a1, unsigned a2, unsigned a3, unsigned b0, unsigned b1, unsigned b2, unsigned b3) {
    float out;
    if (GAP == 0)
        asm volatile(
            "v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %3\n\tv_mov_b32 v103, %4\n\t"
            "v_mov_b32 v104, %5\n\tv_mov_b32 v105, %6\n\tv_mov_b32 v106, %7\n\tv_mov_b32 v107, %8\n\t"
            "s_nop 7\n\t"
            "v_mfma_f32_32x32x16_f16 v[108:123], v[100:103], v[104:107], 0\n\t"
            "v_mov_b32 v100, 0x7e007e00\n\tv_mov_b32 v101, 0x7e007e00\n\tv_mov_b32 v102, 0x7e007e00\n\tv_mov_b32 v103, 0x7e007e00\n\t"
            "v_mov_b32 v104, 0x7e007e00\n\tv_mov_b32 v105, 0x7e007e00\n\tv_mov_b32 v106, 0x7e007e00\n\tv_mov_b32 v107, 0x7e007e00\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_add_f32 %0, v108, v123\n\t"
            : "=v"(out)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3)
            : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117",
              "v118", "v119", "v120", "v121", "v122", "v123");
    else
        asm volatile(
            "v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %3\n\tv_mov_b32 v103, %4\n\t"
            "v_mov_b32 v104, %5\n\tv_mov_b32 v105, %6\n\tv_mov_b32 v106, %7\n\tv_mov_b32 v107, %8\n\t"
            "s_nop 7\n\t"
            "v_mfma_f32_32x32x16_f16 v[108:123], v[100:103], v[104:107], 0\n\t"
            "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
            "v_mov_b32 v100, 0x7e007e00\n\tv_mov_b32 v101, 0x7e007e00\n\tv_mov_b32 v102, 0x7e007e00\n\tv_mov_b32 v103, 0x7e007e00\n\t"
            "v_mov_b32 v104, 0x7e007e00\n\tv_mov_b32 v105, 0x7e007e00\n\tv_mov_b32 v106, 0x7e007e00\n\tv_mov_b32 v107, 0x7e007e00\n\t"
            "s_nop 15\n\ts_nop 15\n\t"
            "v_add_f32 %0, v108, v123\n\t"
            : "=v"(out)
            : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3)
            : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117",
              "v118", "v119", "v120", "v121", "v122", "v123");
    return out;
}

// The same with a DEPENDENT pair: the second MFMA accumulates onto the first one's result (it has to wait for it in the matrix pipe) and its
// SrcA / SrcB are overwritten right behind it -- the shape the compiler produced in attention_x3_kernel (two products onto one accumulator, then the
// address arithmetic of the next DMA into the operand registers).
template <int GAP>
__device__ __forceinline__ float victim_pair(unsigned a0, unsigned a1, unsigned a2, unsigned a3, unsigned b0, unsigned b1, unsigned b2, unsigned b3) {
    float out;
#define LOADS "v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %3\n\tv_mov_b32 v103, %4\n\tv_mov_b32 v104, %5\n\tv_mov_b32 v105, %6\n\tv_mov_b32 v106, %7\n\tv_mov_b32 v107, %8\n\t" \
              "v_mov_b32 v124, %2\n\tv_mov_b32 v125, %1\n\tv_mov_b32 v126, %4\n\tv_mov_b32 v127, %3\n\ts_nop 7\n\t"
#define PAIR "v_mfma_f32_32x32x16_f16 v[108:123], v[124:127], v[104:107], 0\n\tv_mfma_f32_32x32x16_f16 v[108:123], v[100:103], v[104:107], v[108:123]\n\t"
#define SMASH "v_mov_b32 v100, 0x7e007e00\n\tv_mov_b32 v101, 0x7e007e00\n\tv_mov_b32 v102, 0x7e007e00\n\tv_mov_b32 v103, 0x7e007e00\n\t" \
              "v_mov_b32 v104, 0x7e007e00\n\tv_mov_b32 v105, 0x7e007e00\n\tv_mov_b32 v106, 0x7e007e00\n\tv_mov_b32 v107, 0x7e007e00\n\t"
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", \
             "v122", "v123", "v124", "v125", "v126", "v127"
    if (GAP == 0)
        asm volatile(LOADS PAIR SMASH "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\tv_add_f32 %0, v108, v123\n\t"
                     : "=v"(out) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3) : CLOB);
    else
        asm volatile(LOADS PAIR "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t" SMASH "s_nop 15\n\ts_nop 15\n\tv_add_f32 %0, v108, v123\n\t"
                     : "=v"(out) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3) : CLOB);
    return out;
}

// blockDim = 64 * (1 + hammer) * 4?  Simpler: one workgroup of 4 waves per "role"; launch (1 + hammer) workgroups per CU. Role by blockIdx parity.
__global__ __launch_bounds__(256) void war_probe(unsigned* bad, float* sink, int iters, int victims_every, int dependent) {
    const int lane = threadIdx.x & 63;
    if (blockIdx.x % victims_every != 0) {  // hammer: keep the matrix pipe of this SIMD busy
        h8 a, b;
        for (int i = 0; i < 8; ++i) a[i] = (_Float16)(0.01f * (lane + i)), b[i] = (_Float16)(0.5f);
        f32x16 c0, c1;
        for (int r = 0; r < 16; ++r) c0[r] = c1[r] = 0.f;
        for (int i = 0; i < iters * 6; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        }
        float s = 0.f;
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
        if (s == 1.2345f) sink[threadIdx.x] = s;
        return;
    }
    unsigned nbad = 0;
    for (int i = 0; i < iters; ++i) {
        // fp16 pairs: small integers so that everything is exact
        const unsigned x = 0x3c003c00u + (((unsigned)(lane + i) & 7u) << 6);  // (1 + k/16, 1 + k/16)-ish patterns
        const unsigned y = 0x40003800u + (((unsigned)(lane * 3 + i) & 3u) << 22);
        const float safe = dependent ? victim_pair<1>(x, x ^ 0x00400040u, x, x ^ 0x00800000u, y, y, y ^ 0x04000000u, y) : victim_once<1>(x, x ^ 0x00400040u, x, x ^ 0x00800000u, y, y, y ^ 0x04000000u, y);
        const float fast = dependent ? victim_pair<0>(x, x ^ 0x00400040u, x, x ^ 0x00800000u, y, y, y ^ 0x04000000u, y) : victim_once<0>(x, x ^ 0x00400040u, x, x ^ 0x00800000u, y, y, y ^ 0x04000000u, y);
        if (!(safe == fast)) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    unsigned* bad;
    float* sink;
    hipMalloc(&bad, 4);
    hipMalloc(&sink, 4096);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    for (int dependent : {0, 1})
    for (int every : {1, 2, 3, 4}) {  // 1: victims only; 2: one hammer workgroup per victim workgroup; ...
        hipMemset(bad, 0, 4);
        const int grid = cus * every * 2;
        hipLaunchKernelGGL(war_probe, dim3(grid), dim3(256), 0, 0, bad, sink, 20000, every, dependent);
        hipDeviceSynchronize();
        unsigned h = 0;
        hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("%s: victim workgroups 1 in %d (the others hammer the matrix pipe), %d workgroups, 20000 per victim lane: %u lane-results differ between overwrite-at-once and overwrite-after-64-nops\n",
               dependent ? "dependent MFMA pair" : "single MFMA", every, grid, h);
    }
    return 0;
}
