// Does a VALU instruction that reads the results of v_permlane32_swap_b32 need wait states the compiler does not insert? The attention kernels combine the
// two lane halves of a query with
//     v_mov v1, v0 ; s_nop 1 ; v_permlane32_swap v0, v1 ; [k other instructions] ; v_add v2, v0, v1
// (at_halves_sum / at_halves_max). This probe runs that sequence with k = 0, 1 (an MFMA), 1 (a VALU), 2, and with 8 idle cycles as the reference, on SIMDs
// that are otherwise idle and on SIMDs kept busy by other waves, and counts lanes whose sum differs from the reference.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_permlane_swap.hip -o tools/bin/probe_permlane_swap && tools/bin/probe_permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define SWAP_HEAD "v_mov_b32 v101, %1\n\tv_mov_b32 v100, %1\n\ts_nop 4\n\tv_permlane32_swap_b32 v100, v101\n\t"
#define SWAP_TAIL "v_add_f32 %0, v100, v101\n\t"
#define MF "v_mfma_f32_32x32x16_f16 v[104:119], v[120:123], v[124:127], v[104:119]\n\t"

template <int KIND>
__device__ __forceinline__ float halves_sum(float x) {
    float out;
    if (KIND == 0) asm volatile(SWAP_HEAD SWAP_TAIL : "=v"(out) : "v"(x) : "v100", "v101");
    if (KIND == 1) asm volatile(SWAP_HEAD MF SWAP_TAIL : "=v"(out) : "v"(x) : "v100", "v101", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119");
    if (KIND == 2) asm volatile(SWAP_HEAD "v_mov_b32 v102, v103\n\t" SWAP_TAIL : "=v"(out) : "v"(x) : "v100", "v101", "v102");
    if (KIND == 3) asm volatile(SWAP_HEAD "v_mov_b32 v102, v103\n\tv_mov_b32 v102, v103\n\t" SWAP_TAIL : "=v"(out) : "v"(x) : "v100", "v101", "v102");
    if (KIND == 4) asm volatile(SWAP_HEAD "s_nop 7\n\t" SWAP_TAIL : "=v"(out) : "v"(x) : "v100", "v101");
    if (KIND == 6) asm volatile("v_mov_b32 v101, %1\n\tv_mov_b32 v100, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 v100, v101\n\ts_nop 7\n\t" SWAP_TAIL : "=v"(out) : "v"(x) : "v100", "v101");
    if (KIND == 7) asm volatile("v_mov_b32 v101, %1\n\tv_mov_b32 v100, %1\n\tv_permlane32_swap_b32 v100, v101\n\ts_nop 7\n\t" SWAP_TAIL : "=v"(out) : "v"(x) : "v100", "v101");
    if (KIND == 8) asm volatile("v_mov_b32 v100, %1\n\tv_mov_b32 v101, v100\n\ts_nop 1\n\tv_permlane32_swap_b32 v100, v101\n\t" MF SWAP_TAIL : "=v"(out) : "v"(x) : "v100", "v101", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119");
    if (KIND == 5) asm volatile("s_barrier\n\t" SWAP_HEAD MF SWAP_TAIL : "=v"(out) : "v"(x) : "v100", "v101", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119");
    return out;
}

__global__ __launch_bounds__(256) void probe(unsigned* bad, float* sink, int iters, int victims_every) {
    const int lane = threadIdx.x & 63;
    if (blockIdx.x % victims_every != 0) {  // partner waves: VALU-only work (no MFMA), so that the victim's MFMA issues at once
        float a = lane * 0.001f, b = 1.0001f;
        for (int i = 0; i < iters * 40; ++i) a = a * b + 0.5f;
        if (a == 1.2345f) sink[threadIdx.x] = a;
        return;
    }
    unsigned nbad[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        const float x = (float)((lane * 7 + i * 13) & 1023) * 0.25f + 1.0f;
        const float ref = halves_sum<4>(x);
        nbad[0] += !(halves_sum<0>(x) == ref);
        nbad[1] += !(halves_sum<1>(x) == ref);
        nbad[2] += !(halves_sum<2>(x) == ref);
        nbad[3] += !(halves_sum<3>(x) == ref);
        nbad[5] += !(halves_sum<5>(x) == ref);
        nbad[6] += !(halves_sum<6>(x) == ref);
        nbad[7] += !(halves_sum<7>(x) == ref);
        nbad[8] += !(halves_sum<8>(x) == ref);
        // and the reference against plain arithmetic: the partner lane's x differs by lane ^ 32
        const float other = (float)((((lane ^ 32) * 7) + i * 13) & 1023) * 0.25f + 1.0f;
        nbad[4] += !(ref == x + other);
    }
    for (int k = 0; k < 9; ++k)
        if (nbad[k]) atomicAdd(bad + k, nbad[k]);
}

int main() {
    unsigned* bad;
    float* sink;
    hipMalloc(&bad, 64);
    hipMalloc(&sink, 4096);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    for (int every : {1, 2, 4}) {
        hipMemset(bad, 0, 64);
        hipLaunchKernelGGL(probe, dim3(cus * 2 * every), dim3(256), 0, 0, bad, sink, 20000, every);
        hipDeviceSynchronize();
        unsigned h[9];
        hipMemcpy(h, bad, 36, hipMemcpyDeviceToHost);
        printf("victim workgroups 1 in %d: lane-results differing from the 8-idle-cycle form: use at once %u, behind one MFMA %u, behind one VALU %u, behind two VALU %u, "
               "behind s_barrier + MFMA %u; producer 2 wait states before the swap (the compiler's form) %u, producer 0 wait states %u, compiler's form + MFMA + use %u; reference != x + x(lane ^ 32): %u   (of %d per lane)\n", every, h[0], h[1], h[2], h[3], h[5], h[6], h[7], h[8], h[4], 20000);
    }
    return 0;
}
