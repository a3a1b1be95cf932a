// Where do MFMA-bound loops with LDS-fed operands top out? (The conv, GEMM and attention kernels of this repo all
// saturate at 83-86 % of the fp32 MFMA peak although tools/mfma_peak.hip sustains 99 % from registers.)
// One k-step = 2 ds_read_b128 (conflict-free, row stride 68 floats) + optionally 1 global 16-byte load per lane from an
// L2-resident stream + 8 v_mfma_f32_32x32x2_f32 on two accumulators, i.e. the inner loop of conv3x3_mfma_kernel without
// staging, barriers or epilogue. Variants: operands read at use / one step ahead with the issue order pinned; global
// stream off / 1 / 3 steps ahead; 1-4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_lds_feed.hip -o tools/mfma_lds_feed && tools/mfma_lds_feed
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define ROW 68

template <int AHEAD, int GLOBAL>
__global__ __launch_bounds__(256) void feed_kernel(const float* __restrict__ w, float* out, int iters, int wsteps) {
    __shared__ __attribute__((aligned(16))) float lds[128 * ROW];
    __shared__ __attribute__((aligned(16))) float wring[4 * 4 * 256];  // GLOBAL == 4: per wave a ring of 4 x 1 KiB weight fragments
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 128 * ROW; i += 256) lds[i] = (float)((i * 2654435761u) >> 20) * 1e-4f - 0.2f;
    __syncthreads();
    const int j = lane & 31, kh = lane >> 5;
    const int a_base0 = ((wave >> 1) * 64 + j) * ROW + kh * 4, a_base1 = a_base0 + 32 * ROW;
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    const float* wp = w + (size_t)(wave & 1) * 256 + lane * 4;  // [step][2 halves][64 lanes][4], L2-resident
    f32x4 bc = {0.01f, -0.02f, 0.03f, 0.015f}, b1 = bc, b2 = bc;
    int s = 0;
    float* ring = wring + wave * 1024;
    if (GLOBAL == 4) {  // LDS-DMA: global -> LDS without passing through VGPRs, fragment read back with ds_read_b128
        __builtin_amdgcn_global_load_lds(wp, ring, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(wp + 512, ring + 256, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(wp + 1024, ring + 512, 16, 0, 0);
        s = 3;
    } else if (GLOBAL) {
        bc = *reinterpret_cast<const f32x4*>(wp);
        b1 = *reinterpret_cast<const f32x4*>(wp + 512);
        b2 = *reinterpret_cast<const f32x4*>(wp + 1024);
        s = 3;
    }
    for (int it = 0; it < iters; ++it) {
        f32x4 a0 = *reinterpret_cast<const f32x4*>(&lds[a_base0]);
        f32x4 a1 = *reinterpret_cast<const f32x4*>(&lds[a_base1]);
        if (AHEAD) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            f32x4 bf = bc;
            if (GLOBAL == 4) {
                // ring slot (c8 + 3) & 3 receives step s; slot c8 & 3 (loaded three steps ago) is consumed now
                __builtin_amdgcn_global_load_lds(wp + (size_t)s * 512, ring + ((c8 + 3) & 3) * 256, 16, 0, 0);
                s = (s + 1 < wsteps) ? s + 1 : 0;
                __builtin_amdgcn_s_waitcnt(0x0f70 | 3);  // vmcnt(3): the DMA issued three steps ago has landed
                bc = *reinterpret_cast<const f32x4*>(ring + (c8 & 3) * 256 + lane * 4);
            } else if (GLOBAL) {
                bf = *reinterpret_cast<const f32x4*>(wp + (size_t)s * 512);
                s = (s + 1 < wsteps) ? s + 1 : 0;
            }
            f32x4 a0n = a0, a1n = a1;
            if (AHEAD) {
                if (c8 < 7) {
                    a0n = *reinterpret_cast<const f32x4*>(&lds[a_base0 + (c8 + 1) * 8]);
                    a1n = *reinterpret_cast<const f32x4*>(&lds[a_base1 + (c8 + 1) * 8]);
                }
            } else if (c8 > 0) {
                a0 = *reinterpret_cast<const f32x4*>(&lds[a_base0 + c8 * 8]);
                a1 = *reinterpret_cast<const f32x4*>(&lds[a_base1 + c8 * 8]);
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bc.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bc.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bc.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bc.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bc.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bc.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bc.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bc.w, acc1, 0, 0, 0);
            if (GLOBAL == 3) {
                bc = b1, b1 = b2, b2 = bf;
            } else if (GLOBAL == 1) {
                bc = bf;
            }
            if (AHEAD) {
                a0 = a0n, a1 = a1n;
                if (GLOBAL && GLOBAL != 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (c8 < 7) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
        }
    }
    float sum = 0.f;
    for (int r = 0; r < 16; ++r) sum += acc0[r] + acc1[r];
    if (sum == 12345.678f) out[blockIdx.x * 256 + tid] = sum;
}

template <int AHEAD, int GLOBAL>
static void run(const char* name, int wgs_per_cu, const float* w, float* out, int cus, int wsteps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = cus * wgs_per_cu, iters = 20000 / wgs_per_cu;
    for (int rep = 0; rep < 2; ++rep) {  // first launch warms the clocks
        hipEventRecord(e0);
        hipLaunchKernelGGL((feed_kernel<AHEAD, GLOBAL>), dim3(grid), dim3(256), 0, 0, w, out, iters, wsteps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double tf = (double)grid * 4 * iters * 64 * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%-58s %d wave(s)/SIMD  %8.2f ms  %6.1f TFLOP/s  %5.1f %%\n", name, wgs_per_cu, ms, tf, 100 * tf / 157.3);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, wsteps = 288;  // 288 steps x 2 KiB = 576 KiB of "weights", L2-resident
    float *w, *out;
    hipMalloc(&w, (size_t)wsteps * 512 * 4);
    hipMemset(w, 0, (size_t)wsteps * 512 * 4);
    hipMalloc(&out, (size_t)cus * 4 * 256 * 4);
    for (int wg = 1; wg <= 4; ++wg) {
        if (wg == 4) continue;  // 34 KiB of LDS per workgroup: three fit comfortably, keep the sweep short
        run<0, 0>("LDS reads at use, free schedule, no global stream", wg, w, out, cus, wsteps);
        run<1, 0>("LDS reads one step ahead, pinned, no global stream", wg, w, out, cus, wsteps);
        run<0, 1>("LDS reads at use, global stream 1 step ahead", wg, w, out, cus, wsteps);
        run<1, 1>("LDS one ahead pinned, global stream 1 step ahead", wg, w, out, cus, wsteps);
        run<1, 3>("LDS one ahead pinned, global stream 3 steps ahead", wg, w, out, cus, wsteps);
        run<0, 4>("LDS reads at use, global stream via LDS-DMA 3 ahead", wg, w, out, cus, wsteps);
    }
    return 0;
}
