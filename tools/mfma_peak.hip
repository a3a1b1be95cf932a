// Sustained fp32 MFMA rate of this MI355X, measured rather than assumed: every SIMD of every CU issues
// v_mfma_f32_32x32x2_f32 back to back from registers (no memory traffic), for launches of ~1 ms up to ~1 s.
// 157.3 TFLOP/s = 65 536 flop/cycle x 2.4 GHz assumes the boost clock; the number this prints is what the matrix pipe
// delivers under the power the kernels of this repo actually draw, i.e. the practical ceiling of roofline.frac.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_burn(float* out, int iters, float a, float b, const float* rnd) {
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
    // constant operands toggle few datapath bits (the chip then sustains a higher clock inside its power budget);
    // rnd != nullptr feeds per-lane N(0,1) operands instead -- what real activations look like
    const float av = rnd ? rnd[threadIdx.x] : a + threadIdx.x * 1e-9f, bv = rnd ? rnd[256 + threadIdx.x] : b;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {  // 32 MFMAs per iteration whatever the number of independent accumulator chains
            if (CHAINS == 4) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c3, 0, 0, 0);
            } else if (CHAINS == 2) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c1, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c0, 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;  // keep the chain alive
}

int main(int argc, char** argv) {
    const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 2;
    const int chains = argc > 2 ? atoi(argv[2]) : 4;  // independent accumulators a wave alternates over (4, 2 or 1)
    const int random_data = argc > 3 ? atoi(argv[3]) : 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out;
    hipMalloc(&out, (size_t)cus * waves_per_simd * 256 * sizeof(float));
    float* rnd = nullptr;
    if (random_data) {
        float h[512];
        unsigned st = 12345u;
        for (int i = 0; i < 512; ++i) {  // sum of 4 uniforms, roughly normal, sign-mixed
            float v = 0.f;
            for (int k = 0; k < 4; ++k) { st = st * 1664525u + 1013904223u; v += (st >> 8) * (1.0f / 16777216.0f) - 0.5f; }
            h[i] = v * 1.7f;
        }
        hipMalloc(&rnd, sizeof(h));
        hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("device %s, %d CUs, clockRate %d kHz, %d wave(s) per SIMD, %d accumulator chain(s) per wave, %s operands\n", prop.name, cus, prop.clockRate, waves_per_simd, chains, random_data ? "random" : "constant");
    const int grid = cus * waves_per_simd;  // 4 waves per workgroup = one per SIMD
    const int iters_list[] = {2000, 2000, 20000, 200000, 2000};
    auto kernel = chains == 4 ? mfma_burn<4> : chains == 2 ? mfma_burn<2> : mfma_burn<1>;
    for (int iters : iters_list) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1e-6f, rnd);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 * iters * 32 * 4096.0;
        const double tf = flops / (ms * 1e-3) / 1e12;
        printf("iters %8d  %10.3f ms  %7.2f TFLOP/s  -> effective matrix clock %.3f GHz (%.1f %% of 157.3)\n", iters, ms, tf,
               tf * 1e12 / (65536.0 * cus / 256) / 1e9, 100.0 * tf / 157.3);
        fflush(stdout);
    }
    return 0;
}
