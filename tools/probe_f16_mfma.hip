// What v_mfma_f32_32x32x16_f16 does with SUBNORMAL fp16 inputs on this MI355X, what v_cvt_pk_f16_f32 returns for values below fp16's normal
// range, and the sustained rate of the fp16 / bf16 / fp32 matrix pipes under random operands -- the three facts the "f16x2" arithmetic
// (f16x2.h: x = hi + lo in two fp16 pieces, three products per block) rests on.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_f16_mfma.hip -o tools/bin/probe_f16_mfma && tools/bin/probe_f16_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// out[0]: sum over k = 16 of a * b through the MFMA, every A element = a, every B element = b
__global__ void probe_mfma(float a, float b, float* out) {
    h8 A, B;
    for (int i = 0; i < 8; ++i) A[i] = (_Float16)a, B[i] = (_Float16)b;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
__global__ void probe_cvt(const float* x, int n, unsigned* bits, float* back) {
    const int i = threadIdx.x;
    if (2 * i + 1 >= n) return;
    const h2 h = __builtin_convertvector(f2{x[2 * i], x[2 * i + 1]}, h2);
    bits[i] = __builtin_bit_cast(unsigned, h);
    back[2 * i] = (float)h.x, back[2 * i + 1] = (float)h.y;
}

template <int KIND>  // 0: f32 32x32x2, 1: bf16 32x32x16, 2: f16 32x32x16
__global__ __launch_bounds__(256) void burn(float* out, int iters, const float* rnd) {
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
    h8 ha, hb;
    b8 ba, bb;
    for (int i = 0; i < 8; ++i) {
        const float u = rnd[(threadIdx.x * 8 + i) & 2047], v = rnd[(threadIdx.x * 8 + i + 1024) & 2047];
        ha[i] = (_Float16)u, hb[i] = (_Float16)v, ba[i] = (__bf16)u, bb[i] = (__bf16)v;
    }
    const float fa = rnd[threadIdx.x], fb = rnd[256 + threadIdx.x];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c3, 0, 0, 0);
            } else if (KIND == 1) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, c3, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c3, 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 1 << 20);
    struct { float a, b; const char* what; } cases[] = {
        {1.0f, 1.0f, "1 x 1 (expect 16)"},
        {ldexpf(1.f, -20), 1.0f, "subnormal 2^-20 x 1 (expect 16 x 2^-20 = 1.5259e-05; 0 if inputs are flushed)"},
        {ldexpf(1.f, -24), 1024.0f, "smallest subnormal 2^-24 x 1024 (expect 9.7656e-04)"},
        {ldexpf(1.f, -20), ldexpf(1.f, -20), "subnormal x subnormal 2^-40 (expect 1.4552e-11)"},
        {ldexpf(1.5f, -15), 1.0f, "subnormal 1.5 x 2^-15 x 1 (expect 7.3242e-04)"},
    };
    for (auto& c : cases) {
        hipLaunchKernelGGL(probe_mfma, dim3(1), dim3(64), 0, 0, c.a, c.b, out);
        float r;
        hipMemcpy(&r, out, 4, hipMemcpyDeviceToHost);
        printf("mfma_f32_32x32x16_f16: %-90s -> %.6e\n", c.what, r);
    }
    {
        const float xs[8] = {ldexpf(1.f, -14), ldexpf(1.f, -15), ldexpf(1.25f, -20), ldexpf(1.f, -24), ldexpf(1.f, -25), ldexpf(1.1f, -25), 65504.f, 70000.f};
        float* dx;
        unsigned* bits;
        float* back;
        hipMalloc(&dx, sizeof(xs)), hipMalloc(&bits, 16), hipMalloc(&back, 32);
        hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe_cvt, dim3(1), dim3(64), 0, 0, dx, 8, bits, back);
        float hb[8];
        hipMemcpy(hb, back, 32, hipMemcpyDeviceToHost);
        for (int i = 0; i < 8; ++i) printf("v_cvt_pk_f16_f32: %.9e -> %.9e\n", xs[i], hb[i]);
    }
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float h[2048];
    unsigned st = 12345u;
    for (int i = 0; i < 2048; ++i) {
        float v = 0.f;
        for (int k = 0; k < 4; ++k) { st = st * 1664525u + 1013904223u; v += (st >> 8) * (1.0f / 16777216.0f) - 0.5f; }
        h[i] = v * 1.7f;
    }
    float* rnd;
    hipMalloc(&rnd, sizeof(h));
    hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int grid = cus * 2;
    const char* names[3] = {"f32 32x32x2 ", "bf16 32x32x16", "f16 32x32x16 "};
    const double flop[3] = {4096.0, 32768.0, 32768.0};
    const double peak[3] = {157.3, 2516.6, 2516.6};
    for (int rep = 0; rep < 2; ++rep)
        for (int kind = 0; kind < 3; ++kind)
            for (int iters : {4000, 40000}) {
                const int it = kind == 0 ? iters / 2 : iters;
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(burn<0>, dim3(grid), dim3(256), 0, 0, out, it, rnd);
                if (kind == 1) hipLaunchKernelGGL(burn<1>, dim3(grid), dim3(256), 0, 0, out, it, rnd);
                if (kind == 2) hipLaunchKernelGGL(burn<2>, dim3(grid), dim3(256), 0, 0, out, it, rnd);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double tf = (double)grid * 4 * it * 32 * flop[kind] / (ms * 1e-3) / 1e12;
                printf("burn %s %6d iterations: %8.3f ms  %8.1f TFLOP/s  (%.3f of the nominal %.1f)\n", names[kind], it, ms, tf, tf / peak[kind], peak[kind]);
            }
    return 0;
}
