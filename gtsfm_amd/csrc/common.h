// Shared helpers for the gfx950 kernels of the GTSfM deep front-end (SuperPoint / SuperGlue / LightGlue).
// CDNA4 only: 64-wide wavefronts, fp32-input MFMA (v_mfma_f32_32x32x2_f32), 160 KiB LDS per CU.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define GTSFM_OK 0
#define GTSFM_ERR_INVALID -1
#define GTSFM_ERR_HIP -2
#define GTSFM_ERR_WORKSPACE -3

void gtsfm_set_error(const char* fmt, ...);
// Compute units of the current device (hipDeviceProp_t::multiProcessorCount, read once per process; 256 on an MI355X and the
// fallback when the query fails, e.g. in a build container without a GPU). Launch-geometry thresholds scale with it.
int gtsfm_cu_count(void);

#define GTSFM_CHECK_ARG(cond, ...)        \
    do {                                  \
        if (!(cond)) {                    \
            gtsfm_set_error(__VA_ARGS__); \
            return GTSFM_ERR_INVALID;     \
        }                                 \
    } while (0)

#define GTSFM_CHECK_LAUNCH(name)                                                              \
    do {                                                                                      \
        hipError_t e_ = hipGetLastError();                                                    \
        if (e_ != hipSuccess) {                                                               \
            gtsfm_set_error("kernel launch failed (%s): %s", name, hipGetErrorString(e_));    \
            return GTSFM_ERR_HIP;                                                             \
        }                                                                                     \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__host__ __device__ static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Wave-wide reductions over all 64 lanes (result valid in every lane).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
