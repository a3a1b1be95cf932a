// Input step in front of SuperPoint on the device: the loader's INTER_CUBIC downsample
// (gtsfm/loader/loader_base.py:160-200 -> gtsfm/utils/images.py:102-129, cv.resize) and the wrapper's RGB -> gray
// conversion (gtsfm/frontend/detector_descriptor/superpoint.py:73 -> gtsfm/utils/images.py:15-42, cv.cvtColor), both on
// uint8 with OpenCV's 8-bit fixed-point arithmetic (scalar code paths of imgproc/src/resize.cpp and color_rgb.simd.hpp;
// PARITY UNPINNED: cv2 is absent here, oracle/imageprep_oracle.py restates the same published algorithms).
// Integer arithmetic only on the device: the float32 tap weights are computed once on the host (gtsfm_prep_cubic_taps).
// HBM-bound byte work: one thread per output pixel, 16 source bytes per channel each (L2-resident source rows).

#include <math.h>

#include "../../include/gtsfm_amd.h"
#include "common.h"

#define PREP_COEF_BITS 11

__global__ __launch_bounds__(256) void prep_rgb_to_gray_kernel(const uint8_t* __restrict__ rgb, int npix, int channels, uint8_t* __restrict__ gray) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const uint8_t* p = rgb + (size_t)i * channels;
        gray[i] = (uint8_t)(((int)p[0] * 9798 + (int)p[1] * 19235 + (int)p[2] * 3735 + (1 << 14)) >> 15);
    }
}

__global__ __launch_bounds__(256) void prep_resize_cubic_kernel(const uint8_t* __restrict__ src, int sh, int sw, int channels,
                                                                const int* __restrict__ xofs, const short* __restrict__ xw,
                                                                const int* __restrict__ yofs, const short* __restrict__ yw,
                                                                uint8_t* __restrict__ dst, int dh, int dw) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    const int sx = xofs[x], sy = yofs[y];
    int cx[4], wx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        cx[t] = min(max(sx - 1 + t, 0), sw - 1) * channels;  // replicate border
        wx[t] = xw[4 * x + t];
    }
    for (int c = 0; c < channels; ++c) {
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint8_t* row = src + (size_t)min(max(sy - 1 + k, 0), sh - 1) * sw * channels + c;
            const int hsum = (int)row[cx[0]] * wx[0] + (int)row[cx[1]] * wx[1] + (int)row[cx[2]] * wx[2] + (int)row[cx[3]] * wx[3];
            acc += hsum * (int)yw[4 * y + k];
        }
        const int v = (acc + (1 << (2 * PREP_COEF_BITS - 1))) >> (2 * PREP_COEF_BITS);
        dst[((size_t)y * dw + x) * channels + c] = (uint8_t)min(max(v, 0), 255);
    }
}

extern "C" int gtsfm_prep_cubic_taps(int dst_size, int src_size, int32_t* first_src_host, int16_t* weights_host) {
    GTSFM_CHECK_ARG(dst_size > 0 && src_size > 0 && first_src_host && weights_host, "prep_cubic_taps: bad arguments");
    const double scale = (double)src_size / (double)dst_size;
    const float A = -0.75f;
    for (int d = 0; d < dst_size; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        const int s = (int)floorf(f);
        const float x = f - (float)s;
        float c[4];
        c[0] = ((A * (x + 1.0f) - 5.0f * A) * (x + 1.0f) + 8.0f * A) * (x + 1.0f) - 4.0f * A;
        c[1] = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
        c[2] = ((A + 2.0f) * (1.0f - x) - (A + 3.0f)) * (1.0f - x) * (1.0f - x) + 1.0f;
        c[3] = 1.0f - c[0] - c[1] - c[2];
        first_src_host[d] = s;
        for (int t = 0; t < 4; ++t) {
            long v = lrintf(c[t] * (float)(1 << PREP_COEF_BITS));  // cvRound: round half to even
            weights_host[4 * d + t] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
        }
    }
    return GTSFM_OK;
}

extern "C" int gtsfm_prep_rgb_to_gray_u8(const uint8_t* rgb_dev, int height, int width, int channels, uint8_t* gray_dev, void* stream) {
    GTSFM_CHECK_ARG(rgb_dev && gray_dev, "prep_rgb_to_gray: null pointer");
    GTSFM_CHECK_ARG(height >= 0 && width >= 0 && (channels == 3 || channels == 4), "prep_rgb_to_gray: needs 3 or 4 channels");
    const long long npix = (long long)height * width;
    if (npix == 0) return GTSFM_OK;
    GTSFM_CHECK_ARG(npix < (1ll << 31), "prep_rgb_to_gray: image too large");
    const int blocks = (int)((npix + 255) / 256 < 8192 ? (npix + 255) / 256 : 8192);
    hipLaunchKernelGGL(prep_rgb_to_gray_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rgb_dev, (int)npix, channels, gray_dev);
    GTSFM_CHECK_LAUNCH("prep_rgb_to_gray_kernel");
    return GTSFM_OK;
}

extern "C" int gtsfm_prep_resize_cubic_u8(const uint8_t* src_dev, int src_h, int src_w, int channels, const int32_t* xofs_dev, const int16_t* xw_dev,
                                          const int32_t* yofs_dev, const int16_t* yw_dev, uint8_t* dst_dev, int dst_h, int dst_w, void* stream) {
    GTSFM_CHECK_ARG(src_dev && dst_dev && xofs_dev && xw_dev && yofs_dev && yw_dev, "prep_resize_cubic: null pointer");
    GTSFM_CHECK_ARG(src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0 && channels >= 1 && channels <= 4, "prep_resize_cubic: bad shape");
    hipLaunchKernelGGL(prep_resize_cubic_kernel, dim3(ceil_div(dst_w, 256), dst_h), dim3(256), 0, (hipStream_t)stream, src_dev, src_h, src_w, channels,
                       xofs_dev, (const short*)xw_dev, yofs_dev, (const short*)yw_dev, dst_dev, dst_h, dst_w);
    GTSFM_CHECK_LAUNCH("prep_resize_cubic_kernel");
    return GTSFM_OK;
}
