// 3x3 convolution launcher and the host-side weight packers (dense_kernels.hip, mfma_tiles.h).
#pragma once

#include "common.h"

struct ConvParams {
    const float* in;   // NHWC activations [B][H][W][in_stride], channels in_coff .. in_coff+Cin-1 are read
    int in_stride, in_coff;
    float* out;        // NHWC [B][Ho][Wo][out_stride], channels out_coff .. out_coff+Cout-1 are written
    int out_stride, out_coff;
    const float* wpack;  // packed weights (pack_conv3x3_weights)
    const float* bias;   // [ceil(Cout/64)*64]
    int B, H, W, Cin, Cout;
    int relu, pool;      // pool: fused 2x2/stride-2 max-pool, Ho = H/2, Wo = W/2 (floor)
    int tiles_x, tiles_y;  // filled by the launcher
    // optional fused first layer: when img != null the input activation is relu(conv1a(img)) computed on the fly
    const void* img;   // [B][H][W] gray image, fp32 or uint8
    int img_is_u8;
    const float* w1a;  // conv1a weights [9 taps][64]
    const float* b1a;  // conv1a bias [64]
};

int launch_conv3x3(const ConvParams& p, hipStream_t stream);

size_t packed_conv3x3_floats(int cin, int cout);
size_t packed_linear_floats(int k, int n);
void pack_conv3x3_weights(const float* w, int cin, int cout, float* out);
void pack_linear_weights(const float* w, int k_real, int k, int n, float* out);
