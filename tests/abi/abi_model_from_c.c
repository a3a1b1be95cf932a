/* The MODEL-LEVEL entry points of the C ABI driven from plain C -- what a foreign binding (cgo / JNI / N-API stub) would call -- with no
 * Python, no torch, no C++ in the process: the 24 SuperPoint state_dict tensors and a gray uint8 image are built from a counter-based integer
 * hash, packed with gtsfm_sp_pack_weights, uploaded with the HIP runtime's C API, the workspace is sized with gtsfm_sp_workspace_bytes and
 * gtsfm_sp_forward (thirdparty/SuperGluePretrainedNetwork/models/superpoint.py:145-202 + the wrapper's /255,
 * gtsfm/frontend/detector_descriptor/superpoint.py:73-75) runs on a stream this program created. The keypoint count, EVERY keypoint's
 * (x, y), every score (1e-4) and the first descriptor values (1e-4) are compared with the numbers oracle/superpoint_oracle.py wrote into
 * abi_model_expected.h (oracle/make_abi_model_expectation.py builds the same tensors from the same hash in numpy).
 * Built with gcc -std=c99 and run on the GPU box by tests/test_abi_from_c.py (-m gpu). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "abi_model_expected.h"
#include "gtsfm_amd.h"

static uint32_t hash32(uint32_t t, uint32_t i, uint32_t seed) { /* element i of tensor t: murmur3's finaliser over a counter */
    uint32_t x = i * 2654435761u + t * 40503u + seed;
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}

static float unit(uint32_t t, uint32_t i, uint32_t seed) { /* [-0.5, 0.5) on a 2^-24 grid: exact in float32 */
    return (float)(hash32(t, i, seed) >> 8) * (1.0f / 16777216.0f) - 0.5f;
}

#define CHECK_HIP(expr, code)                                   \
    do {                                                        \
        if ((expr) != hipSuccess) {                             \
            fprintf(stderr, "HIP call failed: %s\n", #expr);    \
            return code;                                        \
        }                                                       \
    } while (0)

int main(void) {
    const int H = ABI_MODEL_H, W = ABI_MODEL_W, cap = 1024;
    float* tensors[24];
    const float* tensor_ptrs[24];
    float *packed, *packed_dev = NULL, *xy_dev = NULL, *score_dev = NULL, *desc_dev = NULL;
    unsigned char *img = (unsigned char*)malloc((size_t)H * W), *img_dev = NULL;
    void* ws_dev = NULL;
    int32_t *count_dev = NULL, count = -1, count_raw = -1, *count_raw_dev = NULL;
    float *xy = (float*)malloc(sizeof(float) * cap * 2), *score = (float*)malloc(sizeof(float) * cap), *desc;
    hipStream_t stream = NULL;
    size_t ws_bytes, nfloats;
    double worst_score = 0.0, worst_desc = 0.0;
    int li, i, y, x, rc, bad_xy = 0;

    for (li = 0; li < 12; ++li) { /* conv{1a..4b, Pa, Pb, Da, Db}.weight [cout][cin][k][k] and .bias [cout], checkpoint order */
        const int cout = abi_model_shapes[li][0], cin = abi_model_shapes[li][1], k = abi_model_shapes[li][2];
        const int nw = cout * cin * k * k;
        tensors[2 * li] = (float*)malloc(sizeof(float) * nw);
        tensors[2 * li + 1] = (float*)malloc(sizeof(float) * cout);
        for (i = 0; i < nw; ++i) tensors[2 * li][i] = unit(2 * li, i, ABI_MODEL_SEED) * abi_model_scales[2 * li];
        for (i = 0; i < cout; ++i) tensors[2 * li + 1][i] = unit(2 * li + 1, i, ABI_MODEL_SEED) * abi_model_scales[2 * li + 1];
        tensor_ptrs[2 * li] = tensors[2 * li];
        tensor_ptrs[2 * li + 1] = tensors[2 * li + 1];
    }
    for (y = 0; y < H; ++y)
        for (x = 0; x < W; ++x) {
            const uint32_t cell = hash32(100u, (uint32_t)((y / 8) * 64 + x / 8), ABI_MODEL_SEED) >> 25;
            const uint32_t fine = hash32(101u, (uint32_t)(y * W + x), ABI_MODEL_SEED) >> 26;
            const uint32_t ramp = (uint32_t)((x * 3 + y * 5) >> 2) & 63u;
            img[y * W + x] = (unsigned char)((cell + fine + ramp) & 255u);
        }

    nfloats = gtsfm_sp_packed_weight_floats();
    packed = (float*)malloc(sizeof(float) * nfloats);
    rc = gtsfm_sp_pack_weights(tensor_ptrs, packed);
    if (rc != GTSFM_OK) {
        fprintf(stderr, "gtsfm_sp_pack_weights: %d (%s)\n", rc, gtsfm_last_error());
        return 2;
    }
    ws_bytes = gtsfm_sp_workspace_bytes(1, H, W);
    CHECK_HIP(hipStreamCreate(&stream), 3);
    CHECK_HIP(hipMalloc((void**)&packed_dev, sizeof(float) * nfloats), 3);
    CHECK_HIP(hipMalloc((void**)&img_dev, (size_t)H * W), 3);
    CHECK_HIP(hipMalloc(&ws_dev, ws_bytes), 3);
    CHECK_HIP(hipMalloc((void**)&count_dev, sizeof(int32_t)), 3);
    CHECK_HIP(hipMalloc((void**)&count_raw_dev, sizeof(int32_t)), 3);
    CHECK_HIP(hipMalloc((void**)&xy_dev, sizeof(float) * cap * 2), 3);
    CHECK_HIP(hipMalloc((void**)&score_dev, sizeof(float) * cap), 3);
    CHECK_HIP(hipMalloc((void**)&desc_dev, sizeof(float) * cap * 256), 3);
    CHECK_HIP(hipMemcpy(packed_dev, packed, sizeof(float) * nfloats, hipMemcpyHostToDevice), 4);
    CHECK_HIP(hipMemcpy(img_dev, img, (size_t)H * W, hipMemcpyHostToDevice), 4);

    /* a workspace that is too small is an error code + message, not a crash */
    if (gtsfm_sp_forward(packed_dev, img_dev, 1, 1, H, W, 0.005f, 4, 4, cap, 0, ws_dev, 16, count_dev, count_raw_dev, xy_dev, score_dev, desc_dev, NULL,
                         NULL, (void*)stream) != GTSFM_ERR_WORKSPACE)
        return 5;
    /* SuperPoint.forward with GTSfM's settings: keypoint_threshold 0.005, nms_radius 4, remove_borders 4, max_keypoints -1 (top_k = 0) */
    rc = gtsfm_sp_forward(packed_dev, img_dev, 1, 1, H, W, 0.005f, 4, 4, cap, 0, ws_dev, ws_bytes, count_dev, count_raw_dev, xy_dev, score_dev, desc_dev,
                          NULL, NULL, (void*)stream);
    if (rc != GTSFM_OK) {
        fprintf(stderr, "gtsfm_sp_forward: %d (%s)\n", rc, gtsfm_last_error());
        return 6;
    }
    CHECK_HIP(hipStreamSynchronize(stream), 7);
    CHECK_HIP(hipMemcpy(&count, count_dev, sizeof(int32_t), hipMemcpyDeviceToHost), 8);
    CHECK_HIP(hipMemcpy(&count_raw, count_raw_dev, sizeof(int32_t), hipMemcpyDeviceToHost), 8);
    if (count != ABI_MODEL_K || count_raw != ABI_MODEL_K) {
        fprintf(stderr, "keypoint count %d (raw %d), the oracle has %d\n", count, count_raw, ABI_MODEL_K);
        return 9;
    }
    desc = (float*)malloc(sizeof(float) * count * 256);
    CHECK_HIP(hipMemcpy(xy, xy_dev, sizeof(float) * count * 2, hipMemcpyDeviceToHost), 8);
    CHECK_HIP(hipMemcpy(score, score_dev, sizeof(float) * count, hipMemcpyDeviceToHost), 8);
    CHECK_HIP(hipMemcpy(desc, desc_dev, sizeof(float) * count * 256, hipMemcpyDeviceToHost), 8);
    for (i = 0; i < count; ++i) {
        int d;
        if (xy[2 * i] != (float)abi_model_xy[i][0] || xy[2 * i + 1] != (float)abi_model_xy[i][1]) ++bad_xy; /* bit-exact, row-major order */
        if (fabs((double)score[i] - abi_model_scores[i]) > worst_score) worst_score = fabs((double)score[i] - abi_model_scores[i]);
        for (d = 0; d < 4; ++d)
            if (fabs((double)desc[i * 256 + d] - abi_model_desc_head[i][d]) > worst_desc) worst_desc = fabs((double)desc[i * 256 + d] - abi_model_desc_head[i][d]);
    }
    hipFree(packed_dev), hipFree(img_dev), hipFree(ws_dev), hipFree(count_dev), hipFree(count_raw_dev), hipFree(xy_dev), hipFree(score_dev), hipFree(desc_dev);
    hipStreamDestroy(stream);
    if (bad_xy) {
        fprintf(stderr, "%d of %d keypoints differ from the oracle's\n", bad_xy, count);
        return 10;
    }
    if (worst_score > 1e-4 || worst_desc > 1e-4) {
        fprintf(stderr, "scores differ by %g, descriptors by %g (tolerance 1e-4)\n", worst_score, worst_desc);
        return 11;
    }
    printf("abi_model_from_c OK (%d keypoints identical to the oracle's; max |dscore| %.3g, max |ddescriptor| %.3g)\n", count, worst_score, worst_desc);
    return 0;
}
