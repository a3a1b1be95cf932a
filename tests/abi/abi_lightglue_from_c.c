/* The MATCHER side of the C ABI driven from plain C -- no Python, no torch, no C++ in the process: what a foreign binding of
 * gtsfm/frontend/matcher/lightglue_matcher.py:75-112 would do. A two-layer LightGlue is built from a counter-based integer hash directly in the
 * logical form the blob packer takes (include/gtsfm_amd.h: q | k | v head-major Wqkv, fused to_qk | to_v, output projections folded into ffn.0),
 * packed with gtsfm_blob_floats / gtsfm_pack_blob and uploaded with the HIP runtime's C API; two keypoint / descriptor sets come from the same hash
 * (200 of image 1's 280 keypoints are shifted, noised copies of image 0's); gtsfm_match_desc_ints / gtsfm_match_build_desc make the batch
 * descriptor, gtsfm_lg_workspace_bytes sizes the workspace, gtsfm_lg_forward runs on a stream this program created. Every keypoint's match index
 * (identical) and matching score (1e-4) on both sides is compared with what oracle/lightglue_oracle.py computed for the upstream-layout state_dict
 * whose load-time preparation gives these logical entries bit for bit (oracle/make_abi_lightglue_expectation.py -> abi_lightglue_expected.h).
 * Built with gcc -std=c99 and run on the GPU box by tests/test_abi_from_c.py (-m gpu). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "abi_lightglue_expected.h"
#include "gtsfm_amd.h"

static uint32_t hash32(uint32_t t, uint32_t i, uint32_t seed) {
    uint32_t x = i * 2654435761u + t * 40503u + seed;
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}

static float unit(uint32_t t, uint32_t i) { /* [-0.5, 0.5) on a 2^-24 grid: exact in float32 */
    return (float)(hash32(t, i, ABI_LG_SEED) >> 8) * (1.0f / 16777216.0f) - 0.5f;
}

#define CHECK_HIP(expr, code)                                \
    do {                                                     \
        if ((expr) != hipSuccess) {                          \
            fprintf(stderr, "HIP call failed: %s\n", #expr); \
            return code;                                     \
        }                                                    \
    } while (0)
#define CHECK_ABI(expr, code)                                                  \
    do {                                                                       \
        int rc_ = (expr);                                                      \
        if (rc_ != GTSFM_OK) {                                                 \
            fprintf(stderr, "%s: %d (%s)\n", #expr, rc_, gtsfm_last_error()); \
            return code;                                                       \
        }                                                                      \
    } while (0)

int main(void) {
    enum { N0 = ABI_LG_N0, N1 = ABI_LG_N1, T = ABI_LG_N0 + ABI_LG_N1, E = ABI_LG_ENTRIES };
    int32_t kinds[E], en[E], ek[E];
    float* w[E];
    float* b[E];
    const float* wp[E];
    const float* bp[E];
    const int32_t n0[1] = {N0}, n1[1] = {N1}, hw[4] = {ABI_LG_H, ABI_LG_W, ABI_LG_H, ABI_LG_W};
    float *k = (float*)malloc(sizeof(float) * T * 2), *d = (float*)malloc(sizeof(float) * T * 256);
    float *blob, *blob_dev = NULL, *k_dev = NULL, *d_dev = NULL, *ms_dev = NULL, *ms = (float*)malloc(sizeof(float) * T);
    int32_t *desc, *desc_dev = NULL, *m_dev = NULL, *m = (int32_t*)malloc(sizeof(int32_t) * T);
    void* ws_dev = NULL;
    hipStream_t stream = NULL;
    size_t blob_floats, desc_ints, ws_bytes;
    double worst = 0.0;
    int t, i, j, c, bad = 0, matches = 0;

    for (t = 0; t < E; ++t) { /* the blob entries in the packer's order: kind 0 = linear W [n][k] + bias [n], kind 1 = n raw floats */
        const int kind = abi_lg_entry_shape[t][0], n = abi_lg_entry_shape[t][1], kk = abi_lg_entry_shape[t][2];
        const float ws = abi_lg_entry_scale[t][0], bs = abi_lg_entry_scale[t][1], off = abi_lg_entry_scale[t][2];
        kinds[t] = kind, en[t] = n, ek[t] = kk;
        b[t] = NULL;
        if (kind == 0) {
            w[t] = (float*)malloc(sizeof(float) * n * kk);
            b[t] = (float*)malloc(sizeof(float) * n);
            for (i = 0; i < n * kk; ++i) w[t][i] = unit(2 * t, i) * ws;
            for (i = 0; i < n; ++i) b[t][i] = unit(2 * t + 1, i) * bs;
        } else {
            w[t] = (float*)malloc(sizeof(float) * n);
            for (i = 0; i < n; ++i) w[t][i] = off + unit(2 * t, i) * ws;
        }
        wp[t] = w[t], bp[t] = b[t];
    }
    /* features, token-major: image 0's N0 rows, then image 1's N1 rows. Image 1's first MATCHED keypoints are keypoint (7 j + 3) mod N0 of image 0,
     * shifted by (7, -5), with descriptor noise; the rest are unrelated */
    for (i = 0; i < N0; ++i) {
        k[2 * i] = floorf((unit(210, i) + 0.5f) * 600.0f) + 4.0f;
        k[2 * i + 1] = floorf((unit(211, i) + 0.5f) * 430.0f) + 10.0f;
        for (c = 0; c < 256; ++c) d[i * 256 + c] = unit(200, i * 256 + c) * 0.125f;
    }
    for (j = 0; j < N1; ++j) {
        float* kj = k + 2 * (N0 + j);
        float* dj = d + 256 * (N0 + j);
        if (j < ABI_LG_MATCHED) {
            const int src = (7 * j + 3) % N0;
            kj[0] = k[2 * src] + 7.0f, kj[1] = k[2 * src + 1] + -5.0f;
            for (c = 0; c < 256; ++c) {
                const float noise = unit(201, j * 256 + c) * 0.01f;
                dj[c] = d[src * 256 + c] + noise;
            }
        } else {
            kj[0] = floorf((unit(212, j) + 0.5f) * 600.0f) + 4.0f;
            kj[1] = floorf((unit(213, j) + 0.5f) * 430.0f) + 10.0f;
            for (c = 0; c < 256; ++c) dj[c] = unit(202, j * 256 + c) * 0.125f;
        }
    }

    blob_floats = gtsfm_blob_floats(E, kinds, en, ek);
    blob = (float*)malloc(sizeof(float) * blob_floats);
    CHECK_ABI(gtsfm_pack_blob(E, kinds, en, ek, wp, bp, blob), 2);
    desc_ints = gtsfm_match_desc_ints(0, 1, n0, n1);
    desc = (int32_t*)malloc(sizeof(int32_t) * desc_ints);
    CHECK_ABI(gtsfm_match_build_desc(0, 1, n0, n1, hw, desc), 2);
    ws_bytes = gtsfm_lg_workspace_bytes(1, n0, n1);

    CHECK_HIP(hipStreamCreate(&stream), 3);
    CHECK_HIP(hipMalloc((void**)&blob_dev, sizeof(float) * blob_floats), 3);
    CHECK_HIP(hipMalloc((void**)&desc_dev, sizeof(int32_t) * desc_ints), 3);
    CHECK_HIP(hipMalloc((void**)&k_dev, sizeof(float) * T * 2), 3);
    CHECK_HIP(hipMalloc((void**)&d_dev, sizeof(float) * T * 256), 3);
    CHECK_HIP(hipMalloc((void**)&m_dev, sizeof(int32_t) * T), 3);
    CHECK_HIP(hipMalloc((void**)&ms_dev, sizeof(float) * T), 3);
    CHECK_HIP(hipMalloc(&ws_dev, ws_bytes), 3);
    CHECK_HIP(hipMemcpy(blob_dev, blob, sizeof(float) * blob_floats, hipMemcpyHostToDevice), 4);
    CHECK_HIP(hipMemcpy(desc_dev, desc, sizeof(int32_t) * desc_ints, hipMemcpyHostToDevice), 4);
    CHECK_HIP(hipMemcpy(k_dev, k, sizeof(float) * T * 2, hipMemcpyHostToDevice), 4);
    CHECK_HIP(hipMemcpy(d_dev, d, sizeof(float) * T * 256, hipMemcpyHostToDevice), 4);

    /* a workspace that is too small is an error code + message, not a crash */
    if (gtsfm_lg_forward(blob_dev, ABI_LG_LAYERS, abi_lg_match_bias, abi_lg_conf_bias, 1, n0, n1, desc_dev, k_dev, d_dev, -1.0f, -1.0f, 0.1f, 0x7fffffff, ws_dev,
                         1024, m_dev, ms_dev, NULL, (void*)stream) != GTSFM_ERR_WORKSPACE)
        return 5;
    /* LightGlue.forward on one pair: adaptive depth and width off (the oracle run behind the expected numbers has them off too), filter threshold 0.1 */
    CHECK_ABI(gtsfm_lg_forward(blob_dev, ABI_LG_LAYERS, abi_lg_match_bias, abi_lg_conf_bias, 1, n0, n1, desc_dev, k_dev, d_dev, -1.0f, -1.0f, 0.1f, 0x7fffffff,
                               ws_dev, ws_bytes, m_dev, ms_dev, NULL, (void*)stream), 6);
    CHECK_HIP(hipStreamSynchronize(stream), 7);
    CHECK_HIP(hipMemcpy(m, m_dev, sizeof(int32_t) * T, hipMemcpyDeviceToHost), 8);
    CHECK_HIP(hipMemcpy(ms, ms_dev, sizeof(float) * T, hipMemcpyDeviceToHost), 8);
    for (i = 0; i < T; ++i) {
        const int want_m = i < N0 ? abi_lg_matches0[i] : abi_lg_matches1[i - N0];
        const double want_s = i < N0 ? abi_lg_scores0[i] : abi_lg_scores1[i - N0];
        if (m[i] != want_m) ++bad;
        if (i < N0 && m[i] >= 0) ++matches;
        if (fabs((double)ms[i] - want_s) > worst) worst = fabs((double)ms[i] - want_s);
    }
    hipFree(blob_dev), hipFree(desc_dev), hipFree(k_dev), hipFree(d_dev), hipFree(m_dev), hipFree(ms_dev), hipFree(ws_dev);
    hipStreamDestroy(stream);
    if (bad) {
        fprintf(stderr, "%d of %d match indices differ from the oracle's\n", bad, T);
        return 9;
    }
    if (worst > 1e-4) {
        fprintf(stderr, "matching scores differ by %g (tolerance 1e-4)\n", worst);
        return 10;
    }
    printf("abi_lightglue_from_c OK (%d matches, all %d match indices identical to the oracle's; max |dscore| %.3g)\n", matches, T, worst);
    return 0;
}
