/* A DEVICE entry point of the C ABI driven from plain C: device memory and the stream come from the HIP runtime's C API, no
 * Python, no torch, no C++. gtsfm_sp_softmax_d2s (softmax over 65 logits per cell, dustbin dropped, depth-to-space 8x8:
 * thirdparty/SuperGluePretrainedNetwork/models/superpoint.py:163-166) is checked against the same arithmetic in C.
 * Built with gcc -std=c99 and run on the GPU box by tests/test_abi_from_c.py (-m gpu). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "gtsfm_amd.h"

#define HC 6
#define WC 9
#define LD 68

int main(void) {
    const int n_in = HC * WC * LD, n_out = 8 * HC * 8 * WC;
    float *logits = (float*)malloc(sizeof(float) * n_in), *got = (float*)malloc(sizeof(float) * n_out);
    float *logits_dev = NULL, *scores_dev = NULL;
    hipStream_t stream = NULL;
    unsigned state = 12345u;
    double worst = 0.0;
    int i, y, x, c, rc;
    for (i = 0; i < n_in; ++i) {
        state = state * 1664525u + 1013904223u;
        logits[i] = (float)((state >> 8) & 0xffff) / 65535.0f * 12.0f - 6.0f;
    }
    if (hipMalloc((void**)&logits_dev, sizeof(float) * n_in) != hipSuccess) return 2;
    if (hipMalloc((void**)&scores_dev, sizeof(float) * n_out) != hipSuccess) return 3;
    if (hipStreamCreate(&stream) != hipSuccess) return 4;
    if (hipMemcpy(logits_dev, logits, sizeof(float) * n_in, hipMemcpyHostToDevice) != hipSuccess) return 5;
    rc = gtsfm_sp_softmax_d2s(logits_dev, LD, 1, HC, WC, scores_dev, (void*)stream);
    if (rc != GTSFM_OK) {
        fprintf(stderr, "gtsfm_sp_softmax_d2s: %d (%s)\n", rc, gtsfm_last_error());
        return 6;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return 7;
    if (hipMemcpy(got, scores_dev, sizeof(float) * n_out, hipMemcpyDeviceToHost) != hipSuccess) return 8;
    for (y = 0; y < HC; ++y)
        for (x = 0; x < WC; ++x) {
            const float* cell = logits + (y * WC + x) * LD;
            float mx = cell[0];
            double sum = 0.0;
            for (c = 1; c < 65; ++c) mx = cell[c] > mx ? cell[c] : mx;
            for (c = 0; c < 65; ++c) sum += exp((double)cell[c] - mx);
            for (c = 0; c < 64; ++c) { /* channel c -> pixel (8 y + c / 8, 8 x + c % 8) */
                const double want = exp((double)cell[c] - mx) / sum;
                const double diff = fabs(want - (double)got[(8 * y + c / 8) * (8 * WC) + 8 * x + c % 8]);
                worst = diff > worst ? diff : worst;
            }
        }
    /* a null device pointer is an argument error (code + message), not a crash */
    if (gtsfm_sp_softmax_d2s(NULL, LD, 1, HC, WC, scores_dev, (void*)stream) != GTSFM_ERR_INVALID) return 9;
    hipFree(logits_dev);
    hipFree(scores_dev);
    hipStreamDestroy(stream);
    if (worst > 1e-6) {
        fprintf(stderr, "softmax_d2s differs from the C restatement by %g\n", worst);
        return 10;
    }
    printf("abi_device_from_c OK (max |diff| %.3g over %d pixels)\n", worst, n_out);
    return 0;
}
