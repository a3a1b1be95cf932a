// Developer tool: per-segment cycle budget of gemm_mfma_kernel at a projection shape (default 131072 x 256 -> 768).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGTSFM_TRACE -Igtsfm_amd/csrc -Iinclude tools/trace_gemm.hip -o tools/trace_gemm
#include <cstdarg>
#include <cstdlib>
#include <vector>
#include "../gtsfm_amd/csrc/gemm_mfma_kernels.hip"

// the register-staged kernel alone: no LDS-DMA dispatch, packer restated
bool gemm_uses_dma(int, int) { return false; }
int launch_gemm_dma(const GemmParams&, hipStream_t) { return GTSFM_ERR_INVALID; }
size_t packed_linear_floats(int k, int n) { return (size_t)ceil_div(n, 64) * (k / 8) * MT_PACK_STEP_FLOATS; }

void gtsfm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 131072, K = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 768;
    const int with_res = argc > 4 ? atoi(argv[4]) : 0;
    float *A, *W, *C, *B, *R;
    hipMalloc(&A, (size_t)M * K * 4);
    hipMalloc(&C, (size_t)M * N * 4);
    hipMalloc(&R, (size_t)M * N * 4);
    const size_t wf = packed_linear_floats(K, N);
    hipMalloc(&W, wf * 4);
    hipMalloc(&B, (size_t)(N + 64) * 4);
    std::vector<float> h((size_t)M * K);
    unsigned st = 1;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = ((st >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4f; }
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hw(wf);
    for (auto& v : hw) { st = st * 1664525u + 1013904223u; v = ((st >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.2f; }
    hipMemcpy(W, hw.data(), wf * 4, hipMemcpyHostToDevice);
    hipMemset(B, 0, (size_t)(N + 64) * 4);
    hipMemset(R, 0, (size_t)M * N * 4);
    GemmParams p{};
    p.A = A; p.lda = K; p.wpack = W; p.bias = B; p.C = C; p.ldc = N; p.c_coff = 0; p.M = M; p.N = N; p.K = K; p.alpha = 1.0f; p.relu = 0;
    p.res = with_res ? R : nullptr; p.ldres = N;
    const int mtiles = (M + 127) / 128;
    unsigned long long* trace;
    const size_t nrec = (size_t)mtiles * ((N + 127) / 128) * 4 * 8;
    hipMalloc(&trace, nrec * 8);
    hipMemset(trace, 0, nrec * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace), &trace, sizeof(trace));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 100; ++i) launch_gemm(p, 0);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch_gemm(p, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double tf = 2.0 * M * K * N / (ms * 1e-3) / 1e12;
    printf("gemm %d x %d -> %d%s: %.3f ms, %.1f TFLOP/s (%.1f %% of 157.3) WITH trace stamps\n", M, K, N, with_res ? " +res" : "", ms, tf, 100 * tf / 157.3);
    std::vector<unsigned long long> t(nrec);
    hipMemcpy(t.data(), trace, nrec * 8, hipMemcpyDeviceToHost);
    double sum[6] = {0}; size_t waves = 0; double blocks = 0, chunks = 0;
    for (size_t w = 0; w < nrec / 8; ++w) {
        const unsigned long long* o = &t[w * 8];
        if (!o[6]) continue;
        for (int k = 0; k < 6; ++k) sum[k] += (double)o[k];
        blocks += (double)o[6]; chunks += (double)(o[6] * o[7]);
        ++waves;
    }
    const double nfirst = blocks, nother = chunks - blocks;
    printf("%zu waves, %.0f column blocks x %.0f chunks each; 128 MFMAs per chunk = 8192 cycles of own pipe time (16384 when shared by 2 waves)\n", waves, blocks / waves, chunks / blocks);
    printf("  MFMA chunk, first of a column block  %8.0f cycles per chunk\n", sum[1] / nfirst);
    if (nother > 0) printf("  MFMA chunk, later ones               %8.0f cycles per chunk\n", sum[0] / nother);
    printf("  epilogue (per column block)          %8.0f\n", sum[2] / blocks);
    printf("  LDS staging + barrier (per chunk)    %8.0f\n", sum[3] / chunks);
    printf("  accumulator init (per column block)  %8.0f\n", sum[4] / blocks);
    printf("  wave lifetime per chunk              %8.0f  (shares: mfma %.1f %%, epilogue %.1f %%, staging+barrier %.1f %%, init %.1f %%)\n", sum[5] / chunks,
           100 * (sum[0] + sum[1]) / sum[5], 100 * sum[2] / sum[5], 100 * sum[3] / sum[5], 100 * sum[4] / sum[5]);
    return 0;
}
