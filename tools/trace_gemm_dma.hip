// Developer tool: per-segment cycle budget of gemm_dma_kernel (LDS-DMA GEMM) at a projection shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGTSFM_TRACE -Igtsfm_amd/csrc -Iinclude tools/trace_gemm_dma.hip -o tools/trace_gemm_dma
#include <cstdarg>
#include <cstdlib>
#include <vector>
#include "../gtsfm_amd/csrc/gemm_dma_kernels.hip"

int gtsfm_cu_count(void) { return 256; }

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
    const int math = argc > 5 ? atoi(argv[5]) : 0;  // 0 exact fp32, 1 bf16x3, 2 f16x2
    float *A, *W, *C, *B, *R;
    hipMalloc(&A, (size_t)M * K * 4);
    hipMalloc(&C, (size_t)M * N * 4);
    hipMalloc(&R, (size_t)M * N * 4);
    hipMalloc(&W, (size_t)N * K * 4);
    hipMalloc(&B, (size_t)(N + 64) * 4);
    std::vector<float> h((size_t)M * K);
    unsigned st = 1;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = ((st >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4f; }
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> hw((size_t)N * K);
    for (auto& v : hw) { st = st * 1664525u + 1013904223u; v = ((st >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.2f; }
    hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemset(B, 0, (size_t)(N + 64) * 4);
    hipMemset(R, 0, (size_t)M * N * 4);
    GemmParams p{};
    p.A = A; p.lda = K; p.wraw = W; p.ldw = K; p.bias = B; p.C = C; p.ldc = N; p.c_coff = 0; p.M = M; p.N = N; p.K = K; p.alpha = 1.0f; p.relu = 0;
    p.res = with_res ? R : nullptr; p.ldres = N;
    p.math = math;
    const size_t nwg = (size_t)((M + 127) / 128 + 7) / 8 * 8 * ((N + 127) / 128);
    const size_t nrec = nwg * 4 * 8;
#ifdef GTSFM_TRACE
    unsigned long long* trace;
    hipMalloc(&trace, nrec * 8);
    hipMemset(trace, 0, nrec * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_trace), &trace, sizeof(trace));
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 100; ++i) launch_gemm_dma(p, 0);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch_gemm_dma(p, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double tf = 2.0 * M * K * N / (ms * 1e-3) / 1e12;
    {  // sampled self-check against a float64 dot product
        std::vector<float> hc((size_t)M * N);
        hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        unsigned s2 = 12345;
        for (int t = 0; t < 512; ++t) {
            s2 = s2 * 1664525u + 1013904223u; const int r = (s2 >> 8) % M;
            s2 = s2 * 1664525u + 1013904223u; const int c = (s2 >> 8) % N;
            double acc = 0;
            for (int k = 0; k < K; ++k) acc += (double)h[(size_t)r * K + k] * (double)hw[(size_t)c * K + k];
            const double d = fabs(acc - (double)hc[(size_t)r * N + c]);
            worst = d > worst ? d : worst;
        }
        printf("self-check: max |error| over 512 samples = %.2e %s\n", worst, worst < 1e-3 ? "ok" : "FAILED");
    }
    printf("math %d: gemm_dma %d x %d -> %d%s: %.3f ms, %.1f TFLOP/s (%.1f %% of 157.3)%s\n", math, M, K, N, with_res ? " +res" : "", ms, tf, 100 * tf / 157.3,
#ifdef GTSFM_TRACE
           " WITH trace stamps");
#else
           "");
    (void)nrec;
    return 0;
#endif
#ifdef GTSFM_TRACE
    std::vector<unsigned long long> t(nrec);
    hipMemcpy(t.data(), trace, nrec * 8, hipMemcpyDeviceToHost);
    double sum[6] = {0}; size_t waves = 0; double stages = 0, tiles = 0;
    for (size_t w = 0; w < nrec / 8; ++w) {
        const unsigned long long* o = &t[w * 8];
        if (!o[6]) continue;
        for (int k = 0; k < 6; ++k) sum[k] += (double)o[k];
        stages += (double)(o[6] * o[7]);
        tiles += (double)o[6];
        ++waves;
    }
    printf("%zu waves, %.1f tiles of 128 x 128 per workgroup, %.0f stages of 32 per tile; 64 MFMAs per stage = 4096 cycles of own pipe time (8192 when shared by 2 waves)\n", waves, tiles / waves, stages / tiles);
    printf("  prologue (first DMA, barrier)         %8.0f cycles per workgroup\n", sum[0] / waves);
    printf("  DMA issue (+ init / aux loads)        %8.0f cycles per stage\n", sum[1] / stages);
    printf("  LDS reads + MFMAs                     %8.0f cycles per stage\n", sum[2] / stages);
    printf("  vmcnt(0) + barrier                    %8.0f cycles per stage\n", sum[3] / stages);
    printf("  epilogue                              %8.0f cycles per tile\n", sum[4] / tiles);
    printf("  wave lifetime                         %8.0f cycles per tile (shares: prologue %.1f %%, dma %.1f %%, mfma %.1f %%, wait %.1f %%, epilogue %.1f %%)\n", sum[5] / tiles,
           100 * sum[0] / sum[5], 100 * sum[1] / sum[5], 100 * sum[2] / sum[5], 100 * sum[3] / sum[5], 100 * sum[4] / sum[5]);
    return 0;
#endif
}
