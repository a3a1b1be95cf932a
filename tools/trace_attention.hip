// Developer tool: per-segment cycle budget of attention_mfma_kernel at the workload shape (64 sequences x 4 heads,
// N = 2048). Builds the kernel source with -DGTSFM_TRACE; no part of this reaches libgtsfm_amd.so.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DGTSFM_TRACE -Igtsfm_amd/csrc -Iinclude tools/trace_attention.hip -o tools/trace_attention
#include <cstdarg>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../gtsfm_amd/csrc/attention_kernels.hip"

void gtsfm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}

int main(int argc, char** argv) {
    const int nseq = argc > 1 ? atoi(argv[1]) : 64, n = argc > 2 ? atoi(argv[2]) : 2048, heads = 4;
    const size_t rows = (size_t)nseq * n;
    float *qkv, *out;
    hipMalloc(&qkv, rows * 768 * sizeof(float));
    hipMalloc(&out, rows * 256 * sizeof(float));
    std::vector<float> h(rows * 768);
    unsigned st = 1;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = ((st >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4f; }
    hipMemcpy(qkv, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    std::vector<AttnProblem> pr(nseq);
    std::vector<int> cnt(nseq, n);
    for (int s = 0; s < nseq; ++s) pr[s] = {s * n, s, s * n, s};
    AttnProblem* dpr; int* dcnt;
    hipMalloc(&dpr, nseq * sizeof(AttnProblem));
    hipMalloc(&dcnt, nseq * sizeof(int));
    hipMemcpy(dpr, pr.data(), nseq * sizeof(AttnProblem), hipMemcpyHostToDevice);
    hipMemcpy(dcnt, cnt.data(), nseq * sizeof(int), hipMemcpyHostToDevice);
    const int qtiles = (n + 127) / 128, wgs = ((heads * nseq + 7) / 8) * 8 * qtiles;
    unsigned long long* trace;
    hipMalloc(&trace, (size_t)wgs * 4 * 8 * sizeof(unsigned long long));
    hipMemset(trace, 0, (size_t)wgs * 4 * 8 * sizeof(unsigned long long));
    hipMemcpyToSymbol(HIP_SYMBOL(g_attn_trace), &trace, sizeof(trace));
    AttnParams p{qkv, 768, qkv + 256, 768, qkv + 512, 768, out, 256, dpr, dcnt, 0.125f, heads, 0, 0};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) launch_attention(p, nseq, n, 0);  // warm the clocks
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) launch_attention(p, nseq, n, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double tf = 1024.0 * n * n * nseq / (ms * 1e-3) / 1e12;
    printf("attention %d seq x %d: %.3f ms per launch, %.1f TFLOP/s (%.1f %% of 157.3) WITH trace stamps\n", nseq, n, ms, tf, 100 * tf / 157.3);
    std::vector<unsigned long long> t((size_t)wgs * 4 * 8);
    hipMemcpy(t.data(), trace, t.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    const char* names[5] = {"S = K Q^T issue (64 MFMA)", "softmax (VALU)", "O += V^T P^T issue (64 MFMA)", "barrier 1 wait", "tile store + barrier 2"};
    double sum[6] = {0, 0, 0, 0, 0, 0}, tiles = 0;
    size_t waves = 0;
    std::vector<double> per_tile_total;
    for (size_t w = 0; w < (size_t)wgs * 4; ++w) {
        const unsigned long long* o = &t[w * 8];
        if (!o[7]) continue;
        for (int k = 0; k < 6; ++k) sum[k] += (double)o[k];
        tiles += (double)o[7];
        per_tile_total.push_back((double)o[5] / o[7]);
        ++waves;
    }
    printf("%zu waves traced, %.0f key tiles each; cycles per tile per wave (mean):\n", waves, tiles / waves);
    double acc = 0;
    for (int k = 0; k < 5; ++k) { printf("  %-32s %8.0f  (%.1f %%)\n", names[k], sum[k] / tiles, 100 * sum[k] / sum[5]); acc += sum[k]; }
    printf("  %-32s %8.0f  (%.1f %%)\n", "prologue + epilogue (per tile)", (sum[5] - acc) / tiles, 100 * (sum[5] - acc) / sum[5]);
    printf("  %-32s %8.0f   [own MFMA time 8192; x3 waves per SIMD = 24576 at 100 %% pipe use]\n", "wave lifetime per tile", sum[5] / tiles);
    std::sort(per_tile_total.begin(), per_tile_total.end());
    printf("  lifetime per tile: p10 %.0f  p50 %.0f  p90 %.0f\n", per_tile_total[per_tile_total.size() / 10], per_tile_total[per_tile_total.size() / 2], per_tile_total[per_tile_total.size() * 9 / 10]);
    return 0;
}
