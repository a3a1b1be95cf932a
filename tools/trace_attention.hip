// Developer tool: per-phase cycle budget of attention_dma_kernel at a workload shape (default: 16 sequences x 4 heads, N = 5000).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -DGTSFM_TRACE -Igtsfm_amd/csrc -Iinclude tools/trace_attention.hip -o tools/bin/trace_attention
#include <cstdarg>
#include <cstdlib>
#include <vector>
#include "../gtsfm_amd/csrc/attention_kernels.hip"

void gtsfm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}

int main(int argc, char** argv) {
    const int nseq = argc > 1 ? atoi(argv[1]) : 16, n = argc > 2 ? atoi(argv[2]) : 5000;
    const int cap = (n + 127) / 128 * 128;
    const size_t rows = (size_t)nseq * cap;
    float *qkv, *out;
    hipMalloc(&qkv, rows * 768 * 4);
    hipMalloc(&out, rows * 256 * 4);
    std::vector<float> h(rows * 768);
    unsigned st = 1;
    for (auto& v : h) { st = st * 1664525u + 1013904223u; v = ((st >> 8) * (1.0f / 16777216.0f) - 0.5f) * 3.4f; }
    hipMemcpy(qkv, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<int> probs(nseq * 4), counts(nseq, n);
    for (int s = 0; s < nseq; ++s) probs[4 * s] = s * cap, probs[4 * s + 1] = s, probs[4 * s + 2] = s * cap, probs[4 * s + 3] = s;
    int *probs_d, *counts_d;
    hipMalloc(&probs_d, probs.size() * 4); hipMalloc(&counts_d, counts.size() * 4);
    hipMemcpy(probs_d, probs.data(), probs.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(counts_d, counts.data(), counts.size() * 4, hipMemcpyHostToDevice);
    AttnParams p = {};
    p.q = qkv, p.ldq = 768, p.k = qkv + 256, p.ldk = 768, p.v = qkv + 512, p.ldv = 768, p.out = out, p.ldo = 256;
    p.problems = (const AttnProblem*)probs_d, p.counts = counts_d, p.scale = 0.125f, p.heads = 4, p.max_k = n, p.force_split = -1;
    p.workspace_floats = attention_workspace_floats(nseq, 4, n, n, rows);
    hipMalloc(&p.workspace, p.workspace_floats * 4 + 16);
    const size_t nwg = (size_t)(nseq * 4 + 7) / 8 * 8 * ((n + 127) / 128);
#ifdef GTSFM_TRACE
    unsigned long long* trace;
    hipMalloc(&trace, nwg * 4 * 10 * 8);
    hipMemset(trace, 0, nwg * 4 * 10 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_attn_trace), &trace, sizeof(trace));
#endif
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_attention(p, nseq, n, 0);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) launch_attention(p, nseq, n, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("attention %d seq x %d: %.3f ms per launch, %.1f TFLOP/s (%.3f of 157.3)\n", nseq, n, ms, 1024.0 * n * n * nseq / (ms * 1e-3) / 1e12,
           1024.0 * n * n * nseq / (ms * 1e-3) / 1e12 / 157.3);
#ifdef GTSFM_TRACE
    std::vector<unsigned long long> t(nwg * 4 * 10);
    hipMemcpy(t.data(), trace, t.size() * 8, hipMemcpyDeviceToHost);
    double sum[8] = {0}; size_t waves = 0;
    for (size_t w = 0; w < nwg * 4; ++w) {
        if (t[w * 10 + 7] == 0) continue;
        ++waves;
        for (int k = 0; k < 8; ++k) sum[k] += (double)t[w * 10 + k];
    }
    const char* names[6] = {"S(t+1) MFMA issue", "mask + softmax VALU", "vmcnt + barrier 1 + K DMA issue", "PV MFMA issue", "vmcnt + barrier 2 + V DMA issue + sc=sn", "segment merge"};
    const double tiles = sum[7] / waves;  // (the effective clock comes from GRBM_GUI_ACTIVE, tools/prof_clock.sh: s_memtime counters of different CUs are not comparable)
    printf("%zu waves, %.1f tiles each; per wave and TILE (shader-clock cycles; two waves share a SIMD's matrix pipe: 2 x 8192 = 16384 when it never idles):\n", waves, tiles);
    double acc = 0;
    for (int k = 0; k < 6; ++k) { printf("  %-42s %9.0f\n", names[k], sum[k] / waves / tiles); acc += sum[k] / waves / tiles; }
    printf("  %-42s %9.0f\n  %-42s %9.0f (prologue + epilogue per wave: %.0f)\n", "sum of the loop", acc, "wave lifetime / tiles", sum[6] / waves / tiles, sum[6] / waves - acc * tiles);
#endif
    return 0;
}
