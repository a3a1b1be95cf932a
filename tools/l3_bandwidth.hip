// Read bandwidth of a buffer that is swept again and again, by size: does a working set under the 256 MiB Infinity Cache stream faster than
// one that comes from HBM every time? The question behind "Sinkhorn with Z resident in the Infinity Cache" (VERDICT round 5, item 2): one
// couplings matrix at the 5000-keypoint cap is 100 MB, and 100 iterations read it 100 times.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/l3_bandwidth tools/l3_bandwidth.hip && tools/bin/l3_bandwidth
//
// Per size: `reps` back-to-back launches of one streaming-sum kernel over the same buffer (float4 buffer loads, 16 B per lane, every workgroup
// walks a contiguous slab -- the sweeps' access pattern), timed with HIP events; the first launch warms the cache and is not timed. Reported:
// GB/s per launch, plain and nontemporal loads, and with the grid sized as ONE Sinkhorn pair offers it (157 / 313 / 626 / 1251 workgroups of
// 256 threads = 32 / 16 / 8 / 4 rows of a 5001-row matrix per workgroup) against a chip-filling grid. The last block times an EMPTY kernel
// launched back to back on one stream: the floor under any scheme that runs one small launch per pair and iteration.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// every workgroup sums a contiguous slab of `slab_f4` float4; 4 loads in flight per lane
template <bool NT>
__global__ __launch_bounds__(256) void stream_sum(const float* __restrict__ buf, size_t total_f4, size_t slab_f4, float* __restrict__ out) {
    const size_t begin = (size_t)blockIdx.x * slab_f4;
    size_t end = begin + slab_f4;
    if (end > total_f4) end = total_f4;
    const f32x4* p = reinterpret_cast<const f32x4*>(buf);
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    size_t i = begin + threadIdx.x;
    for (; i + 768 < end; i += 1024) {
        f32x4 x0, x1, x2, x3;
        if (NT) {
            x0 = __builtin_nontemporal_load(p + i);
            x1 = __builtin_nontemporal_load(p + i + 256);
            x2 = __builtin_nontemporal_load(p + i + 512);
            x3 = __builtin_nontemporal_load(p + i + 768);
        } else {
            x0 = p[i], x1 = p[i + 256], x2 = p[i + 512], x3 = p[i + 768];
        }
        a0 += x0, a1 += x1, a2 += x2, a3 += x3;
    }
    for (; i < end; i += 256) a0 += p[i];
    const f32x4 s = (a0 + a1) + (a2 + a3);
    const float t = s.x + s.y + s.z + s.w;
    if (t == 12345.678f) out[blockIdx.x] = t;  // never true for the fill below; keeps the loads alive
}

__global__ void empty_kernel(float* out) {
    if (out == nullptr) out[0] = 1.f;
}

template <bool NT>
static double time_stream(const float* buf, size_t bytes, int blocks, float* out, int reps, hipStream_t s) {
    const size_t total_f4 = bytes / 16;
    const size_t slab = (total_f4 + blocks - 1) / blocks;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream_sum<NT>), dim3(blocks), dim3(256), 0, s, buf, total_f4, slab, out);
    CHECK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((stream_sum<NT>), dim3(blocks), dim3(256), 0, s, buf, total_f4, slab, out);
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms / reps;
}

int main() {
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    const size_t max_bytes = (size_t)2048 << 20;
    float *buf, *out;
    CHECK(hipMalloc(&buf, max_bytes));
    CHECK(hipMalloc(&out, 1 << 20));
    CHECK(hipMemsetAsync(buf, 0, max_bytes, s));
    CHECK(hipStreamSynchronize(s));
    const int sizes_mb[] = {16, 32, 64, 100, 128, 160, 200, 256, 320, 512, 1024, 1600, 2048};
    printf("# streaming read of one buffer, swept repeatedly (20 launches after 2 warm-up launches); 4096 workgroups x 256 threads\n");
    printf("# size_MB  plain_us  plain_GBs  nt_us  nt_GBs\n");
    for (int mb : sizes_mb) {
        const size_t bytes = (size_t)mb * 1000 * 1000 / 16 * 16;
        const double a = time_stream<false>(buf, bytes, 4096, out, 20, s), b = time_stream<true>(buf, bytes, 4096, out, 20, s);
        printf("%7d  %8.2f  %9.1f  %8.2f  %8.1f\n", mb, a * 1e3, bytes / (a * 1e-3) / 1e9, b * 1e3, bytes / (b * 1e-3) / 1e9);
    }
    printf("# 100 MB (one couplings matrix at the 5000 cap) by grid size: the workgroups ONE pair offers at 32 / 16 / 8 / 4 rows per workgroup\n");
    printf("# blocks  plain_us  plain_GBs\n");
    const size_t one = (size_t)5001 * 5004 * 4;
    for (int blocks : {157, 313, 626, 1251, 2501, 4096}) {
        const double a = time_stream<false>(buf, one, blocks, out, 20, s);
        printf("%7d  %8.2f  %9.1f\n", blocks, a * 1e3, one / (a * 1e-3) / 1e9);
    }
    printf("# 200 MB (two matrices) by grid size\n");
    for (int blocks : {314, 626, 1252, 2502, 4096}) {
        const double a = time_stream<false>(buf, 2 * one, blocks, out, 20, s);
        printf("%7d  %8.2f  %9.1f\n", blocks, a * 1e3, 2 * one / (a * 1e-3) / 1e9);
    }
    // launch floor: empty kernels back to back on one stream, plain launches and as one captured graph
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int n = 2000;
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(empty_kernel, dim3(626), dim3(256), 0, s, out);
    CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(626), dim3(256), 0, s, out);
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("# empty kernel (626 x 256), %d back-to-back launches on one stream: %.2f us per launch\n", n, ms * 1e3 / n);
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(empty_kernel, dim3(626), dim3(256), 0, s, out);
    CHECK(hipStreamEndCapture(s, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CHECK(hipGraphLaunch(exec, s));
    CHECK(hipStreamSynchronize(s));
    CHECK(hipEventRecord(e0, s));
    CHECK(hipGraphLaunch(exec, s));
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("# the same %d launches replayed as one captured hipGraph: %.2f us per launch\n", n, ms * 1e3 / n);
    return 0;
}
