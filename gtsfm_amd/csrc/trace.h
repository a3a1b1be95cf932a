// Developer timeline macros: tools/trace_gemm*.hip build the kernel sources with -DGTSFM_TRACE (s_memtime stamps at segment
// boundaries, fenced with sched_barrier); nothing of it is in the product build.
#pragma once

#ifdef GTSFM_TRACE
static __device__ unsigned long long* g_gemm_trace;  // [workgroup][wave][8]
#define GT_DECL unsigned gt_prev = (unsigned)__builtin_amdgcn_s_memtime(); const unsigned gt_begin = gt_prev; unsigned gseg[6] = {0, 0, 0, 0, 0, 0};
#define GT_SEG(k)                                                      \
    {                                                                  \
        __builtin_amdgcn_sched_barrier(0);                             \
        const unsigned gt_now = (unsigned)__builtin_amdgcn_s_memtime(); \
        gseg[k] += gt_now - gt_prev;                                   \
        gt_prev = gt_now;                                              \
        __builtin_amdgcn_sched_barrier(0);                             \
    }
#else
#define GT_DECL
#define GT_SEG(k)
#endif

