// fp32-MFMA multi-head attention launcher (see attention_kernels.hip).
#pragma once

#include "common.h"

// One attention problem: queries are rows [q_off, q_off + counts[q_cnt_idx]) of the q/out arrays, keys/values are
// rows [k_off, k_off + counts[k_cnt_idx]) of the k/v arrays. Counts live in device memory so that LightGlue's point
// pruning can shrink them without host synchronisation.
struct AttnProblem {
    int q_off, q_cnt_idx, k_off, k_cnt_idx;
};

struct AttnParams {
    const float* q;  // [rows][ldq], head h = columns [64h, 64h + 64)
    int ldq;
    const float* k;
    int ldk;
    const float* v;
    int ldv;
    float* out;  // [rows][ldo]
    int ldo;
    const AttnProblem* problems;  // device
    const int* counts;            // device
    float scale;                  // softmax(scale * q k^T)
    int heads;
    // split schedule (one workgroup per query tile AND key segment + a combine kernel; bit-identical to the fused schedule):
    int max_k;                     // most keys of any problem (0: max_q); sizes the segment dimension of the split grid
    float* workspace;              // device, attention_workspace_floats(...) floats: the split schedule's partial states, or the fused
    size_t workspace_floats;       // schedule's parking space for the merged state between key segments; nullptr: fused, parked in LDS
    size_t part_rows;              // rows of the q / out arrays (a segment's partial O is [part_rows][heads * 64])
    int force_split;               // 0: by launch geometry, 1: always (if more than one segment fits), -1: never
    int math;                      // ATTN_MATH_F32 (exact fp32 MFMA, the default) ATTN_MATH_BF16X3 (three-way bf16 split of both products, fp32 accumulate) or ATTN_MATH_F16X2 (two-way fp16 split)
    int qtiles, nproblems, nseg;   // filled by the launcher
    int lds_has_oc;                // "
    int xcd_rep;                   // " (XCDs one (problem, head) group's workgroups are dealt over: > 1 for launches with fewer than 8 groups)
    float* part_o;                 // "
    float* part_ml;                // "
    float* park;                   // "
    float* x3;                     // " (bf16x3 / f16x2: the split K / V^T tiles, at the start of the workspace)
    int x3_tiles;                  // " (bf16x3 / f16x2: key tiles per problem in that buffer)
};

#define ATTN_MATH_F32 0
#define ATTN_MATH_BF16X3 1
#define ATTN_MATH_F16X2 2  // two fp16 pieces per operand, three fp16 MFMA products per block (f16x2.h)

// Floats of workspace the launch (nproblems, heads, max_q, max_k) over `rows` token rows can use (0 when every problem is one segment).
size_t attention_workspace_floats(int nproblems, int heads, int max_q, int max_k, size_t rows, int math);
int attention_math_from_env();      // GTSFM_ATTENTION_MATH=bf16x3 -> ATTN_MATH_BF16X3, f16x2 -> ATTN_MATH_F16X2, else ATTN_MATH_F32 (read per call)
int attention_segments(int max_k);  // key segments a launch must provide for when no problem has more than max_k keys (an upper bound: 512-key segments)
bool attention_parks_in_workspace();  // the double-buffered build: fused launches on two streams must not share one workspace
int launch_attention(const AttnParams& p, int nproblems, int max_q, hipStream_t stream);
