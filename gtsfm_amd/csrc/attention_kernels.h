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
    int qtiles, nproblems;        // filled by the launcher
};

int launch_attention(const AttnParams& p, int nproblems, int max_q, hipStream_t stream);
