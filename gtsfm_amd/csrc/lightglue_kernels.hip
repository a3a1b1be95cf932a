// LightGlue-specific kernels: input staging into the 128-row-aligned ragged layout, adaptive-depth stop test,
// adaptive-width point pruning (stable compaction), and the scatter of the final matches back to the original
// keypoint indices. All control flow stays on the device: a pair that stops early gets its live row counts zeroed, so
// every later GEMM tile / attention block / row kernel of that pair exits immediately -- no host synchronisation.
// Restates upstream cvg/LightGlue lightglue.py (check_if_stop, get_pruning_mask, the index_select pruning in
// _forward, and the final re-indexing through ind0 / ind1); the reference tree only holds the call sites
// (gtsfm/frontend/matcher/lightglue_matcher.py:37-110).

#include "lightglue_kernels.h"

// X[row_off + i][0:256] = desc[in_off + i]; ind[row_off + i] = i
__global__ __launch_bounds__(256) void lg_load_inputs_kernel(const float* __restrict__ desc, const SeqDesc* __restrict__ seqs,
                                                             const int* __restrict__ counts, float* __restrict__ X, int ldx,
                                                             int* __restrict__ ind) {
    const SeqDesc sq = seqs[blockIdx.y];
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= counts[sq.cnt_idx]) return;
    const int lane = threadIdx.x & 63;
    *reinterpret_cast<f32x4*>(X + (size_t)(sq.row_off + i) * ldx + lane * 4) =
        *reinterpret_cast<const f32x4*>(desc + (size_t)(sq.in_off + i) * 256 + lane * 4);
    if (lane == 0) ind[sq.row_off + i] = i;
}

// out[in_off + i][0:256] = X[row_off + i][0:256]: the inverse of lg_load_inputs (x after the first self block, per image)
__global__ __launch_bounds__(256) void lg_store_rows_kernel(const float* __restrict__ X, int ldx, const SeqDesc* __restrict__ seqs,
                                                            const int* __restrict__ counts, float* __restrict__ out) {
    const SeqDesc sq = seqs[blockIdx.y];
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= counts[sq.cnt_idx]) return;
    const int lane = threadIdx.x & 63;
    *reinterpret_cast<f32x4*>(out + (size_t)(sq.in_off + i) * 256 + lane * 4) =
        *reinterpret_cast<const f32x4*>(X + (size_t)(sq.row_off + i) * ldx + lane * 4);
}

// check_if_stop for layer `layer`: ratio of confident points (pruned points count as confident) > depth_confidence.
// One workgroup per pair. Also maintains the per-layer "assign" counts: the keypoint sets of the pairs that stop at
// this layer (or at the last layer) get their final / assign counts set and their live counts zeroed, and their kept-index lists
// are copied to a stable array (later pruning steps ping-pong the live buffers; a kernel of its own until round 4).
__global__ __launch_bounds__(256) void lg_stop_check_kernel(const float* __restrict__ conf, const SeqDesc* __restrict__ seqs,
                                                            int* __restrict__ live, int* __restrict__ final_cnt, int* __restrict__ assign,
                                                            const int* __restrict__ orig, int* __restrict__ stop_layer, int layer,
                                                            int last_layer, float conf_threshold, float depth_confidence,
                                                            const int* __restrict__ ind_cur, int* __restrict__ ind_final) {
    __shared__ int wsum[4];
    const int p = blockIdx.x;
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int c0 = live[2 * p], c1 = live[2 * p + 1];
    const int stopped_at = stop_layer[p];
    // every wave has read the counts and the stop layer before thread 0 overwrites them below: on the last layer there is no other
    // barrier between these loads and those stores, and a wave that ran late would read 0 / 0 with stop_layer >= 0, leave at the
    // early-out and skip its share of the index copy
    __syncthreads();
    if (threadIdx.x == 0) assign[2 * p] = assign[2 * p + 1] = 0;
    if (c0 == 0 && c1 == 0 && stopped_at >= 0) return;  // stopped earlier (uniform)
    bool stop = (layer == last_layer);
    if (!stop && depth_confidence > 0.f) {
        int cnt = 0;
        for (int i = threadIdx.x; i < c0; i += 256) cnt += (conf[s0.row_off + i] < conf_threshold) ? 1 : 0;
        for (int i = threadIdx.x; i < c1; i += 256) cnt += (conf[s1.row_off + i] < conf_threshold) ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
        __syncthreads();
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const float ratio = 1.0f - (float)total / (float)(orig[2 * p] + orig[2 * p + 1]);
        stop = ratio > depth_confidence;
    }
    if (!stop) return;  // uniform
    for (int i = threadIdx.x; i < c0; i += 256) ind_final[s0.row_off + i] = ind_cur[s0.row_off + i];
    for (int i = threadIdx.x; i < c1; i += 256) ind_final[s1.row_off + i] = ind_cur[s1.row_off + i];
    if (threadIdx.x == 0) {
        stop_layer[p] = layer;
        final_cnt[2 * p] = assign[2 * p] = c0;
        final_cnt[2 * p + 1] = assign[2 * p + 1] = c1;
        live[2 * p] = live[2 * p + 1] = 0;
    }
}

// get_pruning_mask + stable compaction plan for one keypoint set (one workgroup per set):
//   keep_i = (matchability_i > 1 - width_confidence) | (confidence_i <= conf_threshold)   if count > pruning_threshold
// pos[row_off + i] = new position (or -1 when pruned); old_cnt = count before; live = count after.
__global__ __launch_bounds__(1024) void lg_prune_plan_kernel(const float* __restrict__ conf, const float* __restrict__ matchability,
                                                             const SeqDesc* __restrict__ seqs, int* __restrict__ live,
                                                             int* __restrict__ old_cnt, int* __restrict__ pos, float conf_threshold,
                                                             float keep_threshold, int pruning_threshold, int use_conf) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const SeqDesc sq = seqs[blockIdx.x];
    const int n = live[sq.cnt_idx];
    const bool prune = n > pruning_threshold;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        int keep = 0;
        if (i < n) {
            keep = 1;
            if (prune) {
                keep = matchability[sq.row_off + i] > keep_threshold;
                if (use_conf) keep |= conf[sq.row_off + i] <= conf_threshold;
            }
        }
        int incl = keep;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        const int carry = carry_s;
        if (i < n) pos[sq.row_off + i] = keep ? carry + wbase + incl - 1 : -1;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wbase + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        old_cnt[sq.cnt_idx] = n;
        live[sq.cnt_idx] = carry_s;
    }
}

// Apply the plan: copy the kept rows of X (first 256 columns), the cached rotary encoding and the index list into the
// alternate buffers (index_select(1, keep) in upstream _forward). One wave per source row.
__global__ __launch_bounds__(256) void lg_prune_apply_kernel(const SeqDesc* __restrict__ seqs, const int* __restrict__ old_cnt,
                                                             const int* __restrict__ pos, const float* __restrict__ Xs, float* __restrict__ Xd,
                                                             int ldx, const float* __restrict__ encs, float* __restrict__ encd,
                                                             const int* __restrict__ inds, int* __restrict__ indd) {
    const SeqDesc sq = seqs[blockIdx.y];
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= old_cnt[sq.cnt_idx]) return;
    const int dst = pos[sq.row_off + i];
    if (dst < 0) return;
    const int lane = threadIdx.x & 63;
    const size_t rs = (size_t)(sq.row_off + i), rd = (size_t)(sq.row_off + dst);
    *reinterpret_cast<f32x4*>(Xd + rd * ldx + lane * 4) = *reinterpret_cast<const f32x4*>(Xs + rs * ldx + lane * 4);
    encd[rd * 64 + lane] = encs[rs * 64 + lane];
    if (lane == 0) indd[rd] = inds[rs];
}

// matches_out / mscores_out default to -1 / 0; kept keypoint i of a set maps to original index ind[i].
__global__ void lg_fill_outputs_kernel(int* __restrict__ matches, float* __restrict__ mscores, int total) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        matches[t] = -1;
        mscores[t] = 0.f;
    }
}

__global__ void lg_scatter_matches_kernel(const SeqDesc* __restrict__ seqs, const int* __restrict__ final_cnt, const int* __restrict__ ind,
                                          const int* __restrict__ m_int, const float* __restrict__ ms_int, int* __restrict__ matches,
                                          float* __restrict__ mscores) {
    const int s = blockIdx.y;
    const SeqDesc sq = seqs[s], other = seqs[s ^ 1];
    const int n = final_cnt[sq.cnt_idx];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int m = m_int[sq.row_off + i];
        const int dst = sq.in_off + ind[sq.row_off + i];
        matches[dst] = (m >= 0) ? ind[other.row_off + m] : -1;
        mscores[dst] = ms_int[sq.row_off + i];
    }
}

int launch_lg_load_inputs(const float* desc, const SeqDesc* seqs, const int* counts, int nseq, int max_n, float* X, int ldx, int* ind,
                          hipStream_t stream) {
    if (nseq <= 0 || max_n <= 0) return GTSFM_OK;
    hipLaunchKernelGGL(lg_load_inputs_kernel, dim3(ceil_div(max_n, 4), nseq), dim3(256), 0, stream, desc, seqs, counts, X, ldx, ind);
    GTSFM_CHECK_LAUNCH("lg_load_inputs_kernel");
    return GTSFM_OK;
}

int launch_lg_store_rows(const float* X, int ldx, const SeqDesc* seqs, const int* counts, int nseq, int max_n, float* out, hipStream_t stream) {
    if (nseq <= 0 || max_n <= 0) return GTSFM_OK;
    hipLaunchKernelGGL(lg_store_rows_kernel, dim3(ceil_div(max_n, 4), nseq), dim3(256), 0, stream, X, ldx, seqs, counts, out);
    GTSFM_CHECK_LAUNCH("lg_store_rows_kernel");
    return GTSFM_OK;
}

int launch_lg_stop_check(const float* conf, const SeqDesc* seqs, int* live, int* final_cnt, int* assign, const int* orig, int* stop_layer,
                         int npairs, int layer, int last_layer, float conf_threshold, float depth_confidence, const int* ind_cur, int* ind_final,
                         hipStream_t stream) {
    if (npairs <= 0) return GTSFM_OK;
    hipLaunchKernelGGL(lg_stop_check_kernel, dim3(npairs), dim3(256), 0, stream, conf, seqs, live, final_cnt, assign, orig, stop_layer, layer,
                       last_layer, conf_threshold, depth_confidence, ind_cur, ind_final);
    GTSFM_CHECK_LAUNCH("lg_stop_check_kernel");
    return GTSFM_OK;
}

int launch_lg_prune(const float* conf, const float* matchability, const SeqDesc* seqs, int* live, int* old_cnt, int* pos, int nseq,
                    int max_n, float conf_threshold, float keep_threshold, int pruning_threshold, int use_conf, const float* Xs, float* Xd,
                    int ldx, const float* encs, float* encd, const int* inds, int* indd, hipStream_t stream) {
    if (nseq <= 0 || max_n <= 0) return GTSFM_OK;
    hipLaunchKernelGGL(lg_prune_plan_kernel, dim3(nseq), dim3(1024), 0, stream, conf, matchability, seqs, live, old_cnt, pos, conf_threshold,
                       keep_threshold, pruning_threshold, use_conf);
    hipLaunchKernelGGL(lg_prune_apply_kernel, dim3(ceil_div(max_n, 4), nseq), dim3(256), 0, stream, seqs, old_cnt, pos, Xs, Xd, ldx, encs, encd,
                       inds, indd);
    GTSFM_CHECK_LAUNCH("lg_prune kernels");
    return GTSFM_OK;
}

int launch_lg_scatter_matches(const SeqDesc* seqs, const int* final_cnt, const int* ind, const int* m_int, const float* ms_int, int nseq,
                              int max_n, int total_out, int* matches, float* mscores, hipStream_t stream) {
    if (total_out <= 0) return GTSFM_OK;
    hipLaunchKernelGGL(lg_fill_outputs_kernel, dim3(ceil_div(total_out, 256) < 1024 ? ceil_div(total_out, 256) : 1024), dim3(256), 0, stream,
                       matches, mscores, total_out);
    if (nseq > 0 && max_n > 0)
        hipLaunchKernelGGL(lg_scatter_matches_kernel, dim3(ceil_div(max_n, 256), nseq), dim3(256), 0, stream, seqs, final_cnt, ind, m_int,
                           ms_int, matches, mscores);
    GTSFM_CHECK_LAUNCH("lg_scatter kernels");
    return GTSFM_OK;
}

