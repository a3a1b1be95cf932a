// Matcher kernels that are not GEMMs: keypoint encoding inputs, rotary embedding, LayerNorm+GELU, log-space Sinkhorn,
// LightGlue's double-softmax assignment, mutual-nearest-neighbour match extraction, point pruning.
// Replaces the ATen op sequences of
//   thirdparty/SuperGluePretrainedNetwork/models/superglue.py:63-70 (normalize_keypoints), :141-170 (optimal
//   transport), :266-276 (match extraction)
// and of upstream cvg/LightGlue lightglue.py (normalize_keypoints, LearnableFourierPositionalEncoding,
// apply_cached_rotary_emb, the FFN's LayerNorm+GELU, sigmoid_log_double_softmax, filter_matches, pruning).
// These are HBM/L2-bound sweeps: coalesced rows, wave-shuffle reductions, one pass over the score matrix per
// Sinkhorn iteration. Built with -ffp-contract=off.

#include "matcher_kernels.h"

// ---------------------------------------------------------------------------------------------------------------
// Keypoint inputs
// ---------------------------------------------------------------------------------------------------------------

// SuperGlue: enc_in[t] = [(x - W/2) / (0.7 max(W,H)), (y - H/2) / (0.7 max(W,H)), score, 0, 0, 0, 0, 0]
__global__ void sg_encode_input_kernel(const float* __restrict__ kpts, const float* __restrict__ scores, const SeqDesc* __restrict__ seqs,
                                       const int* __restrict__ counts, int nseq, float* __restrict__ enc_in) {
    const int s = blockIdx.y;
    const SeqDesc sq = seqs[s];
    const int n = counts[sq.cnt_idx];
    const float w = (float)sq.W, h = (float)sq.H;
    const float cx = w / 2.0f, cy = h / 2.0f;
    const float scaling = fmaxf(w, h) * 0.7f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const size_t t = (size_t)sq.row_off + i, ti = (size_t)sq.in_off + i;
        float* o = enc_in + t * 8;
        o[0] = (kpts[ti * 2 + 0] - cx) / scaling;
        o[1] = (kpts[ti * 2 + 1] - cy) / scaling;
        o[2] = scores[ti];
        o[3] = o[4] = o[5] = o[6] = o[7] = 0.f;
    }
}

// LightGlue: normalised keypoints (k - size/2) / (max(size)/2) -> Fourier features cos/sin(Wr k): enc[t] = [f][cos, sin], f < 32
__global__ void lg_posenc_kernel(const float* __restrict__ kpts, const SeqDesc* __restrict__ seqs, const int* __restrict__ counts,
                                 const float* __restrict__ Wr /*[32][2]*/, float* __restrict__ enc /*[T][64]*/) {
    const int s = blockIdx.y;
    const SeqDesc sq = seqs[s];
    const int n = counts[sq.cnt_idx];
    const float w = (float)sq.W, h = (float)sq.H;
    const float sx = w / 2.0f, sy = h / 2.0f;
    const float scale = fmaxf(w, h) / 2.0f;
    const int f = threadIdx.x & 31;
    const float w0 = Wr[f * 2 + 0], w1 = Wr[f * 2 + 1];
    for (int i = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5); i < n; i += gridDim.x * (blockDim.x >> 5)) {
        const size_t t = (size_t)sq.row_off + i, ti = (size_t)sq.in_off + i;
        const float x = (kpts[ti * 2 + 0] - sx) / scale;
        const float y = (kpts[ti * 2 + 1] - sy) / scale;
        const float pr = x * w0 + y * w1;  // F.linear without bias: sum over 2 inputs
        enc[t * 64 + 2 * f] = cosf(pr);  // [f][cos, sin]: one 16-byte load gives the GEMM epilogue two feature pairs' factors
        enc[t * 64 + 2 * f + 1] = sinf(pr);
    }
}

// Rotary embedding on the q and k parts of a packed [T][ld] buffer (head-major, 4 heads x 64). Stand-alone form: the
// LDS-DMA GEMM applies it in the Wqkv epilogue (gemm_dma_walk_kernel<false, true>); this kernel serves the register-staged
// GEMM (GTSFM_GEMM=mfma).
//   out[2f] = t[2f] cos_f - t[2f+1] sin_f ; out[2f+1] = t[2f+1] cos_f + t[2f] sin_f     (same freqs for every head)
__global__ void lg_rotary_kernel(float* __restrict__ qkv, int ld, int ncols /*512: q and k*/, const float* __restrict__ enc,
                                 const SeqDesc* __restrict__ seqs, const int* __restrict__ counts) {
    const int s = blockIdx.y;
    const SeqDesc sq = seqs[s];
    const int n = counts[sq.cnt_idx];
    const int pairs_per_row = ncols >> 1;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n * pairs_per_row; idx += gridDim.x * blockDim.x) {
        const int i = idx / pairs_per_row, pp = idx % pairs_per_row;
        const int f = pp & 31;  // frequency index within the head
        const size_t t = (size_t)sq.row_off + i;
        float* x = qkv + t * ld + pp * 2;
        const float c = enc[t * 64 + 2 * f], sn = enc[t * 64 + 2 * f + 1];
        const float x1 = x[0], x2 = x[1];
        x[0] = (x1 * c) + ((-x2) * sn);
        x[1] = (x2 * c) + (x1 * sn);
    }
}

// y = GELU(LayerNorm(x)) over rows of 512 (eps 1e-5, affine), in place. One wave owns LN_ROWS consecutive rows (all
// loads issued up front for memory-level parallelism), 8 elements per lane per row.
#define LN_ROWS 4
__global__ __launch_bounds__(256) void layernorm_gelu_kernel(float* __restrict__ x, int ld, const SeqDesc* __restrict__ seqs,
                                                             const int* __restrict__ counts, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta) {
    const SeqDesc sq = seqs[blockIdx.y];
    const int n = counts[sq.cnt_idx];
    const int i0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * LN_ROWS;
    if (i0 >= n) return;
    const int lane = threadIdx.x & 63;
    f32x4 a[LN_ROWS], b[LN_ROWS];
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        const int i = min(i0 + r, n - 1);
        const float* p = x + (size_t)(sq.row_off + i) * ld + lane * 8;
        a[r] = *reinterpret_cast<const f32x4*>(p);
        b[r] = *reinterpret_cast<const f32x4*>(p + 4);
    }
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + lane * 8), g1 = *reinterpret_cast<const f32x4*>(gamma + lane * 8 + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + lane * 8), b1 = *reinterpret_cast<const f32x4*>(beta + lane * 8 + 4);
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        if (i0 + r >= n) break;
        float v[8] = {a[r].x, a[r].y, a[r].z, a[r].w, b[r].x, b[r].y, b[r].z, b[r].w};
        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += v[e];
        const float mean = wave_sum(sum) / 512.0f;
        float ssq = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = v[e] - mean;
            ssq += d * d;
        }
        const float var = wave_sum(ssq) / 512.0f;
        const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float y = (v[e] - mean) * rstd * gm[e] + bt[e];
            v[e] = 0.5f * y * (1.0f + erff(y * 0.70710678118654752440f));
        }
        float* p = x + (size_t)(sq.row_off + i0 + r) * ld + lane * 8;
        *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
}

// LightGlue's two Linear(256, 1) heads over the live tokens of a layer in ONE pass over x (lightglue.py: TokenConfidence.forward,
// MatchAssignment.get_matchability): conf[t] = sigmoid(x_t . w_conf + b_conf) (adaptive depth / width; absent on the last layer),
// z[t] = x_t . w_match + b_match (the matchability LOGIT the final assignment uses if the pair stops at this layer: stopped pairs have a
// live count of 0 from then on, so their rows keep the values of their last layer) and mval[t] = sigmoid(z[t]) (point pruning).
// One wave per token. Rounds 1-3 ran three launches of a one-head kernel here; the arithmetic per head is the same.
__global__ __launch_bounds__(256) void lg_heads_kernel(const float* __restrict__ x, int ld, const SeqDesc* __restrict__ seqs,
                                                       const int* __restrict__ counts, const float* __restrict__ w_conf, float b_conf,
                                                       const float* __restrict__ w_match, float b_match, float* __restrict__ conf,
                                                       float* __restrict__ z, float* __restrict__ mval) {
    const SeqDesc sq = seqs[blockIdx.y];
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= counts[sq.cnt_idx]) return;
    const int row = sq.row_off + i;
    const int lane = threadIdx.x & 63;
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + (size_t)row * ld + lane * 4);
    const f32x4 wm = *reinterpret_cast<const f32x4*>(w_match + lane * 4);
    float sm = a.x * wm.x + a.y * wm.y + a.z * wm.z + a.w * wm.w;
    sm = wave_sum(sm) + b_match;
    if (w_conf) {
        const f32x4 wc = *reinterpret_cast<const f32x4*>(w_conf + lane * 4);
        float sc = a.x * wc.x + a.y * wc.y + a.z * wc.z + a.w * wc.w;
        sc = wave_sum(sc) + b_conf;
        if (lane == 0) conf[row] = 1.0f / (1.0f + expf(-sc));
    }
    if (lane == 0) {
        z[row] = sm;
        if (mval) mval[row] = 1.0f / (1.0f + expf(-sm));
    }
}

__global__ __launch_bounds__(256) void copy_rows256_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int rows) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r < rows) *reinterpret_cast<f32x4*>(dst + (size_t)r * ldd + lane * 4) = *reinterpret_cast<const f32x4*>(src + (size_t)r * lds + lane * 4);
}

int launch_copy_rows256(const float* src, int lds, float* dst, int ldd, int rows, hipStream_t stream) {
    if (rows <= 0) return GTSFM_OK;
    hipLaunchKernelGGL(copy_rows256_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, src, lds, dst, ldd, rows);
    GTSFM_CHECK_LAUNCH("copy_rows256_kernel");
    return GTSFM_OK;
}

// dst block b <- src block src_index[b] (gather) or dst block dst_index[b] <- src block b (scatter); a block = block_floats contiguous
// floats (an image's rows of the feature table), a multiple of 2. One workgroup row per block, float2 granularity.
__global__ __launch_bounds__(256) void move_blocks_kernel(const float* __restrict__ src, const int* __restrict__ src_index, float* __restrict__ dst,
                                                          const int* __restrict__ dst_index, long long block_floats) {
    const int b = blockIdx.y;
    const float2* s = reinterpret_cast<const float2*>(src + (size_t)(src_index ? src_index[b] : b) * block_floats);
    float2* d = reinterpret_cast<float2*>(dst + (size_t)(dst_index ? dst_index[b] : b) * block_floats);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < block_floats / 2; i += (long long)gridDim.x * 256) d[i] = s[i];
}

int launch_move_blocks(const float* src, const int* src_index, float* dst, const int* dst_index, int nblocks, long long block_floats, hipStream_t stream) {
    if (nblocks <= 0 || block_floats <= 0) return GTSFM_OK;
    const long long per_block = ceil_div((int)((block_floats / 2 + 255) / 256), 1);
    hipLaunchKernelGGL(move_blocks_kernel, dim3((unsigned)(per_block < 64 ? per_block : 64), nblocks), dim3(256), 0, stream, src, src_index, dst, dst_index, block_floats);
    GTSFM_CHECK_LAUNCH("move_blocks_kernel");
    return GTSFM_OK;
}

int launch_sg_encode_input(const float* kpts, const float* scores, const SeqDesc* seqs, const int* counts, int nseq, int max_n,
                           float* enc_in, hipStream_t stream) {
    if (nseq <= 0 || max_n <= 0) return GTSFM_OK;
    hipLaunchKernelGGL(sg_encode_input_kernel, dim3(ceil_div(max_n, 256), nseq), dim3(256), 0, stream, kpts, scores, seqs, counts, nseq, enc_in);
    GTSFM_CHECK_LAUNCH("sg_encode_input_kernel");
    return GTSFM_OK;
}

int launch_lg_posenc(const float* kpts, const SeqDesc* seqs, const int* counts, int nseq, int max_n, const float* Wr, float* enc,
                     hipStream_t stream) {
    if (nseq <= 0 || max_n <= 0) return GTSFM_OK;
    hipLaunchKernelGGL(lg_posenc_kernel, dim3(ceil_div(max_n, 8), nseq), dim3(256), 0, stream, kpts, seqs, counts, Wr, enc);
    GTSFM_CHECK_LAUNCH("lg_posenc_kernel");
    return GTSFM_OK;
}

int launch_lg_rotary(float* qkv, int ld, int ncols, const float* enc, const SeqDesc* seqs, const int* counts, int nseq, int max_n,
                     hipStream_t stream) {
    if (nseq <= 0 || max_n <= 0) return GTSFM_OK;
    const int blocks = ceil_div(max_n * (ncols / 2), 256);
    hipLaunchKernelGGL(lg_rotary_kernel, dim3(blocks < 2048 ? blocks : 2048, nseq), dim3(256), 0, stream, qkv, ld, ncols, enc, seqs, counts);
    GTSFM_CHECK_LAUNCH("lg_rotary_kernel");
    return GTSFM_OK;
}

int launch_layernorm_gelu(float* x, int ld, const SeqDesc* seqs, const int* counts, int nseq, int max_n, const float* gamma,
                          const float* beta, hipStream_t stream) {
    if (nseq <= 0 || max_n <= 0) return GTSFM_OK;
    hipLaunchKernelGGL(layernorm_gelu_kernel, dim3(ceil_div(max_n, 4 * LN_ROWS), nseq), dim3(256), 0, stream, x, ld, seqs, counts, gamma, beta);
    GTSFM_CHECK_LAUNCH("layernorm_gelu_kernel");
    return GTSFM_OK;
}

int launch_lg_heads(const float* x, int ld, const SeqDesc* seqs, const int* counts, int nseq, int max_n, const float* w_conf, float b_conf,
                    const float* w_match, float b_match, float* conf, float* z, float* mval, hipStream_t stream) {
    if (nseq <= 0 || max_n <= 0) return GTSFM_OK;
    hipLaunchKernelGGL(lg_heads_kernel, dim3(ceil_div(max_n, 4), nseq), dim3(256), 0, stream, x, ld, seqs, counts, w_conf, b_conf, w_match, b_match,
                       conf, z, mval);
    GTSFM_CHECK_LAUNCH("lg_heads_kernel");
    return GTSFM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Score-matrix sweeps. Per pair p the matrix Z lives at zbuf + z_off, (m + EXT) x (n + EXT) with row stride ld,
// where EXT = 1 for SuperGlue (dustbin row/column, superglue.py:150-170) and 0 for LightGlue. Row vectors (u, row
// log-sum-exp, ...) are indexed by vec0 + i, column vectors by vec1 + j with vecS = row_off(seq) + seq index.
//
// One Sinkhorn iteration (superglue.py:141-147) reads Z ONCE:
//   rows kernel : a workgroup owns 16 rows; phase A: u_i = log_mu_i - logsumexp_j(Z_ij + v_j) (wave per row, two-pass);
//                 phase B: column partials (max, sum) of Z_ij + u_i over its 16 rows (L1/L2-hot re-read)
//   cols kernel : v_j = log_nu_j - logsumexp over the row-block partials
// LightGlue's double softmax uses the same two kernels once with u = v = 0 and plain log-sum-exp outputs.
// ---------------------------------------------------------------------------------------------------------------

// Rows per workgroup of the row sweep: as many as fit a 64 KiB LDS slab (2 workgroups per CU), at most 16.
int sweep_rows_per_block(int max_cols) {
    const int ld = (max_cols + 3) / 4 * 4;
    int r = (64 * 1024 / 4 - 16) / ld - 1;  // one extra row of LDS holds the column vector v
    return r < 1 ? 1 : (r > 16 ? 16 : r);
}

template <bool SG>
__global__ __launch_bounds__(256) void lse_rows_kernel(const float* __restrict__ zbuf, const PairDesc* __restrict__ pairs,
                                                       const SeqDesc* __restrict__ seqs, const int* __restrict__ counts,
                                                       float* __restrict__ rowvec, const float* __restrict__ colvec,
                                                       float* __restrict__ partials, int R) {
    // The workgroup's R rows are staged ONCE into LDS (coalesced 16-byte loads, all in flight together); the row
    // log-sum-exp (phase A) and the column partials (phase B) both run out of LDS, so Z is read from HBM exactly once
    // per Sinkhorn iteration.
    extern __shared__ __attribute__((aligned(16))) float zs[];  // [R][ld] + v[ld] + c[R]
    const int p = blockIdx.y;
    const PairDesc pd = pairs[p];
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    const int rows = m + (SG ? 1 : 0), cols = n + (SG ? 1 : 0);
    const int r0 = blockIdx.x * R;
    if (r0 >= rows) return;
    const int nr = min(R, rows - r0);
    const int ld = pd.ld, ld4 = ld >> 2;
    const float* Z = zbuf + pd.z_off + (size_t)r0 * ld;
    const int vec0 = vec_off(s0, 2 * p), vec1 = vec_off(s1, 2 * p + 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float norm = SG ? -logf((float)m + (float)n) : 0.f;
    const float NEG = neg_inf();
    float* v_s = zs + (size_t)R * ld;  // staged column vector (SuperGlue's v)
    float* c_s = v_s + ld;  // per-row exponent offsets of the e values (phase A -> phase B)
    if (SG)
        for (int j = threadIdx.x * 4; j < cols; j += 1024) *reinterpret_cast<f32x4*>(v_s + j) = *reinterpret_cast<const f32x4*>(colvec + vec1 + j);
    // stage: the nr rows are contiguous in memory (row stride ld), nr * ld4 float4 in total
    for (int base = 0; base < nr * ld4; base += 256 * 8) {
        f32x4 t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = base + e * 256 + threadIdx.x;
            t[e] = (idx < nr * ld4) ? reinterpret_cast<const f32x4*>(Z)[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = base + e * 256 + threadIdx.x;
            if (idx < nr * ld4) reinterpret_cast<f32x4*>(zs)[idx] = t[e];
        }
    }
    __syncthreads();
    // phase A: wave w reduces rows w, w+4, ... : row maximum, then e_ij = exp(z_ij + v_j - mx_i) written back IN PLACE
    // (LDS copy only) and summed. One hardware exponential (v_exp_f32 on a pre-scaled argument) per element; the column
    // pass below reuses e_ij instead of exponentiating again:
    //     exp(z_ij + u_i) = e_ij * exp(-v_j) * exp(c_i),   c_i = u_i + mx_i        (SuperGlue; LightGlue: c_i = mx_i, v = 0)
    // (round 1 spent two library expf per element here and was VALU-bound)
    constexpr float LOG2E = 1.44269504088896340736f;
    for (int rr = wave; rr < nr; rr += 4) {
        const int i = r0 + rr;
        float* zr = zs + (size_t)rr * ld;
        float mx = NEG;
#pragma unroll 4
        for (int j = lane * 4; j < cols; j += 256) {
            f32x4 z = *reinterpret_cast<const f32x4*>(zr + j);
            if (SG) z = z + *reinterpret_cast<const f32x4*>(v_s + j);
            mx = fmaxf(mx, z.x);
            if (j + 1 < cols) mx = fmaxf(mx, z.y);
            if (j + 2 < cols) mx = fmaxf(mx, z.z);
            if (j + 3 < cols) mx = fmaxf(mx, z.w);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll 4
        for (int j = lane * 4; j < cols; j += 256) {
            f32x4 z = *reinterpret_cast<const f32x4*>(zr + j);
            if (SG) z = z + *reinterpret_cast<const f32x4*>(v_s + j);
            f32x4 e;
            e.x = __builtin_amdgcn_exp2f((z.x - mx) * LOG2E);
            e.y = (j + 1 < cols) ? __builtin_amdgcn_exp2f((z.y - mx) * LOG2E) : 0.f;
            e.z = (j + 2 < cols) ? __builtin_amdgcn_exp2f((z.z - mx) * LOG2E) : 0.f;
            e.w = (j + 3 < cols) ? __builtin_amdgcn_exp2f((z.w - mx) * LOG2E) : 0.f;
            sum += (e.x + e.y) + (e.z + e.w);
            *reinterpret_cast<f32x4*>(zr + j) = e;
        }
        sum = wave_sum(sum);
        const float lse = logf(sum) + mx;
        float ui;
        if (SG) {
            const float log_mu = (i < m) ? norm : logf((float)n) + norm;
            ui = log_mu - lse;
        } else {
            ui = lse;
        }
        if (lane == 0) {
            rowvec[vec0 + i] = ui;
            c_s[rr] = SG ? ui + mx : mx;
        }
    }
    __syncthreads();
    // phase B: column partials of this block's rows as (M_b, sum_i e_ij exp(c_i - M_b)), M_b = max_i c_i: the same
    // (max, sum) pairs lse_cols_kernel combines; SuperGlue's exp(-v_j) factor is applied there
    float mb = NEG;
    for (int rr = 0; rr < nr; ++rr) mb = fmaxf(mb, c_s[rr]);
    float* part = partials + pd.part_off + (size_t)blockIdx.x * ld * 2;
    for (int j = threadIdx.x * 4; j < cols; j += 1024) {
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int rr = 0; rr < nr; ++rr) {
            const f32x4 e = *reinterpret_cast<const f32x4*>(zs + (size_t)rr * ld + j);
            const float w = __builtin_amdgcn_exp2f((c_s[rr] - mb) * LOG2E);
            sum.x = fmaf(e.x, w, sum.x), sum.y = fmaf(e.y, w, sum.y), sum.z = fmaf(e.z, w, sum.z), sum.w = fmaf(e.w, w, sum.w);
        }
        // columns beyond `cols` (row padding) produce garbage partials that are never read
        *reinterpret_cast<f32x4*>(part + (size_t)j * 2) = f32x4{mb, sum.x, mb, sum.y};
        *reinterpret_cast<f32x4*>(part + (size_t)j * 2 + 4) = f32x4{mb, sum.z, mb, sum.w};
    }
}

// Round-1 form of the row sweep: two passes per reduction and a second exponential in the column pass. Exact for any
// value range, so LightGlue's double log-softmax (no dustbin row that bounds the column sums; similarities of +-100)
// keeps it: reusing the row exponentials for the column sums underflows when a column's largest entries lie > 87
// below their rows' maxima (8 LightGlue parity tests caught exactly that on the first GPU run of round 2).
template <bool SG>
__global__ __launch_bounds__(256) void lse_rows_twopass_kernel(const float* __restrict__ zbuf, const PairDesc* __restrict__ pairs,
                                                       const SeqDesc* __restrict__ seqs, const int* __restrict__ counts,
                                                       float* __restrict__ rowvec, const float* __restrict__ colvec,
                                                       float* __restrict__ partials, int R) {
    // The workgroup's R rows are staged ONCE into LDS (coalesced 16-byte loads, all in flight together); the row
    // log-sum-exp (phase A) and the column partials (phase B) both run out of LDS, so Z is read from HBM exactly once
    // per Sinkhorn iteration.
    extern __shared__ __attribute__((aligned(16))) float zs[];  // [R][ld] + u[R]
    const int p = blockIdx.y;
    const PairDesc pd = pairs[p];
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    const int rows = m + (SG ? 1 : 0), cols = n + (SG ? 1 : 0);
    const int r0 = blockIdx.x * R;
    if (r0 >= rows) return;
    const int nr = min(R, rows - r0);
    const int ld = pd.ld, ld4 = ld >> 2;
    const float* Z = zbuf + pd.z_off + (size_t)r0 * ld;
    const int vec0 = vec_off(s0, 2 * p), vec1 = vec_off(s1, 2 * p + 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float norm = SG ? -logf((float)m + (float)n) : 0.f;
    const float NEG = neg_inf();
    float* v_s = zs + (size_t)R * ld;  // staged column vector (SuperGlue's v)
    float* u_s = v_s + ld;
    if (SG)
        for (int j = threadIdx.x * 4; j < cols; j += 1024) *reinterpret_cast<f32x4*>(v_s + j) = *reinterpret_cast<const f32x4*>(colvec + vec1 + j);
    // stage: the nr rows are contiguous in memory (row stride ld), nr * ld4 float4 in total
    for (int base = 0; base < nr * ld4; base += 256 * 8) {
        f32x4 t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = base + e * 256 + threadIdx.x;
            t[e] = (idx < nr * ld4) ? reinterpret_cast<const f32x4*>(Z)[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = base + e * 256 + threadIdx.x;
            if (idx < nr * ld4) reinterpret_cast<f32x4*>(zs)[idx] = t[e];
        }
    }
    __syncthreads();
    // phase A: wave w reduces rows w, w+4, ... (two passes over LDS: max, then sum of exp -- as torch.logsumexp does)
    for (int rr = wave; rr < nr; rr += 4) {
        const int i = r0 + rr;
        const float* zr = zs + (size_t)rr * ld;
        float mx = NEG;
#pragma unroll 4
        for (int j = lane * 4; j < cols; j += 256) {
            f32x4 z = *reinterpret_cast<const f32x4*>(zr + j);
            if (SG) z = z + *reinterpret_cast<const f32x4*>(v_s + j);
            mx = fmaxf(mx, z.x);
            if (j + 1 < cols) mx = fmaxf(mx, z.y);
            if (j + 2 < cols) mx = fmaxf(mx, z.z);
            if (j + 3 < cols) mx = fmaxf(mx, z.w);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll 4
        for (int j = lane * 4; j < cols; j += 256) {
            f32x4 z = *reinterpret_cast<const f32x4*>(zr + j);
            if (SG) z = z + *reinterpret_cast<const f32x4*>(v_s + j);
            sum += expf(z.x - mx);
            if (j + 1 < cols) sum += expf(z.y - mx);
            if (j + 2 < cols) sum += expf(z.z - mx);
            if (j + 3 < cols) sum += expf(z.w - mx);
        }
        sum = wave_sum(sum);
        const float lse = logf(sum) + mx;
        float ui;
        if (SG) {
            const float log_mu = (i < m) ? norm : logf((float)n) + norm;
            ui = log_mu - lse;
        } else {
            ui = lse;
        }
        if (lane == 0) {
            rowvec[vec0 + i] = ui;
            u_s[rr] = ui;
        }
    }
    __syncthreads();
    // phase B: column partials (max, sum) of Z + u over this block's rows, 4 columns per thread, from LDS
    float* part = partials + pd.part_off + (size_t)blockIdx.x * ld * 2;
    for (int j = threadIdx.x * 4; j < cols; j += 1024) {
        f32x4 mx = {NEG, NEG, NEG, NEG};
        for (int rr = 0; rr < nr; ++rr) {
            f32x4 v = *reinterpret_cast<const f32x4*>(zs + (size_t)rr * ld + j);
            if (SG) {
                const float u = u_s[rr];
                v.x += u, v.y += u, v.z += u, v.w += u;
            }
            mx.x = fmaxf(mx.x, v.x), mx.y = fmaxf(mx.y, v.y), mx.z = fmaxf(mx.z, v.z), mx.w = fmaxf(mx.w, v.w);
        }
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
        for (int rr = 0; rr < nr; ++rr) {
            f32x4 v = *reinterpret_cast<const f32x4*>(zs + (size_t)rr * ld + j);
            if (SG) {
                const float u = u_s[rr];
                v.x += u, v.y += u, v.z += u, v.w += u;
            }
            sum.x += expf(v.x - mx.x), sum.y += expf(v.y - mx.y), sum.z += expf(v.z - mx.z), sum.w += expf(v.w - mx.w);
        }
        // columns beyond `cols` (row padding) produce garbage partials that are never read
        *reinterpret_cast<f32x4*>(part + (size_t)j * 2) = f32x4{mx.x, sum.x, mx.y, sum.y};
        *reinterpret_cast<f32x4*>(part + (size_t)j * 2 + 4) = f32x4{mx.z, sum.z, mx.w, sum.w};
    }
}

// Combine the row-block partials of 64 columns: 4 thread groups stride over the blocks, then merge through LDS.
template <bool SG>
__global__ __launch_bounds__(256) void lse_cols_kernel(const PairDesc* __restrict__ pairs, const SeqDesc* __restrict__ seqs,
                                                       const int* __restrict__ counts, const float* __restrict__ partials,
                                                       float* __restrict__ colvec, int R) {
    __shared__ float pm[4][64], ps[4][64];
    const int p = blockIdx.y;
    const PairDesc pd = pairs[p];
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    const int rows = m + (SG ? 1 : 0), cols = n + (SG ? 1 : 0);
    if (blockIdx.x * 64 >= cols) return;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const int nblk = ceil_div(rows, R);
    float mx = neg_inf(), sum = 0.f;
    if (j < cols) {
        const float* part = partials + pd.part_off + (size_t)j * 2;
        const size_t bstride = (size_t)pd.ld * 2;
#pragma unroll 4
        for (int b = grp; b < nblk; b += 4) {
            const float2 ms = *reinterpret_cast<const float2*>(part + b * bstride);
            const float nm = fmaxf(mx, ms.x);
            sum = sum * expf(mx - nm) + ms.y * expf(ms.x - nm);
            mx = nm;
        }
    }
    pm[grp][lane] = mx;
    ps[grp][lane] = sum;
    __syncthreads();
    if (grp != 0 || j >= cols) return;
#pragma unroll
    for (int g = 1; g < 4; ++g) {
        const float om = pm[g][lane], os = ps[g][lane];
        const float nm = fmaxf(mx, om);
        if (nm != neg_inf()) sum = sum * expf(mx - nm) + os * expf(om - nm);
        mx = nm;
    }
    // the partial sums carry exp(v_j) (SuperGlue): log sum_i exp(z_ij + u_i) = log(sum) + mx - v_j. A column whose terms
    // all underflowed (> 87 below their rows' maxima) gets the smallest normal number instead of log(0).
    const float lse = logf(fmaxf(sum, 1.17549435e-38f)) + mx;
    const int vec1 = vec_off(s1, 2 * p + 1);
    if (SG) {
        const float norm = -logf((float)m + (float)n);
        const float log_nu = (j < n) ? norm : logf((float)m) + norm;
        colvec[vec1 + j] = log_nu - (lse - colvec[vec1 + j]);
    } else {
        colvec[vec1 + j] = lse;
    }
}

// SuperGlue couplings: dustbin row / column = bin_score, v = 0 (superglue.py:156-160,143)
__global__ void sg_fill_bins_kernel(float* __restrict__ zbuf, const PairDesc* __restrict__ pairs, const SeqDesc* __restrict__ seqs,
                                    const int* __restrict__ counts, float alpha, float* __restrict__ colvec) {
    const int p = blockIdx.y;
    const PairDesc pd = pairs[p];
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    float* Z = zbuf + pd.z_off;
    const int vec1 = vec_off(s1, 2 * p + 1);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t <= max(m, n); t += gridDim.x * blockDim.x) {
        if (t <= m) Z[(size_t)t * pd.ld + n] = alpha;
        if (t <= n) {
            Z[(size_t)m * pd.ld + t] = alpha;
            colvec[vec1 + t] = 0.f;
        }
    }
}

// Row-wise max / first-argmax over the inner m x n block. One wave per row.
template <bool SG>
__global__ __launch_bounds__(256) void best_rows_kernel(const float* __restrict__ zbuf, const PairDesc* __restrict__ pairs,
                                                        const SeqDesc* __restrict__ seqs, const int* __restrict__ counts,
                                                        const float* __restrict__ rowvec, const float* __restrict__ colvec,
                                                        const float* __restrict__ zlogit, float* __restrict__ max0,
                                                        int* __restrict__ idx0) {
    const int p = blockIdx.y;
    const PairDesc pd = pairs[p];
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= m) return;
    const int lane = threadIdx.x & 63;
    const float* zr = zbuf + pd.z_off + (size_t)i * pd.ld;
    const int vec0 = vec_off(s0, 2 * p), vec1 = vec_off(s1, 2 * p + 1);
    const float norm = SG ? -logf((float)m + (float)n) : 0.f;
    const float a_i = rowvec[vec0 + i];
    const float c_i = SG ? 0.f : logsigmoid(zlogit[s0.row_off + i]);
    float best = neg_inf();
    int bj = 0x7fffffff;
    for (int j = lane; j < n; j += 64) {
        const float c_j = SG ? 0.f : logsigmoid(zlogit[s1.row_off + j]);
        const float v = assign_value<SG>(zr[j], a_i, colvec[vec1 + j], norm, c_i, c_j);
        if (v > best) {
            best = v;
            bj = j;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oj = __shfl_xor(bj, off, 64);
        if (ob > best || (ob == best && oj < bj)) {
            best = ob;
            bj = oj;
        }
    }
    if (lane == 0) {
        max0[s0.row_off + i] = best;
        idx0[s0.row_off + i] = (bj == 0x7fffffff) ? 0 : bj;  // all-NaN row: stay in range
    }
}

// Column-wise max / first-argmax. Block = 64 columns x 4 row groups.
template <bool SG>
__global__ __launch_bounds__(256) void best_cols_kernel(const float* __restrict__ zbuf, const PairDesc* __restrict__ pairs,
                                                        const SeqDesc* __restrict__ seqs, const int* __restrict__ counts,
                                                        const float* __restrict__ rowvec, const float* __restrict__ colvec,
                                                        const float* __restrict__ zlogit, int* __restrict__ idx1) {
    __shared__ float bv[4][64];
    __shared__ int bi[4][64];
    const int p = blockIdx.y;
    const PairDesc pd = pairs[p];
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    if (blockIdx.x * 64 >= n) return;
    const float* Z = zbuf + pd.z_off;
    const int vec0 = vec_off(s0, 2 * p), vec1 = vec_off(s1, 2 * p + 1);
    const float norm = SG ? -logf((float)m + (float)n) : 0.f;
    float best = neg_inf();
    int bidx = 0x7fffffff;
    if (j < n) {
        const float b_j = colvec[vec1 + j];
        const float c_j = SG ? 0.f : logsigmoid(zlogit[s1.row_off + j]);
        for (int i = rg; i < m; i += 4) {
            const float c_i = SG ? 0.f : logsigmoid(zlogit[s0.row_off + i]);
            const float v = assign_value<SG>(Z[(size_t)i * pd.ld + j], rowvec[vec0 + i], b_j, norm, c_i, c_j);
            if (v > best) {
                best = v;
                bidx = i;
            }
        }
    }
    bv[rg][lane] = best;
    bi[rg][lane] = bidx;
    __syncthreads();
    if (rg == 0 && j < n) {
#pragma unroll
        for (int g = 1; g < 4; ++g) {
            const float ob = bv[g][lane];
            const int oi = bi[g][lane];
            if (ob > best || (ob == best && oi < bidx)) {
                best = ob;
                bidx = oi;
            }
        }
        idx1[s1.row_off + j] = (bidx == 0x7fffffff) ? 0 : bidx;  // all-NaN column: stay in range
    }
}

// Mutual check, exp, threshold (superglue.py:268-276; LightGlue filter_matches). matches are -1 when invalid.
__global__ void mutual_matches_kernel(const SeqDesc* __restrict__ seqs, const int* __restrict__ counts, const float* __restrict__ max0,
                                      const int* __restrict__ idx0, const int* __restrict__ idx1, float threshold,
                                      int* __restrict__ matches, float* __restrict__ mscores) {
    const int p = blockIdx.y;
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < max(m, n); t += gridDim.x * blockDim.x) {
        if (t < m) {
            const int j = idx0[s0.row_off + t];
            const bool mutual = idx1[s1.row_off + j] == t;
            const float mx = max0[s0.row_off + t];
            float ms = mutual ? expf(mx) : 0.f;
            // a row whose best log-score is not a finite number saw NaN / inf arithmetic (an operand beyond fp16's range under the opt-in f16x2 switches,
            // non-finite inputs): its score says so instead of reading "unmatched" (the engines raise on it under f16x2: check_split_arithmetic_range)
            if (!(fabsf(mx) <= 3.402823466e+38f)) ms = __builtin_nanf("");
            const bool valid = mutual && (ms > threshold);
            matches[s0.row_off + t] = valid ? j : -1;
            mscores[s0.row_off + t] = ms;
        }
        if (t < n) {
            const int i = idx1[s1.row_off + t];
            const bool mutual1 = idx0[s0.row_off + i] == t;
            const float ms0 = mutual1 ? expf(max0[s0.row_off + i]) : 0.f;  // mutual1 implies mutual0(i)
            const bool valid = mutual1 && (ms0 > threshold);
            matches[s1.row_off + t] = valid ? i : -1;
            mscores[s1.row_off + t] = ms0;
        }
    }
}

// Materialise the final log-assignment matrix (parity tests only): out has the layout of zbuf.
template <bool SG>
__global__ void materialize_assignment_kernel(const float* __restrict__ zbuf, const PairDesc* __restrict__ pairs,
                                              const SeqDesc* __restrict__ seqs, const int* __restrict__ counts,
                                              const float* __restrict__ rowvec, const float* __restrict__ colvec,
                                              const float* __restrict__ zlogit, float* __restrict__ out) {
    const int p = blockIdx.z;
    const PairDesc pd = pairs[p];
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    const int rows = m + (SG ? 1 : 0), cols = n + (SG ? 1 : 0);
    const int i = blockIdx.y, vec0 = vec_off(s0, 2 * p), vec1 = vec_off(s1, 2 * p + 1);
    if (i >= rows) return;
    const float norm = SG ? -logf((float)m + (float)n) : 0.f;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < cols; j += gridDim.x * blockDim.x) {
        const float c_i = SG ? 0.f : logsigmoid(zlogit[s0.row_off + i]);
        const float c_j = SG ? 0.f : logsigmoid(zlogit[s1.row_off + j]);
        out[pd.z_off + (size_t)i * pd.ld + j] =
            assign_value<SG>(zbuf[pd.z_off + (size_t)i * pd.ld + j], rowvec[vec0 + i], colvec[vec1 + j], norm, c_i, c_j);
    }
}

int launch_materialize_assignment(const SweepArgs& a, int superglue, const float* zlogit, float* out, hipStream_t stream) {
    if (a.npairs <= 0) return GTSFM_OK;
    const int ext = superglue ? 1 : 0;
    dim3 grid(ceil_div(a.max_n + ext, 256), a.max_m + ext, a.npairs);
    if (superglue)
        hipLaunchKernelGGL(materialize_assignment_kernel<true>, grid, dim3(256), 0, stream, a.zbuf, a.pairs, a.seqs, a.counts, a.rowvec,
                           a.colvec, zlogit, out);
    else
        hipLaunchKernelGGL(materialize_assignment_kernel<false>, grid, dim3(256), 0, stream, a.zbuf, a.pairs, a.seqs, a.counts, a.rowvec,
                           a.colvec, zlogit, out);
    GTSFM_CHECK_LAUNCH("materialize_assignment_kernel");
    return GTSFM_OK;
}

template <bool SG>
static int sweep_impl(const SweepArgs& a, int iters, hipStream_t stream) {
    if (a.npairs <= 0) return GTSFM_OK;
    const int ext = SG ? 1 : 0;
    const int R = sweep_rows_per_block(a.max_n + ext);
    const int ld_max = (a.max_n + ext + 3) / 4 * 4;
    const size_t lds_bytes = ((size_t)(R + 1) * ld_max + R) * sizeof(float);
    dim3 grid_rows(ceil_div(a.max_m + ext, R), a.npairs);
    dim3 grid_cols(ceil_div(a.max_n + ext, 64), a.npairs);
    for (int it = 0; it < iters; ++it) {
        if (SG)
            hipLaunchKernelGGL(lse_rows_kernel<true>, grid_rows, dim3(256), lds_bytes, stream, a.zbuf, a.pairs, a.seqs, a.counts, a.rowvec,
                               a.colvec, a.partials, R);
        else
            hipLaunchKernelGGL(lse_rows_twopass_kernel<false>, grid_rows, dim3(256), lds_bytes, stream, a.zbuf, a.pairs, a.seqs, a.counts,
                               a.rowvec, a.colvec, a.partials, R);
        hipLaunchKernelGGL(lse_cols_kernel<SG>, grid_cols, dim3(256), 0, stream, a.pairs, a.seqs, a.counts, a.partials, a.colvec, R);
    }
    GTSFM_CHECK_LAUNCH("lse_rows/cols_kernel");
    return GTSFM_OK;
}

int launch_sinkhorn_lds(const SweepArgs& a, float bin_score, int iters, hipStream_t stream) {
    if (a.npairs <= 0) return GTSFM_OK;
    int rc = launch_sg_fill_bins(a, bin_score, stream);
    if (rc != GTSFM_OK) return rc;
    return sweep_impl<true>(a, iters, stream);
}

int launch_sg_fill_bins(const SweepArgs& a, float bin_score, hipStream_t stream) {
    if (a.npairs <= 0) return GTSFM_OK;
    hipLaunchKernelGGL(sg_fill_bins_kernel, dim3(ceil_div(max(a.max_m, a.max_n) + 1, 256), a.npairs), dim3(256), 0, stream, a.zbuf, a.pairs,
                       a.seqs, a.counts, bin_score, a.colvec);
    GTSFM_CHECK_LAUNCH("sg_fill_bins_kernel");
    return GTSFM_OK;
}

int launch_mutual_matches(const SweepArgs& a, float threshold, const float* max0, const int* idx0, const int* idx1, int* matches,
                          float* mscores, hipStream_t stream) {
    if (a.npairs <= 0 || a.max_m <= 0 || a.max_n <= 0) return GTSFM_OK;
    hipLaunchKernelGGL(mutual_matches_kernel, dim3(ceil_div(max(a.max_m, a.max_n), 256), a.npairs), dim3(256), 0, stream, a.seqs, a.counts,
                       max0, idx0, idx1, threshold, matches, mscores);
    GTSFM_CHECK_LAUNCH("mutual_matches_kernel");
    return GTSFM_OK;
}

int launch_double_softmax_lse_lds(const SweepArgs& a, hipStream_t stream) { return sweep_impl<false>(a, 1, stream); }

int launch_extract_matches_lds(const SweepArgs& a, int superglue, const float* zlogit, float threshold, float* max0, int* idx0, int* idx1,
                               int* matches, float* mscores, hipStream_t stream) {
    if (a.npairs <= 0 || a.max_m <= 0 || a.max_n <= 0) return GTSFM_OK;
    dim3 gr(ceil_div(a.max_m, 4), a.npairs), gc(ceil_div(a.max_n, 64), a.npairs);
    if (superglue) {
        hipLaunchKernelGGL(best_rows_kernel<true>, gr, dim3(256), 0, stream, a.zbuf, a.pairs, a.seqs, a.counts, a.rowvec, a.colvec, zlogit, max0, idx0);
        hipLaunchKernelGGL(best_cols_kernel<true>, gc, dim3(256), 0, stream, a.zbuf, a.pairs, a.seqs, a.counts, a.rowvec, a.colvec, zlogit, idx1);
    } else {
        hipLaunchKernelGGL(best_rows_kernel<false>, gr, dim3(256), 0, stream, a.zbuf, a.pairs, a.seqs, a.counts, a.rowvec, a.colvec, zlogit, max0, idx0);
        hipLaunchKernelGGL(best_cols_kernel<false>, gc, dim3(256), 0, stream, a.zbuf, a.pairs, a.seqs, a.counts, a.rowvec, a.colvec, zlogit, idx1);
    }
    GTSFM_CHECK_LAUNCH("best_rows / best_cols kernels");
    return launch_mutual_matches(a, threshold, max0, idx0, idx1, matches, mscores, stream);
}
