// Descriptors and launchers of the non-GEMM matcher kernels (see matcher_kernels.hip).
#pragma once

#include "common.h"

// One keypoint set ("sequence"): rows [row_off, row_off + counts[cnt_idx]) of every internal token-major array and rows
// [in_off, in_off + n) of the caller's input / output arrays. SuperGlue packs sequences back to back (in_off == row_off);
// LightGlue aligns every sequence to 128 rows (cap = rows reserved) so that GEMM tiles can be masked per sequence.
// Pair p consists of sequences 2p (image i1) and 2p + 1 (image i2).
struct SeqDesc {
    int row_off, cnt_idx, H, W, in_off, cap;
};

// Score matrix of one pair: zbuf + z_off, row stride ld; Sinkhorn column partials at partials + part_off.
struct PairDesc {
    long long z_off;
    long long part_off;
    int ld, pad;
};

struct SweepArgs {
    const PairDesc* pairs;  // device
    const SeqDesc* seqs;    // device
    const int* counts;      // device
    int npairs, max_m, max_n;
    float* zbuf;
    float* rowvec;  // [T + 16P + 8]: entry vec(s) + i with vec(s) = align4(row_off(s)) + 8 s
    float* colvec;  // same indexing (a sequence is a "row" set in one role and a "column" set in the other)
    float* partials;
};

#ifdef __HIPCC__
__device__ __forceinline__ float neg_inf() { return -__builtin_inff(); }

// Row / column vectors of a pair live at a 16-byte-aligned offset per sequence (float4 loads of v in the row sweeps).
__device__ __forceinline__ int vec_off(const SeqDesc& sq, int s) { return ((sq.row_off + 3) & ~3) + 8 * s; }

// Element value of the final assignment matrix.
//   SG: ((Z + u_i) + v_j) - norm                                   (superglue.py:147,169)
//   LG: ((sim - rowlse_i) + (sim - collse_j)) + (c0_i + c1_j)      (sigmoid_log_double_softmax)
template <bool SG>
__device__ __forceinline__ float assign_value(float z, float a_i, float b_j, float norm, float c_i, float c_j) {
    if (SG) return ((z + a_i) + b_j) - norm;
    return ((z - a_i) + (z - b_j)) + (c_i + c_j);
}

__device__ __forceinline__ float logsigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
#endif

// dst[r][0:256] = src[r][0:256] for r < rows (row strides lds / ldd floats)
int launch_copy_rows256(const float* src, int lds, float* dst, int ldd, int rows, hipStream_t stream);
int launch_move_blocks(const float* src, const int* src_index, float* dst, const int* dst_index, int nblocks, long long block_floats, hipStream_t stream);
int launch_sg_encode_input(const float* kpts, const float* scores, const SeqDesc* seqs, const int* counts, int nseq, int max_n,
                           float* enc_in, hipStream_t stream);
int launch_lg_posenc(const float* kpts, const SeqDesc* seqs, const int* counts, int nseq, int max_n, const float* Wr, float* enc,
                     hipStream_t stream);
int launch_lg_rotary(float* qkv, int ld, int ncols, const float* enc, const SeqDesc* seqs, const int* counts, int nseq, int max_n,
                     hipStream_t stream);
int launch_layernorm_gelu(float* x, int ld, const SeqDesc* seqs, const int* counts, int nseq, int max_n, const float* gamma,
                          const float* beta, hipStream_t stream);
int launch_lg_heads(const float* x, int ld, const SeqDesc* seqs, const int* counts, int nseq, int max_n, const float* w_conf, float b_conf,
                    const float* w_match, float b_match, float* conf, float* z, float* mval, hipStream_t stream);
// Score-matrix sweeps (sweep_kernels.hip): register-resident rows (a wave per row up to 2048 columns, a workgroup per row up to 10240), LDS-staged rows beyond
int launch_sinkhorn(const SweepArgs& a, float bin_score, int iters, hipStream_t stream);
int launch_double_softmax_lse(const SweepArgs& a, hipStream_t stream);
int launch_extract_matches(const SweepArgs& a, int superglue, const float* zlogit, float threshold, float* max0, int* idx0, int* idx1,
                           int* matches, float* mscores, hipStream_t stream);
// LDS-staged forms (matcher_kernels.hip): any number of columns
int launch_sinkhorn_lds(const SweepArgs& a, float bin_score, int iters, hipStream_t stream);
int launch_double_softmax_lse_lds(const SweepArgs& a, hipStream_t stream);
int launch_extract_matches_lds(const SweepArgs& a, int superglue, const float* zlogit, float threshold, float* max0, int* idx0, int* idx1,
                               int* matches, float* mscores, hipStream_t stream);
int launch_sg_fill_bins(const SweepArgs& a, float bin_score, hipStream_t stream);
int launch_mutual_matches(const SweepArgs& a, float threshold, const float* max0, const int* idx0, const int* idx1, int* matches,
                          float* mscores, hipStream_t stream);
int launch_materialize_assignment(const SweepArgs& a, int superglue, const float* zlogit, float* out, hipStream_t stream);
int sweep_rows_per_block(int max_cols);  // rows of the score matrix one workgroup of the LDS-staged row sweep owns
// rows behind one block of column partials for a batch whose widest pair has max_n columns (+ ext dustbin column): sizes `partials`
int sweep_partial_rows(int max_n, int ext);
