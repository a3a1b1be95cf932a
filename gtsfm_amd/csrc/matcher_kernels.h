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

int launch_sg_encode_input(const float* kpts, const float* scores, const SeqDesc* seqs, const int* counts, int nseq, int max_n,
                           float* enc_in, hipStream_t stream);
int launch_lg_posenc(const float* kpts, const SeqDesc* seqs, const int* counts, int nseq, int max_n, const float* Wr, float* enc,
                     hipStream_t stream);
int launch_lg_rotary(float* qkv, int ld, int ncols, const float* enc, const SeqDesc* seqs, const int* counts, int nseq, int max_n,
                     hipStream_t stream);
int launch_layernorm_gelu(float* x, int ld, const SeqDesc* seqs, const int* counts, int nseq, int max_n, const float* gamma,
                          const float* beta, hipStream_t stream);
int launch_rowdot(const float* x, int ld, const SeqDesc* seqs, const int* counts, int nseq, int max_n, const float* w, float b, int act,
                  float* out, hipStream_t stream);
int launch_sinkhorn(const SweepArgs& a, float bin_score, int iters, hipStream_t stream);
int launch_double_softmax_lse(const SweepArgs& a, hipStream_t stream);
int launch_extract_matches(const SweepArgs& a, int superglue, const float* zlogit, float threshold, float* max0, int* idx0, int* idx1,
                           int* matches, float* mscores, hipStream_t stream);
int launch_materialize_assignment(const SweepArgs& a, int superglue, const float* zlogit, float* out, hipStream_t stream);
int sweep_rows_per_block(int max_cols);  // rows of the score matrix one workgroup of the row sweep owns
