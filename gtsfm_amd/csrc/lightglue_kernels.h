// Launchers of the LightGlue-specific kernels (see lightglue_kernels.hip).
#pragma once

#include "matcher_kernels.h"

int launch_lg_store_rows(const float* X, int ldx, const SeqDesc* seqs, const int* counts, int nseq, int max_n, float* out, hipStream_t stream);
int launch_lg_load_inputs(const float* desc, const SeqDesc* seqs, const int* counts, int nseq, int max_n, float* X, int ldx, int* ind,
                          hipStream_t stream);
int launch_lg_stop_check(const float* conf, const SeqDesc* seqs, int* live, int* final_cnt, int* assign, const int* orig, int* stop_layer,
                         int npairs, int layer, int last_layer, float conf_threshold, float depth_confidence, const int* ind_cur, int* ind_final,
                         hipStream_t stream);
int launch_lg_prune(const float* conf, const float* matchability, const SeqDesc* seqs, int* live, int* old_cnt, int* pos, int nseq,
                    int max_n, float conf_threshold, float keep_threshold, int pruning_threshold, int use_conf, const float* Xs, float* Xd,
                    int ldx, const float* encs, float* encd, const int* inds, int* indd, hipStream_t stream);
int launch_lg_scatter_matches(const SeqDesc* seqs, const int* final_cnt, const int* ind, const int* m_int, const float* ms_int, int nseq,
                              int max_n, int total_out, int* matches, float* mscores, hipStream_t stream);
