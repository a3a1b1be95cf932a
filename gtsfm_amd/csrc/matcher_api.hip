// C-ABI entry points of the matchers: weight-blob packing, batch descriptors, stand-alone attention, and the
// SuperGlue forward pass (replaces thirdparty/SuperGluePretrainedNetwork/models/superglue.py:228-283) for a ragged
// batch of image pairs. See include/gtsfm_amd.h.

#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/gtsfm_amd.h"
#include "attention_kernels.h"
#include "dense_kernels.h"
#include "gemm_batch.h"
#include "lightglue_kernels.h"
#include "matcher_kernels.h"

#define TRY(expr)                          \
    do {                                   \
        int rc_ = (expr);                  \
        if (rc_ != GTSFM_OK) return rc_;   \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------
// Weight blobs: a sequence of entries, each 64-float aligned.
//   kind 0 (linear): packed W[n][k_pad] (pack_linear_weights), the bias padded to a multiple of 64, then W once more
//                    row-major [n][k_pad] for the LDS-DMA GEMM (k_pad = k rounded up to 8, zero-filled)
//   kind 1 (raw)   : n floats copied verbatim
// ---------------------------------------------------------------------------------------------------------------

static size_t round64(size_t x) { return (x + 63) / 64 * 64; }
static int kpad8(int k) { return (k + 7) / 8 * 8; }

static size_t blob_entry_floats(int kind, int n, int k) {
    if (kind == 0) return round64(packed_linear_floats(kpad8(k), n)) + round64(n) + round64((size_t)n * kpad8(k));
    return round64(n);
}

struct BlobCursor {
    const float* base;
    size_t off;
    // linear entry -> (packed W, bias, row-major W); advances
    void linear(int n, int k, const float** w, const float** b, const float** raw) {
        *w = base + off;
        *b = base + off + round64(packed_linear_floats(kpad8(k), n));
        *raw = *b + round64(n);
        off += blob_entry_floats(0, n, k);
    }
    const float* raw(int n) {
        const float* r = base + off;
        off += round64(n);
        return r;
    }
};

extern "C" size_t gtsfm_blob_floats(int count, const int32_t* kinds, const int32_t* n, const int32_t* k) {
    size_t total = 0;
    for (int i = 0; i < count; ++i) total += blob_entry_floats(kinds[i], n[i], k[i]);
    return total;
}

extern "C" int gtsfm_pack_blob(int count, const int32_t* kinds, const int32_t* n, const int32_t* k, const float* const* w,
                               const float* const* b, float* out) {
    GTSFM_CHECK_ARG(count >= 0 && kinds && n && k && w && b && out, "pack_blob: null pointer");
    size_t off = 0;
    for (int i = 0; i < count; ++i) {
        const size_t sz = blob_entry_floats(kinds[i], n[i], k[i]);
        memset(out + off, 0, sz * sizeof(float));
        GTSFM_CHECK_ARG(w[i], "pack_blob: entry %d has no data", i);
        if (kinds[i] == 0) {
            GTSFM_CHECK_ARG(n[i] > 0 && k[i] > 0, "pack_blob: entry %d has bad dims", i);
            pack_linear_weights(w[i], k[i], kpad8(k[i]), n[i], out + off);
            float* bias_at = out + off + round64(packed_linear_floats(kpad8(k[i]), n[i]));
            if (b[i]) memcpy(bias_at, b[i], n[i] * sizeof(float));
            float* raw_at = bias_at + round64(n[i]);
            for (int r = 0; r < n[i]; ++r) memcpy(raw_at + (size_t)r * kpad8(k[i]), w[i] + (size_t)r * k[i], k[i] * sizeof(float));
        } else {
            memcpy(out + off, w[i], n[i] * sizeof(float));
        }
        off += sz;
    }
    return GTSFM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Batch descriptors (int32 block built on the host, uploaded by the caller). P pairs, NT = number of 128-row tiles
// (LightGlue; 0 for SuperGlue):
//   live counts [2P] | final counts [2P] | assign counts [2P] | original counts [2P] | old counts [2P] | stop layer [P]
//   | seqs [2P][6] | pairs [P][6] | self-attention problems [2P][4] | cross-attention problems [2P][4]
//   | score-GEMM problems [P][8] | tile -> count index [NT] | tile -> first row within its sequence [NT]
// ---------------------------------------------------------------------------------------------------------------

struct DescLayout {
    size_t live, final_cnt, assign, orig, old_cnt, stop, seqs, pairs, self_p, cross_p, score_p, tile_idx, tile_row0, total;
};

static DescLayout desc_layout(int P, int NT) {
    DescLayout L;
    size_t o = 0;
    L.live = o, o += (size_t)2 * P;
    L.final_cnt = o, o += (size_t)2 * P;
    L.assign = o, o += (size_t)2 * P;
    L.orig = o, o += (size_t)2 * P;
    L.old_cnt = o, o += (size_t)2 * P;
    L.stop = o, o += (size_t)P;
    o = (o + 1) / 2 * 2;
    L.seqs = o, o += (size_t)12 * P;
    o = (o + 1) / 2 * 2;
    L.pairs = o, o += (size_t)6 * P;
    L.self_p = o, o += (size_t)8 * P;
    L.cross_p = o, o += (size_t)8 * P;
    o = (o + 1) / 2 * 2;
    L.score_p = o, o += (size_t)8 * P;
    L.tile_idx = o, o += (size_t)NT;
    L.tile_row0 = o, o += (size_t)NT;
    L.total = o;
    return L;
}

static int z_ld(int n1, int ext) { return (n1 + ext + 3) / 4 * 4; }
static int cap128(int n) { return (n + 127) / 128 * 128; }

static int count_tiles(int superglue, int P, const int32_t* n0, const int32_t* n1) {
    if (superglue) return 0;
    int nt = 0;
    for (int p = 0; p < P; ++p) nt += cap128(n0[p]) / 128 + cap128(n1[p]) / 128;
    return nt;
}

extern "C" size_t gtsfm_match_desc_ints(int superglue, int npairs, const int32_t* n0, const int32_t* n1) {
    if (npairs <= 0 || !n0 || !n1) return 0;
    return desc_layout(npairs, count_tiles(superglue, npairs, n0, n1)).total;
}

extern "C" int gtsfm_match_build_desc(int superglue, int npairs, const int32_t* n0, const int32_t* n1, const int32_t* hw,
                                      int32_t* out) {
    GTSFM_CHECK_ARG(npairs > 0 && n0 && n1 && hw && out, "match_build_desc: bad arguments");
    const DescLayout L = desc_layout(npairs, count_tiles(superglue, npairs, n0, n1));
    memset(out, 0, L.total * sizeof(int32_t));
    const int ext = superglue ? 1 : 0;
    int row = 0, in_row = 0, tile = 0, max_n1 = 0;
    long long zoff = 0, poff = 0;
    for (int p = 0; p < npairs; ++p) max_n1 = max_n1 > n1[p] ? max_n1 : n1[p];
    const int R = sweep_partial_rows(max_n1, ext);
    for (int p = 0; p < npairs; ++p) {
        // an empty SECOND set is a valid descriptor: the per-image phase (gtsfm_{sg,lg}_forward_phase, phase 1) takes keypoint sets two at a
        // time and an odd number of images leaves the last slot empty (no tiles, no rows, count 0: every kernel skips it). The whole-pair
        // phases reject it themselves.
        GTSFM_CHECK_ARG(n0[p] > 0 && n1[p] >= 0, "match_build_desc: pair %d has an empty keypoint set", p);
        const int ns[2] = {n0[p], n1[p]};
        int offs[2];
        out[L.stop + p] = -1;
        for (int s = 0; s < 2; ++s) {
            const int si = 2 * p + s;
            const int cap = superglue ? ns[s] : cap128(ns[s]);
            out[L.live + si] = out[L.orig + si] = ns[s];
            if (superglue) out[L.final_cnt + si] = ns[s];
            SeqDesc sd = {row, si, hw[4 * p + 2 * s], hw[4 * p + 2 * s + 1], in_row, cap};
            memcpy(out + L.seqs + 6 * si, &sd, sizeof(sd));
            offs[s] = row;
            if (!superglue)
                for (int t = 0; t < cap / 128; ++t, ++tile) {
                    out[L.tile_idx + tile] = si;
                    out[L.tile_row0 + tile] = t * 128;
                }
            row += cap;
            in_row += ns[s];
        }
        PairDesc pd;
        pd.z_off = zoff, pd.part_off = poff, pd.ld = z_ld(n1[p], ext), pd.pad = 0;
        memcpy(out + L.pairs + 6 * p, &pd, sizeof(pd));
        GemmProblem gp = {zoff, offs[0], offs[1], 2 * p, 2 * p + 1, pd.ld, 0};  // scores of pair p: rows of image 0 x rows of image 1 -> Z
        memcpy(out + L.score_p + 8 * p, &gp, sizeof(gp));
        zoff += (long long)(n0[p] + ext) * pd.ld;
        poff += (long long)ceil_div(n0[p] + ext, R) * pd.ld * 2;
        for (int s = 0; s < 2; ++s) {
            AttnProblem self = {offs[s], 2 * p + s, offs[s], 2 * p + s};
            AttnProblem cross = {offs[s], 2 * p + s, offs[1 - s], 2 * p + 1 - s};
            memcpy(out + L.self_p + 4 * (2 * p + s), &self, sizeof(self));
            memcpy(out + L.cross_p + 4 * (2 * p + s), &cross, sizeof(cross));
        }
    }
    return GTSFM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Feature-table plumbing of the batched pipeline: the keypoint sets of a chunk's images are gathered from / scattered into the
// device-resident table [images][max_keypoints][...] by image index
// ---------------------------------------------------------------------------------------------------------------

extern "C" int gtsfm_move_blocks_f32(const float* src_dev, const int32_t* src_index_dev, float* dst_dev, const int32_t* dst_index_dev, int nblocks,
                                     int64_t block_floats, void* stream) {
    GTSFM_CHECK_ARG(src_dev && dst_dev && nblocks >= 0 && nblocks <= 65535 && block_floats >= 0 && block_floats % 2 == 0, "move_blocks: bad arguments (at most 65535 blocks of an even number of floats)");
    GTSFM_CHECK_ARG(((uintptr_t)src_dev | (uintptr_t)dst_dev) % 8 == 0, "move_blocks: arrays must be 8-byte aligned");
    return launch_move_blocks(src_dev, src_index_dev, dst_dev, dst_index_dev, nblocks, block_floats, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Stand-alone attention (parity tests)
// ---------------------------------------------------------------------------------------------------------------

extern "C" int gtsfm_attention_f32(const float* q_dev, int ldq, const float* k_dev, int ldk, const float* v_dev, int ldv, float* out_dev,
                                   int ldo, const int32_t* problems_dev, const int32_t* counts_dev, int nproblems, int max_q, int heads,
                                   float scale, void* stream) {
    // no workspace, key counts unknown to the host (max_k = 0): fused schedule, merged state parked in LDS
    return gtsfm_attention_split_f32(q_dev, ldq, k_dev, ldk, v_dev, ldv, out_dev, ldo, problems_dev, counts_dev, nproblems, max_q, 0, heads, scale, -1,
                                     0, nullptr, 0, stream);
}

extern "C" size_t gtsfm_attention_split_workspace_bytes(int nproblems, int max_q, int max_k, int heads, size_t rows) {
    // enough for either schedule: the split schedule's partial states (one unnormalised O row + (m, l) per head, key segment and
    // token row) or the parking space of the fused schedule's double-buffered build (34 floats per thread of the launch; the default
    // build parks in LDS and ignores it)
    const int nseg = attention_segments(max_k);
    const size_t split = (size_t)nseg * rows * ((size_t)heads * 64 + (size_t)heads * 2);
    const size_t park = (size_t)((heads * nproblems + 7) / 8 * 8) * ((max_q + 127) / 128) * 34 * 256;
    return (split > park ? split : park) * sizeof(float);
}

extern "C" int gtsfm_attention_split_f32(const float* q_dev, int ldq, const float* k_dev, int ldk, const float* v_dev, int ldv, float* out_dev,
                                         int ldo, const int32_t* problems_dev, const int32_t* counts_dev, int nproblems, int max_q, int max_k,
                                         int heads, float scale, int mode, size_t rows, void* workspace_dev, size_t workspace_bytes, void* stream) {
    return gtsfm_attention_math_f32(q_dev, ldq, k_dev, ldk, v_dev, ldv, out_dev, ldo, problems_dev, counts_dev, nproblems, max_q, max_k, heads, scale, mode,
                                    0, rows, workspace_dev, workspace_bytes, stream);
}

extern "C" size_t gtsfm_attention_math_workspace_bytes(int nproblems, int max_q, int max_k, int heads, size_t rows, int math) {
    // either schedule of the chosen arithmetic: the larger of the split schedule's partial states and the fused schedule's parking space,
    // plus (bf16x3 / f16x2) the split K / V^T tiles: 3 / 2 pieces of 8 KiB per 64-key tile, tensor, head and problem
    if (math != ATTN_MATH_BF16X3 && math != ATTN_MATH_F16X2) return gtsfm_attention_split_workspace_bytes(nproblems, max_q, max_k, heads, rows);
    const size_t key_tiles = (size_t)((max_k < 1 ? 1 : max_k) + 63) / 64;
    const size_t tiles = (size_t)(math == ATTN_MATH_F16X2 ? 4 : 6) * heads * nproblems * key_tiles * 8192;
    return tiles + gtsfm_attention_split_workspace_bytes(nproblems, max_q, max_k, heads, rows);
}

extern "C" int gtsfm_attention_math_f32(const float* q_dev, int ldq, const float* k_dev, int ldk, const float* v_dev, int ldv, float* out_dev,
                                        int ldo, const int32_t* problems_dev, const int32_t* counts_dev, int nproblems, int max_q, int max_k,
                                        int heads, float scale, int mode, int math, size_t rows, void* workspace_dev, size_t workspace_bytes, void* stream) {
    GTSFM_CHECK_ARG(q_dev && k_dev && v_dev && out_dev && problems_dev && counts_dev, "attention: null pointer");
    GTSFM_CHECK_ARG(mode >= -1 && mode <= 1, "attention: mode is -1 (fused), 0 (by launch geometry) or 1 (split)");
    GTSFM_CHECK_ARG(math >= 0 && math <= 2, "attention: math is 0 (exact fp32), 1 (bf16x3) or 2 (f16x2)");
    AttnParams p = {};
    p.q = q_dev, p.ldq = ldq, p.k = k_dev, p.ldk = ldk, p.v = v_dev, p.ldv = ldv, p.out = out_dev, p.ldo = ldo;
    p.problems = (const AttnProblem*)problems_dev, p.counts = counts_dev, p.scale = scale, p.heads = heads;
    p.max_k = max_k, p.force_split = mode, p.workspace = (float*)workspace_dev, p.workspace_floats = workspace_bytes / sizeof(float), p.part_rows = rows;
    p.math = math;
    return launch_attention(p, nproblems, max_q, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// SuperGlue
// ---------------------------------------------------------------------------------------------------------------

namespace {

struct BatchDims {
    int P, T, max_n, max_n0, max_n1;
    size_t z_floats, part_floats, pack_floats;
};

BatchDims batch_dims(int P, const int32_t* n0, const int32_t* n1, int ext) {
    BatchDims d = {P, 0, 0, 0, 0, 0, 0, 0};
    int mx1 = 0;
    for (int p = 0; p < P; ++p) mx1 = mx1 > n1[p] ? mx1 : n1[p];
    const int R = sweep_partial_rows(mx1, ext);
    for (int p = 0; p < P; ++p) {
        d.T += n0[p] + n1[p];
        d.max_n0 = d.max_n0 > n0[p] ? d.max_n0 : n0[p];
        d.max_n1 = d.max_n1 > n1[p] ? d.max_n1 : n1[p];
        const int ld = z_ld(n1[p], ext);
        d.z_floats += (size_t)(n0[p] + ext) * ld;
        d.part_floats += (size_t)ceil_div(n0[p] + ext, R) * ld * 2;
        const size_t pk = packed_linear_floats(256, n1[p]);
        d.pack_floats = d.pack_floats > pk ? d.pack_floats : pk;
    }
    d.max_n = d.max_n0 > d.max_n1 ? d.max_n0 : d.max_n1;
    return d;
}

struct SgWorkspace {
    size_t enc_in, ka, kb, x, qkv, mlp, md, pack, z, part, uv_row, uv_col, max0, idx0, idx1, attn, attn_floats, total;
};

SgWorkspace sg_workspace_layout(const BatchDims& d, int attn_math) {
    SgWorkspace w;
    size_t o = 0;
    auto take = [&](size_t floats) {
        size_t r = o;
        o += align_up(floats * 4, 256);
        return r;
    };
    const size_t T = d.T;
    w.enc_in = take(T * 8);
    w.ka = take(T * 256);
    w.kb = take(T * 256);
    w.x = take(T * 512);
    w.qkv = take(T * 768);
    w.mlp = take(T * 512);
    w.md = take(T * 256);
    w.pack = take(d.pack_floats);
    w.z = take(d.z_floats);
    w.part = take(d.part_floats);
    w.uv_row = take(T + 16 * d.P + 8);
    w.uv_col = take(T + 16 * d.P + 8);
    w.max0 = take(T);
    w.idx0 = take(T);
    w.idx1 = take(T);
    w.attn_floats = attention_workspace_floats(2 * d.P, 4, d.max_n, d.max_n, T, attn_math);  // split partials (small batches) or fused parking space; 0 below 1025 keypoints
    w.attn = take(w.attn_floats);
    w.total = o;
    return w;
}

}  // namespace

extern "C" size_t gtsfm_sg_workspace_bytes(int npairs, const int32_t* n0, const int32_t* n1) {
    if (npairs <= 0 || !n0 || !n1) return 256;
    return sg_workspace_layout(batch_dims(npairs, n0, n1, 1), attention_math_from_env()).total;
}

// phase 0: the whole forward. phase 1: only what depends on ONE image -- keypoint encoder and the first (self) GNN layer --
// written to x_out_dev [T][256] in the input's row order. phase 2: descriptors_dev holds that result; the encoder and the first
// layer are skipped. An image that takes part in many pairs runs phase 1 once (FrontEndPipeline); the arithmetic per row is the
// same in either split, so the results are bit-identical to phase 0.
static int sg_forward_phased(const float* wts, int num_layers, float bin_score, int npairs, const int32_t* n0, const int32_t* n1,
                             const int32_t* desc_dev, const float* kpts_dev, const float* scores_dev, const float* descriptors_dev,
                             int sinkhorn_iters, float match_threshold, void* workspace_dev, size_t workspace_bytes,
                             int32_t* matches_dev, float* mscores_dev, float* ot_dev, int phase, float* x_out_dev, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GTSFM_CHECK_ARG(wts && n0 && n1 && desc_dev && descriptors_dev && workspace_dev, "sg_forward: null pointer");
    GTSFM_CHECK_ARG(phase == 1 ? x_out_dev != nullptr : (matches_dev && mscores_dev), "sg_forward: null output");
    GTSFM_CHECK_ARG(phase == 2 || (kpts_dev && scores_dev), "sg_forward: null keypoints / scores");
    GTSFM_CHECK_ARG(phase >= 0 && phase <= 2 && (phase == 0 || num_layers >= 1), "sg_forward: bad phase");
    GTSFM_CHECK_ARG(npairs > 0 && num_layers >= 0 && sinkhorn_iters >= 0, "sg_forward: bad arguments");
    for (int p = 0; p < npairs; ++p) GTSFM_CHECK_ARG(n0[p] > 0 && (n1[p] > 0 || (phase == 1 && n1[p] == 0)), "sg_forward: pair %d has an empty keypoint set", p);
    const BatchDims d = batch_dims(npairs, n0, n1, 1);
    const int attn_math = attention_math_from_env();  // GTSFM_ATTENTION_MATH, read per call (sizing and launches of one call agree)
    const int gemm_math = gemm_math_from_env();       // GTSFM_GEMM_MATH: the matchers' projection / FFN / score GEMMs only
    const SgWorkspace ws = sg_workspace_layout(d, attn_math);
    if (workspace_bytes < ws.total) {
        gtsfm_set_error("sg_forward: workspace too small (%zu < %zu bytes)", workspace_bytes, ws.total);
        return GTSFM_ERR_WORKSPACE;
    }
    const DescLayout DL = desc_layout(npairs, 0);
    const int* counts = desc_dev + DL.live;
    const SeqDesc* seqs = (const SeqDesc*)(desc_dev + DL.seqs);
    const PairDesc* pairs = (const PairDesc*)(desc_dev + DL.pairs);
    const AttnProblem* self_p = (const AttnProblem*)(desc_dev + DL.self_p);
    const AttnProblem* cross_p = (const AttnProblem*)(desc_dev + DL.cross_p);
    char* wsp = (char*)workspace_dev;
    float* enc_in = (float*)(wsp + ws.enc_in);
    float* ka = (float*)(wsp + ws.ka);
    float* kb = (float*)(wsp + ws.kb);
    float* X = (float*)(wsp + ws.x);
    float* QKV = (float*)(wsp + ws.qkv);
    float* MLP = (float*)(wsp + ws.mlp);
    float* MD = (float*)(wsp + ws.md);
    float* PACK = (float*)(wsp + ws.pack);
    float* Z = (float*)(wsp + ws.z);
    float* PART = (float*)(wsp + ws.part);
    float* rowvec = (float*)(wsp + ws.uv_row);
    float* colvec = (float*)(wsp + ws.uv_col);
    float* max0 = (float*)(wsp + ws.max0);
    int* idx0 = (int*)(wsp + ws.idx0);
    int* idx1 = (int*)(wsp + ws.idx1);
    const int T = d.T;

    BlobCursor cur = {wts, 0};
    auto gemm = [&](const float* A, int lda, int K, int N, float* C, int ldc, int coff, const float* res, int ldres, int relu) -> int {
        const float *w, *b, *raw;
        cur.linear(N, K, &w, &b, &raw);
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = A, g.lda = lda, g.M = T, g.K = kpad8(K), g.wpack = w, g.wraw = raw, g.ldw = kpad8(K), g.bias = b, g.N = N;
        g.C = C, g.ldc = ldc, g.c_coff = coff, g.res = res, g.ldres = ldres, g.alpha = 1.0f, g.relu = relu;
        g.math = gemm_math;
        return launch_gemm(g, stream);
    };

    auto skip = [&](int N, int K) {
        const float *w, *b, *raw;
        cur.linear(N, K, &w, &b, &raw);
    };
    if (phase == 2) {  // x after the encoder and the first layer comes in; step over their weights
        skip(32, 3), skip(64, 32), skip(128, 64), skip(256, 128), skip(256, 256);
        skip(768, 256), skip(512, 512), skip(256, 512);
        TRY(launch_copy_rows256(descriptors_dev, 256, X, 512, T, stream));
    } else {
        // keypoint encoder (superglue.py:73-82, BatchNorm folded into the convolutions at load time)
        TRY(launch_sg_encode_input(kpts_dev, scores_dev, seqs, counts, 2 * npairs, d.max_n, enc_in, stream));
        TRY(gemm(enc_in, 8, 3, 32, ka, 256, 0, nullptr, 0, 1));
        TRY(gemm(ka, 256, 32, 64, kb, 256, 0, nullptr, 0, 1));
        TRY(gemm(kb, 256, 64, 128, ka, 256, 0, nullptr, 0, 1));
        TRY(gemm(ka, 256, 128, 256, kb, 256, 0, nullptr, 0, 1));
        TRY(gemm(kb, 256, 256, 256, X, 512, 0, descriptors_dev, 256, 0));  // desc + kenc(kpts, scores)
    }

    // attentional GNN (superglue.py:122-138): alternating self / cross layers, both images per launch
    for (int l = (phase == 2 ? 1 : 0); l < (phase == 1 ? 1 : num_layers); ++l) {
        TRY(gemm(X, 512, 256, 768, QKV, 768, 0, nullptr, 0, 0));
        AttnParams ap = {};
        // the attention output lands in the second half of cat([x, .]); attn.merge is folded into mlp.0 at load time
        ap.q = QKV, ap.ldq = 768, ap.k = QKV + 256, ap.ldk = 768, ap.v = QKV + 512, ap.ldv = 768, ap.out = X + 256, ap.ldo = 512;
        ap.problems = (l % 2 == 0) ? self_p : cross_p, ap.counts = counts, ap.scale = 0.125f, ap.heads = 4;
        ap.max_k = d.max_n, ap.workspace = ws.attn_floats ? (float*)(wsp + ws.attn) : nullptr, ap.workspace_floats = ws.attn_floats, ap.part_rows = T;
        ap.math = attn_math;
        TRY(launch_attention(ap, 2 * npairs, d.max_n, stream));
        TRY(gemm(X, 512, 512, 512, MLP, 512, 0, nullptr, 0, 1));       // mlp.0 (+BN, merge folded) + ReLU
        TRY(gemm(MLP, 512, 512, 256, X, 512, 0, X, 512, 0));           // mlp.3, desc += delta
    }
    if (phase == 1) {
        TRY(launch_copy_rows256(X, 512, x_out_dev, 256, T, stream));
        return GTSFM_OK;
    }
    TRY(gemm(X, 512, 256, 256, MD, 256, 0, nullptr, 0, 0));  // final_proj

    // scores = mdesc0^T mdesc1 / sqrt(256) into the (m+1) x (n+1) couplings matrix (superglue.py:257-258,156-160): one ragged
    // launch for all pairs when the LDS-DMA GEMM is in use (image 1's descriptor rows are its "weights" as they are)
    if (gemm_uses_dma(256, 256)) {
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = MD, g.lda = 256, g.M = d.max_n0, g.K = 256, g.wraw = MD, g.ldw = 256, g.N = d.max_n1, g.C = Z, g.alpha = 0.0625f;
        g.math = gemm_math;
        GemmBatch bt = {(const GemmProblem*)(desc_dev + DL.score_p), counts, npairs};
        TRY(launch_gemm_dma_batched(g, bt, stream));
    } else {
        int row = 0;
        size_t zoff = 0;
        for (int p = 0; p < npairs; ++p) {
            const int r0 = row, r1 = row + n0[p];
            const int ld = z_ld(n1[p], 1);
            TRY(launch_pack_rows(MD + (size_t)r1 * 256, 256, n1[p], nullptr, 256, PACK, stream));
            GemmParams g;
            memset(&g, 0, sizeof(g));
            g.A = MD + (size_t)r0 * 256, g.lda = 256, g.M = n0[p], g.K = 256, g.wpack = PACK, g.bias = nullptr, g.N = n1[p];
            g.C = Z + zoff, g.ldc = ld, g.alpha = 0.0625f;
            TRY(launch_gemm(g, stream));
            zoff += (size_t)(n0[p] + 1) * ld;
            row += n0[p] + n1[p];
        }
    }
    SweepArgs sa;
    sa.pairs = pairs, sa.seqs = seqs, sa.counts = counts, sa.npairs = npairs, sa.max_m = d.max_n0, sa.max_n = d.max_n1;
    sa.zbuf = Z, sa.rowvec = rowvec, sa.colvec = colvec, sa.partials = PART;
    TRY(launch_sinkhorn(sa, bin_score, sinkhorn_iters, stream));
    if (sinkhorn_iters == 0) {  // u stays 0 (superglue.py:143)
        if (hipMemsetAsync(rowvec, 0, sizeof(float) * (T + 16 * npairs + 8), stream) != hipSuccess) return GTSFM_ERR_HIP;
    }
    TRY(launch_extract_matches(sa, 1, nullptr, match_threshold, max0, idx0, idx1, matches_dev, mscores_dev, stream));
    if (ot_dev) TRY(launch_materialize_assignment(sa, 1, nullptr, ot_dev, stream));
    return GTSFM_OK;
}

extern "C" int gtsfm_sg_forward(const float* wts, int num_layers, float bin_score, int npairs, const int32_t* n0, const int32_t* n1,
                                const int32_t* desc_dev, const float* kpts_dev, const float* scores_dev, const float* descriptors_dev,
                                int sinkhorn_iters, float match_threshold, void* workspace_dev, size_t workspace_bytes,
                                int32_t* matches_dev, float* mscores_dev, float* ot_dev, void* stream_) {
    return sg_forward_phased(wts, num_layers, bin_score, npairs, n0, n1, desc_dev, kpts_dev, scores_dev, descriptors_dev, sinkhorn_iters,
                             match_threshold, workspace_dev, workspace_bytes, matches_dev, mscores_dev, ot_dev, 0, nullptr, stream_);
}

extern "C" int gtsfm_sg_forward_phase(const float* wts, int num_layers, float bin_score, int npairs, const int32_t* n0, const int32_t* n1,
                                      const int32_t* desc_dev, const float* kpts_dev, const float* scores_dev, const float* descriptors_dev,
                                      int sinkhorn_iters, float match_threshold, void* workspace_dev, size_t workspace_bytes,
                                      int32_t* matches_dev, float* mscores_dev, float* ot_dev, int phase, float* x_out_dev, void* stream_) {
    return sg_forward_phased(wts, num_layers, bin_score, npairs, n0, n1, desc_dev, kpts_dev, scores_dev, descriptors_dev, sinkhorn_iters,
                             match_threshold, workspace_dev, workspace_bytes, matches_dev, mscores_dev, ot_dev, phase, x_out_dev, stream_);
}

// ---------------------------------------------------------------------------------------------------------------
// Stand-alone Sinkhorn (parity tests against torch.logsumexp iterations; bench.py's roofline of the sweep kernels)
// ---------------------------------------------------------------------------------------------------------------

namespace {

struct SkWorkspace {
    size_t desc, part, uv_row, uv_col, total;
};

SkWorkspace sk_workspace_layout(int P, const int32_t* m, const int32_t* n) {
    const BatchDims d = batch_dims(P, m, n, 1);
    SkWorkspace w;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t r = o;
        o += align_up(bytes, 256);
        return r;
    };
    w.desc = take(desc_layout(P, 0).total * sizeof(int32_t));
    w.part = take(d.part_floats * 4);
    w.uv_row = take(((size_t)d.T + 16 * P + 8) * 4);
    w.uv_col = take(((size_t)d.T + 16 * P + 8) * 4);
    w.total = o;
    return w;
}

__global__ void sk_copy_vectors_kernel(const SeqDesc* __restrict__ seqs, const int* __restrict__ counts, const float* __restrict__ rowvec,
                                       const float* __restrict__ colvec, int stride_u, int stride_v, float* __restrict__ u, float* __restrict__ v) {
    const int p = blockIdx.y;
    const SeqDesc s0 = seqs[2 * p], s1 = seqs[2 * p + 1];
    const int m = counts[s0.cnt_idx], n = counts[s1.cnt_idx];
    const int vec0 = vec_off(s0, 2 * p), vec1 = vec_off(s1, 2 * p + 1);
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t <= max(m, n); t += gridDim.x * blockDim.x) {
        if (t <= m) u[(size_t)p * stride_u + t] = rowvec[vec0 + t];
        if (t <= n) v[(size_t)p * stride_v + t] = colvec[vec1 + t];
    }
}

}  // namespace

extern "C" size_t gtsfm_sinkhorn_workspace_bytes(int npairs, const int32_t* m, const int32_t* n) {
    if (npairs <= 0 || !m || !n) return 256;
    return sk_workspace_layout(npairs, m, n).total;
}

extern "C" int gtsfm_sinkhorn_f32(float* z_dev, int npairs, const int32_t* m, const int32_t* n, float bin_score, int iters,
                                  void* workspace_dev, size_t workspace_bytes, float* u_dev, float* v_dev, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GTSFM_CHECK_ARG(z_dev && m && n && workspace_dev && u_dev && v_dev, "sinkhorn: null pointer");
    GTSFM_CHECK_ARG(npairs > 0 && iters >= 0, "sinkhorn: bad arguments");
    const SkWorkspace ws = sk_workspace_layout(npairs, m, n);
    if (workspace_bytes < ws.total) {
        gtsfm_set_error("sinkhorn: workspace too small (%zu < %zu bytes)", workspace_bytes, ws.total);
        return GTSFM_ERR_WORKSPACE;
    }
    const BatchDims d = batch_dims(npairs, m, n, 1);
    const DescLayout DL = desc_layout(npairs, 0);
    std::vector<int32_t> host(DL.total), hw((size_t)4 * npairs, 1);
    TRY(gtsfm_match_build_desc(1, npairs, m, n, hw.data(), host.data()));
    char* wsp = (char*)workspace_dev;
    int32_t* desc_dev = (int32_t*)(wsp + ws.desc);
    if (hipMemcpyAsync(desc_dev, host.data(), DL.total * sizeof(int32_t), hipMemcpyHostToDevice, stream) != hipSuccess) return GTSFM_ERR_HIP;
    if (hipStreamSynchronize(stream) != hipSuccess) return GTSFM_ERR_HIP;  // `host` goes out of scope below
    SweepArgs sa;
    sa.pairs = (const PairDesc*)(desc_dev + DL.pairs), sa.seqs = (const SeqDesc*)(desc_dev + DL.seqs), sa.counts = desc_dev + DL.live;
    sa.npairs = npairs, sa.max_m = d.max_n0, sa.max_n = d.max_n1;
    sa.zbuf = z_dev, sa.rowvec = (float*)(wsp + ws.uv_row), sa.colvec = (float*)(wsp + ws.uv_col), sa.partials = (float*)(wsp + ws.part);
    if (hipMemsetAsync(sa.rowvec, 0, ((size_t)d.T + 16 * npairs + 8) * 4, stream) != hipSuccess) return GTSFM_ERR_HIP;  // u = 0 for iters = 0
    TRY(launch_sinkhorn(sa, bin_score, iters, stream));
    hipLaunchKernelGGL(sk_copy_vectors_kernel, dim3(ceil_div((d.max_n0 > d.max_n1 ? d.max_n0 : d.max_n1) + 1, 256), npairs), dim3(256), 0, stream,
                       sa.seqs, sa.counts, sa.rowvec, sa.colvec, d.max_n0 + 1, d.max_n1 + 1, u_dev, v_dev);
    GTSFM_CHECK_LAUNCH("sk_copy_vectors_kernel");
    return GTSFM_OK;
}

// Stand-alone score matrices: the ragged batched score GEMM exactly as sg_forward_phased launches it (parity against per-pair matmuls,
// bench.py's roofline of the launch as the workload issues it -- one pair alone is 1600 tiles on 512 workgroup slots and says little
// about a 16-pair chunk)
extern "C" size_t gtsfm_score_matrices_workspace_bytes(int npairs) {
    return npairs <= 0 ? 256 : align_up(desc_layout(npairs, 0).total * sizeof(int32_t), 256);
}

extern "C" int gtsfm_score_matrices_f32(const float* mdesc_dev, int npairs, const int32_t* m, const int32_t* n, float alpha, float* z_dev,
                                        void* workspace_dev, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GTSFM_CHECK_ARG(mdesc_dev && m && n && z_dev && workspace_dev, "score_matrices: null pointer");
    GTSFM_CHECK_ARG(npairs > 0, "score_matrices: bad arguments");
    GTSFM_CHECK_ARG(gemm_uses_dma(256, 256), "score_matrices: the LDS-DMA GEMM is switched off");
    const DescLayout DL = desc_layout(npairs, 0);
    if (workspace_bytes < DL.total * sizeof(int32_t)) {
        gtsfm_set_error("score_matrices: workspace too small (%zu < %zu bytes)", workspace_bytes, DL.total * sizeof(int32_t));
        return GTSFM_ERR_WORKSPACE;
    }
    const BatchDims d = batch_dims(npairs, m, n, 1);
    std::vector<int32_t> host(DL.total), hw((size_t)4 * npairs, 1);
    TRY(gtsfm_match_build_desc(1, npairs, m, n, hw.data(), host.data()));
    int32_t* desc_dev = (int32_t*)workspace_dev;
    if (hipMemcpyAsync(desc_dev, host.data(), DL.total * sizeof(int32_t), hipMemcpyHostToDevice, stream) != hipSuccess) return GTSFM_ERR_HIP;
    if (hipStreamSynchronize(stream) != hipSuccess) return GTSFM_ERR_HIP;  // `host` goes out of scope below
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.A = mdesc_dev, g.lda = 256, g.M = d.max_n0, g.K = 256, g.wraw = mdesc_dev, g.ldw = 256, g.N = d.max_n1, g.C = z_dev, g.alpha = alpha;
    g.math = gemm_math_from_env();
    GemmBatch bt = {(const GemmProblem*)(desc_dev + DL.score_p), desc_dev + DL.live, npairs};
    return launch_gemm_dma_batched(g, bt, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Stand-alone LightGlue assignment (sigmoid_log_double_softmax + filter_matches on given similarity matrices) and stand-alone
// LayerNorm + GELU: parity tests against torch, bench.py's rooflines of the sweep / row kernels the forward launches
// ---------------------------------------------------------------------------------------------------------------

namespace {

struct LgaWorkspace {
    size_t desc, part, uv_row, uv_col, max0, idx0, idx1, total;
};

LgaWorkspace lga_workspace_layout(int P, const int32_t* m, const int32_t* n) {
    LgaWorkspace w;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t r = o;
        o += align_up(bytes, 256);
        return r;
    };
    size_t Tp = 0, part = 0;
    int mx = 0;
    for (int p = 0; p < P; ++p) mx = mx > n[p] ? mx : n[p];
    const int R = sweep_partial_rows(mx, 0);
    for (int p = 0; p < P; ++p) {
        Tp += (size_t)cap128(m[p]) + cap128(n[p]);
        part += (size_t)ceil_div(m[p], R) * z_ld(n[p], 0) * 2;
    }
    w.desc = take(desc_layout(P, count_tiles(0, P, m, n)).total * sizeof(int32_t));
    w.part = take(part * 4);
    w.uv_row = take((Tp + 16 * (size_t)P + 8) * 4), w.uv_col = take((Tp + 16 * (size_t)P + 8) * 4);
    w.max0 = take(Tp * 4), w.idx0 = take(Tp * 4), w.idx1 = take(Tp * 4);
    w.total = o;
    return w;
}

}  // namespace

extern "C" size_t gtsfm_lg_assignment_workspace_bytes(int npairs, const int32_t* m, const int32_t* n) {
    if (npairs <= 0 || !m || !n) return 256;
    return lga_workspace_layout(npairs, m, n).total;
}

extern "C" int gtsfm_lg_assignment_f32(const float* sim_dev, int npairs, const int32_t* m, const int32_t* n, const float* zlogit_dev,
                                       float filter_threshold, int stages, void* workspace_dev, size_t workspace_bytes, int32_t* matches_dev,
                                       float* mscores_dev, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GTSFM_CHECK_ARG(sim_dev && m && n && zlogit_dev && workspace_dev && matches_dev && mscores_dev, "lg_assignment: null pointer");
    GTSFM_CHECK_ARG(npairs > 0 && (stages & 3) != 0 && stages <= 7, "lg_assignment: stages is 1 (double-softmax sweeps), 2 (extraction; after a call with 1 on the same workspace) or 3 (both), + 4 to reuse the workspace's batch descriptor");
    const LgaWorkspace ws = lga_workspace_layout(npairs, m, n);
    if (workspace_bytes < ws.total) {
        gtsfm_set_error("lg_assignment: workspace too small (%zu < %zu bytes)", workspace_bytes, ws.total);
        return GTSFM_ERR_WORKSPACE;
    }
    const DescLayout DL = desc_layout(npairs, count_tiles(0, npairs, m, n));
    char* wsp = (char*)workspace_dev;
    int32_t* desc_dev = (int32_t*)(wsp + ws.desc);
    if (!(stages & 4)) {  // 4: the workspace holds the descriptor of an earlier call with the same shapes (timing loops: no upload, no synchronisation)
        std::vector<int32_t> host(DL.total), hw((size_t)4 * npairs, 1);
        TRY(gtsfm_match_build_desc(0, npairs, m, n, hw.data(), host.data()));
        for (int s = 0; s < 2 * npairs; ++s) host[DL.final_cnt + s] = host[DL.live + s];  // the assignment runs over the kept keypoints: all of them here
        if (hipMemcpyAsync(desc_dev, host.data(), DL.total * sizeof(int32_t), hipMemcpyHostToDevice, stream) != hipSuccess) return GTSFM_ERR_HIP;
        if (hipStreamSynchronize(stream) != hipSuccess) return GTSFM_ERR_HIP;  // `host` goes out of scope below
    }
    int max_m = 0, max_n = 0;
    for (int p = 0; p < npairs; ++p) max_m = max_m > m[p] ? max_m : m[p], max_n = max_n > n[p] ? max_n : n[p];
    SweepArgs sa;
    sa.pairs = (const PairDesc*)(desc_dev + DL.pairs), sa.seqs = (const SeqDesc*)(desc_dev + DL.seqs), sa.counts = desc_dev + DL.final_cnt;
    sa.npairs = npairs, sa.max_m = max_m, sa.max_n = max_n;
    sa.zbuf = const_cast<float*>(sim_dev), sa.rowvec = (float*)(wsp + ws.uv_row), sa.colvec = (float*)(wsp + ws.uv_col), sa.partials = (float*)(wsp + ws.part);
    if (stages & 1) TRY(launch_double_softmax_lse(sa, stream));
    if (stages & 2)
        TRY(launch_extract_matches(sa, 0, zlogit_dev, filter_threshold, (float*)(wsp + ws.max0), (int*)(wsp + ws.idx0), (int*)(wsp + ws.idx1), matches_dev,
                                   mscores_dev, stream));
    return GTSFM_OK;
}

extern "C" int gtsfm_layernorm_gelu_f32(float* x_dev, int ld, int rows, const float* gamma_dev, const float* beta_dev, void* scratch_dev, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GTSFM_CHECK_ARG(x_dev && gamma_dev && beta_dev && scratch_dev && rows >= 0 && ld >= 512, "layernorm_gelu: bad arguments (512 columns per row, 64 bytes of scratch)");
    if (rows == 0) return GTSFM_OK;
    const int32_t host[8] = {0, 6, 0, 0, 0, rows, rows, 0};  // SeqDesc {row_off 0, cnt_idx 6, H, W, in_off, cap} followed by the count it points at
    if (hipMemcpyAsync(scratch_dev, host, sizeof(host), hipMemcpyHostToDevice, stream) != hipSuccess) return GTSFM_ERR_HIP;
    if (hipStreamSynchronize(stream) != hipSuccess) return GTSFM_ERR_HIP;  // `host` is a stack array
    return launch_layernorm_gelu(x_dev, ld, (const SeqDesc*)scratch_dev, (const int*)scratch_dev, 1, rows, gamma_dev, beta_dev, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// LightGlue (features = "superpoint": 9 layers, 4 heads x 64, descriptor_dim 256)
// ---------------------------------------------------------------------------------------------------------------

namespace {

struct LgDims {
    int P, T, Tp, NT, max_n, max_n0, max_n1;
    size_t z_floats, part_floats, pack_floats;
};

LgDims lg_dims(int P, const int32_t* n0, const int32_t* n1) {
    LgDims d = {P, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int mx1 = 0;
    for (int p = 0; p < P; ++p) mx1 = mx1 > n1[p] ? mx1 : n1[p];
    const int R = sweep_partial_rows(mx1, 0);
    for (int p = 0; p < P; ++p) {
        d.T += n0[p] + n1[p];
        d.Tp += cap128(n0[p]) + cap128(n1[p]);
        d.max_n0 = d.max_n0 > n0[p] ? d.max_n0 : n0[p];
        d.max_n1 = d.max_n1 > n1[p] ? d.max_n1 : n1[p];
        const int ld = z_ld(n1[p], 0);
        d.z_floats += (size_t)n0[p] * ld;
        d.part_floats += (size_t)ceil_div(n0[p], R) * ld * 2;
        const size_t pk = packed_linear_floats(256, n1[p]);
        d.pack_floats = d.pack_floats > pk ? d.pack_floats : pk;
    }
    d.NT = d.Tp / 128;
    d.max_n = d.max_n0 > d.max_n1 ? d.max_n0 : d.max_n1;
    return d;
}

struct LgWorkspace {
    size_t xa, xb, qkv, mlp, md, enca, encb, inda, indb, indf, conf, mval, z_logit, pos, pack, z, part, uv_row, uv_col, max0, idx0, idx1,
        m_int, ms_int, attn, attn_floats, total;
};

LgWorkspace lg_workspace_layout(const LgDims& d, int attn_math) {
    LgWorkspace w;
    size_t o = 0;
    auto take = [&](size_t floats) {
        size_t r = o;
        o += align_up(floats * 4, 256);
        return r;
    };
    const size_t T = d.Tp;
    w.xa = take(T * 512), w.xb = take(T * 512), w.qkv = take(T * 768), w.mlp = take(T * 512), w.md = take(T * 256);
    w.enca = take(T * 64), w.encb = take(T * 64), w.inda = take(T), w.indb = take(T), w.indf = take(T);
    w.conf = take(T), w.mval = take(T), w.z_logit = take(T), w.pos = take(T);
    w.pack = take(d.pack_floats), w.z = take(d.z_floats), w.part = take(d.part_floats);
    w.uv_row = take(T + 16 * d.P + 8), w.uv_col = take(T + 16 * d.P + 8);
    w.max0 = take(T), w.idx0 = take(T), w.idx1 = take(T), w.m_int = take(T), w.ms_int = take(T);
    w.attn_floats = attention_workspace_floats(2 * d.P, 4, d.max_n, d.max_n, T, attn_math);  // split partials (small batches) or fused parking space; 0 below 1025 keypoints
    w.attn = take(w.attn_floats);
    w.total = o;
    return w;
}

}  // namespace

extern "C" size_t gtsfm_lg_workspace_bytes(int npairs, const int32_t* n0, const int32_t* n1) {
    if (npairs <= 0 || !n0 || !n1) return 256;
    return lg_workspace_layout(lg_dims(npairs, n0, n1), attention_math_from_env()).total;
}

// Phases as for SuperGlue: 1 = only the first layer's SELF block (the part of LightGlue that sees one image), x to x_out_dev
// [T][256] in the input's row order; 2 = descriptors_dev holds that x, the first self block is skipped; bit-identical to phase 0.
//
// side_stream (round 5, optional): ONE pair's launch sequence as TWO -- everything LightGlue does per image (Wqkv + rotary, self attention, the
// FFNs, to_qk | to_v, the confidence / matchability heads) is enqueued per keypoint set, image 0's on `stream` and image 1's on `side_stream`;
// the cross attention of a set waits for the other set's to_qk | to_v by an event, and the two sequences meet once per layer for the part that
// sees the pair (stop test, final projection, pruning). One pair's launches leave the chip partly idle (an attention launch of one pair is
// 3.1 rounds of the chip's workgroup slots, the fourth nearly empty; two dozen launches of a few microseconds per layer): the second sequence
// fills it, as a second caller thread does for whole pairs. Every launch covers the same rows with the same kernels' arithmetic (the schedules
// chosen from the launch geometry are bit-identical by construction): results are bit-identical to the one-stream form.
// Taken only for npairs == 1 in the exact-fp32 attention arithmetic (the bf16x3 tiles of a launch are indexed by problem, not by row).
namespace {

struct LgEvents {  // a handful of HIP events for one call; destroyed when the call returns (the runtime keeps them until they completed)
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ok = true;
    void create() {
        for (auto& e : ev)
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) ok = false, e = nullptr;
    }
    ~LgEvents() {
        for (auto e : ev)
            if (e) (void)hipEventDestroy(e);
    }
};

}  // namespace

#define HIP_TRY(expr)                             \
    do {                                          \
        if ((expr) != hipSuccess) {               \
            gtsfm_set_error("%s failed", #expr);  \
            return GTSFM_ERR_HIP;                 \
        }                                         \
    } while (0)

static int lg_forward_phased(const float* wts, int num_layers, const float* match_bias_host, const float* conf_bias_host, int npairs,
                             const int32_t* n0, const int32_t* n1, int32_t* desc_dev, const float* kpts_dev, const float* descriptors_dev,
                             float depth_confidence, float width_confidence, float filter_threshold, int pruning_threshold,
                             void* workspace_dev, size_t workspace_bytes, int32_t* matches_dev, float* mscores_dev, float* sim_dev,
                             int phase, float* x_out_dev, void* stream_, void* side_stream_ = nullptr) {
    hipStream_t stream = (hipStream_t)stream_;
    GTSFM_CHECK_ARG(wts && match_bias_host && n0 && n1 && desc_dev && kpts_dev && descriptors_dev && workspace_dev, "lg_forward: null pointer");
    GTSFM_CHECK_ARG(phase >= 0 && phase <= 2 && (phase == 1 ? x_out_dev != nullptr : (matches_dev && mscores_dev)), "lg_forward: bad phase / null output");
    GTSFM_CHECK_ARG(npairs > 0 && num_layers > 0 && (num_layers == 1 || conf_bias_host), "lg_forward: bad arguments");
    for (int p = 0; p < npairs; ++p) GTSFM_CHECK_ARG(n0[p] > 0 && (n1[p] > 0 || (phase == 1 && n1[p] == 0)), "lg_forward: pair %d has an empty keypoint set", p);
    const LgDims d = lg_dims(npairs, n0, n1);
    const int attn_math = attention_math_from_env();  // GTSFM_ATTENTION_MATH, read per call (sizing and launches of one call agree)
    const int gemm_math = gemm_math_from_env();       // GTSFM_GEMM_MATH: the matchers' projection / FFN / score GEMMs only
    const LgWorkspace ws = lg_workspace_layout(d, attn_math);
    if (workspace_bytes < ws.total) {
        gtsfm_set_error("lg_forward: workspace too small (%zu < %zu bytes)", workspace_bytes, ws.total);
        return GTSFM_ERR_WORKSPACE;
    }
    const DescLayout DL = desc_layout(npairs, d.NT);
    int* live = desc_dev + DL.live;
    int* final_cnt = desc_dev + DL.final_cnt;
    int* assign = desc_dev + DL.assign;
    const int* orig = desc_dev + DL.orig;
    int* old_cnt = desc_dev + DL.old_cnt;
    int* stop_layer = desc_dev + DL.stop;
    const SeqDesc* seqs = (const SeqDesc*)(desc_dev + DL.seqs);
    const PairDesc* pairs = (const PairDesc*)(desc_dev + DL.pairs);
    const AttnProblem* self_p = (const AttnProblem*)(desc_dev + DL.self_p);
    const AttnProblem* cross_p = (const AttnProblem*)(desc_dev + DL.cross_p);
    const int* tile_idx = desc_dev + DL.tile_idx;
    const int* tile_row0 = desc_dev + DL.tile_row0;
    char* wsp = (char*)workspace_dev;
    float* X = (float*)(wsp + ws.xa);
    float* Xalt = (float*)(wsp + ws.xb);
    float* QKV = (float*)(wsp + ws.qkv);
    float* MLP = (float*)(wsp + ws.mlp);
    float* MD = (float*)(wsp + ws.md);
    float* enc = (float*)(wsp + ws.enca);
    float* enc_alt = (float*)(wsp + ws.encb);
    int* ind = (int*)(wsp + ws.inda);
    int* ind_alt = (int*)(wsp + ws.indb);
    int* ind_final = (int*)(wsp + ws.indf);
    float* conf = (float*)(wsp + ws.conf);
    float* mval = (float*)(wsp + ws.mval);
    float* z_logit = (float*)(wsp + ws.z_logit);
    int* pos = (int*)(wsp + ws.pos);
    float* PACK = (float*)(wsp + ws.pack);
    float* Z = sim_dev ? sim_dev : (float*)(wsp + ws.z);
    float* PART = (float*)(wsp + ws.part);
    float* rowvec = (float*)(wsp + ws.uv_row);
    float* colvec = (float*)(wsp + ws.uv_col);
    float* max0 = (float*)(wsp + ws.max0);
    int* idx0 = (int*)(wsp + ws.idx0);
    int* idx1 = (int*)(wsp + ws.idx1);
    int* m_int = (int*)(wsp + ws.m_int);
    float* ms_int = (float*)(wsp + ws.ms_int);
    const int nseq = 2 * npairs;
    const bool do_prune = width_confidence > 0.f && pruning_threshold != 0x7fffffff;

    // A "view" is the set of keypoint sets one launch sequence covers: all of them on one stream (the batched form), or ONE set of the single
    // pair per stream. Sets are 128-row aligned, so a view is a row range of every token-major array plus a range of the tile / sequence /
    // problem tables.
    struct View {
        int row0, rows, tile0, seq0, nseq;
        hipStream_t stream;
    };
    // (the double-buffered attention build parks every fused launch's merged state in one workspace slab: no two launches side by side there)
    const bool two = side_stream_ != nullptr && npairs == 1 && phase != 1 && attn_math == ATTN_MATH_F32 && !attention_parks_in_workspace();
    // Any early return below (a failed launch, a failed event call) leaves kernels queued on side_stream that `stream` never waited for, while the
    // caller owns workspace and outputs in `stream`'s order only: drain the side stream before the error leaves this function.
    struct SideStreamGuard {
        hipStream_t side;
        bool armed;
        ~SideStreamGuard() {
            if (armed) (void)hipStreamSynchronize(side);
        }
    } side_guard = {(hipStream_t)side_stream_, two};
    View views[2];
    int nviews = 1;
    views[0] = {0, d.Tp, 0, 0, nseq, stream};
    if (two) {
        nviews = 2;
        views[0] = {0, cap128(n0[0]), 0, 0, 1, stream};
        views[1] = {cap128(n0[0]), cap128(n1[0]), cap128(n0[0]) / 128, 1, 1, (hipStream_t)side_stream_};
    }
    LgEvents E;
    if (two) {
        E.create();
        GTSFM_CHECK_ARG(E.ok, "lg_forward: could not create events for the two-stream form");
    }
    hipEvent_t ev_fork = E.ev[0], ev_qk[2] = {E.ev[1], E.ev[2]}, ev_side_done = E.ev[3], ev_tail = E.ev[4];

    struct Lin {
        const float *w, *b, *raw;
        int n, k;
    };
    BlobCursor cur = {wts, 0};
    auto take = [&](int N, int K) {
        Lin l;
        cur.linear(N, K, &l.w, &l.b, &l.raw);
        l.n = N, l.k = K;
        return l;
    };
    // masked GEMM over the padded token rows of a view; `cnt` selects which count array gates the tiles
    auto gemm = [&](const View& v, const Lin& W, const float* A, int lda, float* C, int ldc, const float* res, int ldres, float alpha, const int* cnt,
                    const float* rot_enc = nullptr, int rot_cols = 0) -> int {
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = A + (size_t)v.row0 * lda, g.lda = lda, g.M = v.rows, g.K = W.k, g.wpack = W.w, g.wraw = W.raw, g.ldw = W.k, g.bias = W.b, g.N = W.n;
        g.C = C + (size_t)v.row0 * ldc, g.ldc = ldc, g.c_coff = 0, g.res = res ? res + (size_t)v.row0 * ldres : nullptr, g.ldres = ldres, g.alpha = alpha, g.relu = 0;
        g.tile_cnt_idx = tile_idx + v.tile0, g.tile_row0 = tile_row0 + v.tile0, g.live_counts = cnt;
        g.rot_enc = rot_enc ? rot_enc + (size_t)v.row0 * 64 : nullptr, g.rot_cols = rot_cols;
        g.math = gemm_math;
        return launch_gemm(g, v.stream);
    };
    struct Ffn {
        Lin f0;
        const float *gamma, *beta;
        Lin f3;
    };
    auto take_ffn = [&]() {
        Ffn f;
        f.f0 = take(512, 512);
        f.gamma = cur.raw(512), f.beta = cur.raw(512);
        f.f3 = take(256, 512);
        return f;
    };
    auto ffn = [&](const View& v, const Ffn& f, float* Xc) -> int {  // x + ffn(cat[x, message])
        // (LayerNorm + GELU inside the ffn.0 GEMM was an opt-in in rounds 2-4; slower batched and single-pair, removed: gemm_dma_kernels.hip)
        TRY(gemm(v, f.f0, Xc, 512, MLP, 512, nullptr, 0, 1.0f, live));
        TRY(launch_layernorm_gelu(MLP, 512, seqs + v.seq0, live, v.nseq, d.max_n, f.gamma, f.beta, v.stream));
        TRY(gemm(v, f.f3, MLP, 512, Xc, 512, Xc, 512, 1.0f, live));
        return GTSFM_OK;
    };
    auto attention = [&](const View& v, const AttnProblem* problems, const float* q, const float* k, const float* vv, float* Xc) -> int {
        AttnParams ap = {};
        // the attention context lands in the second half of cat([x, .]); out_proj / to_out are folded into ffn.0 at load time
        ap.q = q, ap.ldq = 768, ap.k = k, ap.ldk = 768, ap.v = vv, ap.ldv = 768, ap.out = Xc + 256, ap.ldo = 512;
        ap.counts = live, ap.scale = 0.125f, ap.heads = 4;
        ap.max_k = d.max_n, ap.workspace = ws.attn_floats ? (float*)(wsp + ws.attn) : nullptr, ap.workspace_floats = ws.attn_floats, ap.part_rows = d.Tp;
        ap.math = attn_math;
        ap.problems = problems + v.seq0;
        return launch_attention(ap, v.nseq, d.max_n, v.stream);
    };

    const float* Wr = cur.raw(64);
    TRY(launch_lg_load_inputs(descriptors_dev, seqs, live, nseq, d.max_n, X, 512, ind, stream));
    TRY(launch_lg_posenc(kpts_dev, seqs, live, nseq, d.max_n, Wr, enc, stream));
    if (two) {
        HIP_TRY(hipEventRecord(ev_fork, stream));
        HIP_TRY(hipStreamWaitEvent(views[1].stream, ev_fork, 0));
    }

    for (int l = 0; l < num_layers; ++l) {
        const Lin wqkv = take(768, 256);
        const Ffn self_ffn = take_ffn();
        const bool skip_self = (l == 0 && phase == 2);  // the first self block was run per image (phase 1): its weights are stepped over
        const bool fused_rotary = gemm_uses_dma(256, 256);  // rotary on q and k: in the Wqkv epilogue of the LDS-DMA GEMM, a kernel of its own otherwise
        if (!skip_self) {
            for (int vi = 0; vi < nviews; ++vi) {  // self block: Wqkv, rotary on q and k, attention, out_proj (folded), ffn
                const View& v = views[vi];
                TRY(gemm(v, wqkv, X, 512, QKV, 768, nullptr, 0, 1.0f, live, fused_rotary ? enc : nullptr, 512));
                if (!fused_rotary) TRY(launch_lg_rotary(QKV, 768, 512, enc, seqs + v.seq0, live, v.nseq, d.max_n, v.stream));
                TRY(attention(v, self_p, QKV, QKV + 256, QKV + 512, X));
                TRY(ffn(v, self_ffn, X));
            }
        }
        if (phase == 1) {
            TRY(launch_lg_store_rows(X, 512, seqs, live, nseq, d.max_n, x_out_dev, stream));
            return GTSFM_OK;
        }
        // cross block: shared to_qk | to_v, both directions of the bidirectional attention, to_out (folded), ffn
        const Lin wcross = take(512, 256);
        const Ffn cross_ffn = take_ffn();
        for (int vi = 0; vi < nviews; ++vi) {
            TRY(gemm(views[vi], wcross, X, 512, QKV, 768, nullptr, 0, 1.0f, live));
            if (two) HIP_TRY(hipEventRecord(ev_qk[vi], views[vi].stream));
        }
        for (int vi = 0; vi < nviews; ++vi) {
            const View& v = views[vi];
            if (two) HIP_TRY(hipStreamWaitEvent(v.stream, ev_qk[1 - vi], 0));  // the other set's keys / values
            TRY(attention(v, cross_p, QKV, QKV, QKV + 256, X));
            TRY(ffn(v, cross_ffn, X));
        }

        // adaptive depth / final assignment inputs
        const Lin wfp = take(256, 256);  // log_assignment[l].final_proj
        const float* w_match = cur.raw(256);
        const float* w_conf = (l < num_layers - 1) ? cur.raw(256) : nullptr;
        const float thr = (float)fmin(fmax(0.8 + 0.1 * exp(-4.0 * l / num_layers), 0.0), 1.0);
        // both heads over the live tokens in one pass: conf (depth / width), the matchability logit (kept by the pairs that stop here) and
        // its sigmoid (pruning); the index lists of the pairs that stop are frozen by the stop check itself
        const bool prune_here = do_prune && l < num_layers - 1;
        for (int vi = 0; vi < nviews; ++vi) {
            const View& v = views[vi];
            TRY(launch_lg_heads(X, 512, seqs + v.seq0, live, v.nseq, d.max_n, w_conf, w_conf ? conf_bias_host[l] : 0.f, w_match, match_bias_host[l], conf, z_logit,
                                prune_here ? mval : nullptr, v.stream));
        }
        if (two) {  // the part of a layer that sees the PAIR runs on `stream` behind both sequences
            HIP_TRY(hipEventRecord(ev_side_done, views[1].stream));
            HIP_TRY(hipStreamWaitEvent(stream, ev_side_done, 0));
        }
        TRY(launch_lg_stop_check(conf, seqs, live, final_cnt, assign, orig, stop_layer, npairs, l, num_layers - 1, thr, depth_confidence, ind, ind_final,
                                 stream));
        // pairs that stopped at this layer: mdesc = final_proj(x) / 256^(1/4)
        const View all = {0, d.Tp, 0, 0, nseq, stream};
        TRY(gemm(all, wfp, X, 512, MD, 256, nullptr, 0, 0.25f, assign));
        if (prune_here) {
            TRY(launch_lg_prune(conf, mval, seqs, live, old_cnt, pos, nseq, d.max_n, thr, (float)(1.0 - (double)width_confidence), pruning_threshold,
                                depth_confidence > 0.f ? 1 : 0, X, Xalt, 512, enc, enc_alt, ind, ind_alt, stream));
            float* tx = X; X = Xalt; Xalt = tx;
            float* te = enc; enc = enc_alt; enc_alt = te;
            int* ti = ind; ind = ind_alt; ind_alt = ti;
        }
        if (two && l < num_layers - 1) {
            HIP_TRY(hipEventRecord(ev_tail, stream));
            HIP_TRY(hipStreamWaitEvent(views[1].stream, ev_tail, 0));
        }
    }

    // sim = mdesc0 mdesc1^T per pair over the final (kept) keypoints: one ragged launch (LDS-DMA GEMM), sizes from final_cnt
    if (gemm_uses_dma(256, 256)) {
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = MD, g.lda = 256, g.M = d.max_n0, g.K = 256, g.wraw = MD, g.ldw = 256, g.N = d.max_n1, g.C = Z, g.alpha = 1.0f;
        g.math = gemm_math;
        GemmBatch bt = {(const GemmProblem*)(desc_dev + DL.score_p), final_cnt, npairs};
        TRY(launch_gemm_dma_batched(g, bt, stream));
    } else {
        size_t zoff = 0;
        int row = 0;
        for (int p = 0; p < npairs; ++p) {
            const int r0 = row, r1 = row + cap128(n0[p]);
            const int ld = z_ld(n1[p], 0);
            TRY(launch_pack_rows(MD + (size_t)r1 * 256, 256, n1[p], final_cnt + 2 * p + 1, 256, PACK, stream));
            GemmParams g;
            memset(&g, 0, sizeof(g));
            g.A = MD + (size_t)r0 * 256, g.lda = 256, g.M = n0[p], g.m_dev = final_cnt + 2 * p, g.K = 256, g.wpack = PACK, g.N = n1[p];
            g.C = Z + zoff, g.ldc = ld, g.alpha = 1.0f;
            TRY(launch_gemm(g, stream));
            zoff += (size_t)n0[p] * ld;
            row += cap128(n0[p]) + cap128(n1[p]);
        }
    }
    SweepArgs sa;
    sa.pairs = pairs, sa.seqs = seqs, sa.counts = final_cnt, sa.npairs = npairs, sa.max_m = d.max_n0, sa.max_n = d.max_n1;
    sa.zbuf = Z, sa.rowvec = rowvec, sa.colvec = colvec, sa.partials = PART;
    TRY(launch_double_softmax_lse(sa, stream));
    TRY(launch_extract_matches(sa, 0, z_logit, filter_threshold, max0, idx0, idx1, m_int, ms_int, stream));
    TRY(launch_lg_scatter_matches(seqs, final_cnt, ind_final, m_int, ms_int, nseq, d.max_n, d.T, matches_dev, mscores_dev, stream));
    side_guard.armed = false;  // the last layer joined the side stream back into `stream`
    return GTSFM_OK;
}

extern "C" int gtsfm_lg_forward(const float* wts, int num_layers, const float* match_bias_host, const float* conf_bias_host, int npairs,
                                const int32_t* n0, const int32_t* n1, int32_t* desc_dev, const float* kpts_dev, const float* descriptors_dev,
                                float depth_confidence, float width_confidence, float filter_threshold, int pruning_threshold,
                                void* workspace_dev, size_t workspace_bytes, int32_t* matches_dev, float* mscores_dev, float* sim_dev,
                                void* stream_) {
    return lg_forward_phased(wts, num_layers, match_bias_host, conf_bias_host, npairs, n0, n1, desc_dev, kpts_dev, descriptors_dev, depth_confidence,
                             width_confidence, filter_threshold, pruning_threshold, workspace_dev, workspace_bytes, matches_dev, mscores_dev, sim_dev,
                             0, nullptr, stream_);
}

extern "C" int gtsfm_lg_forward_phase(const float* wts, int num_layers, const float* match_bias_host, const float* conf_bias_host, int npairs,
                                      const int32_t* n0, const int32_t* n1, int32_t* desc_dev, const float* kpts_dev, const float* descriptors_dev,
                                      float depth_confidence, float width_confidence, float filter_threshold, int pruning_threshold,
                                      void* workspace_dev, size_t workspace_bytes, int32_t* matches_dev, float* mscores_dev, float* sim_dev,
                                      int phase, float* x_out_dev, void* stream_) {
    return lg_forward_phased(wts, num_layers, match_bias_host, conf_bias_host, npairs, n0, n1, desc_dev, kpts_dev, descriptors_dev, depth_confidence,
                             width_confidence, filter_threshold, pruning_threshold, workspace_dev, workspace_bytes, matches_dev, mscores_dev, sim_dev,
                             phase, x_out_dev, stream_);
}

extern "C" int gtsfm_lg_forward_streams(const float* wts, int num_layers, const float* match_bias_host, const float* conf_bias_host, int npairs,
                                        const int32_t* n0, const int32_t* n1, int32_t* desc_dev, const float* kpts_dev, const float* descriptors_dev,
                                        float depth_confidence, float width_confidence, float filter_threshold, int pruning_threshold,
                                        void* workspace_dev, size_t workspace_bytes, int32_t* matches_dev, float* mscores_dev, float* sim_dev,
                                        int phase, float* x_out_dev, void* stream_, void* side_stream_) {
    return lg_forward_phased(wts, num_layers, match_bias_host, conf_bias_host, npairs, n0, n1, desc_dev, kpts_dev, descriptors_dev, depth_confidence,
                             width_confidence, filter_threshold, pruning_threshold, workspace_dev, workspace_bytes, matches_dev, mscores_dev, sim_dev,
                             phase, x_out_dev, stream_, side_stream_);
}
