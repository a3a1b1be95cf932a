// C-ABI entry points of the matchers: weight-blob packing, batch descriptors, stand-alone attention, and the
// SuperGlue forward pass (replaces thirdparty/SuperGluePretrainedNetwork/models/superglue.py:228-283) for a ragged
// batch of image pairs. See include/gtsfm_amd.h.

#include <string.h>

#include <vector>

#include "../../include/gtsfm_amd.h"
#include "attention_kernels.h"
#include "dense_kernels.h"
#include "matcher_kernels.h"

#define TRY(expr)                          \
    do {                                   \
        int rc_ = (expr);                  \
        if (rc_ != GTSFM_OK) return rc_;   \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------
// Weight blobs: a sequence of entries, each 64-float aligned.
//   kind 0 (linear): packed W[n][k_pad] (pack_linear_weights) followed by the bias padded to a multiple of 64
//   kind 1 (raw)   : n floats copied verbatim
// ---------------------------------------------------------------------------------------------------------------

static size_t round64(size_t x) { return (x + 63) / 64 * 64; }
static int kpad8(int k) { return (k + 7) / 8 * 8; }

static size_t blob_entry_floats(int kind, int n, int k) {
    if (kind == 0) return round64(packed_linear_floats(kpad8(k), n)) + round64(n);
    return round64(n);
}

struct BlobCursor {
    const float* base;
    size_t off;
    // linear entry -> (packed W, bias); advances
    void linear(int n, int k, const float** w, const float** b) {
        *w = base + off;
        *b = base + off + round64(packed_linear_floats(kpad8(k), n));
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
            if (b[i]) memcpy(out + off + round64(packed_linear_floats(kpad8(k[i]), n[i])), b[i], n[i] * sizeof(float));
        } else {
            memcpy(out + off, w[i], n[i] * sizeof(float));
        }
        off += sz;
    }
    return GTSFM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Batch descriptors (int32 block built on the host, uploaded by the caller):
//   counts [2P] | seqs [2P][4] | pairs [P][6] | self-attention problems [2P][4] | cross-attention problems [2P][4]
// ---------------------------------------------------------------------------------------------------------------

struct DescLayout {
    size_t counts, seqs, pairs, self_p, cross_p, total;
};

static DescLayout desc_layout(int P) {
    DescLayout L;
    size_t o = 0;
    L.counts = o, o += (size_t)2 * P;
    o = (o + 1) / 2 * 2;
    L.seqs = o, o += (size_t)8 * P;
    L.pairs = o, o += (size_t)6 * P;
    L.self_p = o, o += (size_t)8 * P;
    L.cross_p = o, o += (size_t)8 * P;
    L.total = o;
    return L;
}

static int z_ld(int n1, int ext) { return (n1 + ext + 3) / 4 * 4; }

extern "C" size_t gtsfm_match_desc_ints(int npairs) { return desc_layout(npairs > 0 ? npairs : 0).total; }

extern "C" int gtsfm_match_build_desc(int superglue, int npairs, const int32_t* n0, const int32_t* n1, const int32_t* hw,
                                      int32_t* out) {
    GTSFM_CHECK_ARG(npairs > 0 && n0 && n1 && hw && out, "match_build_desc: bad arguments");
    const DescLayout L = desc_layout(npairs);
    memset(out, 0, L.total * sizeof(int32_t));
    const int ext = superglue ? 1 : 0;
    int row = 0;
    long long zoff = 0, poff = 0;
    for (int p = 0; p < npairs; ++p) {
        GTSFM_CHECK_ARG(n0[p] > 0 && n1[p] > 0, "match_build_desc: pair %d has an empty keypoint set", p);
        const int ns[2] = {n0[p], n1[p]};
        int offs[2];
        for (int s = 0; s < 2; ++s) {
            const int si = 2 * p + s;
            out[L.counts + si] = ns[s];
            SeqDesc sd = {row, si, hw[4 * p + 2 * s], hw[4 * p + 2 * s + 1]};
            memcpy(out + L.seqs + 4 * si, &sd, sizeof(sd));
            offs[s] = row;
            row += ns[s];
        }
        PairDesc pd;
        pd.z_off = zoff, pd.part_off = poff, pd.ld = z_ld(n1[p], ext), pd.pad = 0;
        memcpy(out + L.pairs + 6 * p, &pd, sizeof(pd));
        zoff += (long long)(n0[p] + ext) * pd.ld;
        poff += (long long)ceil_div(n0[p] + ext, 16) * pd.ld * 2;
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
// Stand-alone attention (parity tests)
// ---------------------------------------------------------------------------------------------------------------

extern "C" int gtsfm_attention_f32(const float* q_dev, int ldq, const float* k_dev, int ldk, const float* v_dev, int ldv, float* out_dev,
                                   int ldo, const int32_t* problems_dev, const int32_t* counts_dev, int nproblems, int max_q, int heads,
                                   float scale, void* stream) {
    GTSFM_CHECK_ARG(q_dev && k_dev && v_dev && out_dev && problems_dev && counts_dev, "attention: null pointer");
    AttnParams p;
    p.q = q_dev, p.ldq = ldq, p.k = k_dev, p.ldk = ldk, p.v = v_dev, p.ldv = ldv, p.out = out_dev, p.ldo = ldo;
    p.problems = (const AttnProblem*)problems_dev, p.counts = counts_dev, p.scale = scale, p.heads = heads;
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
    for (int p = 0; p < P; ++p) {
        d.T += n0[p] + n1[p];
        d.max_n0 = d.max_n0 > n0[p] ? d.max_n0 : n0[p];
        d.max_n1 = d.max_n1 > n1[p] ? d.max_n1 : n1[p];
        const int ld = z_ld(n1[p], ext);
        d.z_floats += (size_t)(n0[p] + ext) * ld;
        d.part_floats += (size_t)ceil_div(n0[p] + ext, 16) * ld * 2;
        const size_t pk = packed_linear_floats(256, n1[p]);
        d.pack_floats = d.pack_floats > pk ? d.pack_floats : pk;
    }
    d.max_n = d.max_n0 > d.max_n1 ? d.max_n0 : d.max_n1;
    return d;
}

struct SgWorkspace {
    size_t enc_in, ka, kb, x, qkv, att, mlp, md, pack, z, part, uv_row, uv_col, max0, idx0, idx1, total;
};

SgWorkspace sg_workspace_layout(const BatchDims& d) {
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
    w.att = take(T * 256);
    w.mlp = take(T * 512);
    w.md = take(T * 256);
    w.pack = take(d.pack_floats);
    w.z = take(d.z_floats);
    w.part = take(d.part_floats);
    w.uv_row = take(T + 2 * d.P);
    w.uv_col = take(T + 2 * d.P);
    w.max0 = take(T);
    w.idx0 = take(T);
    w.idx1 = take(T);
    w.total = o;
    return w;
}

}  // namespace

extern "C" size_t gtsfm_sg_workspace_bytes(int npairs, const int32_t* n0, const int32_t* n1) {
    if (npairs <= 0 || !n0 || !n1) return 256;
    return sg_workspace_layout(batch_dims(npairs, n0, n1, 1)).total;
}

extern "C" int gtsfm_sg_forward(const float* wts, int num_layers, float bin_score, int npairs, const int32_t* n0, const int32_t* n1,
                                const int32_t* desc_dev, const float* kpts_dev, const float* scores_dev, const float* descriptors_dev,
                                int sinkhorn_iters, float match_threshold, void* workspace_dev, size_t workspace_bytes,
                                int32_t* matches_dev, float* mscores_dev, float* ot_dev, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GTSFM_CHECK_ARG(wts && n0 && n1 && desc_dev && kpts_dev && scores_dev && descriptors_dev && workspace_dev && matches_dev && mscores_dev,
                    "sg_forward: null pointer");
    GTSFM_CHECK_ARG(npairs > 0 && num_layers >= 0 && sinkhorn_iters >= 0, "sg_forward: bad arguments");
    const BatchDims d = batch_dims(npairs, n0, n1, 1);
    const SgWorkspace ws = sg_workspace_layout(d);
    if (workspace_bytes < ws.total) {
        gtsfm_set_error("sg_forward: workspace too small (%zu < %zu bytes)", workspace_bytes, ws.total);
        return GTSFM_ERR_WORKSPACE;
    }
    const DescLayout DL = desc_layout(npairs);
    const int* counts = desc_dev + DL.counts;
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
    float* ATT = (float*)(wsp + ws.att);
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
        const float *w, *b;
        cur.linear(N, K, &w, &b);
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = A, g.lda = lda, g.M = T, g.K = kpad8(K), g.wpack = w, g.bias = b, g.N = N;
        g.C = C, g.ldc = ldc, g.c_coff = coff, g.res = res, g.ldres = ldres, g.alpha = 1.0f, g.relu = relu;
        return launch_gemm(g, stream);
    };

    // keypoint encoder (superglue.py:73-82, BatchNorm folded into the convolutions at load time)
    TRY(launch_sg_encode_input(kpts_dev, scores_dev, seqs, counts, 2 * npairs, d.max_n, enc_in, stream));
    TRY(gemm(enc_in, 8, 3, 32, ka, 256, 0, nullptr, 0, 1));
    TRY(gemm(ka, 256, 32, 64, kb, 256, 0, nullptr, 0, 1));
    TRY(gemm(kb, 256, 64, 128, ka, 256, 0, nullptr, 0, 1));
    TRY(gemm(ka, 256, 128, 256, kb, 256, 0, nullptr, 0, 1));
    TRY(gemm(kb, 256, 256, 256, X, 512, 0, descriptors_dev, 256, 0));  // desc + kenc(kpts, scores)

    // attentional GNN (superglue.py:122-138): alternating self / cross layers, both images per launch
    for (int l = 0; l < num_layers; ++l) {
        TRY(gemm(X, 512, 256, 768, QKV, 768, 0, nullptr, 0, 0));
        AttnParams ap;
        ap.q = QKV, ap.ldq = 768, ap.k = QKV + 256, ap.ldk = 768, ap.v = QKV + 512, ap.ldv = 768, ap.out = ATT, ap.ldo = 256;
        ap.problems = (l % 2 == 0) ? self_p : cross_p, ap.counts = counts, ap.scale = 0.125f, ap.heads = 4;
        TRY(launch_attention(ap, 2 * npairs, d.max_n, stream));
        TRY(gemm(ATT, 256, 256, 256, X, 512, 256, nullptr, 0, 0));   // merge -> message half of cat([x, message])
        TRY(gemm(X, 512, 512, 512, MLP, 512, 0, nullptr, 0, 1));       // mlp.0 (+BN folded) + ReLU
        TRY(gemm(MLP, 512, 512, 256, X, 512, 0, X, 512, 0));           // mlp.3, desc += delta
    }
    TRY(gemm(X, 512, 256, 256, MD, 256, 0, nullptr, 0, 0));  // final_proj

    // scores = mdesc0^T mdesc1 / sqrt(256) into the (m+1) x (n+1) couplings matrix (superglue.py:257-258,156-160)
    {
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
        if (hipMemsetAsync(rowvec, 0, sizeof(float) * (T + 2 * npairs), stream) != hipSuccess) return GTSFM_ERR_HIP;
    }
    TRY(launch_extract_matches(sa, 1, nullptr, match_threshold, max0, idx0, idx1, matches_dev, mscores_dev, stream));
    if (ot_dev) TRY(launch_materialize_assignment(sa, 1, nullptr, ot_dev, stream));
    return GTSFM_OK;
}
