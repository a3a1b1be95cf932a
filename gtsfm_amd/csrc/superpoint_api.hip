// C-ABI entry points: error handling, generic dense ops, and the SuperPoint forward pass
// (replaces thirdparty/SuperGluePretrainedNetwork/models/superpoint.py:145-202). See include/gtsfm_amd.h.

#include <stdarg.h>
#include <string.h>

#include <vector>

#include "../../include/gtsfm_amd.h"
#include "dense_kernels.h"
#include "superpoint_kernels.h"

static thread_local char g_error[512] = "";

void gtsfm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int gtsfm_cu_count(void) {
    static const int cus = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) {
            (void)hipGetLastError();
            return 256;
        }
        return (int)prop.multiProcessorCount;
    }();
    return cus;
}

extern "C" int gtsfm_abi_version(void) { return 1; }
extern "C" const char* gtsfm_last_error(void) { return g_error; }

// ---------------------------------------------------------------------------------------------------------------
// generic dense ops
// ---------------------------------------------------------------------------------------------------------------

extern "C" size_t gtsfm_packed_conv3x3_floats(int cin, int cout) { return packed_conv3x3_floats(cin, cout); }
extern "C" size_t gtsfm_packed_linear_floats(int k_pad, int n) { return packed_linear_floats(k_pad, n); }

extern "C" int gtsfm_pack_conv3x3(const float* w_host, int cin, int cout, float* packed_host) {
    GTSFM_CHECK_ARG(w_host && packed_host && cin >= 64 && cin % 64 == 0 && cout > 0, "pack_conv3x3: bad arguments");
    pack_conv3x3_weights(w_host, cin, cout, packed_host);
    return GTSFM_OK;
}

extern "C" int gtsfm_pack_linear(const float* w_host, int k_real, int k_pad, int n, float* packed_host) {
    GTSFM_CHECK_ARG(w_host && packed_host && k_real > 0 && k_pad >= k_real && k_pad % 8 == 0 && n > 0, "pack_linear: bad arguments");
    pack_linear_weights(w_host, k_real, k_pad, n, packed_host);
    return GTSFM_OK;
}

extern "C" int gtsfm_conv3x3_f32(const float* in_dev, int in_stride, int in_coff, float* out_dev, int out_stride, int out_coff,
                                 const float* packed_w_dev, const float* bias_dev, int batch, int h, int w, int cin, int cout, int relu,
                                 int pool, void* stream) {
    GTSFM_CHECK_ARG(in_dev && out_dev && packed_w_dev && bias_dev, "conv3x3: null pointer");
    GTSFM_CHECK_ARG(batch >= 0 && h >= 0 && w >= 0 && cout > 0, "conv3x3: bad shape");
    if (batch == 0 || h == 0 || w == 0) return GTSFM_OK;
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.in = in_dev, p.in_stride = in_stride, p.in_coff = in_coff;
    p.out = out_dev, p.out_stride = out_stride, p.out_coff = out_coff;
    p.wpack = packed_w_dev, p.bias = bias_dev;
    p.B = batch, p.H = h, p.W = w, p.Cin = cin, p.Cout = cout, p.relu = relu, p.pool = pool;
    return launch_conv3x3(p, (hipStream_t)stream);
}

extern "C" int gtsfm_conv1_fused_f32(const void* image_dev, int image_is_u8, const float* w1a_dev, const float* b1a_dev,
                                     const float* packed_w1b_dev, const float* bias1b_dev, int batch, int h, int w, int pool, float* out_dev,
                                     void* stream) {
    GTSFM_CHECK_ARG(image_dev && w1a_dev && b1a_dev && packed_w1b_dev && bias1b_dev && out_dev, "conv1_fused: null pointer");
    GTSFM_CHECK_ARG(batch >= 0 && h >= 0 && w >= 0, "conv1_fused: bad shape");
    if (batch == 0 || h == 0 || w == 0) return GTSFM_OK;
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.in = nullptr, p.in_stride = 64, p.out = out_dev, p.out_stride = 64;
    p.wpack = packed_w1b_dev, p.bias = bias1b_dev;
    p.B = batch, p.H = h, p.W = w, p.Cin = 64, p.Cout = 64, p.relu = 1, p.pool = pool;
    p.img = image_dev, p.img_is_u8 = image_is_u8, p.w1a = w1a_dev, p.b1a = b1a_dev;
    return launch_conv3x3(p, (hipStream_t)stream);
}

extern "C" int gtsfm_linear_f32(const float* a_dev, int lda, int m, const int32_t* m_dev, int k, const float* packed_w_dev,
                                const float* bias_dev, int n, float* c_dev, int ldc, int c_coff, const float* res_dev, int ldres,
                                float alpha, int relu, void* stream) {
    GTSFM_CHECK_ARG(a_dev && packed_w_dev && c_dev, "linear: null pointer");
    GTSFM_CHECK_ARG(m >= 0 && n > 0, "linear: bad shape");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = a_dev, p.lda = lda, p.M = m, p.K = k, p.m_dev = m_dev;
    p.wpack = packed_w_dev, p.bias = bias_dev, p.N = n;
    p.C = c_dev, p.ldc = ldc, p.c_coff = c_coff, p.res = res_dev, p.ldres = ldres, p.alpha = alpha, p.relu = relu;
    return launch_gemm(p, (hipStream_t)stream);
}

extern "C" int gtsfm_linear_rowmajor_f32(const float* a_dev, int lda, int m, const int32_t* m_dev, int k, const float* w_dev, int ldw,
                                         const float* bias_dev, int n, const int32_t* n_dev, float* c_dev, int ldc, int c_coff,
                                         const float* res_dev, int ldres, float alpha, int relu, void* stream) {
    GTSFM_CHECK_ARG(a_dev && w_dev && c_dev, "linear_rowmajor: null pointer");
    GTSFM_CHECK_ARG(m >= 0 && n > 0, "linear_rowmajor: bad shape");
    GTSFM_CHECK_ARG(gemm_uses_dma(k, ldw), "linear_rowmajor: needs k %% 32 == 0 and ldw %% 4 == 0 (got k = %d, ldw = %d)", k, ldw);
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = a_dev, p.lda = lda, p.M = m, p.K = k, p.m_dev = m_dev;
    p.wraw = w_dev, p.ldw = ldw, p.n_dev = n_dev, p.bias = bias_dev, p.N = n;
    p.C = c_dev, p.ldc = ldc, p.c_coff = c_coff, p.res = res_dev, p.ldres = ldres, p.alpha = alpha, p.relu = relu;
    p.math = gemm_math_from_env();  // the stand-alone product the arithmetic tests and bench.py's GEMM rooflines call
    return launch_gemm(p, (hipStream_t)stream);
}

extern "C" int gtsfm_pack_rows_f32(const float* b_dev, int ldb, int n, const int32_t* n_dev, int k, float* packed_dev, void* stream) {
    GTSFM_CHECK_ARG(b_dev && packed_dev, "pack_rows: null pointer");
    return launch_pack_rows(b_dev, ldb, n, n_dev, k, packed_dev, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------
// SuperPoint weight blob
// ---------------------------------------------------------------------------------------------------------------

namespace {

struct SpLayer {
    int cin, cout, k;
};
// checkpoint order (SP:119-134)
const SpLayer kSpLayers[12] = {{1, 64, 3},    {64, 64, 3},    {64, 64, 3},    {64, 64, 3},  {64, 128, 3},  {128, 128, 3},
                               {128, 128, 3}, {128, 128, 3},  {128, 256, 3},  {256, 65, 1}, {128, 256, 3}, {256, 256, 1}};
enum { L1A, L1B, L2A, L2B, L3A, L3B, L4A, L4B, LPA, LPB, LDA, LDB };

struct SpBlob {
    // float offsets into the packed blob
    size_t w1a, b1a;
    size_t w[8], b[8];  // index by layer id for conv1b..conv4b (L1B..L4B)
    size_t wPD, bPD;    // convPa | convDa fused: 128 -> 512
    size_t wPb, bPb, wDb, bDb;
    size_t rPb, rDb;    // convPb / convDb weights once more, row-major [cout][256], for the LDS-DMA GEMM
    size_t total;
};

size_t pad64(int n) { return (size_t)ceil_div(n, 64) * 64; }

SpBlob sp_blob_layout() {
    SpBlob L;
    size_t o = 0;
    auto take = [&](size_t n) {
        size_t r = o;
        o += (n + 63) / 64 * 64;  // keep every section 256-byte aligned
        return r;
    };
    L.w1a = take(9 * 64);
    L.b1a = take(64);
    for (int l = L1B; l <= L4B; ++l) {
        L.w[l] = take(packed_conv3x3_floats(kSpLayers[l].cin, kSpLayers[l].cout));
        L.b[l] = take(pad64(kSpLayers[l].cout));
    }
    L.wPD = take(packed_conv3x3_floats(128, 512));
    L.bPD = take(512);
    L.wPb = take(packed_linear_floats(256, 65));
    L.bPb = take(pad64(65));
    L.wDb = take(packed_linear_floats(256, 256));
    L.bDb = take(256);
    L.rPb = take(65 * 256);
    L.rDb = take(256 * 256);
    L.total = o;
    return L;
}

}  // namespace

extern "C" size_t gtsfm_sp_packed_weight_floats(void) { return sp_blob_layout().total; }

extern "C" int gtsfm_sp_pack_weights(const float* const* t, float* out) {
    GTSFM_CHECK_ARG(t && out, "sp_pack_weights: null pointer");
    for (int i = 0; i < 24; ++i) GTSFM_CHECK_ARG(t[i], "sp_pack_weights: tensor %d is null", i);
    const SpBlob L = sp_blob_layout();
    memset(out, 0, L.total * sizeof(float));
    // conv1a: [64][1][3][3] -> [9][64]
    for (int c = 0; c < 64; ++c)
        for (int tap = 0; tap < 9; ++tap) out[L.w1a + tap * 64 + c] = t[0][c * 9 + tap];
    memcpy(out + L.b1a, t[1], 64 * sizeof(float));
    for (int l = L1B; l <= L4B; ++l) {
        pack_conv3x3_weights(t[2 * l], kSpLayers[l].cin, kSpLayers[l].cout, out + L.w[l]);
        memcpy(out + L.b[l], t[2 * l + 1], kSpLayers[l].cout * sizeof(float));
    }
    // convPa and convDa share their input: one 128 -> 512 convolution (channels 0..255 = Pa, 256..511 = Da)
    {
        std::vector<float> w((size_t)512 * 128 * 9);
        memcpy(w.data(), t[2 * LPA], (size_t)256 * 128 * 9 * sizeof(float));
        memcpy(w.data() + (size_t)256 * 128 * 9, t[2 * LDA], (size_t)256 * 128 * 9 * sizeof(float));
        pack_conv3x3_weights(w.data(), 128, 512, out + L.wPD);
        memcpy(out + L.bPD, t[2 * LPA + 1], 256 * sizeof(float));
        memcpy(out + L.bPD + 256, t[2 * LDA + 1], 256 * sizeof(float));
    }
    pack_linear_weights(t[2 * LPB], 256, 256, 65, out + L.wPb);
    memcpy(out + L.bPb, t[2 * LPB + 1], 65 * sizeof(float));
    pack_linear_weights(t[2 * LDB], 256, 256, 256, out + L.wDb);
    memcpy(out + L.bDb, t[2 * LDB + 1], 256 * sizeof(float));
    memcpy(out + L.rPb, t[2 * LPB], (size_t)65 * 256 * sizeof(float));   // [65][256][1][1] is already row-major
    memcpy(out + L.rDb, t[2 * LDB], (size_t)256 * 256 * sizeof(float));
    return GTSFM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// SuperPoint forward
// ---------------------------------------------------------------------------------------------------------------

namespace {

struct SpWorkspace {
    size_t ping0, ping1, logits, dense, scores, nms, ss, mask, supp, rows, xy_full, sc_full, cnt_full, total;
    int full_capacity;
};

SpWorkspace sp_workspace_layout(int B, int H, int W) {
    const size_t H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2, Hc = H4 / 2, Wc = W4 / 2;
    const size_t cells = (size_t)B * Hc * Wc, pix8 = cells * 64;
    SpWorkspace ws;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t r = o;
        o += align_up(bytes, 256);
        return r;
    };
    size_t p0 = (size_t)B * H2 * W2 * 64;  // conv2a out (conv1a's output lives only in LDS, fused into conv1b)
    p0 = p0 > (size_t)B * H4 * W4 * 128 ? p0 : (size_t)B * H4 * W4 * 128;
    p0 = p0 > cells * 512 ? p0 : cells * 512;
    size_t p1 = (size_t)B * H2 * W2 * 64;
    p1 = p1 > cells * 128 ? p1 : cells * 128;
    ws.ping0 = take(p0 * 4);
    ws.ping1 = take(p1 * 4);
    ws.logits = take(cells * 65 * 4);
    ws.dense = take(cells * 256 * 4);
    ws.scores = take(pix8 * 4);
    ws.nms = take(pix8 * 4);
    ws.ss = take(pix8 * 4);
    ws.mask = take(pix8);
    ws.supp = take(pix8);
    ws.rows = take(((size_t)B * Hc * 8 * 2 + B) * 4);
    // scratch for the top-k path: one slot per pixel, so that no image can overflow it -- NMS leaves far fewer keypoints
    // on natural images, but a flat or saturated image ties everywhere and keeps every pixel above the threshold; a smaller
    // scratch would drop the bottom rows silently before the top-k. 12 bytes per pixel and image.
    ws.full_capacity = (int)(Hc * 8 * Wc * 8);
    ws.xy_full = take((size_t)B * ws.full_capacity * 2 * 4);
    ws.sc_full = take((size_t)B * ws.full_capacity * 4);
    ws.cnt_full = take((size_t)B * 4);
    ws.total = o;
    return ws;
}

}  // namespace

extern "C" size_t gtsfm_sp_workspace_bytes(int batch, int height, int width) {
    if (batch <= 0 || height <= 0 || width <= 0) return 256;
    return sp_workspace_layout(batch, height, width).total;
}

extern "C" size_t gtsfm_sp_nms_scratch_bytes(int batch, int h, int w) {
    const size_t n = (size_t)batch * h * w;
    return align_up(n * 4, 256) + 2 * align_up(n, 256);
}

extern "C" int gtsfm_sp_softmax_d2s(const float* logits_dev, int ld, int batch, int hc, int wc, float* scores_dev, void* stream) {
    GTSFM_CHECK_ARG(logits_dev && scores_dev && ld >= 65, "sp_softmax_d2s: bad arguments");
    return launch_softmax_d2s(logits_dev, ld, batch, hc, wc, scores_dev, (hipStream_t)stream);
}

extern "C" int gtsfm_sp_simple_nms(const float* scores_dev, int batch, int h, int w, int radius, void* scratch_dev, float* out_dev,
                                   void* stream) {
    GTSFM_CHECK_ARG(scores_dev && scratch_dev && out_dev, "sp_simple_nms: null pointer");
    const size_t n = (size_t)batch * h * w;
    char* s = (char*)scratch_dev;
    float* ss = (float*)s;
    uint8_t* mask = (uint8_t*)(s + align_up(n * 4, 256));
    uint8_t* supp = mask + align_up(n, 256);
    return launch_simple_nms(scores_dev, batch, h, w, radius, mask, supp, ss, out_dev, (hipStream_t)stream);
}

extern "C" int gtsfm_sp_extract_keypoints(const float* nms_dev, int batch, int h, int w, float threshold, int border, int capacity,
                                          int32_t* scratch_dev, int32_t* kp_count_dev, int32_t* kp_count_raw_dev, float* kp_xy_dev,
                                          float* kp_score_dev, void* stream) {
    GTSFM_CHECK_ARG(nms_dev && scratch_dev && kp_count_dev && kp_count_raw_dev && kp_xy_dev && kp_score_dev, "sp_extract_keypoints: null pointer");
    return launch_extract_keypoints(nms_dev, batch, h, w, threshold, border, capacity, scratch_dev, scratch_dev + (size_t)batch * h,
                                    kp_count_dev, kp_count_raw_dev, kp_xy_dev, kp_score_dev, (hipStream_t)stream);
}

extern "C" int gtsfm_sp_sample_descriptors(const float* dense_dev, int ld, int batch, int hc, int wc, const float* kp_xy_dev,
                                           const int32_t* kp_count_dev, int capacity, float* desc_dev, void* stream) {
    GTSFM_CHECK_ARG(dense_dev && kp_xy_dev && kp_count_dev && desc_dev, "sp_sample_descriptors: null pointer");
    return launch_sample_descriptors(dense_dev, ld, batch, hc, wc, kp_xy_dev, kp_count_dev, capacity, desc_dev, (hipStream_t)stream);
}

extern "C" int gtsfm_sp_select_topk(const float* kp_score_dev, const float* kp_xy_dev, const int32_t* kp_count_dev, int batch, int capacity,
                                    int top_k, float* out_xy_dev, float* out_score_dev, int32_t* out_count_dev, void* stream) {
    GTSFM_CHECK_ARG(kp_score_dev && kp_xy_dev && kp_count_dev && out_xy_dev && out_score_dev && out_count_dev, "sp_select_topk: null pointer");
    return launch_select_topk(kp_score_dev, kp_count_dev, batch, capacity, top_k, kp_xy_dev, out_xy_dev, out_score_dev, out_count_dev,
                              (hipStream_t)stream);
}

#define SP_TRY(expr)                 \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != GTSFM_OK) return rc_; \
    } while (0)

static int sp_forward_impl(const float* wts, const void* image_dev, int image_is_u8, int B, int H, int W, float thr, int nms_radius,
                           int border, int capacity, int top_k, void* workspace_dev, size_t workspace_bytes, int32_t* kp_count_dev,
                           int32_t* kp_count_raw_dev, float* kp_xy_dev, float* kp_score_dev, float* desc_dev,
                           float* dense_scores_dev, float* nms_scores_dev, const uint8_t* valid_mask_dev, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GTSFM_CHECK_ARG(wts && image_dev && workspace_dev && kp_count_dev && kp_xy_dev && kp_score_dev && desc_dev, "sp_forward: null pointer");
    GTSFM_CHECK_ARG(B > 0 && H > 0 && W > 0 && capacity > 0, "sp_forward: bad shape (batch %d, %d x %d, capacity %d)", B, H, W, capacity);
    GTSFM_CHECK_ARG(top_k <= 0 || top_k == capacity, "sp_forward: with top_k > 0 the output arrays must have capacity == top_k");
    GTSFM_CHECK_ARG(top_k <= 0 || nms_radius >= 1, "sp_forward: the device top-k path needs nms_radius >= 1 (scratch sizing)");
    const SpWorkspace ws = sp_workspace_layout(B, H, W);
    if (workspace_bytes < ws.total) {
        gtsfm_set_error("sp_forward: workspace too small (%zu < %zu bytes)", workspace_bytes, ws.total);
        return GTSFM_ERR_WORKSPACE;
    }
    const SpBlob L = sp_blob_layout();
    char* wsp = (char*)workspace_dev;
    float* p0 = (float*)(wsp + ws.ping0);
    float* p1 = (float*)(wsp + ws.ping1);
    float* logits = (float*)(wsp + ws.logits);
    float* dense = (float*)(wsp + ws.dense);
    float* scores = dense_scores_dev ? dense_scores_dev : (float*)(wsp + ws.scores);
    float* nms = nms_scores_dev ? nms_scores_dev : (float*)(wsp + ws.nms);
    float* ss = (float*)(wsp + ws.ss);
    uint8_t* mask = (uint8_t*)(wsp + ws.mask);
    uint8_t* supp = (uint8_t*)(wsp + ws.supp);
    int* rows = (int*)(wsp + ws.rows);
    const int H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2, Hc = H4 / 2, Wc = W4 / 2;
    const int H8 = Hc * 8, W8 = Wc * 8;

    int32_t* count_raw = kp_count_raw_dev ? kp_count_raw_dev : rows + (size_t)2 * B * H8;  // tail of the rows scratch
    if (Hc == 0 || Wc == 0) {  // image smaller than one 8x8 cell: no keypoints
        if (hipMemsetAsync(kp_count_dev, 0, sizeof(int32_t) * B, stream) != hipSuccess) return GTSFM_ERR_HIP;
        if (kp_count_raw_dev && hipMemsetAsync(kp_count_raw_dev, 0, sizeof(int32_t) * B, stream) != hipSuccess) return GTSFM_ERR_HIP;
        return GTSFM_OK;
    }

    auto conv = [&](int layer, const float* in, int cin_stride, float* out, int h, int w, int pool) -> int {
        ConvParams p;
        memset(&p, 0, sizeof(p));
        p.in = in, p.in_stride = cin_stride, p.in_coff = 0;
        p.out = out, p.out_stride = kSpLayers[layer].cout, p.out_coff = 0;
        p.wpack = wts + L.w[layer], p.bias = wts + L.b[layer];
        p.B = B, p.H = h, p.W = w, p.Cin = kSpLayers[layer].cin, p.Cout = kSpLayers[layer].cout, p.relu = 1, p.pool = pool;
        return launch_conv3x3(p, stream);
    };

    {  // conv1a + ReLU fused into conv1b's halo staging (conv1a's output never touches HBM), + conv1b + ReLU + pool
        ConvParams p;
        memset(&p, 0, sizeof(p));
        p.in = p0, p.in_stride = 64, p.out = p1, p.out_stride = 64;
        p.wpack = wts + L.w[L1B], p.bias = wts + L.b[L1B];
        p.B = B, p.H = H, p.W = W, p.Cin = 64, p.Cout = 64, p.relu = 1, p.pool = 1;
        p.img = image_dev, p.img_is_u8 = image_is_u8, p.w1a = wts + L.w1a, p.b1a = wts + L.b1a;
        SP_TRY(launch_conv3x3(p, stream));
    }
    SP_TRY(conv(L2A, p1, 64, p0, H2, W2, 0));
    SP_TRY(conv(L2B, p0, 64, p1, H2, W2, 1));
    SP_TRY(conv(L3A, p1, 64, p0, H4, W4, 0));
    SP_TRY(conv(L3B, p0, 128, p1, H4, W4, 1));
    SP_TRY(conv(L4A, p1, 128, p0, Hc, Wc, 0));
    SP_TRY(conv(L4B, p0, 128, p1, Hc, Wc, 0));
    {  // convPa | convDa fused: 128 -> 512, ReLU
        ConvParams p;
        memset(&p, 0, sizeof(p));
        p.in = p1, p.in_stride = 128, p.out = p0, p.out_stride = 512;
        p.wpack = wts + L.wPD, p.bias = wts + L.bPD;
        p.B = B, p.H = Hc, p.W = Wc, p.Cin = 128, p.Cout = 512, p.relu = 1, p.pool = 0;
        SP_TRY(launch_conv3x3(p, stream));
    }
    const int cells = B * Hc * Wc;
    {  // convPb: 1x1, 256 -> 65
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = p0, g.lda = 512, g.M = cells, g.K = 256, g.wpack = wts + L.wPb, g.wraw = wts + L.rPb, g.ldw = 256, g.bias = wts + L.bPb, g.N = 65;
        g.C = logits, g.ldc = 65, g.alpha = 1.0f;
        SP_TRY(launch_gemm(g, stream));
    }
    {  // convDb: 1x1, 256 -> 256
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = p0 + 256, g.lda = 512, g.M = cells, g.K = 256, g.wpack = wts + L.wDb, g.wraw = wts + L.rDb, g.ldw = 256, g.bias = wts + L.bDb, g.N = 256;
        g.C = dense, g.ldc = 256, g.alpha = 1.0f;
        SP_TRY(launch_gemm(g, stream));
    }
    SP_TRY(launch_softmax_d2s(logits, 65, B, Hc, Wc, scores, stream));
    SP_TRY(launch_simple_nms(scores, B, H8, W8, nms_radius, mask, supp, ss, nms, stream));
    // Keypoints.filter_by_mask (gtsfm/common/keypoints.py:112-127) ahead of the top-k, as the wrapper orders them
    // (gtsfm/frontend/detector_descriptor/superpoint.py:76-91): a keypoint at pixel (x, y) survives iff mask[y][x] == 1
    if (valid_mask_dev) SP_TRY(launch_apply_keypoint_mask(nms, B, H8, W8, valid_mask_dev, H, W, stream));
    if (top_k <= 0) {
        SP_TRY(launch_extract_keypoints(nms, B, H8, W8, thr, border, capacity, rows, rows + (size_t)B * H8, kp_count_dev, count_raw,
                                        kp_xy_dev, kp_score_dev, stream));
        SP_TRY(launch_sample_descriptors(dense, 256, B, Hc, Wc, kp_xy_dev, kp_count_dev, capacity, desc_dev, stream));
    } else {
        // extract everything into the workspace, keep the top_k responses (row-major order), describe only those
        float* xy_full = (float*)(wsp + ws.xy_full);
        float* sc_full = (float*)(wsp + ws.sc_full);
        int* cnt_full = (int*)(wsp + ws.cnt_full);
        SP_TRY(launch_extract_keypoints(nms, B, H8, W8, thr, border, ws.full_capacity, rows, rows + (size_t)B * H8, cnt_full, count_raw,
                                        xy_full, sc_full, stream));
        SP_TRY(launch_select_topk(sc_full, cnt_full, B, ws.full_capacity, top_k, xy_full, kp_xy_dev, kp_score_dev, kp_count_dev, stream));
        SP_TRY(launch_sample_descriptors(dense, 256, B, Hc, Wc, kp_xy_dev, kp_count_dev, top_k, desc_dev, stream));
    }
    return GTSFM_OK;
}

extern "C" int gtsfm_sp_forward(const float* wts, const void* image_dev, int image_is_u8, int B, int H, int W, float thr, int nms_radius,
                                int border, int capacity, int top_k, void* workspace_dev, size_t workspace_bytes, int32_t* kp_count_dev,
                                int32_t* kp_count_raw_dev, float* kp_xy_dev, float* kp_score_dev, float* desc_dev,
                                float* dense_scores_dev, float* nms_scores_dev, void* stream_) {
    return sp_forward_impl(wts, image_dev, image_is_u8, B, H, W, thr, nms_radius, border, capacity, top_k, workspace_dev, workspace_bytes, kp_count_dev,
                           kp_count_raw_dev, kp_xy_dev, kp_score_dev, desc_dev, dense_scores_dev, nms_scores_dev, nullptr, stream_);
}

extern "C" int gtsfm_sp_forward_masked(const float* wts, const void* image_dev, int image_is_u8, int B, int H, int W, float thr, int nms_radius,
                                       int border, int capacity, int top_k, void* workspace_dev, size_t workspace_bytes, int32_t* kp_count_dev,
                                       int32_t* kp_count_raw_dev, float* kp_xy_dev, float* kp_score_dev, float* desc_dev,
                                       float* dense_scores_dev, float* nms_scores_dev, const uint8_t* valid_mask_dev, void* stream_) {
    GTSFM_CHECK_ARG(!valid_mask_dev || thr > 0.f, "sp_forward_masked: a mask needs a positive keypoint threshold");
    return sp_forward_impl(wts, image_dev, image_is_u8, B, H, W, thr, nms_radius, border, capacity, top_k, workspace_dev, workspace_bytes, kp_count_dev,
                           kp_count_raw_dev, kp_xy_dev, kp_score_dev, desc_dev, dense_scores_dev, nms_scores_dev, valid_mask_dev, stream_);
}
