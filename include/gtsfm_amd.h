/*
 * gtsfm_amd.h -- C ABI of libgtsfm_amd.so: the MI355X (gfx950) deep front-end of GTSfM.
 *
 * This is the drop-in boundary for ONE hot path of borglab/gtsfm: the SuperPoint detector/descriptor and the
 * SuperGlue / LightGlue matchers behind gtsfm/frontend's DetectorDescriptorBase / MatcherBase plugins. The reference
 * has no native code on this path (it calls ATen ops from Python); each entry point below names the reference
 * Python it replaces (paths relative to the reference repository root, abbreviated
 *   SP = thirdparty/SuperGluePretrainedNetwork/models/superpoint.py
 *   SG = thirdparty/SuperGluePretrainedNetwork/models/superglue.py
 *   LG = thirdparty/LightGlue/lightglue/lightglue.py (un-vendored submodule; call sites in
 *        gtsfm/frontend/matcher/lightglue_matcher.py:37-110)).
 *
 * Conventions
 *   - plain C: pointers and sizes only; no torch types. "_dev" pointers are device (HBM) addresses, e.g.
 *     torch.Tensor.data_ptr(); "_host" pointers are host addresses. `stream` is a hipStream_t passed as void*.
 *   - every function returns 0 on success, a negative GTSFM_ERR_* code otherwise, and never throws across the ABI;
 *     gtsfm_last_error() returns a thread-local message for the last failure.
 *   - no hidden device allocations: callers pass workspaces sized by the *_workspace_bytes() queries.
 *   - no global mutable state besides the thread-local error string: any number of callers (one per GPU / process)
 *     may coexist. Kernels are enqueued on `stream` and are asynchronous with respect to the host.
 *   - activations are fp32, channels-last (NHWC for images, [tokens][channels] for keypoint sets).
 */
#ifndef GTSFM_AMD_H
#define GTSFM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GTSFM_OK 0
#define GTSFM_ERR_INVALID -1
#define GTSFM_ERR_HIP -2
#define GTSFM_ERR_WORKSPACE -3

/* ABI version of this header; bumped on incompatible changes. */
int gtsfm_abi_version(void);
/* Thread-local description of the last error returned on this thread ("" if none). */
const char* gtsfm_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * Generic fp32 dense ops (exact-fp32 MFMA). Exposed so that callers and parity tests can exercise each kernel
 * in isolation against the reference's ATen op.
 * ---------------------------------------------------------------------------------------------------------- */

/* Packed-weight sizes (floats) and host-side packers.
 * conv3x3: w_host is torch Conv2d layout [cout][cin][3][3], cin % 64 == 0.
 * linear : w_host is [n][k_real] row-major (nn.Linear / Conv1d(k=1)); k_pad = k_real rounded up to 8.
 * bias (if any) must be passed to the ops as a device array padded with zeros to a multiple of 64 entries. */
size_t gtsfm_packed_conv3x3_floats(int cin, int cout);
size_t gtsfm_packed_linear_floats(int k_pad, int n);
int gtsfm_pack_conv3x3(const float* w_host, int cin, int cout, float* packed_host);
int gtsfm_pack_linear(const float* w_host, int k_real, int k_pad, int n, float* packed_host);

/* y = [maxpool2x2](relu?(conv3x3(x) + bias)), stride 1, zero pad 1, NHWC.            replaces SP:148-161,190
 * in : [batch][h][w][in_stride], channels in_coff .. in_coff+cin-1
 * out: [batch][ho][wo][out_stride], channels out_coff .. out_coff+cout-1; (ho,wo) = (h,w) or (h/2,w/2) if pool */
int gtsfm_conv3x3_f32(const float* in_dev, int in_stride, int in_coff, float* out_dev, int out_stride, int out_coff,
                      const float* packed_w_dev, const float* bias_dev, int batch, int h, int w, int cin, int cout,
                      int relu, int pool, void* stream);

/* SuperPoint's first two layers as gtsfm_sp_forward runs them: y = [maxpool2x2] relu(conv1b(relu(conv1a(image)))), conv1a
 * recomputed inside conv1b's halo staging (its 64-channel output never reaches HBM).            replaces SP:148-150
 * image: [batch][h][w] uint8 (read as x / 255) or float32; w1a: [9 taps][64] (tap = 3 ky + kx), b1a: [64];
 * packed_w1b: gtsfm_pack_conv3x3(conv1b.weight, 64, 64); bias1b: [64]; out: [batch][ho][wo][64] */
int gtsfm_conv1_fused_f32(const void* image_dev, int image_is_u8, const float* w1a_dev, const float* b1a_dev,
                          const float* packed_w1b_dev, const float* bias1b_dev, int batch, int h, int w, int pool,
                          float* out_dev, void* stream);

/* C[:, c_coff:c_coff+n] = (res +) relu?(alpha * (A[:, :k] W^T + bias))              replaces SP:162,191, SG:49-60,
 * A: [m][lda]; C: [m][ldc]; res (optional): [m][ldres]; m_dev (optional): row count in device memory (<= m).
 * nn.Linear / Conv1d(kernel_size=1) / Conv2d(kernel_size=1).                                     98-119,254 */
int gtsfm_linear_f32(const float* a_dev, int lda, int m, const int32_t* m_dev, int k, const float* packed_w_dev,
                     const float* bias_dev, int n, float* c_dev, int ldc, int c_coff, const float* res_dev, int ldres,
                     float alpha, int relu, void* stream);

/* Same operation with the weights (or a second activation matrix) given row-major, W[n][ldw] as nn.Linear stores them:
 * both operands then travel by LDS-DMA (k % 32 == 0 and ldw % 4 == 0 required). n_dev (optional): column count in
 * device memory (<= n).                                                        same reference lines as gtsfm_linear_f32
 * Environment (read per launch): GTSFM_GEMM_MATH=bf16x3 runs the product (either tiling) in the opt-in arithmetic of
 * gtsfm_attention_math_f32 (operands split exactly into three bf16 pieces, six bf16 MFMA products per block, fp32 accumulation:
 * fp32-class error, NOT the default's bits); GTSFM_GEMM_MATH=f16x2 runs it in the second opt-in arithmetic (two fp16 pieces per operand,
 * three fp16 MFMA products per block; operands beyond +-65504 give NaN); the default is exact fp32 and every parity statement is made
 * with it. */
int gtsfm_linear_rowmajor_f32(const float* a_dev, int lda, int m, const int32_t* m_dev, int k, const float* w_dev, int ldw,
                              const float* bias_dev, int n, const int32_t* n_dev, float* c_dev, int ldc, int c_coff,
                              const float* res_dev, int ldres, float alpha, int relu, void* stream);

/* Pack a device activation matrix B[n][k] (row stride ldb) as the "weight" operand of gtsfm_linear_f32, so that
 * A B^T products of two activation matrices (score matrices, SG:257) use the same kernel. */
int gtsfm_pack_rows_f32(const float* b_dev, int ldb, int n, const int32_t* n_dev, int k, float* packed_dev, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * SuperPoint
 * ---------------------------------------------------------------------------------------------------------- */

/* Number of floats of the packed SuperPoint weight blob, and the packer. `tensors_host` are the 24 state_dict
 * tensors in checkpoint order: conv1a.weight, conv1a.bias, conv1b.weight, ..., convDb.weight, convDb.bias
 * (SP:119-134; torch layouts). */
size_t gtsfm_sp_packed_weight_floats(void);
int gtsfm_sp_pack_weights(const float* const* tensors_host, float* packed_host);

/* Bytes of device workspace gtsfm_sp_forward needs for `batch` images of height x width. */
size_t gtsfm_sp_workspace_bytes(int batch, int height, int width);

/* SuperPoint.forward for a batch of equally-sized gray images.                              replaces SP:145-202
 * image_dev       : [batch][height][width], fp32 in [0,1] (image_is_u8 = 0) or uint8 (image_is_u8 = 1; converted as
 *                   astype(float32) / 255.0 like gtsfm/frontend/detector_descriptor/superpoint.py:73-75)
 * capacity        : rows available per image in the output arrays; keypoints beyond it are dropped (row-major
 *                   order), kp_count_raw_dev still reports the true count
 * kp_count_dev    : [batch] int32, min(count, capacity)
 * kp_count_raw_dev: [batch] int32, true keypoint count (may be NULL)
 * kp_xy_dev       : [batch][capacity][2] fp32 (x, y) pixel coordinates, row-major (y, x) detection order = the order
 *                   of torch.nonzero (SP:170-173,187)
 * kp_score_dev    : [batch][capacity] fp32
 * desc_dev        : [batch][capacity][256] fp32, row i <-> keypoint i (the transposed layout the wrapper builds at
 *                   gtsfm/frontend/detector_descriptor/superpoint.py:84)
 * top_k           : <= 0 returns every keypoint (what SP returns with max_keypoints = -1, as GTSfM runs it). > 0 (then
 *                   capacity must equal top_k) keeps the top_k responses on the device, in detection order, and
 *                   describes only those -- the selection gtsfm/common/keypoints.py:89-110 (get_top_k) makes on
 *                   the host; used by the GPU-resident detect+match pipeline.
 * Optional taps for parity tests (NULL to skip): dense_scores_dev [batch][8*(h/8)][8*(w/8)] (pre-NMS, SP:163-166),
 * nms_scores_dev same shape (SP:167). */
int gtsfm_sp_forward(const float* packed_weights_dev, const void* image_dev, int image_is_u8, int batch, int height,
                     int width, float keypoint_threshold, int nms_radius, int remove_borders, int capacity, int top_k,
                     void* workspace_dev, size_t workspace_bytes, int32_t* kp_count_dev, int32_t* kp_count_raw_dev,
                     float* kp_xy_dev, float* kp_score_dev, float* desc_dev, float* dense_scores_dev,
                     float* nms_scores_dev, void* stream);

/* The same with the image masks GTSfM attaches to images (gtsfm/common/image.py `mask`): valid_mask_dev [batch][height][width]
 * uint8, 1 = valid (NULL = no mask). A keypoint at pixel (x, y) is kept iff valid_mask[y][x] == 1 -- Keypoints.filter_by_mask,
 * gtsfm/common/keypoints.py:112-127 -- applied BEFORE the top-k, as gtsfm/frontend/detector_descriptor/superpoint.py:76-91
 * orders them. (With a mask the nms_scores_dev tap shows the masked scores; keypoint_threshold must be positive.) */
int gtsfm_sp_forward_masked(const float* packed_weights_dev, const void* image_dev, int image_is_u8, int batch, int height,
                            int width, float keypoint_threshold, int nms_radius, int remove_borders, int capacity, int top_k,
                            void* workspace_dev, size_t workspace_bytes, int32_t* kp_count_dev, int32_t* kp_count_raw_dev,
                            float* kp_xy_dev, float* kp_score_dev, float* desc_dev, float* dense_scores_dev,
                            float* nms_scores_dev, const uint8_t* valid_mask_dev, void* stream);

/* The individual SuperPoint stages (same kernels gtsfm_sp_forward launches), for stage-wise parity tests. */

/* softmax over 65 logits per cell, drop dustbin, depth-to-space 8x8.                           replaces SP:163-166
 * logits: [batch][hc][wc][ld] (ld >= 65); scores: [batch][8*hc][8*wc] */
int gtsfm_sp_softmax_d2s(const float* logits_dev, int ld, int batch, int hc, int wc, float* scores_dev, void* stream);
/* simple_nms. scratch: 2*n bytes + n floats where n = batch*h*w (gtsfm_sp_nms_scratch_bytes).    replaces SP:47-62 */
size_t gtsfm_sp_nms_scratch_bytes(int batch, int h, int w);
int gtsfm_sp_simple_nms(const float* scores_dev, int batch, int h, int w, int radius, void* scratch_dev, float* out_dev,
                        void* stream);
/* nonzero(score > thr) + remove_borders + flip. scratch: 2*batch*h int32.            replaces SP:170-178,187 */
int gtsfm_sp_extract_keypoints(const float* nms_dev, int batch, int h, int w, float threshold, int border, int capacity,
                               int32_t* scratch_dev, int32_t* kp_count_dev, int32_t* kp_count_raw_dev, float* kp_xy_dev,
                               float* kp_score_dev, void* stream);
/* Top-k by response, survivors in detection order (host analogue: Keypoints.get_top_k,   gtsfm/common/keypoints.py:89-110).
 * in: kp_score [batch][capacity], kp_xy [batch][capacity][2], kp_count [batch]; out arrays have top_k rows per image. */
int gtsfm_sp_select_topk(const float* kp_score_dev, const float* kp_xy_dev, const int32_t* kp_count_dev, int batch,
                         int capacity, int top_k, float* out_xy_dev, float* out_score_dev, int32_t* out_count_dev,
                         void* stream);
/* L2-normalise dense descriptors, bilinear sample (align_corners=True), L2-normalise.   replaces SP:80-92,192-196
 * dense: [batch][hc*wc][ld] raw convDb output (256 channels). */
int gtsfm_sp_sample_descriptors(const float* dense_dev, int ld, int batch, int hc, int wc, const float* kp_xy_dev,
                                const int32_t* kp_count_dev, int capacity, float* desc_dev, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Matchers (SuperGlue / LightGlue): ragged batches of image pairs, token-major activations
 * ---------------------------------------------------------------------------------------------------------- */

/* Weight blobs. A blob is a sequence of 64-float-aligned entries: kind 0 = linear (W [n][k] row-major + optional
 * bias [n], stored packed for gtsfm_linear_f32), kind 1 = raw vector of n floats. The entry sequences of the two
 * matchers are documented in gtsfm_amd/runtime/matcher_engine.py (BatchNorm folded, attention heads made
 * contiguous); the forward passes walk the same sequence. */
size_t gtsfm_blob_floats(int count, const int32_t* kinds, const int32_t* n, const int32_t* k);
int gtsfm_pack_blob(int count, const int32_t* kinds, const int32_t* n, const int32_t* k, const float* const* w_host,
                    const float* const* b_host, float* packed_host);

/* Batch descriptors. A batch is `npairs` image pairs; pair p has n0[p] / n1[p] keypoints (all > 0; n1[p] == 0 is accepted for
 * the per-image phase 1 of gtsfm_{sg,lg}_forward_phase only: an odd number of images leaves the last second slot empty) and image shapes
 * hw[p] = {H0, W0, H1, W1}. Token-major inputs concatenate the keypoint sets in the order pair0/img0, pair0/img1,
 * pair1/img0, ... (T = sum of all counts rows). The int32 descriptor block (counts, per-set row offsets / image
 * shapes, per-pair score-matrix offsets, attention problem lists) is built on the host and uploaded by the caller;
 * LightGlue's point pruning rewrites the counts section on the device. */
size_t gtsfm_match_desc_ints(int superglue, int npairs, const int32_t* n0_host, const int32_t* n1_host);
int gtsfm_match_build_desc(int superglue, int npairs, const int32_t* n0_host, const int32_t* n1_host,
                           const int32_t* hw_host, int32_t* desc_host);

/* Block moves inside device memory: dst block b <- src block src_index[b] (src_index_dev == NULL: b), written at dst block
 * dst_index[b] (dst_index_dev == NULL: b); a block is block_floats contiguous floats (even).    replaces the per-pair numpy -> torch
 * marshalling of gtsfm/frontend/matcher/{superglue,lightglue}_matcher.py:75-102 in the batched pipeline: the keypoint sets of a
 * pair chunk are gathered from the resident feature table [images][max_keypoints][2 | 1 | 256] by image index. */
int gtsfm_move_blocks_f32(const float* src_dev, const int32_t* src_index_dev, float* dst_dev, const int32_t* dst_index_dev,
                          int nblocks, int64_t block_floats, void* stream);

/* out = softmax(scale * q k^T) v per head (head h = columns [64h, 64h+64)).          replaces SG:85-89,98-106
 * problems_dev: [nproblems][4] int32 {q_row_off, q_count_idx, k_row_off, k_count_idx}; counts_dev: int32 array the
 * *_count_idx fields index; max_q: upper bound of the query counts (grid sizing). */
int gtsfm_attention_f32(const float* q_dev, int ldq, const float* k_dev, int ldk, const float* v_dev, int ldv,
                        float* out_dev, int ldo, const int32_t* problems_dev, const int32_t* counts_dev, int nproblems,
                        int max_q, int heads, float scale, void* stream);
/* The same attention with its two SCHEDULES selectable (results are bit-identical): mode -1 = fused (one workgroup walks all keys of
 * its 128 queries; what gtsfm_attention_f32 runs), 1 = split (one workgroup per query tile and 1024-key segment writes an
 * unnormalised partial (O, m, l) to the workspace, a second kernel merges the segments in ascending order -- fills the chip for a
 * single pair, the per-call plugin API), 0 = chosen from the launch geometry as the matchers do. max_k: upper bound of the key
 * counts (0: unknown); rows: rows of the q / out arrays; workspace_dev: gtsfm_attention_split_workspace_bytes(...) bytes -- the
 * split schedule's partial states or, for the fused schedule, parking space for the merged state between key segments (NULL with
 * mode -1: parked in LDS, one workgroup per CU instead of two). */
size_t gtsfm_attention_split_workspace_bytes(int nproblems, int max_q, int max_k, int heads, size_t rows);
int gtsfm_attention_split_f32(const float* q_dev, int ldq, const float* k_dev, int ldk, const float* v_dev, int ldv,
                              float* out_dev, int ldo, const int32_t* problems_dev, const int32_t* counts_dev, int nproblems,
                              int max_q, int max_k, int heads, float scale, int mode, size_t rows, void* workspace_dev,
                              size_t workspace_bytes, void* stream);
/* The same attention with its ARITHMETIC selectable as well. math 0 = exact fp32 (v_mfma_f32_32x32x2_f32, bit-for-bit an fmaf chain:
 * the default everywhere and the arithmetic every parity statement of this package is made with). math 1 = "bf16x3": both products
 * (K Q^T and P V) on v_mfma_f32_32x32x16_bf16 with each fp32 operand split EXACTLY into three bf16 pieces (8 + 8 + 8 significand
 * bits) and six of the nine piece products executed, fp32 accumulation -- fp32-class error per product term (the dropped terms are
 * ~2^-24 of it, below 2^-21 in the worst case), NOT the same bits as math 0, 3/8 of its matrix-pipe time. Opt-in: the matchers take it from the environment
 * variable GTSFM_ATTENTION_MATH=bf16x3 (read per call). math 2 = "f16x2" (GTSFM_ATTENTION_MATH=f16x2): each operand carried as TWO fp16
 * pieces (hi = RN16(x), lo = RN16(x - hi)) and three products (lo hi, hi lo, hi hi) on v_mfma_f32_32x32x16_f16, fp32 accumulation --
 * per product term < 2^-20.9 |x y| worst case, 2^-24 on average (the class of math 1), 3/16 of math 0's matrix-pipe time; fp16 has no
 * exponent headroom: the softmax weights are kept <= 2^15 by the kernel itself, and a q / k / v value beyond +-65504 gives NaN output rows
 * (never a clamped value). max_k must be given (> 0) for math 1 and 2; the workspace additionally holds the split K / V tiles
 * (6 resp. 4 x heads x nproblems x ceil(max_k / 64) x 8 KiB). */
size_t gtsfm_attention_math_workspace_bytes(int nproblems, int max_q, int max_k, int heads, size_t rows, int math);
int gtsfm_attention_math_f32(const float* q_dev, int ldq, const float* k_dev, int ldk, const float* v_dev, int ldv,
                             float* out_dev, int ldo, const int32_t* problems_dev, const int32_t* counts_dev, int nproblems,
                             int max_q, int max_k, int heads, float scale, int mode, int math, size_t rows, void* workspace_dev,
                             size_t workspace_bytes, void* stream);

/* ---- input step in front of SuperPoint (uint8, OpenCV's 8-bit fixed-point arithmetic) ----
 * RGB(A) -> gray.          replaces gtsfm/utils/images.py:15-42 (cv.cvtColor COLOR_RGB2GRAY / COLOR_RGBA2GRAY), called from
 * gtsfm/frontend/detector_descriptor/superpoint.py:73. rgb_dev [H][W][channels], gray_dev [H][W]. */
int gtsfm_prep_rgb_to_gray_u8(const uint8_t* rgb_dev, int height, int width, int channels, uint8_t* gray_dev, void* stream);
/* INTER_CUBIC resize.      replaces gtsfm/utils/images.py:102-129 (cv.resize), called from gtsfm/loader/loader_base.py:160-200.
 * gtsfm_prep_cubic_taps (host): per destination index the source index of the second of the four taps and the four 11-bit
 * integer weights (float32 cubic weights, A = -0.75, rounded half to even); the caller uploads the tables of both axes.
 * src_dev [src_h][src_w][channels] -> dst_dev [dst_h][dst_w][channels], taps clamped to the image. */
int gtsfm_prep_cubic_taps(int dst_size, int src_size, int32_t* first_src_host, int16_t* weights_host);
int gtsfm_prep_resize_cubic_u8(const uint8_t* src_dev, int src_h, int src_w, int channels, const int32_t* xofs_dev, const int16_t* xw_dev,
                               const int32_t* yofs_dev, const int16_t* yw_dev, uint8_t* dst_dev, int dst_h, int dst_w, void* stream);

/* SuperGlue.forward for a batch of pairs.                                                    replaces SG:228-283
 * kpts_dev [T][2] (x, y) pixels, scores_dev [T], descriptors_dev [T][256] (the wrapper's (N, 256) layout,
 * gtsfm/frontend/matcher/superglue_matcher.py:94-99 without the transposes).
 * matches_dev [T] int32: for a keypoint of image i1 the matched index in image i2 (matches0), for a keypoint of
 * image i2 the matched index in image i1 (matches1), -1 if unmatched; mscores_dev [T]: matching_scores0/1.
 * ot_dev (optional, parity tests): final log optimal-transport matrices, pair p at the offset / row stride of the
 * descriptor block ((n0+1) x (n1+1), SG:263). Pairs with an empty keypoint set must be handled by the caller
 * (SG:233-240). */
size_t gtsfm_sg_workspace_bytes(int npairs, const int32_t* n0_host, const int32_t* n1_host);
int gtsfm_sg_forward(const float* blob_dev, int num_layers, float bin_score, int npairs, const int32_t* n0_host,
                     const int32_t* n1_host, const int32_t* desc_dev, const float* kpts_dev, const float* scores_dev,
                     const float* descriptors_dev, int sinkhorn_iters, float match_threshold, void* workspace_dev,
                     size_t workspace_bytes, int32_t* matches_dev, float* mscores_dev, float* ot_dev, void* stream);

/* Log-space Sinkhorn iterations on their own (parity tests, roofline measurement).        replaces SG:141-147,150-170
 * z_dev: couplings matrices back to back, pair p is (m[p]+1) x (n[p]+1) with row stride ld = (n[p]+1 rounded up to 4); the
 * inner m x n block holds the scores, the dustbin row / column are filled with bin_score here (SG:156-160). After `iters`
 * iterations u_dev [npairs][max(m)+1] and v_dev [npairs][max(n)+1] hold SG:143-146's u and v (Z + u + v - norm is the
 * log optimal-transport matrix). Builds and uploads its own batch descriptor: synchronises `stream` once. */
size_t gtsfm_sinkhorn_workspace_bytes(int npairs, const int32_t* m_host, const int32_t* n_host);
int gtsfm_sinkhorn_f32(float* z_dev, int npairs, const int32_t* m_host, const int32_t* n_host, float bin_score, int iters,
                       void* workspace_dev, size_t workspace_bytes, float* u_dev, float* v_dev, void* stream);

/* The score matrices of a batch of pairs in ONE ragged launch of the LDS-DMA GEMM, as gtsfm_sg_forward issues them (SG:257-258:
 * scores = einsum('bdn,bdm->bnm', mdesc0, mdesc1) / 256^.5): pair p's image-0 descriptor rows times its image-1 rows, image 1's rows
 * standing in for the weights as they lie (no packing pass). mdesc_dev: [sum(m) + sum(n)][256], pair p's m[p] image-0 rows followed by
 * its n[p] image-1 rows, pairs back to back. z_dev: couplings matrices in gtsfm_sinkhorn_f32's layout (pair p: (m[p]+1) rows of stride
 * ld = (n[p]+1 rounded up to 4)); only the inner m x n block is written: alpha * <mdesc0[i], mdesc1[j]>. Parity tests of the batched
 * product against per-pair matmuls; bench.py's roofline of the launch. Builds and uploads its own batch descriptor: synchronises
 * `stream` once. */
size_t gtsfm_score_matrices_workspace_bytes(int npairs);
int gtsfm_score_matrices_f32(const float* mdesc_dev, int npairs, const int32_t* m_host, const int32_t* n_host, float alpha, float* z_dev,
                             void* workspace_dev, size_t workspace_bytes, void* stream);

/* The same, split at the point where a pair's two images first see each other (for callers that match one image against
 * many: the per-image part runs once per image instead of once per pair; results are bit-identical to gtsfm_sg_forward).
 * phase 1: keypoint encoder + the first (self) GNN layer of every keypoint set of the batch (superglue.py:243-248, first
 * iteration of :126-137) -> x_out_dev [T][256] in the input's row order; matches / scores outputs unused (may be NULL);
 * the keypoint sets are independent here, so an odd number of images is passed with n1 = 0 in the last slot.
 * phase 2: descriptors_dev holds that x; kpts_dev / scores_dev unused; the rest of the forward. phase 0 = gtsfm_sg_forward. */
int gtsfm_sg_forward_phase(const float* blob_dev, int num_layers, float bin_score, int npairs, const int32_t* n0_host,
                           const int32_t* n1_host, const int32_t* desc_dev, const float* kpts_dev, const float* scores_dev,
                           const float* descriptors_dev, int sinkhorn_iterations, float match_threshold, void* workspace_dev,
                           size_t workspace_bytes, int32_t* matches_dev, float* mscores_dev, float* ot_dev, int phase, float* x_out_dev,
                           void* stream);

/* LightGlue(features="superpoint").forward for a batch of pairs.          replaces LG (upstream LightGlue._forward;
 * reference call site gtsfm/frontend/matcher/lightglue_matcher.py:88-110). PARITY UNPINNED: the reference does not
 * vendor LightGlue's source; this follows the published upstream algorithm (see oracle/lightglue_oracle.py).
 * kpts_dev [T][2], descriptors_dev [T][256] as above. match_bias_host [num_layers] / conf_bias_host [num_layers-1]:
 * the scalar biases of the matchability / token-confidence heads. desc_dev is READ-WRITE (live counts, stop layers).
 * depth_confidence <= 0 disables early stopping; pruning_threshold = INT32_MAX disables point pruning (upstream:
 * -1 on CPU = always prune, 1024 / 1536 on CUDA without / with flash attention).
 * matches_dev [T] int32 / mscores_dev [T] as for SuperGlue (indices refer to the ORIGINAL keypoint order).
 * sim_dev (optional, parity tests): raw similarity matrices over the kept keypoints. After the call the descriptor
 * block holds the per-pair stop layer and the final (kept) keypoint counts. */
size_t gtsfm_lg_workspace_bytes(int npairs, const int32_t* n0_host, const int32_t* n1_host);
int gtsfm_lg_forward(const float* blob_dev, int num_layers, const float* match_bias_host, const float* conf_bias_host,
                     int npairs, const int32_t* n0_host, const int32_t* n1_host, int32_t* desc_dev,
                     const float* kpts_dev, const float* descriptors_dev, float depth_confidence,
                     float width_confidence, float filter_threshold, int pruning_threshold, void* workspace_dev,
                     size_t workspace_bytes, int32_t* matches_dev, float* mscores_dev, float* sim_dev, void* stream);

/* gtsfm_lg_forward_phase with ONE pair's launch sequence split over two streams (round 5; the per-call plugin path, where a caller
 * hands over one pair at a time as gtsfm/frontend/matcher/lightglue_matcher.py:75-112 does): everything LightGlue computes per image is
 * enqueued per keypoint set -- image 0's on `stream`, image 1's on `side_stream` -- with event waits where a set needs the other's keys /
 * values and once per layer for the pair-level part; all work is joined back into `stream` before the call returns (the caller
 * synchronises `stream` only). Bit-identical to the one-stream form. Taken for npairs == 1, phase 0 / 2, exact-fp32 attention; in every
 * other case (and with side_stream == NULL) it IS gtsfm_lg_forward_phase. */
int gtsfm_lg_forward_streams(const float* blob_dev, int num_layers, const float* match_bias_host, const float* conf_bias_host,
                             int npairs, const int32_t* n0_host, const int32_t* n1_host, int32_t* desc_dev,
                             const float* kpts_dev, const float* descriptors_dev, float depth_confidence,
                             float width_confidence, float filter_threshold, int pruning_threshold, void* workspace_dev,
                             size_t workspace_bytes, int32_t* matches_dev, float* mscores_dev, float* sim_dev, int phase,
                             float* x_out_dev, void* stream, void* side_stream);

/* LightGlue's assignment stage alone, on given similarity matrices (parity tests; bench.py's rooflines of the sweep kernels the forward
 * launches).                                  replaces upstream sigmoid_log_double_softmax + filter_matches (call site
 *                                             gtsfm/frontend/matcher/lightglue_matcher.py:104-110)
 * sim_dev    : pair p's m[p] x n[p] similarities at float offset sum_{q<p} m[q] * ld[q], row stride ld[p] = n[p] rounded up to 4
 * zlogit_dev : matchability logits, token-major with every keypoint set aligned to 128 rows (pair0/img0, pair0/img1, pair1/img0, ...:
 *              set s starts at row sum of the previous sets' counts each rounded up to 128); matches_dev / mscores_dev: same layout,
 *              match index within the other set or -1, exp(score) as upstream's matching_scores
 * stages     : 1 = the two log-softmax sweeps (row / column log-sum-exp into the workspace), 2 = mutual arg-max extraction + filter
 *              (needs the workspace of a call with 1 on the same inputs), 3 = both. Builds and uploads its own batch descriptor:
 *              synchronises `stream` once -- unless 4 is added: the workspace then still holds the descriptor of an earlier call with
 *              the same shapes (timing loops: launches only). */
size_t gtsfm_lg_assignment_workspace_bytes(int npairs, const int32_t* m_host, const int32_t* n_host);
int gtsfm_lg_assignment_f32(const float* sim_dev, int npairs, const int32_t* m_host, const int32_t* n_host, const float* zlogit_dev,
                            float filter_threshold, int stages, void* workspace_dev, size_t workspace_bytes, int32_t* matches_dev,
                            float* mscores_dev, void* stream);

/* x = gelu(layer_norm(x) * gamma + beta) in place over rows of 512 columns (row stride ld >= 512): the FFN's normalisation /
 * activation of a LightGlue block (upstream nn.LayerNorm(512) + nn.GELU between ffn.0 and ffn.3), stand-alone for parity tests and
 * bench.py's roofline. scratch_dev: 64 bytes of device memory; synchronises `stream` once. */
int gtsfm_layernorm_gelu_f32(float* x_dev, int ld, int rows, const float* gamma_dev, const float* beta_dev, void* scratch_dev, void* stream);

/* LightGlue split the same way: phase 1 = the first layer's SELF block (rotary self-attention + FFN of one image) -> x_out_dev;
 * phase 2 = descriptors_dev holds that x, the first self block is skipped. Bit-identical to gtsfm_lg_forward (phase 0). */
int gtsfm_lg_forward_phase(const float* blob_dev, int num_layers, const float* match_bias_host, const float* conf_bias_host,
                           int npairs, const int32_t* n0_host, const int32_t* n1_host, int32_t* desc_dev, const float* kpts_dev,
                           const float* descriptors_dev, float depth_confidence, float width_confidence, float filter_threshold,
                           int pruning_threshold, void* workspace_dev, size_t workspace_bytes, int32_t* matches_dev,
                           float* mscores_dev, float* sim_dev, int phase, float* x_out_dev, void* stream);

/* ---- verifier stage behind the matcher (float64) ----
 * OpencvVerifierBase.verify with use_intrinsics_in_verification=True for a batch of pairs.
 *                                  replaces gtsfm/frontend/verifier/opencv_verifier_base.py:47-111 + ransac.py:52-84
 *                                  (cv2.findEssentialMat) + gtsfm/utils/verification.py:54-96 (cv.recoverPose),
 *                                  called from gtsfm/two_view_estimator.py:391-397.
 * PARITY UNPINNED (OpenCV absent, its sampler not reproducible): Nister's five-point solver, squared Sampson error
 * (gtsfm/utils/verification.py:172-220), RANSAC with a counter-based sampler (splitmix64 of seed / hypothesis / attempt),
 * 256 hypotheses per round scored by MSAC, at most 4 rounds, stop when (1 - w^5)^n <= 1e-6, then one round of 256 samples
 * drawn from the winner's inliers (LO-RANSAC inner sampling), cheirality choice, and six Gauss-Newton steps on the inliers'
 * Sampson error over the pose (kept when the MSAC cost drops); see oracle/verifier_oracle.py.
 * kp_xy_dev [*][2] float32 pixel coordinates of all keypoint tables; pair p uses the tables starting at rows kp_off1_dev[p]
 * (image i1) and kp_off2_dev[p] (image i2). match_idx_dev [total_matches][2] int32 (row in i1's table, row in i2's table),
 * pair p owning rows match_off_dev[p] .. match_off_dev[p+1], or only the first match_count_dev[p] of them when match_count_dev
 * is given (capacity layout, filled by gtsfm_verify_compact_matches; total_matches = match_off_dev[num_pairs]). intrinsics_dev [num_pairs][8] = fx, fy, cx, cy of i1 then i2
 * (pinhole; lens distortion is the caller's to remove). threshold_px is divided by max(fx1, fx2) as the reference does.
 * Outputs per pair: essential_dev [9] i2Ei1 (unnormalised), rotation_dev [9] i2Ri1 row-major, translation_dev [3] unit i2Ui1,
 * inlier_mask_dev [total_matches] (1 = verified), stats_dev [8] int32 = inliers, hypotheses drawn, winning hypothesis,
 * winning root, cheirality counts of (R1,t) (R2,t) (R1,-t) (R2,-t). Pairs with fewer than 6 matches or no model: inliers = 0,
 * NaN matrices (the reference's failure tuple is built by the caller). */
size_t gtsfm_verify_workspace_bytes(long long total_matches);
int gtsfm_verify_essential_f64(const float* kp_xy_dev, const long long* kp_off1_dev, const long long* kp_off2_dev,
                               const int32_t* match_idx_dev, const long long* match_off_dev, const int32_t* match_count_dev,
                               long long total_matches, const double* intrinsics_dev, const unsigned long long* seeds_dev, double threshold_px,
                               int num_pairs, void* workspace_dev, size_t workspace_bytes, double* essential_dev,
                               double* rotation_dev, double* translation_dev, uint8_t* inlier_mask_dev, int32_t* stats_dev,
                               void* stream);

/* The same with use_intrinsics_in_verification=False: fundamental matrix from pixel coordinates, then E = K2^T F K1.
 *                                  replaces opencv_verifier_base.py:91-97 + ransac.py:86-112 (cv2.findFundamentalMat(FM_RANSAC),
 *                                  seven-point samples) + gtsfm/utils/verification.py:99-112 (fundamental_to_essential_matrix).
 * Residual = OpenCV's for this estimator, the larger squared point-to-epipolar-line distance, against threshold_px^2 (not
 * divided by a focal length); minimal sample 7, at least 8 matches (NUM_MATCHES_REQ_F_MATRIX). fundamental_dev [9] = i2Fi1;
 * essential_dev [9] = K2^T F K1; pose and cheirality on the normalised coordinates as above. PARITY UNPINNED as above. */
int gtsfm_verify_fundamental_f64(const float* kp_xy_dev, const long long* kp_off1_dev, const long long* kp_off2_dev,
                                 const int32_t* match_idx_dev, const long long* match_off_dev, const int32_t* match_count_dev,
                                 long long total_matches, const double* intrinsics_dev, const unsigned long long* seeds_dev,
                                 double threshold_px, int num_pairs, void* workspace_dev, size_t workspace_bytes, double* fundamental_dev,
                                 double* essential_dev, double* rotation_dev, double* translation_dev, uint8_t* inlier_mask_dev,
                                 int32_t* stats_dev, void* stream);

/* Matcher output -> the verifier's match lists on the device.   replaces the host marshalling of
 * gtsfm/frontend/matcher/superglue_matcher.py:100-102 / lightglue_matcher.py:104-110 ((K, 2) arrays per pair) between the
 * matcher and the verifier. matches_dev: gtsfm_sg_forward / gtsfm_lg_forward output; pair p's matches0 block starts at row
 * row_off_dev[p] and has n0_dev[p] rows. Writes the (row, matches0[row]) pairs with matches0[row] > -1, in row order, at
 * match_idx_dev + 2 * match_off_dev[p] (capacity n0_dev[p]) and their number to match_count_dev[p]. */
int gtsfm_verify_compact_matches(const int32_t* matches_dev, const long long* row_off_dev, const int32_t* n0_dev,
                                 const long long* match_off_dev, int num_pairs, int32_t* match_idx_dev, int32_t* match_count_dev,
                                 void* stream);

#ifdef __cplusplus
}
#endif

#endif /* GTSFM_AMD_H */
