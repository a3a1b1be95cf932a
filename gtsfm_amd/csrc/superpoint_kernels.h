// Launchers for the SuperPoint-specific kernels (see superpoint_kernels.hip).
#pragma once

#include "common.h"

int launch_conv1a(const void* img, int is_u8, int B, int H, int W, const float* wpack, const float* bias, float* out,
                  hipStream_t stream);
int launch_softmax_d2s(const float* logits, int ld, int B, int Hc, int Wc, float* scores, hipStream_t stream);
int launch_simple_nms(const float* S, int B, int H, int W, int radius, uint8_t* mask, uint8_t* supp, float* SS, float* out,
                      hipStream_t stream);
// nms[b][y][x] = 0 where valid_mask[b][y][x] != 1 (valid_mask: [B][img_h][img_w] uint8; nms: [B][H][W], H <= img_h, W <= img_w)
int launch_apply_keypoint_mask(float* nms, int B, int H, int W, const uint8_t* valid_mask, int img_h, int img_w, hipStream_t stream);
int launch_extract_keypoints(const float* nms, int B, int H, int W, float thr, int border, int capacity, int* rowcnt, int* rowoff,
                             int* count, int* count_raw, float* kp_xy, float* kp_score, hipStream_t stream);
int launch_sample_descriptors(const float* dense, int ld, int B, int Hc, int Wc, const float* kp_xy, const int* count, int capacity,
                              float* desc, hipStream_t stream);
int launch_select_topk(const float* scores, const int* count, int B, int cap, int k, const float* xy, float* out_xy, float* out_score,
                       int* out_count, hipStream_t stream);
