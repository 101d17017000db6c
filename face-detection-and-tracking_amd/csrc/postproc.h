// Internal (library-side) launch interface of postproc.hip; device pointers, async on `st`.
#pragma once
#include <hip/hip_runtime.h>

namespace fdt {

struct DetectPlan {
  int B = 0, P = 0, K = 0, Kp = 0;
  long long key_stride = 0;
  long long off_cand = 0, off_keys = 0, off_sbox = 0, off_sarea = 0, off_sscore = 0, off_sidx = 0,
            off_mask = 0, bytes = 0;
};

DetectPlan make_detect_plan(int B, int P, int nms_top_k);

int launch_priorbox(int width, int height, int stride, int box, int n_scales, const double* ar_dev,
                    int n_ar, int f_w, int f_h, float* out_dev, hipStream_t st);
int launch_decode(const float* loc, const float* pri, int P, float v0, float v1, float* out,
                  hipStream_t st);
int launch_detect(const DetectPlan& pl, void* ws, const float* loc, const float* conf,
                  const float* pri, int num_classes, int top_k, float conf_t, float nms_t, float v0,
                  float v1, float* out, int* counts, hipStream_t st);
int launch_nms(const DetectPlan& pl, void* ws, const float* boxes, const float* scores,
               float overlap, long long* keep, int* count, hipStream_t st);
int launch_facebox_decode(const DetectPlan& pl, void* ws, const float* loc, const float* conf,
                          const float* anchors, float conf_t, float nms_t, float* boxes, float* probs,
                          int* counts, hipStream_t st);
int launch_facebox_anchors(float* out, hipStream_t st);
// measure 0: calculate_iou, 1: calculate_distance (utils/calc_performance.py:54-74 / :34-51)
int launch_pairwise_iou(const void* a, int A, const void* b, int B, int dtype, void* out,
                        hipStream_t st, int measure = 0);

}  // namespace fdt
