// SSD post-processing on gfx950: prior boxes, decode, threshold+compaction, sort, greedy NMS as a
// suppression bit-matrix + wave scan, top-k pack, and pairwise IoU.
//
// Reference semantics (all arithmetic in the reference's operand order, FMA contraction OFF for this
// file so that every f32/f64 result is bit-identical to the reference's IEEE basic operations):
//   layers/functions/prior_box.py:28-44   PriorBoxLayer.__call__
//   layers/box_utils.py:238-258           decode
//   layers/box_utils.py:275-340           nms
//   layers/functions/detection.py:34-84   Detect.__call__
//   utils/calc_performance.py:4-31,54-74  intersect / calculate_iou
//
// All of this is HBM/latency-bound integer + f32 work (2.1 MB read per 1024x1024 frame); nothing here
// is shaped into a GEMM.  Wave64 everywhere: ballots and readlanes are 64 wide.
#include "common.h"
#include "postproc.h"

namespace fdt {

namespace {

constexpr int kSortLds = 4096;  // keys sorted entirely in LDS (32 KiB) when the candidate count fits

__device__ __forceinline__ unsigned ord_f32(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---------------------------------------------------------------------------------------- priors
// prior_box.py:31-41: cx=(j+0.5)*stride/width etc. in f64, one f32 rounding at the end (:43).
__global__ void priorbox_kernel(int width, int height, int stride, int box, int n_scales,
                                const double* __restrict__ ar, int n_ar, int f_w, int f_h,
                                float* __restrict__ out) {
  const int per_cell = n_scales * (1 + n_ar);
  const long long total = (long long)f_w * f_h * per_cell;
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  int k = (int)(t % per_cell);
  long long cell = t / per_cell;
  int j = (int)(cell % f_w), i = (int)(cell / f_w);
  int scale = k / (1 + n_ar), a = k % (1 + n_ar);
  double cx = (j + 0.5) * stride / width;
  double cy = (i + 0.5) * stride / height;
  // (2 ** (1/3)) ** scale, with Python's float pow semantics: scale==0 -> exactly 1.0
  const double cbrt2 = 1.2599210498948732;  // 2 ** (1 / 3) as Python evaluates it
  double box_scale = scale == 0 ? 1.0 : (scale == 1 ? cbrt2 : pow(cbrt2, (double)scale));
  double sx = box * box_scale / width;
  double sy = box * box_scale / height;
  if (a > 0) {
    double r = sqrt(ar[a - 1]);
    sx = sx / r;
    sy = sy * r;
  }
  float* o = out + t * 4;
  o[0] = (float)cx;
  o[1] = (float)cy;
  o[2] = (float)sx;
  o[3] = (float)sy;
}

// ---------------------------------------------------------------------------------------- decode
// box_utils.py:254-258.  (loc*v0)*p_wh then add; wh = p_wh*exp(loc*v1); x1y1 = cxcy - wh/2;
// x2y2 = wh + x1y1.
__device__ __forceinline__ float4 decode_one(float4 l, float4 p, float v0, float v1) {
  float cx = p.x + (l.x * v0) * p.z;
  float cy = p.y + (l.y * v0) * p.w;
  float w = p.z * expf(l.z * v1);
  float h = p.w * expf(l.w * v1);
  float x1 = cx - w / 2.0f;
  float y1 = cy - h / 2.0f;
  return make_float4(x1, y1, w + x1, h + y1);
}

// FaceBox: DataEncoder.decode_np (FACEBOX/encoderl.py:318-320): cxcy = loc*0.1*a_wh + a_xy;
// wh = exp(loc*0.2)*a_wh; box = [cxcy - wh/2, cxcy + wh/2]   (x2 = cx + w/2, unlike decode()).
__device__ __forceinline__ float4 decode_facebox(float4 l, float4 p) {
  float cx = (l.x * 0.1f) * p.z + p.x;
  float cy = (l.y * 0.1f) * p.w + p.y;
  float w = expf(l.z * 0.2f) * p.z;
  float h = expf(l.w * 0.2f) * p.w;
  return make_float4(cx - w / 2.0f, cy - h / 2.0f, cx + w / 2.0f, cy + h / 2.0f);
}

// FaceBox anchors: DataEncoder.__init__ (FACEBOX/encoderl.py:21-47), f64 then one rounding to f32.
// level 0: 32x32 cells x (16 anchors of 32 px on a 4x4 sub-grid, 4 of 64 px on 2x2, 1 of 128 px);
// levels 1,2: one anchor of 256 / 512 px per cell of the 16x16 / 8x8 maps.  21 824 anchors.
__global__ void facebox_anchors_kernel(float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 21824) return;
  const double scale = 1024.0;
  double cx, cy, sz;
  if (t < 32 * 32 * 21) {
    const int cell = t / 21, k = t % 21;
    const int hh = cell / 32, ww = cell % 32;
    const double step = 32.0 / scale, s = 32.0 / scale;
    cx = (ww + 0.5) * step;
    cy = (hh + 0.5) * step;
    double ar, dx, dy;
    if (k < 16) {          // ar = 1, density [-3,-1,1,3]; itertools.product: dx outer, dy inner
      ar = 1.0; dx = -3.0 + 2.0 * (k / 4); dy = -3.0 + 2.0 * (k % 4);
    } else if (k < 20) {   // ar = 2, density [-1,1]
      ar = 2.0; dx = -1.0 + 2.0 * ((k - 16) / 2); dy = -1.0 + 2.0 * ((k - 16) % 2);
    } else {               // ar = 4, density [0]
      ar = 4.0; dx = 0.0; dy = 0.0;
    }
    cx = cx + dx / 8. * s * ar;
    cy = cy + dy / 8. * s * ar;
    sz = s * ar;
  } else {
    int r = t - 32 * 32 * 21;
    int fm = 16;
    double step = 64.0 / scale, s = 256.0 / scale;
    if (r >= 256) { r -= 256; fm = 8; step = 128.0 / scale; s = 512.0 / scale; }
    const int hh = r / fm, ww = r % fm;
    cx = (ww + 0.5) * step;
    cy = (hh + 0.5) * step;
    sz = s * 1.0;
  }
  reinterpret_cast<float4*>(out)[t] = make_float4((float)cx, (float)cy, (float)sz, (float)sz);
}

__global__ void decode_kernel(const float4* __restrict__ loc, const float4* __restrict__ pri, int P,
                              float v0, float v1, float4* __restrict__ out) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) out[p] = decode_one(loc[p], pri[p], v0, v1);
}

// ---------------------------------------------------------------------------------------- detect
__global__ void reset_kernel(float* __restrict__ out, long long n_out, int* __restrict__ counts,
                             int n_counts, int* __restrict__ cand, int n_img) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long step = (long long)gridDim.x * blockDim.x;
  for (long long i = t; i < n_out; i += step) out[i] = 0.0f;
  if (counts)
    for (long long i = t; i < n_counts; i += step) counts[i] = 0;
  for (long long i = t; i < n_img; i += step) cand[i] = 0;
}

// detection.py:64: mask = conf_scores[cl].gt(conf_thresh) (strict, f32).  Unordered compaction; the
// order is fixed by the sort on the unique 64-bit key (ordered score bits << 32 | prior index).
__global__ void compact_kernel(const float* __restrict__ scores, int P, int score_stride, int score_off,
                               long long img_stride, float thr, int use_thr,
                               unsigned long long* __restrict__ keys, long long key_stride,
                               int* __restrict__ cand) {
  int b = blockIdx.y;
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float s = scores[b * img_stride + (long long)p * score_stride + score_off];
  if (!use_thr || s > thr) {
    int pos = atomicAdd(&cand[b], 1);
    keys[b * key_stride + pos] = ((unsigned long long)ord_f32(s) << 32) | (unsigned)p;
  }
}

__device__ __forceinline__ void cmpx_desc(unsigned long long& a, unsigned long long& b, bool desc) {
  if (desc ? (a < b) : (a > b)) {
    unsigned long long t = a;
    a = b;
    b = t;
  }
}

// box_utils.py:296-298: ascending sort, keep the last top_k, walk from the end  ==  descending
// order by (score, index).  One 1024-thread workgroup per image: bitonic network in LDS when the
// candidate count fits, otherwise in global memory (rare: > 4096 candidates).
__global__ __launch_bounds__(1024) void sort_kernel(unsigned long long* __restrict__ keys,
                                                    long long key_stride,
                                                    const int* __restrict__ cand) {
  __shared__ unsigned long long sk[kSortLds];
  const int b = blockIdx.x;
  const int n = cand[b];
  unsigned long long* k = keys + b * key_stride;
  if (n <= 1) return;
  int N = 2;
  while (N < n) N <<= 1;
  const int tid = threadIdx.x;
  if (N <= kSortLds) {
    for (int i = tid; i < N; i += 1024) sk[i] = (i < n) ? k[i] : 0ull;
    __syncthreads();
    for (int kk = 2; kk <= N; kk <<= 1) {
      for (int j = kk >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < N; i += 1024) {
          int l = i ^ j;
          if (l > i) {
            unsigned long long a = sk[i], c = sk[l];
            cmpx_desc(a, c, (i & kk) == 0);
            sk[i] = a;
            sk[l] = c;
          }
        }
        __syncthreads();
      }
    }
    for (int i = tid; i < n; i += 1024) k[i] = sk[i];
  } else {
    for (int i = n + tid; i < N; i += 1024) k[i] = 0ull;  // key_stride >= pow2ceil(P)
    __syncthreads();
    for (int kk = 2; kk <= N; kk <<= 1) {
      for (int j = kk >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < N; i += 1024) {
          int l = i ^ j;
          if (l > i) {
            unsigned long long a = k[i], c = k[l];
            cmpx_desc(a, c, (i & kk) == 0);
            k[i] = a;
            k[l] = c;
          }
        }
        __syncthreads();  // workgroup-scope release/acquire: one CU, one L1
      }
    }
  }
}

// Gather the m = min(n, K) best candidates in score order; boxes are decoded here (Detect) or copied
// (stand-alone nms).  area = (x2-x1)*(y2-y1), no +1 (box_utils.py:295).
__global__ void gather_kernel(const unsigned long long* __restrict__ keys, long long key_stride,
                              const int* __restrict__ cand, int K, int Kp,
                              const float4* __restrict__ loc, const float4* __restrict__ pri,
                              const float4* __restrict__ boxes_in, int P, float v0, float v1, int facebox,
                              const float* __restrict__ scores, int score_stride, int score_off,
                              long long score_img_stride, float4* __restrict__ sbox,
                              float* __restrict__ sarea, float* __restrict__ sscore,
                              int* __restrict__ sidx) {
  int b = blockIdx.y;
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  int n = cand[b];
  int m = n < K ? n : K;
  if (r >= m) return;
  unsigned p = (unsigned)(keys[b * key_stride + r] & 0xffffffffull);
  float4 bx;
  if (boxes_in)
    bx = boxes_in[(long long)b * P + p];
  else if (facebox)
    bx = decode_facebox(loc[(long long)b * P + p], pri[p]);
  else
    bx = decode_one(loc[(long long)b * P + p], pri[p], v0, v1);
  long long o = (long long)b * Kp + r;
  sbox[o] = bx;
  sarea[o] = (bx.z - bx.x) * (bx.w - bx.y);
  sscore[o] = scores[b * score_img_stride + (long long)p * score_stride + score_off];
  sidx[o] = (int)p;
}

// Suppression bit-matrix.  Bit (i, j), j > i (i has the higher rank), is set when box j would be
// removed after keeping box i: NOT (IoU < overlap), with the reference's operand order
// (box_utils.py:321-339): inter = clamp(min(x2)-max(x1),0)*clamp(..); union = (area_j - inter) + area_i.
// One wave per 64x64 tile; the column boxes sit in LDS and every lane builds one 64-bit row word.
__global__ __launch_bounds__(64) void mask_kernel(const float4* __restrict__ sbox,
                                                  const float* __restrict__ sarea,
                                                  const int* __restrict__ cand, int K, int Kp,
                                                  float overlap, int facebox,
                                                  unsigned long long* __restrict__ mask) {
  // The grid is capped (launch_* below): each workgroup strides over the 64x64 tiles of the upper triangle that
  // the image's candidate count really has -- FaceBoxes sizes K for all 21824 anchors (342 x 342 tiles per image),
  // of which a real frame fills a handful.
  const int b = blockIdx.z;
  const int n = cand[b];
  const int m = n < K ? n : K;
  const int mt = (m + 63) >> 6;
  __shared__ float4 cb[64];
  __shared__ float ca[64];
  const int lane = threadIdx.x;
  const long long base = (long long)b * Kp;
  const int nw = Kp >> 6;
  for (int bi = blockIdx.y; bi < mt; bi += gridDim.y) {
    const int i = bi * 64 + lane;
    float4 bi4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float ai = 0.f;
    if (i < m) {
      bi4 = sbox[base + i];
      ai = sarea[base + i];
    }
    for (int bj = bi + (blockIdx.x + gridDim.x - bi % gridDim.x) % gridDim.x; bj < mt; bj += gridDim.x) {
      __syncthreads();
      {
        int j = bj * 64 + lane;
        if (j < m) {
          cb[lane] = sbox[base + j];
          ca[lane] = sarea[base + j];
        }
      }
      __syncthreads();
      if (i >= m) continue;
      unsigned long long word = 0;
      const int jmax = (m - bj * 64) < 64 ? (m - bj * 64) : 64;
      for (int c = 0; c < jmax; ++c) {
        int j = bj * 64 + c;
        if (j <= i) continue;
        float4 bb = cb[c];
        float xx1 = fmaxf(bb.x, bi4.x);
        float yy1 = fmaxf(bb.y, bi4.y);
        float xx2 = fminf(bb.z, bi4.z);
        float yy2 = fminf(bb.w, bi4.w);
        float w = xx2 - xx1;
        float h = yy2 - yy1;
        w = (w < 0.0f) ? 0.0f : w;
        h = (h < 0.0f) ? 0.0f : h;
        float inter = w * h;
        // box_utils.py:336: (area_j - inter) + area_i;  nms_np (encoderl.py:251): (area_i + area_j) - inter
        float uni = facebox ? ((ai + ca[c]) - inter) : ((ca[c] - inter) + ai);
        float iou = inter / uni;
        if (!(iou < overlap)) word |= (1ull << c);
      }
      mask[(base + i) * nw + bj] = word;
    }
  }
}

// Greedy scan over the bit-matrix, one wave per image.  Per 64-row chunk: resolve the chunk serially
// from its diagonal words (readlane), then OR the kept rows' words into the running `removed`
// bitmap, one word per lane.  Equals the reference's sequential loop exactly (box_utils.py:308-339).
// Emits Detect rows [score,x1,y1,x2,y2] (detection.py:80-82) and/or the kept indices.
__global__ __launch_bounds__(64) void scan_kernel(
    const unsigned long long* __restrict__ mask, const float4* __restrict__ sbox,
    const float* __restrict__ sscore, const int* __restrict__ sidx, const int* __restrict__ cand,
    int K, int Kp, int max_keep, int skip_single, float* __restrict__ out, long long out_img_stride,
    int* __restrict__ counts, int count_stride, int count_off, long long* __restrict__ keep_idx,
    long long keep_stride, float* __restrict__ fb_boxes, float* __restrict__ fb_probs, int fb_stride) {
  extern __shared__ unsigned long long removed[];  // Kp/64 words
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  const int n = cand[b];
  int m = n < K ? n : K;
  // detection.py:66-72: exactly one candidate -> 0-dim tensor -> `continue`: nothing emitted
  if (skip_single && n == 1) m = 0;
  const int nwt = Kp >> 6;
  const int nw = (m + 63) >> 6;
  for (int w = lane; w < nw; w += 64) removed[w] = 0ull;
  __syncthreads();
  const long long base = (long long)b * Kp;
  int kept_total = 0;
  const int limit = max_keep < m ? max_keep : m;
  for (int c = 0; c < nw && kept_total < limit; ++c) {
    unsigned long long rem = removed[c];
    const int i = c * 64 + lane;
    unsigned long long diag = (i < m) ? mask[(base + i) * nwt + c] : 0ull;
    unsigned long long keptbits = 0;
    const int rows = (m - c * 64) < 64 ? (m - c * 64) : 64;
    int kt = kept_total;
    // Only the rows still alive are visited, in order: the next kept row is the lowest alive bit, its diagonal word removes
    // rows behind it.  Same decisions as testing the 64 rows one by one (which made dense frames -- thousands of candidates,
    // a handful kept per chunk -- pay 64 cross-lane reads per chunk); everything here is wave-uniform.
    unsigned long long alive = ~rem & (rows == 64 ? ~0ull : ((1ull << rows) - 1ull));
    while (alive && kt < limit) {
      const int r = __ffsll((long long)alive) - 1;
      keptbits |= (1ull << r);
      ++kt;
      rem |= __shfl(diag, r, 64);
      alive &= ~rem;                                   // suppressed by a kept row
      alive &= ~((2ull << r) - 1ull);                  // rows <= r are decided (r = 63: the mask is all ones)
    }
    // emit this chunk's kept rows in rank order
    if ((keptbits >> lane) & 1ull) {
      int row = kept_total + __popcll(keptbits & ((1ull << lane) - 1ull));
      long long src = base + i;
      if (out) {
        float4 bx = sbox[src];
        float* o = out + b * out_img_stride + (long long)row * 5;
        o[0] = sscore[src];
        o[1] = bx.x;
        o[2] = bx.y;
        o[3] = bx.z;
        o[4] = bx.w;
      }
      if (keep_idx) keep_idx[b * keep_stride + row] = (long long)sidx[src];
      if (fb_boxes) {   // boxes[keep], score[ids][keep]  (encoderl.py:325)
        reinterpret_cast<float4*>(fb_boxes)[(long long)b * fb_stride + row] = sbox[src];
        fb_probs[(long long)b * fb_stride + row] = sscore[src];
      }
    }
    kept_total = kt;
    if (kept_total >= limit) break;
    // fold the kept rows into the removed bitmap of the later chunks
    // (eight rows per round trip: one dependent load per kept row made this loop the whole kernel's time)
#ifndef FDT_SCAN_EXP
#define FDT_SCAN_EXP 0   // timing experiment: 1 = the kept rows are not folded into the later words (wrong results)
#endif
    for (int w = c + 1 + lane; w < (FDT_SCAN_EXP == 1 ? 0 : nw); w += 64) {
      unsigned long long acc = 0;
      unsigned long long kb = keptbits;
      while (kb) {
        int rr[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          rr[q] = kb ? __ffsll((long long)kb) - 1 : -1;
          kb &= kb - (kb ? 1ull : 0ull);
        }
        unsigned long long v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = rr[q] >= 0 ? mask[(base + c * 64 + rr[q]) * nwt + w] : 0ull;
#pragma unroll
        for (int q = 0; q < 8; ++q) acc |= v[q];
      }
      removed[w] |= acc;
    }
    __syncthreads();
  }
  if (lane == 0 && counts) counts[b * count_stride + count_off] = kept_total;
}

// ---------------------------------------------------------------------------------------- IoU
template <typename T>
__device__ __forceinline__ T np_min(T x, T y) {  // numpy.minimum: NaN propagates
  return (x != x) ? x : ((y != y) ? y : (x < y ? x : y));
}
template <typename T>
__device__ __forceinline__ T np_max(T x, T y) {
  return (x != x) ? x : ((y != y) ? y : (x > y ? x : y));
}

// calc_performance.py:4-31,54-74: inter = prod(max(min(a_hi,b_hi) - max(a_lo,b_lo), 0));
// union = area_a + area_b - inter; no epsilon (0/0 -> NaN).
template <typename T>
__device__ __forceinline__ T iou_one(const T* a, const T* b) {
  T dx = np_min(a[2], b[2]) - np_max(a[0], b[0]);
  T dy = np_min(a[3], b[3]) - np_max(a[1], b[1]);
  dx = np_max(dx, (T)0);
  dy = np_max(dy, (T)0);
  T inter = dx * dy;
  T area_a = (a[2] - a[0]) * (a[3] - a[1]);
  T area_b = (b[2] - b[0]) * (b[3] - b[1]);
  T uni = area_a + area_b - inter;
  return inter / uni;
}

template <typename T>
__global__ void pairwise_iou_kernel(const T* __restrict__ a, int A, const T* __restrict__ b, int B,
                                    T* __restrict__ out) {
  // 64 consecutive lanes walk the B axis (coalesced stores); box_a is wave-uniform per row.
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int i = blockIdx.y;
  if (j >= B || i >= A) return;
  T av[4] = {a[i * 4 + 0], a[i * 4 + 1], a[i * 4 + 2], a[i * 4 + 3]};
  T bv[4] = {b[j * 4 + 0], b[j * 4 + 1], b[j * 4 + 2], b[j * 4 + 3]};
  out[(long long)i * B + j] = iou_one<T>(av, bv);
}

// calculate_distance(box_a, box_b)  utils/calc_performance.py:34-51, the association measure of the tracker's
// use_iou = False branch (iouTracke_cal.py:136-138): with w/h extents d = box[2:] - box[:2] and centres c = (box[2:] + box[:2]) / 2,
//   delt_z = ((da.x - db.x) + (da.y - db.y)) / 2;   dis = (delt_z^2 + (cb.x - ca.x)^2 + (cb.y - ca.y)^2) ** 0.25
// in the reference's operand order (this TU is compiled with -ffp-contract=off).  `** 0.25` is numpy's pow, which is
// libm / SIMD-library dependent (within 1 ulp, NOT correctly rounded: measured 5 % of f64 inputs off by one ulp in the build
// container's numpy 2.2); here the fourth root IS correctly rounded -- two IEEE square roots plus one exact-residual Newton
// step -- so it agrees with any faithful pow to 1 ulp and with a correctly rounded one bit for bit.
__device__ __forceinline__ double root4_cr(double x) {
  const double r = sqrt(sqrt(x));
  if (!(r > 0.0) || r > 1.7e308) return r;           // 0, NaN, inf: sqrt already gave numpy's answer
  const double r2 = r * r, r2e = fma(r, r, -r2);      // r^2 = r2 + r2e exactly
  const double r4 = r2 * r2, r4e = fma(r2, r2, -r4) + 2.0 * r2 * r2e;
  const double e = (x - r4) - r4e;                    // x - r^4, to well below an ulp of x
  return r + e / (4.0 * r2 * r);
}
template <typename T>
__device__ __forceinline__ T distance_one(const T* a, const T* b) {
  const T adx = a[2] - a[0], ady = a[3] - a[1], bdx = b[2] - b[0], bdy = b[3] - b[1];
  const T cax = (a[2] + a[0]) / (T)2, cay = (a[3] + a[1]) / (T)2;
  const T cbx = (b[2] + b[0]) / (T)2, cby = (b[3] + b[1]) / (T)2;
  const T dx = cbx - cax, dy = cby - cay;
  const T dz = ((adx - bdx) + (ady - bdy)) / (T)2;
  const T dis = dz * dz + dx * dx + dy * dy;
  return (T)root4_cr((double)dis);
}

template <typename T>
__global__ void pairwise_distance_kernel(const T* __restrict__ a, int A, const T* __restrict__ b, int B,
                                         T* __restrict__ out) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int i = blockIdx.y;
  if (j >= B || i >= A) return;
  T av[4] = {a[i * 4 + 0], a[i * 4 + 1], a[i * 4 + 2], a[i * 4 + 3]};
  T bv[4] = {b[j * 4 + 0], b[j * 4 + 1], b[j * 4 + 2], b[j * 4 + 3]};
  out[(long long)i * B + j] = distance_one<T>(av, bv);
}

inline long long align_up(long long x, long long a) { return (x + a - 1) / a * a; }

}  // namespace

// ------------------------------------------------------------------------------------------------
DetectPlan make_detect_plan(int B, int P, int K) {
  DetectPlan pl;
  pl.B = B;
  pl.P = P;
  pl.K = K < P ? K : P;
  if (pl.K < 1) pl.K = 1;
  pl.Kp = (int)align_up(pl.K, 64);
  long long pp = 2;
  while (pp < P) pp <<= 1;
  pl.key_stride = pp;
  long long off = 0;
  pl.off_cand = off;
  off = align_up(off + (long long)B * 4, 256);
  pl.off_keys = off;
  off = align_up(off + (long long)B * pl.key_stride * 8, 256);
  pl.off_sbox = off;
  off = align_up(off + (long long)B * pl.Kp * 16, 256);
  pl.off_sarea = off;
  off = align_up(off + (long long)B * pl.Kp * 4, 256);
  pl.off_sscore = off;
  off = align_up(off + (long long)B * pl.Kp * 4, 256);
  pl.off_sidx = off;
  off = align_up(off + (long long)B * pl.Kp * 4, 256);
  pl.off_mask = off;
  off = align_up(off + (long long)B * pl.Kp * (pl.Kp / 64) * 8, 256);
  pl.bytes = off;
  return pl;
}

int launch_priorbox(int width, int height, int stride, int box, int n_scales, const double* ar_dev,
                    int n_ar, int f_w, int f_h, float* out_dev, hipStream_t st) {
  long long total = (long long)f_w * f_h * n_scales * (1 + n_ar);
  if (total == 0) return FDT_OK;
  int grid = (int)ceil_div_ll(total, 256);
  hipLaunchKernelGGL(priorbox_kernel, dim3(grid), dim3(256), 0, st, width, height, stride, box,
                     n_scales, ar_dev, n_ar, f_w, f_h, out_dev);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

int launch_decode(const float* loc, const float* pri, int P, float v0, float v1, float* out,
                  hipStream_t st) {
  if (P == 0) return FDT_OK;
  hipLaunchKernelGGL(decode_kernel, dim3(ceil_div(P, 256)), dim3(256), 0, st,
                     (const float4*)loc, (const float4*)pri, P, v0, v1, (float4*)out);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

// Shared tail of Detect and stand-alone nms: sort -> gather -> mask -> scan.
static int run_sorted_nms(const DetectPlan& pl, char* ws, const float* loc, const float* pri,
                          const float* boxes_in, float v0, float v1, int facebox, float* fb_boxes,
                          float* fb_probs, const float* scores,
                          int score_stride, int score_off, long long score_img_stride, float overlap,
                          int max_keep, int skip_single, float* out, long long out_img_stride,
                          int* counts, int count_stride, int count_off, long long* keep_idx,
                          long long keep_stride, hipStream_t st) {
  int* cand = (int*)(ws + pl.off_cand);
  unsigned long long* keys = (unsigned long long*)(ws + pl.off_keys);
  float4* sbox = (float4*)(ws + pl.off_sbox);
  float* sarea = (float*)(ws + pl.off_sarea);
  float* sscore = (float*)(ws + pl.off_sscore);
  int* sidx = (int*)(ws + pl.off_sidx);
  unsigned long long* mask = (unsigned long long*)(ws + pl.off_mask);
  hipLaunchKernelGGL(sort_kernel, dim3(pl.B), dim3(1024), 0, st, keys, pl.key_stride, cand);
  FDT_LAUNCH_CHECK();
  hipLaunchKernelGGL(gather_kernel, dim3(ceil_div(pl.K, 256), pl.B), dim3(256), 0, st, keys,
                     pl.key_stride, cand, pl.K, pl.Kp, (const float4*)loc, (const float4*)pri,
                     (const float4*)boxes_in, pl.P, v0, v1, facebox, scores, score_stride, score_off,
                     score_img_stride, sbox, sarea, sscore, sidx);
  FDT_LAUNCH_CHECK();
  int nw = pl.Kp / 64;
  const int gcap = nw < 80 ? nw : 80;   // 80 x 80 tiles = 5120 candidates in one pass; more are strided over
  hipLaunchKernelGGL(mask_kernel, dim3(gcap, gcap, pl.B), dim3(64), 0, st, sbox, sarea, cand, pl.K,
                     pl.Kp, overlap, facebox, mask);
  FDT_LAUNCH_CHECK();
  hipLaunchKernelGGL(scan_kernel, dim3(pl.B), dim3(64), (size_t)nw * 8, st, mask, sbox, sscore,
                     sidx, cand, pl.K, pl.Kp, max_keep, skip_single, out, out_img_stride, counts,
                     count_stride, count_off, keep_idx, keep_stride, fb_boxes, fb_probs, pl.P);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

int launch_detect(const DetectPlan& pl, void* ws_v, const float* loc, const float* conf,
                  const float* pri, int num_classes, int top_k, float conf_t, float nms_t, float v0,
                  float v1, float* out, int* counts, hipStream_t st) {
  char* ws = (char*)ws_v;
  int* cand = (int*)(ws + pl.off_cand);
  long long n_out = (long long)pl.B * num_classes * top_k * 5;
  // class 0 (background) plane stays zero: detection.py:63 loops cl from 1.
  for (int cl = 1; cl < num_classes; ++cl) {
    hipLaunchKernelGGL(reset_kernel, dim3(cl == 1 ? 256 : 1), dim3(256), 0, st,
                       cl == 1 ? out : nullptr, cl == 1 ? n_out : 0, cl == 1 ? counts : nullptr,
                       cl == 1 ? pl.B * num_classes : 0, cand, pl.B);
    FDT_LAUNCH_CHECK();
    hipLaunchKernelGGL(compact_kernel, dim3(ceil_div(pl.P, 256), pl.B), dim3(256), 0, st, conf, pl.P,
                       num_classes, cl, (long long)pl.P * num_classes, conf_t, 1,
                       (unsigned long long*)(ws + pl.off_keys), pl.key_stride, cand);
    FDT_LAUNCH_CHECK();
    FDT_TRY(run_sorted_nms(pl, ws, loc, pri, nullptr, v0, v1, 0, nullptr, nullptr, conf, num_classes, cl,
                           (long long)pl.P * num_classes, nms_t, top_k, 1,
                           out + (long long)cl * top_k * 5, (long long)num_classes * top_k * 5,
                           counts, num_classes, cl, nullptr, 0, st));
  }
  if (num_classes < 2) {
    hipLaunchKernelGGL(reset_kernel, dim3(256), dim3(256), 0, st, out, n_out, counts,
                       pl.B * num_classes, cand, pl.B);
    FDT_LAUNCH_CHECK();
  }
  return FDT_OK;
}

int launch_nms(const DetectPlan& pl, void* ws_v, const float* boxes, const float* scores,
               float overlap, long long* keep, int* count, hipStream_t st) {
  char* ws = (char*)ws_v;
  int* cand = (int*)(ws + pl.off_cand);
  hipLaunchKernelGGL(reset_kernel, dim3(64), dim3(256), 0, st, (float*)keep, (long long)pl.P * 2,
                     count, 1, cand, 1);  // int64 zeros == 2x f32 zeros
  FDT_LAUNCH_CHECK();
  hipLaunchKernelGGL(compact_kernel, dim3(ceil_div(pl.P, 256), 1), dim3(256), 0, st, scores, pl.P, 1,
                     0, (long long)pl.P, 0.0f, 0, (unsigned long long*)(ws + pl.off_keys),
                     pl.key_stride, cand);
  FDT_LAUNCH_CHECK();
  return run_sorted_nms(pl, ws, nullptr, nullptr, boxes, 0.f, 0.f, 0, nullptr, nullptr, scores, 1, 0,
                        (long long)pl.P, overlap, pl.K, 0, nullptr, 0, count, 1, 0, keep, pl.P, st);
}

// DataEncoder.decode_np (FACEBOX/encoderl.py:308-325): score > conf_thres, decode, nms_np(thr) with no
// candidate cap and no output cap; boxes/probs come back in keep order.  `pl` must be built with K = P.
int launch_facebox_decode(const DetectPlan& pl, void* ws_v, const float* loc, const float* conf,
                          const float* anchors, float conf_t, float nms_t, float* boxes, float* probs,
                          int* counts, hipStream_t st) {
  char* ws = (char*)ws_v;
  int* cand = (int*)(ws + pl.off_cand);
  hipLaunchKernelGGL(reset_kernel, dim3(1), dim3(256), 0, st, nullptr, 0ll, counts, pl.B, cand, pl.B);
  FDT_LAUNCH_CHECK();
  hipLaunchKernelGGL(compact_kernel, dim3(ceil_div(pl.P, 256), pl.B), dim3(256), 0, st, conf, pl.P, 2, 1,
                     (long long)pl.P * 2, conf_t, 1, (unsigned long long*)(ws + pl.off_keys), pl.key_stride,
                     cand);
  FDT_LAUNCH_CHECK();
  return run_sorted_nms(pl, ws, loc, anchors, nullptr, 0.1f, 0.2f, 1, boxes, probs, conf, 2, 1,
                        (long long)pl.P * 2, nms_t, pl.K, 0, nullptr, 0, counts, 1, 0, nullptr, 0, st);
}

int launch_facebox_anchors(float* out, hipStream_t st) {
  hipLaunchKernelGGL(facebox_anchors_kernel, dim3(ceil_div(21824, 256)), dim3(256), 0, st, out);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

int launch_pairwise_iou(const void* a, int A, const void* b, int B, int dtype, void* out,
                        hipStream_t st, int measure) {
  if (A == 0 || B == 0) return FDT_OK;
  dim3 grid(ceil_div(B, 64), A);
  if (measure == 1) {
    if (dtype == FDT_F64)
      hipLaunchKernelGGL(pairwise_distance_kernel<double>, grid, dim3(64), 0, st, (const double*)a, A,
                         (const double*)b, B, (double*)out);
    else
      hipLaunchKernelGGL(pairwise_distance_kernel<float>, grid, dim3(64), 0, st, (const float*)a, A,
                         (const float*)b, B, (float*)out);
    FDT_LAUNCH_CHECK();
    return FDT_OK;
  }
  if (dtype == FDT_F64)
    hipLaunchKernelGGL(pairwise_iou_kernel<double>, grid, dim3(64), 0, st, (const double*)a, A,
                       (const double*)b, B, (double*)out);
  else
    hipLaunchKernelGGL(pairwise_iou_kernel<float>, grid, dim3(64), 0, st, (const float*)a, A,
                       (const float*)b, B, (float*)out);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

}  // namespace fdt

// ================================================================================== C ABI (host)
using namespace fdt;

extern "C" int fdt_priorbox(int width, int height, int stride, int box, int n_scales,
                            const double* aspect_ratios, int n_ar, int f_w, int f_h, float* out) {
  const hipStream_t st = fdt::thread_stream();   // never the legacy stream (common.h)
  FDT_REQUIRE(width > 0 && height > 0 && f_w >= 0 && f_h >= 0 && n_scales >= 0 && n_ar >= 0 && out,
              FDT_ERR_ARG, "fdt_priorbox: bad argument");
  long long total = (long long)f_w * f_h * n_scales * (1 + n_ar);
  if (total == 0) return FDT_OK;
  FDT_REQUIRE(st, FDT_ERR_HIP, "%s: could not create the calling thread's private stream", __func__);
  DevBuf d_out, d_ar;
  FDT_TRY(d_out.alloc(total * 16));
  if (n_ar) {
    FDT_TRY(d_ar.alloc(n_ar * 8));
    FDT_HIP(copy_sync(d_ar.p, aspect_ratios, n_ar * 8, hipMemcpyHostToDevice, st));
  }
  FDT_TRY(launch_priorbox(width, height, stride, box, n_scales, d_ar.as<double>(), n_ar, f_w, f_h,
                          d_out.as<float>(), st));
  FDT_HIP(copy_sync(out, d_out.p, total * 16, hipMemcpyDeviceToHost, st));
  return FDT_OK;
}

extern "C" int fdt_decode(const float* loc, const float* priors, int P, float var0, float var1,
                          float* boxes) {
  const hipStream_t st = fdt::thread_stream();   // never the legacy stream (common.h)
  FDT_REQUIRE(P >= 0 && loc && priors && boxes, FDT_ERR_ARG, "fdt_decode: bad argument");
  if (P == 0) return FDT_OK;
  FDT_REQUIRE(st, FDT_ERR_HIP, "%s: could not create the calling thread's private stream", __func__);
  DevBuf dl, dp, dbx;
  FDT_TRY(dl.alloc((size_t)P * 16));
  FDT_TRY(dp.alloc((size_t)P * 16));
  FDT_TRY(dbx.alloc((size_t)P * 16));
  FDT_HIP(copy_sync(dl.p, loc, (size_t)P * 16, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(dp.p, priors, (size_t)P * 16, hipMemcpyHostToDevice, st));
  FDT_TRY(launch_decode(dl.as<float>(), dp.as<float>(), P, var0, var1, dbx.as<float>(), st));
  FDT_HIP(copy_sync(boxes, dbx.p, (size_t)P * 16, hipMemcpyDeviceToHost, st));
  return FDT_OK;
}

extern "C" int fdt_nms(const float* boxes, const float* scores, int n, float overlap, int top_k,
                       long long* keep, int* count) {
  const hipStream_t st = fdt::thread_stream();   // never the legacy stream (common.h)
  FDT_REQUIRE(n >= 0 && keep && count && (n == 0 || (boxes && scores)), FDT_ERR_ARG,
              "fdt_nms: bad argument");
  *count = 0;
  if (n == 0) return FDT_OK;  // box_utils.py:290-291
  FDT_REQUIRE(top_k >= 1, FDT_ERR_ARG, "fdt_nms: top_k must be >= 1");
  DetectPlan pl = make_detect_plan(1, n, top_k);
  FDT_REQUIRE(st, FDT_ERR_HIP, "%s: could not create the calling thread's private stream", __func__);
  DevBuf ws, db, ds, dk, dc;
  FDT_TRY(ws.alloc(pl.bytes));
  FDT_TRY(db.alloc((size_t)n * 16));
  FDT_TRY(ds.alloc((size_t)n * 4));
  FDT_TRY(dk.alloc((size_t)n * 8));
  FDT_TRY(dc.alloc(4));
  FDT_HIP(copy_sync(db.p, boxes, (size_t)n * 16, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(ds.p, scores, (size_t)n * 4, hipMemcpyHostToDevice, st));
  FDT_TRY(launch_nms(pl, ws.p, db.as<float>(), ds.as<float>(), overlap, dk.as<long long>(),
                     dc.as<int>(), st));
  FDT_HIP(copy_sync(keep, dk.p, (size_t)n * 8, hipMemcpyDeviceToHost, st));
  FDT_HIP(copy_sync(count, dc.p, 4, hipMemcpyDeviceToHost, st));
  return FDT_OK;
}

extern "C" long long fdt_detect_workspace_bytes(int B, int P, int nms_top_k) {
  if (B < 1 || P < 1 || nms_top_k < 1) return 0;
  return make_detect_plan(B, P, nms_top_k).bytes;
}

static int check_detect_args(int B, int P, int num_classes, int top_k, float nms_thresh,
                             int nms_top_k) {
  FDT_REQUIRE(B >= 1 && P >= 1 && num_classes >= 1 && top_k >= 1 && nms_top_k >= 1, FDT_ERR_ARG,
              "fdt_detect: bad shape argument");
  // detection.py:28-29
  FDT_REQUIRE(nms_thresh > 0.0f, FDT_ERR_ARG, "nms_threshold must be non negative.");
  return FDT_OK;
}

extern "C" int fdt_detect_dev(const float* loc, const float* conf, const float* priors, int B, int P,
                              int num_classes, int top_k, float conf_thresh, float nms_thresh,
                              int nms_top_k, float var0, float var1, float* out, int* counts,
                              void* workspace, long long workspace_bytes, void* stream) {
  FDT_TRY(check_detect_args(B, P, num_classes, top_k, nms_thresh, nms_top_k));
  DetectPlan pl = make_detect_plan(B, P, nms_top_k);
  FDT_REQUIRE(workspace && workspace_bytes >= pl.bytes, FDT_ERR_ARG,
              "fdt_detect_dev: workspace too small (%lld < %lld)", workspace_bytes, pl.bytes);
  const hipStream_t st = stream ? (hipStream_t)stream : fdt::thread_stream();   // NULL = fdt_thread_stream(), fdt.h
  FDT_REQUIRE(st, FDT_ERR_HIP, "fdt_detect_dev: could not create the calling thread's private stream");
  return launch_detect(pl, workspace, loc, conf, priors, num_classes, top_k, conf_thresh,
                       nms_thresh, var0, var1, out, counts, st);
}

extern "C" int fdt_detect(const float* loc, const float* conf, const float* priors, int B, int P,
                          int num_classes, int top_k, float conf_thresh, float nms_thresh,
                          int nms_top_k, float var0, float var1, float* out, int* counts) {
  const hipStream_t st = fdt::thread_stream();   // never the legacy stream (common.h)
  FDT_TRY(check_detect_args(B, P, num_classes, top_k, nms_thresh, nms_top_k));
  FDT_REQUIRE(loc && conf && priors && out, FDT_ERR_ARG, "fdt_detect: null pointer");
  DetectPlan pl = make_detect_plan(B, P, nms_top_k);
  FDT_REQUIRE(st, FDT_ERR_HIP, "%s: could not create the calling thread's private stream", __func__);
  DevBuf ws, dl, dcf, dp, dout, dcnt;
  size_t n_out = (size_t)B * num_classes * top_k * 5;
  FDT_TRY(ws.alloc(pl.bytes));
  FDT_TRY(dl.alloc((size_t)B * P * 16));
  FDT_TRY(dcf.alloc((size_t)B * P * num_classes * 4));
  FDT_TRY(dp.alloc((size_t)P * 16));
  FDT_TRY(dout.alloc(n_out * 4));
  FDT_TRY(dcnt.alloc((size_t)B * num_classes * 4));
  FDT_HIP(copy_sync(dl.p, loc, (size_t)B * P * 16, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(dcf.p, conf, (size_t)B * P * num_classes * 4, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(dp.p, priors, (size_t)P * 16, hipMemcpyHostToDevice, st));
  FDT_TRY(launch_detect(pl, ws.p, dl.as<float>(), dcf.as<float>(), dp.as<float>(), num_classes,
                        top_k, conf_thresh, nms_thresh, var0, var1, dout.as<float>(),
                        dcnt.as<int>(), st));
  FDT_HIP(copy_sync(out, dout.p, n_out * 4, hipMemcpyDeviceToHost, st));
  if (counts)
    FDT_HIP(copy_sync(counts, dcnt.p, (size_t)B * num_classes * 4, hipMemcpyDeviceToHost, st));
  return FDT_OK;
}

static int pairwise_host(const void* a, int A, const void* b, int B, int dtype, void* out, int measure) {
  const hipStream_t st = fdt::thread_stream();   // never the legacy stream (common.h)
  FDT_REQUIRE(A >= 0 && B >= 0 && (dtype == FDT_F32 || dtype == FDT_F64), FDT_ERR_ARG,
              "fdt_pairwise_iou / fdt_pairwise_distance: bad argument");
  if (A == 0 || B == 0) return FDT_OK;
  FDT_REQUIRE(a && b && out, FDT_ERR_ARG, "fdt_pairwise_iou / fdt_pairwise_distance: null pointer");
  size_t es = dtype == FDT_F64 ? 8 : 4;
  FDT_REQUIRE(st, FDT_ERR_HIP, "%s: could not create the calling thread's private stream", __func__);
  DevBuf da, db, dout;
  FDT_TRY(da.alloc((size_t)A * 4 * es));
  FDT_TRY(db.alloc((size_t)B * 4 * es));
  FDT_TRY(dout.alloc((size_t)A * B * es));
  FDT_HIP(copy_sync(da.p, a, (size_t)A * 4 * es, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(db.p, b, (size_t)B * 4 * es, hipMemcpyHostToDevice, st));
  FDT_TRY(launch_pairwise_iou(da.p, A, db.p, B, dtype, dout.p, st, measure));
  FDT_HIP(copy_sync(out, dout.p, (size_t)A * B * es, hipMemcpyDeviceToHost, st));
  return FDT_OK;
}
extern "C" int fdt_pairwise_iou(const void* a, int A, const void* b, int B, int dtype, void* out) {
  return pairwise_host(a, A, b, B, dtype, out, 0);
}
extern "C" int fdt_pairwise_distance(const void* a, int A, const void* b, int B, int dtype, void* out) {
  return pairwise_host(a, A, b, B, dtype, out, 1);
}

extern "C" int fdt_facebox_anchors(float* out) {
  const hipStream_t st = fdt::thread_stream();   // never the legacy stream (common.h)
  FDT_REQUIRE(out, FDT_ERR_ARG, "fdt_facebox_anchors: null pointer");
  FDT_REQUIRE(st, FDT_ERR_HIP, "%s: could not create the calling thread's private stream", __func__);
  DevBuf d;
  FDT_TRY(d.alloc(21824 * 16));
  FDT_TRY(launch_facebox_anchors(d.as<float>(), st));
  FDT_HIP(copy_sync(out, d.p, 21824 * 16, hipMemcpyDeviceToHost, st));
  return FDT_OK;
}

extern "C" int fdt_facebox_decode(const float* loc, const float* conf, const float* anchors, int P,
                                  float conf_thresh, float nms_thresh, float* boxes, float* probs,
                                  int* count) {
  const hipStream_t st = fdt::thread_stream();   // never the legacy stream (common.h)
  FDT_REQUIRE(P >= 1 && loc && conf && anchors && boxes && probs && count, FDT_ERR_ARG,
              "fdt_facebox_decode: bad argument");
  DetectPlan pl = make_detect_plan(1, P, P);
  FDT_REQUIRE(st, FDT_ERR_HIP, "%s: could not create the calling thread's private stream", __func__);
  DevBuf ws, dl, dc, da, db, dp, dn;
  FDT_TRY(ws.alloc(pl.bytes));
  FDT_TRY(dl.alloc((size_t)P * 16));
  FDT_TRY(dc.alloc((size_t)P * 8));
  FDT_TRY(da.alloc((size_t)P * 16));
  FDT_TRY(db.alloc((size_t)P * 16));
  FDT_TRY(dp.alloc((size_t)P * 4));
  FDT_TRY(dn.alloc(4));
  FDT_HIP(copy_sync(dl.p, loc, (size_t)P * 16, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(dc.p, conf, (size_t)P * 8, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(da.p, anchors, (size_t)P * 16, hipMemcpyHostToDevice, st));
  FDT_TRY(launch_facebox_decode(pl, ws.p, dl.as<float>(), dc.as<float>(), da.as<float>(), conf_thresh,
                                nms_thresh, db.as<float>(), dp.as<float>(), dn.as<int>(), st));
  FDT_HIP(copy_sync(count, dn.p, 4, hipMemcpyDeviceToHost, st));
  FDT_HIP(copy_sync(boxes, db.p, (size_t)P * 16, hipMemcpyDeviceToHost, st));
  FDT_HIP(copy_sync(probs, dp.p, (size_t)P * 4, hipMemcpyDeviceToHost, st));
  return FDT_OK;
}
