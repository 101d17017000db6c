// HBM-bound helper kernels (see ops.h).  One thread per output element, lanes along the contiguous
// W axis so every wave reads/writes whole 256-byte row segments of the NCHW maps.
#include "common.h"
#include "ops.h"

namespace fdt {
namespace {

__global__ void preprocess_kernel(const unsigned char* __restrict__ in, int H, int W, float m0, float m1,
                                  float m2, float scale, float* __restrict__ out) {
  const int b = blockIdx.y;
  const long long hw = (long long)H * W;
  long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= hw) return;
  const unsigned char* px = in + ((long long)b * hw + p) * 3;
  float* o = out + (long long)b * 3 * hw + p;
  // x.astype(f32) - mean (exact in f32), optional scale
  float v0 = (float)px[0] - m0, v1 = (float)px[1] - m1, v2 = (float)px[2] - m2;
  if (scale != 1.0f) { v0 /= scale; v1 /= scale; v2 /= scale; }   // im_tensor.float().div(255)
  o[0] = v0;
  o[hw] = v1;
  o[2 * hw] = v2;
}

// cv2.resize(src, (W, H)) with INTER_LINEAR on 8UC3, fused with the mean-subtract / NCHW ingest.
// Restates OpenCV's generic 8-bit path (imgproc/resize.cpp: resizeGeneric_ with HResizeLinear and
// VResizeLinear<uchar,int,short>): source coordinate (d + 0.5) * scale - 0.5, edge clamp, coefficients
// quantised to 1/2048 (INTER_RESIZE_COEF_BITS = 11), horizontal pass in int, vertical pass
// ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.  cv2 is absent from the build image, so
// agreement with cv2 itself is UNPINNED; the kernel is bit-exact against oracle/ingest.py.
__device__ __forceinline__ void resize_coef(int d, double scale, int ssize, int& s0, int& s1, int& a0, int& a1) {
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  s0 = s;
  s1 = s + 1 < ssize ? s + 1 : s;
  // saturate_cast<short>(v * 2048): round half to even like cvRound
  a0 = (int)rintf((1.f - f) * 2048.f);
  a1 = (int)rintf(f * 2048.f);
}

__global__ void resize_preprocess_kernel(const unsigned char* __restrict__ in, int SH, int SW, int H, int W,
                                         double scale_y, double scale_x, float m0, float m1, float m2,
                                         float div, float* __restrict__ out, unsigned char* __restrict__ out_u8) {
  const int b = blockIdx.z;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= W) return;
  int sx0, sx1, ax0, ax1, sy0, sy1, by0, by1;
  resize_coef(x, scale_x, SW, sx0, sx1, ax0, ax1);
  resize_coef(y, scale_y, SH, sy0, sy1, by0, by1);
  const unsigned char* src = in + (long long)b * SH * SW * 3;
  const unsigned char* r0 = src + (long long)sy0 * SW * 3;
  const unsigned char* r1 = src + (long long)sy1 * SW * 3;
  const float mean[3] = {m0, m1, m2};
  const long long hw = (long long)H * W;
  float* o = out + (long long)b * 3 * hw + (long long)y * W + x;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int h0 = r0[sx0 * 3 + c] * ax0 + r0[sx1 * 3 + c] * ax1;
    int h1 = r1[sx0 * 3 + c] * ax0 + r1[sx1 * 3 + c] * ax1;
    int v = ((((by0 * (h0 >> 4)) >> 16) + ((by1 * (h1 >> 4)) >> 16) + 2) >> 2);
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    if (out_u8) {      // the resized image itself, HWC like its source: the raw-frame stem conv (conv_stem_u8.h) converts it
      out_u8[((long long)b * hw + (long long)y * W + x) * 3 + c] = (unsigned char)v;
      continue;
    }
    float f = (float)v - mean[c];
    if (div != 1.0f) f /= div;
    o[c * hw] = f;
  }
}

// The same resize, one workgroup per OUTPUT ROW: the two source rows it blends are copied to LDS by 16-byte LDS-DMA (1 KB per wave
// instruction, every piece in flight at once) and the taps are LDS byte reads.  The per-pixel kernel above issues twelve scattered
// byte loads per output pixel (3.6 TB/s on 4K -> 1024^2 sources); here the source is read once, in whole lines.  Same integer
// arithmetic, same bits.  Needs 16-byte aligned rows (SW * 3 % 16 == 0, aligned base) and two padded rows in LDS (<= 64 KB).
__global__ __launch_bounds__(256) void resize_rows_kernel(const unsigned char* __restrict__ in, int SH, int SW, int H, int W,
                                                          double scale_y, double scale_x, float m0, float m1, float m2,
                                                          float div, float* __restrict__ out, unsigned char* __restrict__ out_u8,
                                                          int row_pad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rows[];   // [2][row_pad]
  const int b = blockIdx.y, y = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int sy0, sy1, by0, by1;
  resize_coef(y, scale_y, SH, sy0, sy1, by0, by1);
  const int rb = SW * 3, n16 = rb >> 4;
  const unsigned char* g0 = in + ((long long)b * SH + sy0) * rb;
  const unsigned char* g1 = in + ((long long)b * SH + sy1) * rb;
  for (int c = wave; c * 64 < n16; c += 4) {
    const int piece = c * 64 + lane;
    if (piece < n16) {
      __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)(g0 + piece * 16),
                                       (__attribute__((address_space(3))) void*)(rows + c * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)(g1 + piece * 16),
                                       (__attribute__((address_space(3))) void*)(rows + row_pad + c * 1024), 16, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned char* r0 = rows;
  const unsigned char* r1 = rows + row_pad;
  const float mean[3] = {m0, m1, m2};
  const long long hw = (long long)H * W;
  for (int x = tid; x < W; x += 256) {
    int sx0, sx1, ax0, ax1;
    resize_coef(x, scale_x, SW, sx0, sx1, ax0, ax1);
    float* o = out + (long long)b * 3 * hw + (long long)y * W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int h0 = r0[sx0 * 3 + c] * ax0 + r0[sx1 * 3 + c] * ax1;
      int h1 = r1[sx0 * 3 + c] * ax0 + r1[sx1 * 3 + c] * ax1;
      int v = ((((by0 * (h0 >> 4)) >> 16) + ((by1 * (h1 >> 4)) >> 16) + 2) >> 2);
      v = v < 0 ? 0 : (v > 255 ? 255 : v);
      if (out_u8) {
        out_u8[((long long)b * hw + (long long)y * W + x) * 3 + c] = (unsigned char)v;
        continue;
      }
      float f = (float)v - mean[c];
      if (div != 1.0f) f /= div;
      o[c * hw] = f;
    }
  }
}

__global__ void maxpool3_kernel(const float* __restrict__ in, int C, int H, int W, int stride, int crelu,
                                float* __restrict__ out, int Ho, int Wo) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x;
  const int oy = blockIdx.y;
  const int Cout = crelu ? 2 * C : C;
  const int bc = blockIdx.z;
  const int b = bc / Cout, co = bc % Cout;
  if (ox >= Wo) return;
  const bool neg = crelu && co >= C;
  const int ci = neg ? co - C : co;
  const float* src = in + ((long long)b * C + ci) * H * W;
  float m = -INFINITY;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    int y = oy * stride + dy;
    if (y < 0 || y >= H) continue;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      int x = ox * stride + dx;
      if (x < 0 || x >= W) continue;
      float v = src[(long long)y * W + x];
      if (crelu) v = fmaxf(neg ? -v : v, 0.0f);
      m = fmaxf(m, v);
    }
  }
  out[((long long)b * Cout + co) * Ho * Wo + (long long)oy * Wo + ox] = m;
}

// Vector form of max_pool2d(kernel 3, padding 1, stride S) with the optional CReLU in front of it
// (FACEBOX/networks.py:91-93: cat([x, -x]) -> relu -> max_pool).  Same strip layout as dwconv3_vec_kernel; one
// read of the input plane yields both CReLU planes: max relu(x) = relu(max x), max relu(-x) = relu(-min x).
template <int S>
__global__ __launch_bounds__(256) void maxpool3_vec_kernel(const float* __restrict__ in, int C, int H, int W, int crelu,
                                                           float* __restrict__ out, int Ho, int Wo, long long total) {
  constexpr int R = (S == 1) ? 4 : 2;
  constexpr int NIR = (R - 1) * S + 3;
  constexpr int NV = 3 * S + 3;
  const long long idx0 = blockIdx.x * 256ll + threadIdx.x;
  const bool live = idx0 < total;
  const long long idx = live ? idx0 : total - 1;
  const int lane = threadIdx.x & 63;
  const int W4 = Wo >> 2, RG = (Ho + R - 1) / R;
  const int c4 = (int)(idx % W4);
  const long long t = idx / W4;
  const int rg = (int)(t % RG);
  const int bc = (int)(t / RG);
  const int b = bc / C, ci = bc - b * C;
  const float* src = in + (long long)bc * H * W;
  const int ox = c4 * 4, oy0 = rg * R, ix0 = ox * S;
  const bool left_pad = c4 == 0, right_pad = c4 == W4 - 1;
  float mx[R][4], mn[R][4];
#pragma unroll
  for (int o = 0; o < R; ++o)
#pragma unroll
    for (int j = 0; j < 4; ++j) { mx[o][j] = -INFINITY; mn[o][j] = INFINITY; }
  float4 A[NIR], Bq[NIR];
#pragma unroll
  for (int r = 0; r < NIR; ++r) {
    const int y = oy0 * S - 1 + r;
    const float* row = src + (long long)((y >= 0 && y < H) ? y : 0) * W;
    A[r] = *reinterpret_cast<const float4*>(row + ix0);
    Bq[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (S == 2) Bq[r] = *reinterpret_cast<const float4*>(row + ix0 + 4);
  }
#pragma unroll
  for (int r = 0; r < NIR; ++r) {
    const int y = oy0 * S - 1 + r;
    const bool inside = y >= 0 && y < H;
    const float* row = src + (long long)(inside ? y : 0) * W;
    const float4 a = A[r], bq = Bq[r];
    float lft = __shfl_up(S == 1 ? a.w : bq.w, 1, 64);
    if (lane == 0 && !left_pad) lft = row[ix0 - 1];
    float v[NV];
    bool ok[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) ok[i] = inside;
    v[0] = lft;
    ok[0] = inside && !left_pad;
    v[1] = a.x; v[2] = a.y; v[3] = a.z; v[4] = a.w;
    if (S == 1) {
      float rgt = __shfl_down(a.x, 1, 64);
      if (lane == 63 && !right_pad) rgt = row[ix0 + 4];
      v[5] = rgt;
      ok[5] = inside && !right_pad;
    } else {
      v[5] = bq.x; v[6] = bq.y; v[7] = bq.z; v[NV - 1] = bq.w;
    }
#pragma unroll
    for (int o = 0; o < R; ++o) {
      const int dy = r - o * S;
      if (dy < 0 || dy > 2) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = j * S + dx;
          mx[o][j] = ok[q] ? fmaxf(mx[o][j], v[q]) : mx[o][j];
          mn[o][j] = ok[q] ? fminf(mn[o][j], v[q]) : mn[o][j];
        }
    }
  }
  if (!live) return;
  const int Cout = crelu ? 2 * C : C;
#pragma unroll
  for (int o = 0; o < R; ++o) {
    const int oy = oy0 + o;
    if (oy >= Ho) break;
    const long long off = (long long)oy * Wo + ox;
    float* dst = out + ((long long)b * Cout + ci) * Ho * Wo + off;
    if (crelu) {
      *reinterpret_cast<float4*>(dst) = make_float4(fmaxf(mx[o][0], 0.f), fmaxf(mx[o][1], 0.f), fmaxf(mx[o][2], 0.f),
                                                    fmaxf(mx[o][3], 0.f));
      *reinterpret_cast<float4*>(dst + (long long)C * Ho * Wo) =
          make_float4(fmaxf(-mn[o][0], 0.f), fmaxf(-mn[o][1], 0.f), fmaxf(-mn[o][2], 0.f), fmaxf(-mn[o][3], 0.f));
    } else {
      *reinterpret_cast<float4*>(dst) = make_float4(mx[o][0], mx[o][1], mx[o][2], mx[o][3]);
    }
  }
}

__global__ void dwconv3_kernel(const float* __restrict__ in, const float* __restrict__ w9,
                               const float* __restrict__ bias, int C, int H, int W, int stride, int act,
                               float* __restrict__ out, int Ho, int Wo) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x;
  const int oy = blockIdx.y;
  const int bc = blockIdx.z;
  const int c = bc % C;
  if (ox >= Wo) return;
  const float* src = in + (long long)bc * H * W;
  const float* k = w9 + c * 9;
  float acc = 0.0f;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    int y = oy * stride - 1 + dy;
    if (y < 0 || y >= H) continue;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      int x = ox * stride - 1 + dx;
      if (x < 0 || x >= W) continue;
      acc = fmaf(src[(long long)y * W + x], k[dy * 3 + dx], acc);
    }
  }
  if (bias) acc += bias[c];
  if (act == 1) acc = fmaxf(acc, 0.0f);
  else if (act == 2) acc = fminf(fmaxf(acc, 0.0f), 6.0f);
  out[(long long)bc * Ho * Wo + (long long)oy * Wo + ox] = acc;
}

// Generic depthwise KxK (K in 3/5/7, any stride / padding / dilation): the Mobilenetv1 / Mobilenetv2 blocks of
// pyramid_mobile_try1.py:84-134 (conv1_my is a 7x7/s2/p3 depthwise on the 3 image channels, layer2-4 use 5x5/s2
// and 3x3 dilation-2 depthwise).  One thread per output element, taps in (dy, dx) order like dwconv3_kernel.
__global__ void dwconv_generic_kernel(const float* __restrict__ in, const float* __restrict__ wk,
                                      const float* __restrict__ bias, int C, int H, int W, int K, int stride, int pad,
                                      int dil, int act, float* __restrict__ out, int Ho, int Wo, long long total) {
  const long long idx = blockIdx.x * 256ll + threadIdx.x;
  if (idx >= total) return;
  const int ox = (int)(idx % Wo);
  const long long t = idx / Wo;
  const int oy = (int)(t % Ho);
  const int bc = (int)(t / Ho);
  const int c = bc % C;
  const float* src = in + (long long)bc * H * W;
  const float* k = wk + (long long)c * K * K;
  float acc = 0.0f;
  for (int dy = 0; dy < K; ++dy) {
    const int y = oy * stride - pad + dy * dil;
    if (y < 0 || y >= H) continue;
    for (int dx = 0; dx < K; ++dx) {
      const int x = ox * stride - pad + dx * dil;
      if (x < 0 || x >= W) continue;
      acc = fmaf(src[(long long)y * W + x], k[dy * K + dx], acc);
    }
  }
  if (bias) acc += bias[c];
  if (act == 1) acc = fmaxf(acc, 0.0f);
  else if (act == 2) acc = fminf(fmaxf(acc, 0.0f), 6.0f);
  out[idx] = acc;
}

// Vector form of the depthwise 3x3 (pyramid_mb2_try3.py:96,113: groups == channels): each thread owns a
// 4-wide strip of R output rows, walks the (R-1)*S+3 input rows it needs once (16-byte loads plus the one
// or two halo words) and stores 16 bytes per row.  HBM-bound: in + out bytes, nothing else.  The taps are
// accumulated in the same (dy, dx) order as dwconv3_kernel, so both forms give identical bits.
template <int S>
__global__ __launch_bounds__(256) void dwconv3_vec_kernel(const float* __restrict__ in, const float* __restrict__ w9,
                                                          const float* __restrict__ bias, int C, int H, int W, int act,
                                                          float* __restrict__ out, int Ho, int Wo, long long total) {
  constexpr int R = (S == 1) ? 4 : 2;
  constexpr int NIR = (R - 1) * S + 3;        // input rows per thread
  constexpr int NV = 3 * S + 3;               // input columns per thread: S = 1: x-1..x+4, S = 2: 2x-1..2x+7
  const long long idx0 = blockIdx.x * 256ll + threadIdx.x;
  const bool live = idx0 < total;
  const long long idx = live ? idx0 : total - 1;      // every lane stays: neighbours exchange halo words below
  const int lane = threadIdx.x & 63;
  const int W4 = Wo >> 2, RG = (Ho + R - 1) / R;
  const int c4 = (int)(idx % W4);
  const long long t = idx / W4;
  const int rg = (int)(t % RG);
  const int bc = (int)(t / RG);
  const int c = bc % C;
  const float* src = in + (long long)bc * H * W;
  float k[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) k[i] = w9[c * 9 + i];
  const int ox = c4 * 4, oy0 = rg * R, ix0 = ox * S;
  // The halo words sit in the neighbouring lanes' 16-byte loads (consecutive lanes are consecutive strips of
  // one row; where the row wraps inside a wave the halo is the zero padding anyway).  Only a wave's first /
  // last lane in the middle of a row has to fetch its own.
  const bool left_pad = c4 == 0, right_pad = c4 == W4 - 1;
  float acc[R][4];
#pragma unroll
  for (int o = 0; o < R; ++o)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[o][j] = 0.0f;
  // all row loads first (independent, all in flight together), halo exchange and arithmetic afterwards
  float4 A[NIR], Bq[NIR];
#pragma unroll
  for (int r = 0; r < NIR; ++r) {
    const int y = oy0 * S - 1 + r;
    const float* row = src + (long long)((y >= 0 && y < H) ? y : 0) * W;
    A[r] = *reinterpret_cast<const float4*>(row + ix0);
    Bq[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (S == 2) Bq[r] = *reinterpret_cast<const float4*>(row + ix0 + 4);
  }
#pragma unroll
  for (int r = 0; r < NIR; ++r) {
    const int y = oy0 * S - 1 + r;
    const bool inside = y >= 0 && y < H;
    const float* row = src + (long long)(inside ? y : 0) * W;
    const float4 a = A[r], b = Bq[r];
    float lft = __shfl_up(S == 1 ? a.w : b.w, 1, 64);
    if (lane == 0 && !left_pad) lft = row[ix0 - 1];
    if (left_pad) lft = 0.0f;
    float v[NV];
    v[0] = lft;
    v[1] = a.x; v[2] = a.y; v[3] = a.z; v[4] = a.w;
    if (S == 1) {
      float rgt = __shfl_down(a.x, 1, 64);
      if (lane == 63 && !right_pad) rgt = row[ix0 + 4];
      if (right_pad) rgt = 0.0f;
      v[5] = rgt;
    } else {
      v[5] = b.x; v[6] = b.y; v[7] = b.z; v[NV - 1] = b.w;
    }
    if (!inside) {
#pragma unroll
      for (int i = 0; i < NV; ++i) v[i] = 0.0f;
    }
#pragma unroll
    for (int o = 0; o < R; ++o) {
      const int dy = r - o * S;
      if (dy < 0 || dy > 2) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[o][j] = fmaf(v[j * S + dx], k[dy * 3 + dx], acc[o][j]);
    }
  }
  if (!live) return;
  const float bv = bias ? bias[c] : 0.0f;
#pragma unroll
  for (int o = 0; o < R; ++o) {
    const int oy = oy0 + o;
    if (oy >= Ho) break;
    float4 y4;
    float* yp = &y4.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = acc[o][j];
      if (bias) a += bv;
      if (act == 1) a = fmaxf(a, 0.0f);
      else if (act == 2) a = fminf(fmaxf(a, 0.0f), 6.0f);
      yp[j] = a;
    }
    *reinterpret_cast<float4*>(out + (long long)bc * Ho * Wo + (long long)oy * Wo + ox) = y4;
  }
}

// zero border of one pixel: nn.Conv2d(c, c, kernel_size=1, padding=1) == 1x1 conv of the padded map
// (pyramid_mb2_try4.py:190-191, pyramid_mb2_try5.py:191)
__global__ void pad1_kernel(const float* __restrict__ in, int H, int W, float* __restrict__ out, long long total) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= total) return;
  const int Wo = W + 2, Ho = H + 2;
  const int x = (int)(i % Wo);
  const long long t = i / Wo;
  const int y = (int)(t % Ho);
  const long long bc = t / Ho;
  const bool inside = x >= 1 && x <= W && y >= 1 && y <= H;
  out[i] = inside ? in[(bc * H + (y - 1)) * W + (x - 1)] : 0.0f;
}

__device__ __forceinline__ void softmax2(float a, float b, float& pa, float& pb) {
  float m = fmaxf(a, b);
  float ea = expf(a - m), eb = expf(b - m);
  float s = ea + eb;
  pa = ea / s;
  pb = eb / s;
}

__global__ void head_finalize_kernel(const float* __restrict__ head, int HW, int level0, int P, int p_off,
                                     float* __restrict__ loc, float* __restrict__ conf,
                                     float* __restrict__ logits) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const float* h = head + (long long)b * 8 * HW + p;
  float4 l = make_float4(h[0], h[(long long)HW], h[2ll * HW], h[3ll * HW]);
  float c0 = h[4ll * HW], c1 = h[5ll * HW], c2 = h[6ll * HW], c3 = h[7ll * HW];
  float neg, pos;
  if (level0) {   // (a,b,c,pos): neg = max(a,b,c)          pyramid.py:292-298
    neg = fmaxf(fmaxf(c0, c1), c2);
    pos = c3;
  } else {        // (neg,a,b,c): pos = max(a,b,c)          pyramid.py:299-305
    neg = c0;
    pos = fmaxf(fmaxf(c1, c2), c3);
  }
  const long long row = (long long)b * P + p_off + p;
  reinterpret_cast<float4*>(loc)[row] = l;
  float pn, pp;
  softmax2(neg, pos, pn, pp);
  reinterpret_cast<float2*>(conf)[row] = make_float2(pn, pp);
  if (logits) reinterpret_cast<float2*>(logits)[row] = make_float2(neg, pos);
}

__global__ void head_finalize_all_kernel(const HeadFinArgs a) {
  const int b = blockIdx.y;
  int l = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i)
    if (i < a.nlev && (int)blockIdx.x >= a.lv[i].blk0) l = i;
  const HeadLevel& L = a.lv[l];
  const int p = ((int)blockIdx.x - L.blk0) * blockDim.x + threadIdx.x;
  if (p >= L.HW) return;
  float v[8];
  if (L.ksplit > 1) {
    const float* s = L.src + (long long)b * L.ksplit * 8 * L.HW + p;
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = 0.0f;
    for (int k = 0; k < L.ksplit; ++k)
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] += s[((long long)k * 8 + c) * L.HW];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] += L.bias ? L.bias[c] : 0.0f;
  } else {
    const float* h = L.src + (long long)b * 8 * L.HW + p;
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = h[(long long)c * L.HW];
  }
  float neg, pos;
  if (L.level0) {   // (a,b,c,pos): neg = max(a,b,c)          pyramid.py:292-298
    neg = fmaxf(fmaxf(v[4], v[5]), v[6]);
    pos = v[7];
  } else {          // (neg,a,b,c): pos = max(a,b,c)          pyramid.py:299-305
    neg = v[4];
    pos = fmaxf(fmaxf(v[5], v[6]), v[7]);
  }
  const long long row = (long long)b * a.P + L.p_off + p;
  reinterpret_cast<float4*>(a.loc)[row] = make_float4(v[0], v[1], v[2], v[3]);
  float pn, pp;
  softmax2(neg, pos, pn, pp);
  reinterpret_cast<float2*>(a.conf)[row] = make_float2(pn, pp);
  if (a.logits) reinterpret_cast<float2*>(a.logits)[row] = make_float2(neg, pos);
}

// FaceBoxes: all multibox levels in one launch (FACEBOX/multibox_layer.py:34-48); thread = cell * A + anchor of its level
__global__ void multibox_finalize_all_kernel(const HeadFinArgs a) {
  const int b = blockIdx.y;
  int l = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i)
    if (i < a.nlev && (int)blockIdx.x >= a.lv[i].blk0) l = i;
  const HeadLevel& L = a.lv[l];
  const int A = L.anchors, C = A * 6;
  const int t = ((int)blockIdx.x - L.blk0) * blockDim.x + threadIdx.x;
  if (t >= L.HW * A) return;
  const int cell = t / A, an = t - cell * A;
  const int ch[6] = {an * 4, an * 4 + 1, an * 4 + 2, an * 4 + 3, A * 4 + an * 2, A * 4 + an * 2 + 1};
  float v[6];
  if (L.ksplit > 1) {
    const float* s = L.src + (long long)b * L.ksplit * C * L.HW + cell;
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = 0.0f;
    for (int k = 0; k < L.ksplit; ++k)
#pragma unroll
      for (int c = 0; c < 6; ++c) v[c] += s[((long long)k * C + ch[c]) * L.HW];
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] += L.bias ? L.bias[ch[c]] : 0.0f;
  } else {
    const float* h = L.src + (long long)b * C * L.HW + cell;
#pragma unroll
    for (int c = 0; c < 6; ++c) v[c] = h[(long long)ch[c] * L.HW];
  }
  float pn, pp;
  softmax2(v[4], v[5], pn, pp);
  const long long row = (long long)b * a.P + L.p_off + t;
  reinterpret_cast<float4*>(a.loc)[row] = make_float4(v[0], v[1], v[2], v[3]);
  reinterpret_cast<float2*>(a.conf)[row] = make_float2(pn, pp);
  if (a.logits) reinterpret_cast<float2*>(a.logits)[row] = make_float2(v[4], v[5]);
}

__global__ void multibox_finalize_kernel(const float* __restrict__ locmap, const float* __restrict__ confmap,
                                         long long img_stride, int A, int HW, int P, int p_off,
                                         float* __restrict__ loc, float* __restrict__ conf,
                                         float* __restrict__ logits) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;   // cell * A + anchor
  if (t >= HW * A) return;
  const int cell = t / A, a = t % A;
  const float* lm = locmap + (long long)b * img_stride + cell;
  const float* cm = confmap + (long long)b * img_stride + cell;
  float4 l = make_float4(lm[(long long)(a * 4 + 0) * HW], lm[(long long)(a * 4 + 1) * HW],
                         lm[(long long)(a * 4 + 2) * HW], lm[(long long)(a * 4 + 3) * HW]);
  float pn, pp;
  const float c0 = cm[(long long)(a * 2 + 0) * HW], c1 = cm[(long long)(a * 2 + 1) * HW];
  softmax2(c0, c1, pn, pp);
  const long long row = (long long)b * P + p_off + t;
  reinterpret_cast<float4*>(loc)[row] = l;
  reinterpret_cast<float2*>(conf)[row] = make_float2(pn, pp);
  if (logits) reinterpret_cast<float2*>(logits)[row] = make_float2(c0, c1);
}

}  // namespace

int launch_preprocess(const unsigned char* frames, int B, int H, int W, float m0, float m1, float m2,
                      float scale, float* out, hipStream_t st) {
  dim3 grid((unsigned)ceil_div_ll((long long)H * W, 256), B);
  hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), 0, st, frames, H, W, m0, m1, m2, scale, out);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

// workgroup-per-row form when the source rows are 16-byte aligned and two of them fit in LDS
static bool resize_rows_ok(const unsigned char* frames, int SW, int H, int B, int& row_pad) {
  const int rb = SW * 3;
  row_pad = (rb / 16 + 63) / 64 * 1024;
  return (rb & 15) == 0 && ((uintptr_t)frames & 15) == 0 && 2 * row_pad <= 65536 && H <= 0x7fffffff && B <= 65535;
}

int launch_resize_preprocess(const unsigned char* frames, int B, int SH, int SW, int H, int W, float m0,
                             float m1, float m2, float div, float* out, hipStream_t st) {
  FDT_REQUIRE(H <= 65535 && B <= 65535, FDT_ERR_ARG, "resize: grid too large");
  int row_pad;
  if (resize_rows_ok(frames, SW, H, B, row_pad)) {
    hipLaunchKernelGGL(resize_rows_kernel, dim3(H, B), dim3(256), 2 * row_pad, st, frames, SH, SW, H, W, (double)SH / H,
                       (double)SW / W, m0, m1, m2, div, out, (unsigned char*)nullptr, row_pad);
    FDT_LAUNCH_CHECK();
    return FDT_OK;
  }
  dim3 grid(ceil_div(W, 64), H, B);
  hipLaunchKernelGGL(resize_preprocess_kernel, grid, dim3(64), 0, st, frames, SH, SW, H, W, (double)SH / H,
                     (double)SW / W, m0, m1, m2, div, out, (unsigned char*)nullptr);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

// cv2.resize(src, (W, H)) alone: the resized uint8 HWC image (what the reference's detect_face() has before its float conversion)
int launch_resize_u8(const unsigned char* frames, int B, int SH, int SW, int H, int W, unsigned char* out, hipStream_t st) {
  FDT_REQUIRE(H <= 65535 && B <= 65535, FDT_ERR_ARG, "resize: grid too large");
  int row_pad;
  if (resize_rows_ok(frames, SW, H, B, row_pad)) {
    hipLaunchKernelGGL(resize_rows_kernel, dim3(H, B), dim3(256), 2 * row_pad, st, frames, SH, SW, H, W, (double)SH / H,
                       (double)SW / W, 0.f, 0.f, 0.f, 1.0f, (float*)nullptr, out, row_pad);
    FDT_LAUNCH_CHECK();
    return FDT_OK;
  }
  dim3 grid(ceil_div(W, 64), H, B);
  hipLaunchKernelGGL(resize_preprocess_kernel, grid, dim3(64), 0, st, frames, SH, SW, H, W, (double)SH / H,
                     (double)SW / W, 0.f, 0.f, 0.f, 1.0f, (float*)nullptr, out);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

int launch_maxpool3(const float* in, int B, int C, int H, int W, int stride, int crelu, float* out,
                    int Ho, int Wo, hipStream_t st) {
  if (Wo % 4 == 0 && W == Wo * stride && (stride == 1 || stride == 2)) {
    const int R = stride == 1 ? 4 : 2;
    const long long total = (long long)B * C * ceil_div(Ho, R) * (Wo / 4);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (stride == 1)
      hipLaunchKernelGGL(maxpool3_vec_kernel<1>, dim3(blocks), dim3(256), 0, st, in, C, H, W, crelu, out, Ho, Wo, total);
    else
      hipLaunchKernelGGL(maxpool3_vec_kernel<2>, dim3(blocks), dim3(256), 0, st, in, C, H, W, crelu, out, Ho, Wo, total);
    FDT_LAUNCH_CHECK();
    return FDT_OK;
  }
  const int Cout = crelu ? 2 * C : C;
  FDT_REQUIRE((long long)B * Cout <= 65535 && Ho <= 65535, FDT_ERR_ARG, "maxpool: grid too large");
  dim3 grid(ceil_div(Wo, 64), Ho, B * Cout);
  hipLaunchKernelGGL(maxpool3_kernel, grid, dim3(64), 0, st, in, C, H, W, stride, crelu, out, Ho, Wo);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

int launch_dwconv3(const float* in, const float* w9, const float* bias, int B, int C, int H, int W,
                   int stride, int act, float* out, int Ho, int Wo, hipStream_t st) {
  if (Wo % 4 == 0 && W == Wo * stride && (stride == 1 || stride == 2)) {
    const int R = stride == 1 ? 4 : 2;
    const long long total = (long long)B * C * ceil_div(Ho, R) * (Wo / 4);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (stride == 1)
      hipLaunchKernelGGL(dwconv3_vec_kernel<1>, dim3(blocks), dim3(256), 0, st, in, w9, bias, C, H, W, act, out, Ho,
                         Wo, total);
    else
      hipLaunchKernelGGL(dwconv3_vec_kernel<2>, dim3(blocks), dim3(256), 0, st, in, w9, bias, C, H, W, act, out, Ho,
                         Wo, total);
    FDT_LAUNCH_CHECK();
    return FDT_OK;
  }
  FDT_REQUIRE((long long)B * C <= 65535 && Ho <= 65535, FDT_ERR_ARG, "dwconv: grid too large");
  dim3 grid(ceil_div(Wo, 64), Ho, B * C);
  hipLaunchKernelGGL(dwconv3_kernel, grid, dim3(64), 0, st, in, w9, bias, C, H, W, stride, act, out, Ho, Wo);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

int launch_dwconv(const float* in, const float* wk, const float* bias, int B, int C, int H, int W, int K,
                  int stride, int pad, int dil, int act, float* out, int Ho, int Wo, hipStream_t st) {
  if (K == 3 && pad == 1 && dil == 1) return launch_dwconv3(in, wk, bias, B, C, H, W, stride, act, out, Ho, Wo, st);
  const long long total = (long long)B * C * Ho * Wo;
  hipLaunchKernelGGL(dwconv_generic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, wk, bias, C,
                     H, W, K, stride, pad, dil, act, out, Ho, Wo, total);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

int launch_pad1(const float* in, int BC, int H, int W, float* out, hipStream_t st) {
  const long long total = (long long)BC * (H + 2) * (W + 2);
  hipLaunchKernelGGL(pad1_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in, H, W, out, total);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

int launch_head_finalize_all(const HeadFinArgs& a, int B, hipStream_t st) {
  FDT_REQUIRE(a.nlev >= 1 && a.nlev <= 8 && a.nblocks >= 1 && B >= 1 && B <= 65535, FDT_ERR_ARG, "launch_head_finalize_all: bad table");
  if (a.lv[0].anchors > 0)
    hipLaunchKernelGGL(multibox_finalize_all_kernel, dim3(a.nblocks, B), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(head_finalize_all_kernel, dim3(a.nblocks, B), dim3(256), 0, st, a);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

int launch_head_finalize(const float* head, int B, int H, int W, int level0, int P, int p_off, float* loc,
                         float* conf, float* logits, hipStream_t st) {
  dim3 grid(ceil_div(H * W, 256), B);
  hipLaunchKernelGGL(head_finalize_kernel, grid, dim3(256), 0, st, head, H * W, level0, P, p_off, loc, conf,
                     logits);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

int launch_multibox_finalize(const float* locmap, const float* confmap, long long img_stride, int B, int A,
                             int H, int W, int P, int p_off, float* loc, float* conf, float* logits,
                             hipStream_t st) {
  dim3 grid(ceil_div(H * W * A, 256), B);
  hipLaunchKernelGGL(multibox_finalize_kernel, grid, dim3(256), 0, st, locmap, confmap, img_stride, A, H * W,
                     P, p_off, loc, conf, logits);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

}  // namespace fdt
