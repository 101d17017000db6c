// Streaming kernels for the HBM-bound front of the MobileNetV2 detectors (try3 / try4 / try5): layers with so few channels
// that an MFMA tile would be mostly padding and whose cost is the bytes they move.  Both run on the vector ALU, one pass
// over their input, weights as wave-uniform (scalar) operands -- nothing is reshaped into a GEMM.
//
//  * stem3x3s2_u8_kernel: features.0 (reference pyramid_mb2_try3.py:11-24,162: conv_bn(3, 32, stride 2) = Conv2d 3x3 / 2
//    pad 1, no bias, + BatchNorm + ReLU6) on the RAW uint8 HWC BGR frame: (float)u8 - mean happens in registers (the two
//    IEEE operations of the ingest kernel, iouTracke_cal.py:40-46), padding is zero in the converted domain.  Replaces
//    [ingest kernel: 25 MB in, 100 MB out per 8 frames] + [MFMA conv that pads K = 27 to 36 and reads the 100 MB back] by
//    one kernel that reads 25 MB and writes the 268 MB of output.
//  * dw_project_kernel: the t = 1 InvertedResidual (features.1, pyramid_mb2_try3.py:84-94: depthwise 3x3 + BN + ReLU6, 1x1
//    project + BN) as ONE kernel: the depthwise output (as large as the input) is consumed in registers by the 1x1 project
//    instead of being written (268 MB per 8 frames at 512^2) and read back.  Depthwise taps in dwconv3_kernel's (dy, dx)
//    order, the project as an ascending-channel fmaf chain, + bias, + residual: the arithmetic of the two stand-alone
//    kernels (the f32 MFMA is such a chain as well).
#include "common.h"
#include "ops.h"

namespace fdt {
namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

// acc.xy = fma(w.xy, p.<half>, acc.xy): both lanes of the packed FMA take the SAME half of the pixel pair (op_sel / op_sel_hi of
// src1), the weight pair and the accumulator lane by lane.  Written as asm because the compiler materialises the splat with two
// v_mov_b32 per use instead of folding it into the operand selects (272 copies per four output channels in this kernel).
__device__ __forceinline__ void pk_fma_lo(v2f& acc, const v2f w, const v2f p) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(p));
}
__device__ __forceinline__ void pk_fma_hi(v2f& acc, const v2f w, const v2f p) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(p));
}

// ---------------------------------------------------------------------------------------------- dw 3x3 + 1x1 project
// thread = a 4-wide strip of R output rows of one image, all OUP output channels; loop over the hid input channels with the
// next channel's rows in flight.  Stride 1, pad 1 (the blocks this serves; a stride-2 block has its own fused kernel).
template <int OUP, int R>
__global__ __launch_bounds__(256) void dw_project_kernel(const float* __restrict__ in, int hid, int H, int W,
                                                         const float* __restrict__ w9, const float* __restrict__ bdw,
                                                         const float* __restrict__ wpt, const float* __restrict__ bp,
                                                         const float* __restrict__ res, float* __restrict__ out,
                                                         long long total) {
  constexpr int NIR = R + 2;
  const long long idx0 = blockIdx.x * 256ll + threadIdx.x;
  const bool live = idx0 < total;
  const long long idx = live ? idx0 : total - 1;      // every lane stays: neighbours exchange halo words
  const int lane = threadIdx.x & 63;
  const int W4 = W >> 2, RG = (H + R - 1) / R;
  const int c4 = (int)(idx % W4);
  const long long t = idx / W4;
  const int rg = (int)(t % RG);
  const int b = (int)(t / RG);
  const int ox = c4 * 4, oy0 = rg * R;
  const bool left_pad = c4 == 0, right_pad = c4 == W4 - 1;
  const long long HW = (long long)H * W;
  const float* img = in + (long long)b * hid * HW;
  // rows oy0 - 1 .. oy0 + R: offsets and validity are the same for every channel
  long long roff[NIR];
  bool rin[NIR];
#pragma unroll
  for (int r = 0; r < NIR; ++r) {
    const int y = oy0 - 1 + r;
    rin[r] = y >= 0 && y < H;
    roff[r] = (long long)(rin[r] ? y : 0) * W + ox;
  }
  const bool need_l = lane == 0 && !left_pad, need_r = lane == 63 && !right_pad;
  float acc[OUP][R][4];
#pragma unroll
  for (int o = 0; o < OUP; ++o)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[o][r][j] = 0.0f;

  float4 A[2][NIR];
  float EL[2][NIR], ER[2][NIR];
  auto fetch = [&](int c, float4* a, float* el, float* er) {
    const float* src = img + (long long)c * HW;
#pragma unroll
    for (int r = 0; r < NIR; ++r) {
      a[r] = *reinterpret_cast<const float4*>(src + roff[r]);
      el[r] = need_l ? src[roff[r] - 1] : 0.0f;
      er[r] = need_r ? src[roff[r] + 4] : 0.0f;
    }
  };
  auto channel = [&](int c, const float4* a, const float* el, const float* er) {
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = w9[c * 9 + i];
    float d[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) d[r][j] = 0.0f;
#pragma unroll
    for (int r = 0; r < NIR; ++r) {
      float lft = __shfl_up(a[r].w, 1, 64);
      if (need_l) lft = el[r];
      if (left_pad) lft = 0.0f;
      float rgt = __shfl_down(a[r].x, 1, 64);
      if (need_r) rgt = er[r];
      if (right_pad) rgt = 0.0f;
      float v[6] = {lft, a[r].x, a[r].y, a[r].z, a[r].w, rgt};
      if (!rin[r]) {
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = 0.0f;
      }
#pragma unroll
      for (int o = 0; o < R; ++o) {
        const int dy = r - o;
        if (dy < 0 || dy > 2) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int j = 0; j < 4; ++j) d[o][j] = fmaf(v[j + dx], k[dy * 3 + dx], d[o][j]);
      }
    }
    const float bv = bdw[c];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) d[r][j] = fminf(fmaxf(d[r][j] + bv, 0.0f), 6.0f);     // + BN bias, ReLU6 (pyramid_mb2_try3.py:86-88)
    const float* wp = wpt + (long long)c * OUP;
#pragma unroll
    for (int o = 0; o < OUP; ++o) {
      const float w = wp[o];
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[o][r][j] = fmaf(w, d[r][j], acc[o][r][j]);
    }
  };
  fetch(0, A[0], EL[0], ER[0]);
  for (int c = 0; c < hid; c += 2) {                 // hid is even (checked by the launcher)
    fetch(c + 1, A[1], EL[1], ER[1]);
    channel(c, A[0], EL[0], ER[0]);
    if (c + 2 < hid) fetch(c + 2, A[0], EL[0], ER[0]);
    channel(c + 1, A[1], EL[1], ER[1]);
  }
  if (!live) return;
#pragma unroll
  for (int o = 0; o < OUP; ++o) {
    const float bv = bp[o];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int oy = oy0 + r;
      if (oy >= H) break;
      const long long off = ((long long)b * OUP + o) * HW + (long long)oy * W + ox;
      float4 y = make_float4(acc[o][r][0] + bv, acc[o][r][1] + bv, acc[o][r][2] + bv, acc[o][r][3] + bv);
      if (res) {
        const float4 rv = *reinterpret_cast<const float4*>(res + off);
        y.x += rv.x; y.y += rv.y; y.z += rv.z; y.w += rv.w;
      }
      *reinterpret_cast<float4*>(out + off) = y;
    }
  }
}

// ---------------------------------------------------------------------------------------------- 3x3 / 2 stem on uint8 frames
// thread = 4 consecutive output pixels of one output row, all 32 output channels.  Its input is columns 2*ox - 1 .. 2*ox + 7 of
// rows 2*oy - 1 .. 2*oy + 1: per row 24 contiguous bytes (8 pixels x BGR, 8-byte aligned when W % 8 == 0) + the pixel to the
// left, which is the last pixel of the neighbouring lane's 24 bytes.  Weights wt [27][32] ((c, dy, dx) major, BN folded).
template <int COUT>
__global__ __launch_bounds__(256, 2) void stem3x3s2_u8_kernel(const unsigned char* __restrict__ frames, int H, int W, float m0, float m1,
                                                           float m2, const float* __restrict__ wt, const float* __restrict__ bias,
                                                           int act, float* __restrict__ out, int Ho, int Wo, long long total) {
  // The 27 x COUT weights live in LDS and are read as broadcasts (every lane the same address) right where they are used: as
  // wave-uniform scalar operands the compiler keeps all 864 of them live and spills SGPRs into VGPR lanes (12 000 v_readlane
  // in the first build of this kernel: 241 us per batch of eight against 143 for the MFMA conv it replaces).
  __shared__ float sw[27 * COUT];
  for (int i = threadIdx.x; i < 27 * COUT; i += 256) sw[i] = wt[i];
  __syncthreads();
  const long long idx0 = blockIdx.x * 256ll + threadIdx.x;
  const bool live = idx0 < total;
  const long long idx = live ? idx0 : total - 1;
  const int lane = threadIdx.x & 63;
  const int W4 = Wo >> 2;
  const int c4 = (int)(idx % W4);
  const long long t = idx / W4;
  const int oy = (int)(t % Ho);
  const int b = (int)(t / Ho);
  const int ox = c4 * 4, ix0 = ox * 2;
  const bool left_pad = c4 == 0;
  const unsigned char* img = frames + (long long)b * H * W * 3;
  const float mean[3] = {m0, m1, m2};
  // all three rows' bytes first
  uint2 Q[3][3];
  unsigned LW[3];
  bool rin[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int y = 2 * oy - 1 + r;
    rin[r] = y >= 0 && y < H;
    const unsigned char* row = img + ((long long)(rin[r] ? y : 0) * W + ix0) * 3;
#pragma unroll
    for (int q = 0; q < 3; ++q) Q[r][q] = *reinterpret_cast<const uint2*>(row + 8 * q);
    LW[r] = (lane == 0 && !left_pad) ? *reinterpret_cast<const unsigned*>(row - 4) : 0u;    // bytes -4 .. -1: the pixel at -3 .. -1
  }
  // vp[r][c][k] = (column 2k, column 2k + 1) of channel c of row 2*oy - 1 + r, columns counted from 2*ox - 1 (0 .. 8; the
  // tenth is never read), converted: 45 register PAIRS that stay live while the output channels are walked four at a time.
  // A pixel operand of the packed FMAs below is one HALF of such a pair, selected by op_sel -- no copies.
  v2f vp[3][3][5];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    unsigned lw = __shfl_up(Q[r][2].y, 1, 64);
    if (lane == 0) lw = LW[r];
    const unsigned wds[6] = {Q[r][0].x, Q[r][0].y, Q[r][1].x, Q[r][1].y, Q[r][2].x, Q[r][2].y};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v[10];
      const float lv = (float)((lw >> (8 * (c + 1))) & 0xffu) - mean[c];
      v[0] = (left_pad || !rin[r]) ? 0.0f : lv;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int byte = i * 3 + c;
        const float pv = (float)((wds[byte >> 2] >> (8 * (byte & 3))) & 0xffu) - mean[c];
        v[i + 1] = rin[r] ? pv : 0.0f;
      }
      v[9] = 0.0f;
#pragma unroll
      for (int k = 0; k < 5; ++k) vp[r][c][k] = (v2f){v[2 * k], v[2 * k + 1]};
    }
  }
  const long long HWo = (long long)Ho * Wo;
  float* dst = out + (long long)b * COUT * HWo + (long long)oy * Wo + ox;
  // four output channels per trip of a REAL loop: the trip's 27 x 4 weights are read (LDS broadcasts) where they are used, and
  // nothing of the next trip can be hoisted in front of it
#pragma unroll 1
  for (int o4 = 0; o4 < COUT / 4; ++o4) {
    // packed along the OUTPUT CHANNELS: (w[q], w[q+1]) are adjacent in the weight read, the pixel is the same for both halves
    // (v_pk_fma_f32 with the pixel operand's low half selected twice) -- packing along the pixels would need copies of every
    // second column of the 81 converted values
    v2f acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j][0] = acc[j][1] = (v2f){0.0f, 0.0f};
    const float4* wk = reinterpret_cast<const float4*>(sw) + o4;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float4 w4 = wk[((c * 3 + r) * 3 + dx) * (COUT / 4)];
          const v2f w01 = {w4.x, w4.y}, w23 = {w4.z, w4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int e = 2 * j + dx;                        // column of this tap for output pixel j
            if (e & 1) {
              pk_fma_hi(acc[j][0], w01, vp[r][c][e >> 1]);
              pk_fma_hi(acc[j][1], w23, vp[r][c][e >> 1]);
            } else {
              pk_fma_lo(acc[j][0], w01, vp[r][c][e >> 1]);
              pk_fma_lo(acc[j][1], w23, vp[r][c][e >> 1]);
            }
          }
        }
    if (live) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int o = o4 * 4 + q;
        const float bv = bias[o];
        float4 y = make_float4(acc[0][q >> 1][q & 1] + bv, acc[1][q >> 1][q & 1] + bv, acc[2][q >> 1][q & 1] + bv, acc[3][q >> 1][q & 1] + bv);
        if (act == 1) {
          y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
        } else if (act == 2) {
          y.x = fminf(fmaxf(y.x, 0.f), 6.f); y.y = fminf(fmaxf(y.y, 0.f), 6.f);
          y.z = fminf(fmaxf(y.z, 0.f), 6.f); y.w = fminf(fmaxf(y.w, 0.f), 6.f);
        }
        *reinterpret_cast<float4*>(dst + (long long)o * HWo) = y;
      }
    }
  }
}

}  // namespace

bool dw_project_supported(int hid, int H, int W, int oup) {
  return (oup == 16 || oup == 24 || oup == 32) && hid >= 2 && (hid & 1) == 0 && (W & 3) == 0 && H >= 1 &&
         (long long)hid * H * W * 4 < (1ll << 40);
}

int launch_dw_project(const float* in, int B, int hid, int H, int W, const float* w9, const float* bdw, const float* wpt,
                      const float* bp, int oup, const float* res, float* out, hipStream_t st) {
  FDT_REQUIRE(dw_project_supported(hid, H, W, oup) && in && w9 && bdw && wpt && bp && out && B >= 1, FDT_ERR_ARG,
              "launch_dw_project: unsupported shape (hid %d, %dx%d, oup %d)", hid, H, W, oup);
  // rows per thread: two at 16 output channels (128 accumulators); one row per thread measured slower there (122 vs 114 us on
  // features.1 of try3 at batch 8: three input rows read per output row instead of two)
  const int R = oup <= 16 ? 2 : 1;
  const long long total = (long long)B * ceil_div(H, R) * (W / 4);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (oup == 16)
    hipLaunchKernelGGL((dw_project_kernel<16, 2>), dim3(blocks), dim3(256), 0, st, in, hid, H, W, w9, bdw, wpt, bp, res, out, total);
  else if (oup == 24)
    hipLaunchKernelGGL((dw_project_kernel<24, 1>), dim3(blocks), dim3(256), 0, st, in, hid, H, W, w9, bdw, wpt, bp, res, out, total);
  else
    hipLaunchKernelGGL((dw_project_kernel<32, 1>), dim3(blocks), dim3(256), 0, st, in, hid, H, W, w9, bdw, wpt, bp, res, out, total);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

bool stem3x3s2_u8_supported(int H, int W, int Cout) {
  const int Wo = (W - 1) / 2 + 1;
  return Cout == 32 && (W & 7) == 0 && (Wo & 3) == 0 && H >= 2;
}

int launch_stem3x3s2_u8(const unsigned char* frames, int B, int H, int W, const float mean[3], const float* wt, const float* bias,
                        int Cout, int act, float* out, hipStream_t st) {
  FDT_REQUIRE(stem3x3s2_u8_supported(H, W, Cout) && frames && wt && bias && out && B >= 1 && ((uintptr_t)frames & 7) == 0,
              FDT_ERR_ARG, "launch_stem3x3s2_u8: unsupported shape (%dx%d -> %d channels) or misaligned frames", H, W, Cout);
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long long total = (long long)B * Ho * (Wo / 4);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  hipLaunchKernelGGL((stem3x3s2_u8_kernel<32>), dim3(blocks), dim3(256), 0, st, frames, H, W, mean[0], mean[1], mean[2], wt, bias, act,
                     out, Ho, Wo, total);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

}  // namespace fdt
