// conv_stem_u8b.h -- the 7x7 stems on RAW uint8 frames on the bf16 matrix pipe (classes CONV_7x7_S2_U8B: Res50's conv1,
// pyramid.py:229 Conv2d(3, 64, 7, stride 2, padding 3); CONV_7x7_S4_U8B: FaceBoxes' conv1, FACEBOX/networks.py:89
// Conv2d(3, 24, 7, stride 4, padding 3)).  Round 5.
//
// Why.  The stem's activations are PIXELS: (float)u8 - mean with the integer means of iouTracke_cal.py:40-46 (104, 117, 123), or
// the bytes themselves for FaceBoxes (My_test_facebox.py:14-15 divides by 255 -- a scalar that commutes with the convolution and
// is applied to the accumulator here).  An integer of magnitude <= 255 is EXACT in one bf16 (8 significant bits).  So of the split-
// bf16 scheme of conv_b3.h (f32 operand = three bf16 planes, six plane products) only the weights need their three planes: THREE
// bf16 MFMAs of 32 cycles per 16 k-values replace eight f32 MFMAs of 64 cycles, every product is exact, the accumulation is f32
// like the f32 MFMA's, and no vector-ALU split is left -- the staging is v_cvt_f32_ubyte, a subtraction, v_cvt_pk_bf16_f32.
//   * k layout of conv_stem_b3.h: one (channel, tap row) of a pixel = 8 consecutive patch columns (taps -1 .. 6, the first one's
//     weight zero) = one bf16x8 operand; a k-step is two such rows; 11 k-steps (the 22nd row pair is zero);
//   * the weights of a wave's 32 output channels stay in REGISTERS for the whole persistent workgroup (11 x 3 operands = 132
//     VGPRs, loaded once): the main loop reads only the patch from LDS;
//   * patch in LDS as ONE bf16 plane [c][PH rows][RP columns], double-buffered across tiles: one barrier per tile; the next tile's
//     12-byte groups (4 pixels x BGR) are fetched under the current tile's MFMAs;
//   * a wave owns MI rows of 32 output pixels of one 32-cout group; register epilogue (x 1 / scale, + bias, activation).
// Not bit-identical to the f32 stems (another summation order; for FaceBoxes the 1/255 moves behind the sum); same tolerance
// (tests/test_gpu_model.py, tests/test_gpu_facebox.py: fused-ingest tests against the oracle).  Needs Cin = 3, Win % 4 == 0,
// integer means of magnitude <= 255.
#pragma once
#include "conv_b3.h"

namespace fdt {
namespace {

template <int S_, int CT_, int MI_>
struct StemU8B {
  static constexpr int S = S_, CT = CT_, MI = MI_;                   // stride; 32-cout groups per workgroup; 32-pixel rows per wave
  static constexpr int RG = 4 / CT;                                  // waves per cout group
  static constexpr int TH = RG * MI, TW = 32, BN = 32 * CT;
  static constexpr int PH = (TH - 1) * S + 7;                        // patch rows
  static constexpr int RP = (31 * S + 8 + 3) / 4 * 4;                // patch columns: 31 S + 8 (the first one left of tap 0), whole groups
  static constexpr int G4 = RP / 4;                                  // 4-pixel groups per patch row
  static constexpr int NGRP = PH * G4;
  static constexpr int NIT = (NGRP + 255) / 256;
  static constexpr int X_B = (3 * PH * RP * 2 + 15) / 16 * 16;       // bytes of one patch buffer
  static constexpr int NSTEP = 11;
  static constexpr int NCH = MI == 1 ? 2 : 1;                        // accumulation chains per pixel row (independent MFMAs back to back)
  static constexpr int WSZ = NSTEP * 3 * 64 * 4;                     // floats per 32-cout group in global memory: [step][plane][lane][8 bf16]
  static constexpr size_t LDS_BYTES = (size_t)2 * X_B;
  static_assert(RP % 4 == 0 && 4 % CT == 0, "whole 4-pixel groups; waves divide over the cout groups");
};

typedef unsigned u8b_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u8b_u32x2 __attribute__((ext_vector_type(2)));

template <class P>
__global__ __launch_bounds__(256, 2) void conv_stem_u8b_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* Xb = reinterpret_cast<char*>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int ct = wave % P::CT, rg = wave / P::CT;
  const int tiles_x = (a.Wout + P::TW - 1) / P::TW;
  const int total = a.B * a.n_sp;
  const int T0 = blockIdx.x * a.tiles_per_wg;
  const int T1 = min(T0 + a.tiles_per_wg, total);
  if (T0 >= T1) return;

  // ---- this wave's weights: 33 operands of 16 bytes per lane, resident in registers ------------------------------------------
  bf16x8 A[P::NSTEP][3];
  {
    const u8b_u32x4* wg = reinterpret_cast<const u8b_u32x4*>(a.w + (long long)(blockIdx.y * P::CT + ct) * P::WSZ) + lane;
    static_for<0, P::NSTEP>([&](auto sc_) {
      constexpr int s_ = decltype(sc_)::value;
#pragma unroll
      for (int p = 0; p < 3; ++p) A[s_][p] = __builtin_bit_cast(bf16x8, wg[(s_ * 3 + p) * 64]);
    });
  }
  const int HWo = a.Hout * a.Wout;
  const unsigned hw4 = (unsigned)HWo * 4u;
  const int co_base = (blockIdx.y * P::CT + ct) * 32;
  const __amdgpu_buffer_rsrc_t brs = buf_rsrc(a.bias, a.bias ? (long long)a.Cout * 4 : 0);
  const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (unsigned)(co_base + l31) * 4u, 0, 0));
  float bias_r[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bias_r[r] = a.bias ? __shfl(bv, (r & 3) + 8 * (r >> 2) + 4 * half, 64) : 0.0f;
  const float act_lo = a.act == ACT_NONE ? -__builtin_huge_valf() : 0.0f;
  const float act_hi = a.act == ACT_RELU6 ? 6.0f : __builtin_huge_valf();
  const float inv = 1.0f / a.u8_scale;                               // 1 for Res50; FaceBoxes: the / 255 of the pixels, behind the sum
  const bool scaled = a.u8_scale != 1.0f;
  const float m0 = a.u8_mean[0], m1 = a.u8_mean[1], m2 = a.u8_mean[2];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  // lane l31 = output column: its operand starts at patch column S * l31 of patch row S * (rg * MI + i) + ky
  unsigned xa[P::MI];
#pragma unroll
  for (int i = 0; i < P::MI; ++i) xa[i] = lds0 + (unsigned)(((rg * P::MI + i) * P::S) * P::RP + P::S * l31) * 2u;

  // ---- staging: group g = 256 k + tid = (patch row yy, 4-pixel group p): 12 bytes of the HWC frame -> three 8-byte LDS writes --
  int g_yy[P::NIT], g_p4[P::NIT];
#pragma unroll
  for (int k = 0; k < P::NIT; ++k) {
    const int g = tid + 256 * k;
    const int yy = g / P::G4;
    g_yy[k] = g < P::NGRP ? yy : -(1 << 20);                         // groups past the patch: never inside the image
    g_p4[k] = 4 * (g - yy * P::G4);
  }
  u8b_u32x2 va[P::NIT];
  unsigned vc[P::NIT];
  unsigned okm = 0;
  auto fetch = [&](int T) {
    const int b = T / a.n_sp, sp = T - b * a.n_sp;
    const int gy0 = (sp / tiles_x) * P::TH * P::S - 3, gx0 = (sp % tiles_x) * P::TW * P::S - 4;
    const __amdgpu_buffer_rsrc_t xrs = buf_rsrc(a.in_u8 + (long long)b * a.Hin * a.Win * 3, (long long)a.Hin * a.Win * 3);
    okm = 0;
#pragma unroll
    for (int k = 0; k < P::NIT; ++k) {
      const int gy = gy0 + g_yy[k], gx = gx0 + g_p4[k];
      const bool ok = gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;     // Win % 4 == 0: a group is inside or outside as a whole
      const unsigned vo = ok ? (unsigned)((gy * a.Win + gx) * 3) : kOob;
      okm |= ok ? (1u << k) : 0u;
      va[k] = __builtin_bit_cast(u8b_u32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, vo, 0, 0));
      vc[k] = __builtin_amdgcn_raw_buffer_load_b32(xrs, vo, 8, 0);
    }
  };
  auto byte_f = [](unsigned w, int j) -> float { return (float)((w >> (8 * j)) & 0xffu); };      // v_cvt_f32_ubyte<j>
  auto stage = [&](int buf) {
#pragma unroll
    for (int k = 0; k < P::NIT; ++k) {
      if (g_yy[k] >= 0) {
        const unsigned w0 = va[k][0], w1 = va[k][1], w2 = vc[k];
        const bool ok = (okm >> k) & 1u;
        // bytes: pixel j, channel c at 3 j + c
        f32x2v c0a = {byte_f(w0, 0) - m0, byte_f(w0, 3) - m0}, c0b = {byte_f(w1, 2) - m0, byte_f(w2, 1) - m0};
        f32x2v c1a = {byte_f(w0, 1) - m1, byte_f(w1, 0) - m1}, c1b = {byte_f(w1, 3) - m1, byte_f(w2, 2) - m1};
        f32x2v c2a = {byte_f(w0, 2) - m2, byte_f(w1, 1) - m2}, c2b = {byte_f(w2, 0) - m2, byte_f(w2, 3) - m2};
        uint2 q0 = make_uint2(cvt_pk_bf16(c0a), cvt_pk_bf16(c0b));
        uint2 q1 = make_uint2(cvt_pk_bf16(c1a), cvt_pk_bf16(c1b));
        uint2 q2 = make_uint2(cvt_pk_bf16(c2a), cvt_pk_bf16(c2b));
        if (!ok) q0 = q1 = q2 = make_uint2(0u, 0u);                  // zero padding of the NORMALISED image
        char* d = Xb + buf * P::X_B + (size_t)(g_yy[k] * P::RP + g_p4[k]) * 2;
        *reinterpret_cast<uint2*>(d) = q0;
        *reinterpret_cast<uint2*>(d + P::PH * P::RP * 2) = q1;
        *reinterpret_cast<uint2*>(d + 2 * P::PH * P::RP * 2) = q2;
      }
    }
  };

  fetch(T0);
  stage(0);
  int buf = 0;
  for (int T = T0; T < T1; ++T) {
    const int b = T / a.n_sp, sp = T - b * a.n_sp;
    const int oy0 = (sp / tiles_x) * P::TH, ox0 = (sp % tiles_x) * P::TW;
    __syncthreads();                       // this tile's patch is complete; every wave is done reading the other buffer
    if (T + 1 < T1) fetch(T + 1);          // in flight under the MFMAs

    // ---- 11 k-steps: lanes 0-31 multiply pair q = 2 s, lanes 32-63 pair 2 s + 1 ((channel, tap row) = (q / 7, q % 7)) ----------
    // one row per wave: products with weight plane 0 / with the two small planes in two independent chains; more rows: one each
    f32x16 acc[P::MI][P::NCH];
#pragma unroll
    for (int i = 0; i < P::MI; ++i)
#pragma unroll
      for (int c = 0; c < P::NCH; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.0f;
    struct Ops {
      u8b_u32x2 lo[P::MI], hi[P::MI];
    };
    const unsigned bo = (unsigned)(buf * P::X_B);
    auto load = [&](Ops& o, auto sc_) {
      constexpr int s_ = decltype(sc_)::value;
      constexpr int q0_ = 2 * s_, q1_ = (2 * s_ + 1 < 21) ? 2 * s_ + 1 : 20;
      constexpr unsigned o0_ = (unsigned)(((q0_ / 7) * P::PH + q0_ % 7) * P::RP) * 2u;
      constexpr unsigned o1_ = (unsigned)(((q1_ / 7) * P::PH + q1_ % 7) * P::RP) * 2u;
#pragma unroll
      for (int i = 0; i < P::MI; ++i) {
        const unsigned ad = xa[i] + bo + (half ? o1_ : o0_);
        if constexpr (P::S % 4 == 0) {     // 8-byte aligned: two qwords
          asm volatile("ds_read_b64 %0, %1" : "=v"(o.lo[i]) : "v"(ad));
          asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(o.hi[i]) : "v"(ad));
        } else {                           // 4-byte aligned: four dwords
          asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(o.lo[i]) : "v"(ad));
          asm volatile("ds_read2_b32 %0, %1 offset0:2 offset1:3" : "=v"(o.hi[i]) : "v"(ad));
        }
      }
    };
    auto wait_for = [&](Ops& o, auto newer_c) {
      constexpr int N_ = decltype(newer_c)::value;
#pragma unroll
      for (int i = 0; i < P::MI; ++i) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(o.lo[i]), "+v"(o.hi[i]) : "n"(N_));
    };
    Ops O0, O1;
    load(O0, std::integral_constant<int, 0>{});
    static_for<0, P::NSTEP>([&](auto sc_) {
      constexpr int s_ = decltype(sc_)::value;
      Ops& o = (s_ & 1) ? O1 : O0;
      Ops& n = (s_ & 1) ? O0 : O1;
      if constexpr (s_ + 1 < P::NSTEP) {
        load(n, std::integral_constant<int, s_ + 1>{});
        wait_for(o, std::integral_constant<int, 2 * P::MI>{});
      } else {
        wait_for(o, std::integral_constant<int, 0>{});
      }
      bf16x8 B[P::MI];
#pragma unroll
      for (int i = 0; i < P::MI; ++i)
        B[i] = __builtin_bit_cast(bf16x8, (u8b_u32x4){o.lo[i][0], o.lo[i][1], o.hi[i][0], o.hi[i][1]});
      // the pixels are exact in bf16: three plane products per k-step, the small weight planes in a chain of their own
#pragma unroll
      for (int i = 0; i < P::MI; ++i)
        acc[i][P::NCH - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s_][2], B[i], acc[i][P::NCH - 1], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < P::MI; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s_][0], B[i], acc[i][0], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < P::MI; ++i)
        acc[i][P::NCH - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[s_][1], B[i], acc[i][P::NCH - 1], 0, 0, 0);
    });

    // ---- epilogue from the accumulator registers: (x 1 / scale,) + bias (folded BN), activation, unconditional buffer stores ---
    const __amdgpu_buffer_rsrc_t ors = buf_rsrc(a.out + ((long long)b * a.out_ctot + a.out_coff) * HWo, (long long)a.Cout * HWo * 4);
#pragma unroll
    for (int i = 0; i < P::MI; ++i) {
      const int gy = oy0 + rg * P::MI + i, gx = ox0 + l31;
      const unsigned voff = (gy < a.Hout && gx < a.Wout) ? (unsigned)(gy * a.Wout + gx) * 4u + (unsigned)(4 * half) * hw4 : kOob;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2);
        float v = P::NCH == 2 ? acc[i][0][r] + acc[i][1][r] : acc[i][0][r];
        if (scaled) v *= inv;
        const float o_ = fminf(fmaxf(v + bias_r[r], act_lo), act_hi);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o_), ors, voff, (unsigned)(co_base + rr) * hw4, 0);
      }
    }
    if (T + 1 < T1) stage(buf ^ 1);        // the next tile's patch into the other buffer (waits for its bytes: the compiler's vmcnt)
    buf ^= 1;
  }
}

using StemU8B_S2 = StemU8B<2, 2, 2>;      // Res50: 64 couts = two groups x two row pairs, tile 4 x 32 output pixels
using StemU8B_S4 = StemU8B<4, 1, 1>;      // FaceBoxes: 24 (32) couts, four rows of 32 output pixels

template <class P>
KernelEntry entry_stem_u8b() {
  return KernelEntry{conv_stem_u8b_kernel<P>, P::LDS_BYTES, 256};
}

}  // namespace
void conv_fill_stem_u8b(void* row_s2, void* row_s4);
}  // namespace fdt
