// conv_stem_u8.h -- the 7x7 stem convolution reading the RAW uint8 HWC BGR frame (round 4; classes CONV_7x7_S2_U8 / _S4_U8).
//
// Reference: iouTracke_cal.py:40-46 turns the frame into float32, subtracts the BGR mean and transposes to NCHW, and
// pyramid.py:229 (Res50: Conv2d(3, 64, 7, stride 2, padding 3)) / FACEBOX/networks.py:89 (Conv2d(3, 24, 7, stride 4, padding 3),
// after My_test_facebox.py:14-15's /255) convolves it.  Rounds 1-3 ran that as two launches: preprocess_kernel wrote a 12.6 MB
// f32 NCHW frame (201 MB for a FaceBoxes batch of 16) that the stem conv read back.  Here the conversion happens in the conv's
// own staging: a workgroup loads the 3 bytes of every pixel of its input patch once, forms (float)u8 - mean (then / scale, the
// same two IEEE operations preprocess_kernel performs, so the staged values are bit-identical) and writes them to LDS as the
// planar f32 patch the MFMA loop reads; padding is 0 in the converted domain, like nn.Conv2d pads the mean-subtracted tensor.
//   * one stage: K = 3 channels x 49 taps, padded to 4 channels (196 = 98 MFMA k-pairs); the whole weight tile of the
//     workgroup (196 x BN floats, LDS-DMA) and the patch are resident, ONE barrier, then 98 fully unrolled k-steps with
//     immediate LDS offsets;
//   * the patch is stored de-interleaved by column phase (x mod stride), so that the 32 pixels a wave reads for one tap are
//     32 consecutive floats: conflict-free ds_read_b32 at stride 2 and 4;
//   * same MFMA order as conv_kernel<G_7x7_S2 / _S4> (channel pair 0 over all taps, then pair 1): bit-identical outputs;
//   * register epilogue: bias (+ folded BN), ReLU, unconditional buffer stores (conv_1x1p.h).
#pragma once
#include "conv_kernel.h"

namespace fdt {
namespace {

template <int S_, int BN_, int WM_, int WN_>
struct StemU8 {
  static constexpr int S = S_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int KS = 7, PAD = 3, TAPS = 49, KC = 4;
  static constexpr int TH = 4, TW = 32, BM = 128;
  static constexpr int MI = BM / (WM * 32), NI = BN / (WN * 32);
  static constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;   // 13 x 69 (stride 2) / 19 x 131 (stride 4) source pixels
  static constexpr int PWQ = (PW + S - 1) / S;                           // columns of one phase
  static constexpr int RP = S * PWQ;                                     // row pitch: S phases of PWQ floats
  static constexpr int PLANE = PH * RP;
  static constexpr int XSZ = KC * PLANE;
  static constexpr int XSZP = (XSZ + 3) / 4 * 4;
  static constexpr int WSZ = KC * TAPS * BN, WSZP = (WSZ + 1023) / 1024 * 1024;   // = tile_weights' stage size for kc = 4
  static constexpr int NW = WSZP / 1024;
  static constexpr size_t LDS_BYTES = (size_t)(XSZP + WSZP) * sizeof(float);
  static_assert(WM * WN == 4 && MI * WM * 32 == BM && NI * WN * 32 == BN && NI == 1, "one 32-cout MFMA tile per wave");
  static_assert(3 * TAPS * BN * 4 < 65536 && (PLANE * 2 + 6 * RP + (S - 1) * PWQ + 2) * 4 < 65536, "ds_read offset fields");
};

template <class P>
__global__ __launch_bounds__(256, 2) void conv_stem_u8_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* X = smem;
  float* Wl = smem + P::XSZP;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / P::WN, wn = wave % P::WN;
  const int half = lane >> 5, l31 = lane & 31;

  const int tiles_x = (a.Wout + P::TW - 1) / P::TW;
  FDT_BLOCK_MAP(a, tile_id, n_tile);
  const int oy0 = (tile_id / tiles_x) * P::TH;
  const int ox0 = (tile_id % tiles_x) * P::TW;
  const int b = blockIdx.z;

  // ---- weights of this channel tile: LDS-DMA, 4 KB per wave-instruction round ------------------------------------------
  {
    const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(a.w + (long long)n_tile * P::WSZP, (long long)P::WSZP * 4);
#pragma unroll
    for (int k = 0; k < P::NW; ++k) bglds16(wrs, Wl + wave * 256 + 1024 * k, (unsigned)tid * 16u, 4096u * k);
  }
  // ---- the patch: u8 HWC BGR -> (float)u8 - mean (/ scale) -> planar, column-phase de-interleaved f32 in LDS --------------
  {
    const unsigned char* src = a.in_u8 + (long long)b * a.Hin * a.Win * 3;
    const int gy0 = oy0 * P::S - P::PAD, gx0 = ox0 * P::S - P::PAD;
    const float m0 = a.u8_mean[0], m1 = a.u8_mean[1], m2 = a.u8_mean[2], sc = a.u8_scale;
    for (int e = tid; e < P::PH * P::PW; e += 256) {
      const int yy = e / P::PW, xx = e - yy * P::PW;
      const int gy = gy0 + yy, gx = gx0 + xx;
      float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
      if (gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) {
        const unsigned char* px = src + ((long long)gy * a.Win + gx) * 3;
        v0 = (float)px[0] - m0;
        v1 = (float)px[1] - m1;
        v2 = (float)px[2] - m2;
        if (sc != 1.0f) { v0 /= sc; v1 /= sc; v2 /= sc; }   // im_tensor.float().div(255): the same division as preprocess_kernel
      }
      float* d = X + yy * P::RP + (xx % P::S) * P::PWQ + xx / P::S;
      d[0] = v0;
      d[P::PLANE] = v1;
      d[2 * P::PLANE] = v2;
      d[3 * P::PLANE] = 0.0f;                               // the padding channel of the second k-pair
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- 98 k-steps: pair p (channels 2p, 2p+1 = the two halves of the wave), tap t ----------------------------------------
  f32x16 acc[P::MI];
#pragma unroll
  for (int i = 0; i < P::MI; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  unsigned xa[P::MI];
#pragma unroll
  for (int i = 0; i < P::MI; ++i) {
    const int py = wm * P::MI + i;                          // TW = 32: one tile row per 32-pixel MFMA tile
    xa[i] = lds0 + (unsigned)(half * P::PLANE + py * P::S * P::RP + l31) * 4u;
  }
  const unsigned wa = lds0 + (unsigned)(P::XSZP + half * P::TAPS * P::BN + wn * 32 + l31) * 4u;
  {
    constexpr int NSTEP = 2 * P::TAPS;
    constexpr int NLD = 1 + P::MI;
    struct Ops {
      float r[NLD];       // [0]: weights, [1, NLD): pixels
    };
    auto load = [&](Ops& o, auto sc_) {
      constexpr int s_ = decltype(sc_)::value;
      constexpr int p_ = s_ / P::TAPS, t_ = s_ % P::TAPS;
      constexpr int ky = t_ / P::KS, kx = t_ % P::KS;
      constexpr int kx_ = 2 * p_ * P::PLANE + ky * P::RP + (kx % P::S) * P::PWQ + kx / P::S;
      constexpr int kw_ = (2 * p_ * P::TAPS + t_) * P::BN;
      lds_read_b32<kw_ * 4>(o.r[0], wa);
#pragma unroll
      for (int i = 0; i < P::MI; ++i) lds_read_b32<kx_ * 4>(o.r[1 + i], xa[i]);
    };
    auto wait_for = [&](Ops& o, auto newer_c) {
      constexpr int N_ = decltype(newer_c)::value;
      if constexpr (NLD == 2)
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(o.r[0]), "+v"(o.r[1]) : "n"(N_));
      else
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(o.r[0]), "+v"(o.r[1]), "+v"(o.r[2]) : "n"(N_));
    };
    static_assert(NLD == 2 || NLD == 3, "operand sets of 2 or 3 registers");
    Ops A, B;
    load(A, std::integral_constant<int, 0>{});
    static_for<0, NSTEP>([&](auto sc_) {
      constexpr int s_ = decltype(sc_)::value;
      Ops& o = (s_ & 1) ? B : A;
      Ops& n = (s_ & 1) ? A : B;
      if constexpr (s_ + 1 < NSTEP) {
        load(n, std::integral_constant<int, s_ + 1>{});
        wait_for(o, std::integral_constant<int, NLD>{});
      } else {
        wait_for(o, std::integral_constant<int, 0>{});
      }
#pragma unroll
      for (int i = 0; i < P::MI; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.r[0], o.r[1 + i], acc[i], 0, 0, 0);
    });
  }

  // ---- epilogue from the accumulator registers (see conv_1x1p.h): bias, activation, unconditional buffer stores ----------
  const int HWo = a.Hout * a.Wout;
  const unsigned hw4 = (unsigned)HWo * 4u;
  const int co_base = n_tile * P::BN + wn * 32;
  const __amdgpu_buffer_rsrc_t ors = buf_rsrc(a.out + ((long long)b * a.out_ctot + a.out_coff) * HWo, (long long)a.Cout * HWo * 4);
  const __amdgpu_buffer_rsrc_t brs = buf_rsrc(a.bias, a.bias ? (long long)a.Cout * 4 : 0);
  const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (unsigned)(co_base + l31) * 4u, 0, 0));
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rr = (r & 3) + 8 * (r >> 2);
    const float bias_r = a.bias ? __shfl(bv, rr + 4 * half, 64) : 0.0f;
#pragma unroll
    for (int i = 0; i < P::MI; ++i) {
      const int gy = oy0 + wm * P::MI + i, gx = ox0 + l31;
      const unsigned voff = (gy < a.Hout && gx < a.Wout) ? (unsigned)(gy * a.Wout + gx) * 4u + (unsigned)(4 * half) * hw4 : kOob;
      float v = acc[i][r] + bias_r;
      if (a.act == ACT_RELU) v = fmaxf(v, 0.0f);
      else if (a.act == ACT_RELU6) v = fminf(fmaxf(v, 0.0f), 6.0f);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ors, voff, (unsigned)(co_base + rr) * hw4, 0);
    }
  }
}

template <class P>
KernelEntry entry_stem() {
  return KernelEntry{conv_stem_u8_kernel<P>, P::LDS_BYTES, 256};
}

//                       stride BN  WM WN
using STEM_S2_N64 = StemU8<2, 64, 2, 2>;     // Res50: 3 -> 64, 7x7 / 2
// (FaceBoxes' 7x7 / 4 stem on the raw frame: conv_stem_s4.h)

}  // namespace
void conv_fill_stem_u8(void* row_s2);
}  // namespace fdt
