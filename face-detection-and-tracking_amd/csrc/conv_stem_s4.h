// conv_stem_s4.h -- the 7x7 / stride-4 stem of FaceBoxes (FACEBOX/networks.py:89: Conv2d(3, 24, 7, stride 4, padding 3)) as a
// kernel of its own (round 4; classes CONV_7x7_S4_K168: f32 NCHW input, and CONV_7x7_S4_U8: the raw resized uint8 image).
//
// The generic direct kernel runs this layer as K = 4 channels x 49 taps (a zero fourth channel: 98 MFMA k-pairs) with 55 KB of
// LDS per workgroup -- two workgroups per CU that load together and compute together: 181 us per batch of 16 at 46 % of the
// matrix peak, most of the rest being the load phase nobody overlaps.  Here:
//   * K = 3 channels x 7 rows x 8 columns (tap columns -1 .. 6, the first one's weights are zero): 84 k-pairs instead of 98.
//     The two k-halves of an MFMA are the columns 2j - 1 and 2j of one (channel, row) -- in the column-phase de-interleaved
//     patch (conv_stem_u8.h; phase = (column + 4) mod 4 counted from the tile's 16-byte aligned left edge ox0 * 4 - 4) those
//     are a CONSTANT 33 floats apart, so both halves use the same immediate offsets;
//   * 51.6 KB of LDS per workgroup (patch 19 x 132 x 3 floats + 21.5 KB of weights): THREE workgroups per CU, one's staging
//     under the other two's MFMAs;
//   * the f32 patch is fetched as 16-byte pieces (a piece = the four phases of one column group; 1881 pieces per workgroup =
//     eight buffer_load_dwordx4 per thread, all in flight together, out-of-image pieces = out-of-range = zeros) and written
//     to its four phase rows with ds_write_b32.  A first version staged it by LDS-DMA, one dword per lane with a stride-4
//     gather: 118 DMA instructions per workgroup, a load phase of ~15 000 cycles per workgroup and the matrix pipe 51 % busy
//     (162 us per batch of 16; profiles/r04/stem_s4/).  Widths that are not a multiple of 4 keep that path;
//   * PERSISTENT workgroups (768 for 256 CUs): the weights are staged once, and the next tile's pieces are fetched into
//     registers before the 84 MFMAs of the current one and written to LDS after them -- the load latency that left the matrix
//     pipe 59 % busy in the one-tile-per-workgroup form is under the MFMAs;
//   * register epilogue as in conv_stem_u8.h.
// Both input forms run the same MFMA sequence: their outputs are bit-identical to each other (tests/test_gpu_facebox.py); against
// the generic kernel the sums are re-associated (f32 rounding only).
#pragma once
#include "conv_kernel.h"

namespace fdt {
namespace {

typedef float stem_f32x4 __attribute__((ext_vector_type(4)));

struct StemS4 {
  static constexpr int S = 4, BN = 32, KS = 7, PAD = 3, KXP = 4;     // KXP: column pairs per tap row (8 columns, the first one zero)
  static constexpr int TH = 4, TW = 32;
  static constexpr int PH = (TH - 1) * S + KS;                       // 19 patch rows
  static constexpr int PWQ = TW + 1;                                 // 33 columns per phase: 4 x 33 = 132 >= 31 x 4 + 8
  static constexpr int RP = S * PWQ;                                 // 132 floats per row
  static constexpr int PLANE = PH * RP;                              // 2508
  static constexpr int XSZ = 3 * PLANE;                              // 7524 floats (a multiple of 4)
  static constexpr int WROW = 8 * BN;                                // one (channel, tap row): 8 columns x 32 couts
  static constexpr int WSZ = 3 * KS * WROW;                          // 5376 floats = 21 rounds of 1 KB: 5 whole 4 KB rounds + 1 KB
  static constexpr int NSTEP = 3 * KS * KXP;                         // 84
  static constexpr size_t LDS_BYTES = (size_t)(XSZ + WSZ) * sizeof(float);
  static_assert(XSZ % 4 == 0 && WSZ % 256 == 0 && WSZ / 1024 == 5 && WSZ % 1024 == 256, "weight rounds");
  static_assert((2 * PLANE + 6 * RP + 3 * PWQ + 1) * 4 < 65536 && WSZ * 4 < 65536, "ds_read offset fields");
};

template <bool U8>
__global__ __launch_bounds__(256, 3) void conv_stem_s4_kernel(const ConvArgs a) {
  using P = StemS4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* X = smem;
  float* Wl = smem + P::XSZ;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles_x = (a.Wout + P::TW - 1) / P::TW;
  const int n_tile = blockIdx.y;                                     // channel tile: its weights stay in LDS for all of the workgroup's tiles
  // persistent: the workgroup walks the (image, spatial tile) pairs T0 .. T1 - 1 (launch_conv sizes the grid for three
  // workgroups per CU); consecutive tiles run along a tile row, so a row's halo is still in L2 when the next row's tile reads it
  const int total = a.B * a.n_sp;
  const int T0 = blockIdx.x * a.tiles_per_wg;
  const int T1 = min(T0 + a.tiles_per_wg, total);

  // ---- weights of this channel tile: 5376 floats, LDS-DMA (five 4 KB rounds of the workgroup + 1 KB of wave 0) -------------
  {
    const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(a.w + (long long)n_tile * P::WSZ, (long long)P::WSZ * 4);
#pragma unroll
    for (int k = 0; k < 5; ++k) bglds16(wrs, Wl + wave * 256 + 1024 * k, (unsigned)tid * 16u, 4096u * k);
    if (wave == 0) bglds16(wrs, Wl + 5120, (unsigned)lane * 16u, 4096u * 5);
  }
  const int HWo = a.Hout * a.Wout;
  const unsigned hw4 = (unsigned)HWo * 4u;
  const int co_base = n_tile * P::BN;
  const __amdgpu_buffer_rsrc_t brs = buf_rsrc(a.bias, a.bias ? (long long)a.Cout * 4 : 0);
  const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (unsigned)(co_base + l31) * 4u, 0, 0));
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  // wave = tile row; lane l31 = output column; the second column of a pair is one phase (33 floats) further
  const unsigned xa = lds0 + (unsigned)(wave * P::S * P::RP + half * P::PWQ + l31) * 4u;
  const unsigned wa = lds0 + (unsigned)(P::XSZ + half * P::BN + l31) * 4u;

  // f32 input, Win % 4 == 0: the patch of the NEXT tile is fetched into registers (16-byte pieces) while this tile's MFMAs run.
  // piece i = (row r = i / 33 of the 57 (channel, patch row) pairs, column group p = i % 33): image columns gx0 + 4 p .. + 3,
  // 16-byte aligned, wholly inside or wholly outside the image; -> X[r][phase][p] for the four phases
  const bool vec = !U8 && (a.Win & 3) == 0;
  constexpr int NP = 3 * P::PH * P::PWQ, NIT = (NP + 255) / 256;
  stem_f32x4 v[NIT];
  auto fetch = [&](int T) {
    const int b = T / a.n_sp, sp = T - b * a.n_sp;
    const int gy0 = (sp / tiles_x) * P::TH * P::S - P::PAD, gx0 = (sp % tiles_x) * P::TW * P::S - P::PAD - 1;
    const __amdgpu_buffer_rsrc_t xrs = buf_rsrc(a.in + (long long)b * 3 * a.Hin * a.Win, (long long)3 * a.Hin * a.Win * 4);
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int i = tid + 256 * k;
      const int r = i / P::PWQ, p = i - r * P::PWQ;
      const int c = r / P::PH, yy = r - c * P::PH;
      const int gy = gy0 + yy, gx = gx0 + 4 * p;
      const bool ok = i < NP && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
      const unsigned vo = ok ? (unsigned)((c * a.Hin + gy) * a.Win + gx) * 4u : kOob;
      v[k] = __builtin_bit_cast(stem_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, 0, 0));
    }
  };
  if (vec && T0 < T1) fetch(T0);

  for (int T = T0; T < T1; ++T) {
    const int b = T / a.n_sp, sp = T - b * a.n_sp;
    const int oy0 = (sp / tiles_x) * P::TH, ox0 = (sp % tiles_x) * P::TW;
    const int gy0 = oy0 * P::S - P::PAD, gx0 = ox0 * P::S - P::PAD - 1;   // the patch starts one column left of the first tap: 16-byte aligned
    // ---- the patch, column-phase de-interleaved: X[c][yy][xx mod 4][xx div 4] -----------------------------------------------
    if constexpr (!U8) {
      if (vec) {
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          const int i = tid + 256 * k;
          if (i < NP) {
            const int r = i / P::PWQ, p = i - r * P::PWQ;
            float* d = X + r * P::RP + p;
            d[0] = v[k][0];
            d[P::PWQ] = v[k][1];
            d[2 * P::PWQ] = v[k][2];
            d[3 * P::PWQ] = v[k][3];
          }
        }
      } else {
        // any width: LDS-DMA, a wave per row, one dword per lane (a stride-4 gather); LDS word e of a row <- column
        // gx0 + 4 (e mod 33) + e div 33
        const __amdgpu_buffer_rsrc_t xrs = buf_rsrc(a.in + (long long)b * 3 * a.Hin * a.Win, (long long)3 * a.Hin * a.Win * 4);
        unsigned vo[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int e = lane + 64 * j;
          const int gx = gx0 + (e % P::PWQ) * 4 + e / P::PWQ;
          vo[j] = (e < P::RP && gx >= 0 && gx < a.Win) ? (unsigned)gx * 4u : kOob;
        }
        for (int r = wave; r < 3 * P::PH; r += 4) {
          const int c = r / P::PH, yy = r - c * P::PH;
          const int gy = gy0 + yy;
          const bool row_ok = gy >= 0 && gy < a.Hin;                                  // wave-uniform
          const unsigned so = row_ok ? (unsigned)((c * a.Hin + gy) * a.Win) * 4u : 0u;
          float* dst = X + c * P::PLANE + yy * P::RP;
          bglds4(xrs, dst, row_ok ? vo[0] : kOob, so);
          bglds4(xrs, dst + 64, row_ok ? vo[1] : kOob, so);
          if (lane < P::RP - 128) bglds4(xrs, dst + 128, row_ok ? vo[2] : kOob, so);
        }
      }
    } else {
      const unsigned char* src = a.in_u8 + (long long)b * a.Hin * a.Win * 3;
      const float m0 = a.u8_mean[0], m1 = a.u8_mean[1], m2 = a.u8_mean[2], sc = a.u8_scale;
      for (int e = tid; e < P::PH * P::RP; e += 256) {
        const int yy = e / P::RP, xx = e - yy * P::RP;
        const int gy = gy0 + yy, gx = gx0 + xx;
        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
        if (gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) {
          const unsigned char* px = src + ((long long)gy * a.Win + gx) * 3;
          v0 = (float)px[0] - m0;
          v1 = (float)px[1] - m1;
          v2 = (float)px[2] - m2;
          if (sc != 1.0f) { v0 /= sc; v1 /= sc; v2 /= sc; }   // im_tensor.float().div(255): the same division as the ingest kernel
        }
        float* d = X + yy * P::RP + (xx % P::S) * P::PWQ + xx / P::S;
        d[0] = v0;
        d[P::PLANE] = v1;
        d[2 * P::PLANE] = v2;
      }
    }
    if (T == T0 || !vec) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA (the weights; the gathered patch) has landed
    __syncthreads();
    if (vec && T + 1 < T1) fetch(T + 1);                      // in flight under the 84 MFMAs below

    // ---- 84 k-steps: step (c, ky, j) multiplies the tap columns 2j - 1 (lanes 0-31) and 2j (lanes 32-63) of row ky, channel c,
    // i.e. the patch columns 4 l31 + 2j and + 2j + 1 ----------------------------------------------------------------------
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    {
      struct Ops {
        float w, x;
      };
      auto load = [&](Ops& o, auto sc_) {
        constexpr int s_ = decltype(sc_)::value;
        constexpr int j_ = s_ % P::KXP, ky_ = (s_ / P::KXP) % P::KS, c_ = s_ / (P::KXP * P::KS);
        constexpr int kx_ = 2 * j_;                                    // patch column offset of the pair's first half: phase in {0, 2}, the second half is phase + 1
        constexpr int ox_ = c_ * P::PLANE + ky_ * P::RP + (kx_ % P::S) * P::PWQ + kx_ / P::S;
        constexpr int ow_ = (c_ * P::KS + ky_) * P::WROW + kx_ * P::BN;
        lds_read_b32<ow_ * 4>(o.w, wa);
        lds_read_b32<ox_ * 4>(o.x, xa);
      };
      auto wait_for = [&](Ops& o, auto newer_c) {
        constexpr int N_ = decltype(newer_c)::value;
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(o.w), "+v"(o.x) : "n"(N_));
      };
      Ops A, B, C;
      load(A, std::integral_constant<int, 0>{});
      load(B, std::integral_constant<int, 1>{});
      static_for<0, P::NSTEP>([&](auto sc_) {
        constexpr int s_ = decltype(sc_)::value;
        Ops& o = (s_ % 3 == 0) ? A : (s_ % 3 == 1) ? B : C;
        Ops& n = (s_ % 3 == 0) ? C : (s_ % 3 == 1) ? A : B;        // two steps ahead
        if constexpr (s_ + 2 < P::NSTEP) {
          load(n, std::integral_constant<int, s_ + 2>{});
          wait_for(o, std::integral_constant<int, 4>{});
        } else if constexpr (s_ + 1 < P::NSTEP) {
          wait_for(o, std::integral_constant<int, 2>{});
        } else {
          wait_for(o, std::integral_constant<int, 0>{});
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(o.w, o.x, acc, 0, 0, 0);
      });
    }

    // ---- epilogue from the accumulator registers: bias (+ folded BN), activation, unconditional buffer stores ----------------
    const __amdgpu_buffer_rsrc_t ors = buf_rsrc(a.out + ((long long)b * a.out_ctot + a.out_coff) * HWo, (long long)a.Cout * HWo * 4);
    const int gy = oy0 + wave, gx = ox0 + l31;
    const unsigned voff = (gy < a.Hout && gx < a.Wout) ? (unsigned)(gy * a.Wout + gx) * 4u + (unsigned)(4 * half) * hw4 : kOob;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2);
      const float bias_r = a.bias ? __shfl(bv, rr + 4 * half, 64) : 0.0f;
      float o_ = acc[r] + bias_r;
      if (a.act == ACT_RELU) o_ = fmaxf(o_, 0.0f);
      else if (a.act == ACT_RELU6) o_ = fminf(fmaxf(o_, 0.0f), 6.0f);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o_), ors, voff, (unsigned)(co_base + rr) * hw4, 0);
    }
    __syncthreads();      // every wave is done reading X: the next tile's patch may overwrite it
  }
}

template <bool U8>
KernelEntry entry_stem_s4() {
  return KernelEntry{conv_stem_s4_kernel<U8>, StemS4::LDS_BYTES, 256};
}

}  // namespace
void conv_fill_stem_s4(void* row_f32, void* row_u8);
}  // namespace fdt
