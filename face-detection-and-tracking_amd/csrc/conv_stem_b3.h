// conv_stem_b3.h -- the 7x7 / stride-4 stem of FaceBoxes (FACEBOX/networks.py:89: Conv2d(3, 24, 7, stride 4, padding 3)) with
// SPLIT-bf16 products on the bf16 matrix pipe (class CONV_7x7_S4_B3, round 5).  Arithmetic: conv_b3.h (every f32 operand is
// exactly the sum of three bf16 planes; the six largest plane products, f32 accumulate: the f32 MFMA's error against f64).
//
// Why here: conv1 is the longest kernel of the FaceBoxes step and MATRIX-bound in the f32 form (conv_stem_s4.h: 84 f32 MFMAs of
// 64 cycles per 128-pixel tile = 133 us per batch of 16, 0.54 of the f32 peak in padded FLOPs) while it moves only 300 MB.  The
// k-dimension of this layer suits the bf16 instruction's K = 16 without any reshuffling: one (channel, tap row) of a pixel is 8
// consecutive input columns (taps -1 .. 6, the first one's weight zero, like the f32 kernel), i.e. one bf16x8 operand, and a
// k-step is two such rows.  66 bf16 MFMAs of 32 cycles per tile instead of 84 of 64, and the vector-ALU work of the staging
// (the split) co-executes with them.
//   * patch in LDS as THREE bf16 planes [plane][c][19 rows][132 columns], NOT phase-de-interleaved: a lane's B operand is the 16
//     bytes at column 4 * n of its row -- 8-byte aligned, one ds_read2_b64;
//   * the next tile's f32 pieces are fetched into registers (eight buffer_load_dwordx4 per thread) under the current tile's
//     MFMAs, split into the planes and written with ds_write_b64 after them;
//   * weights: [plane][11 k-steps][k-half][32 couts][8 columns] bf16, split on the host (tile_weights), resident for all tiles of
//     the persistent workgroup; pair 21 (the second half of the last k-step) is zero;
//   * 78.9 KB of LDS: two workgroups per CU; register epilogue of conv_stem_s4.h.
// Needs Cin = 3, Win % 4 == 0, f32 NCHW input.  Not bit-identical to the f32 classes; same tolerance (test_facebox_stem_b3).
#pragma once
#include "conv_b3.h"

namespace fdt {
namespace {

typedef float stemb3_f32x4 __attribute__((ext_vector_type(4)));

struct StemB3 {
  static constexpr int S = 4, BN = 32, KS = 7, PAD = 3;
  static constexpr int TH = 4, TW = 32;
  static constexpr int PH = (TH - 1) * S + KS;                       // 19 patch rows
  static constexpr int RP = TW * S + 4;                              // 132 columns per row (the first one left of tap 0)
  static constexpr int NROW = 3 * PH;                                // 57 (channel, patch row) pairs
  static constexpr int PLANE_B = NROW * RP * 2;                      // bytes of one bf16 plane of the patch: 15 048
  static constexpr int X_B = (3 * PLANE_B + 15) / 16 * 16;          // 45 144 -> 45 152: the weights behind it are read 16 bytes at a time
  static constexpr int NSTEP = 11;                                   // k-steps of 16: 22 (channel, tap row) pairs, the last one zero
  static constexpr int WPLANE_B = NSTEP * 2 * BN * 16;               // bytes of one weight plane: 11 264
  static constexpr int W_B = 3 * WPLANE_B;                           // 33 792
  static constexpr int WSZ = W_B / 4;                                // floats per channel tile in global memory
  static constexpr size_t LDS_BYTES = (size_t)X_B + W_B;             // 78 944
  static constexpr int NPIECE = NROW * (RP / 4);                     // 1881 16-byte pieces of f32 input per tile
  static constexpr int NIT = (NPIECE + 255) / 256;                   // 8 per thread
  static_assert(X_B % 16 == 0 && W_B % 4096 == 1024 && W_B / 4096 == 8, "LDS-DMA rounds of the weights: eight of 4 KB + 1 KB");
};

__device__ __forceinline__ void lds_read2_b64(bf16x8& v, unsigned addr) {      // the 16 bytes at an 8-byte aligned address
  asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(v) : "v"(addr));
}

__global__ __launch_bounds__(256, 2) void conv_stem_s4_b3_kernel(const ConvArgs a) {
  using P = StemB3;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* Xb = reinterpret_cast<char*>(smem);
  float* Wl = smem + P::X_B / 4;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles_x = (a.Wout + P::TW - 1) / P::TW;
  const int n_tile = blockIdx.y;
  const int total = a.B * a.n_sp;
  const int T0 = blockIdx.x * a.tiles_per_wg;
  const int T1 = min(T0 + a.tiles_per_wg, total);

  // weights of this channel tile: 33 792 bytes by LDS-DMA (eight 4 KB rounds of the workgroup + 1 KB of wave 0)
  {
    const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(a.w + (long long)n_tile * P::WSZ, (long long)P::W_B);
#pragma unroll
    for (int k = 0; k < 8; ++k) bglds16(wrs, Wl + wave * 256 + 1024 * k, (unsigned)tid * 16u, 4096u * k);
    if (wave == 0) bglds16(wrs, Wl + 8192, (unsigned)lane * 16u, 4096u * 8);
  }
  const int HWo = a.Hout * a.Wout;
  const unsigned hw4 = (unsigned)HWo * 4u;
  const int co_base = n_tile * P::BN;
  const __amdgpu_buffer_rsrc_t brs = buf_rsrc(a.bias, a.bias ? (long long)a.Cout * 4 : 0);
  const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (unsigned)(co_base + l31) * 4u, 0, 0));
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  // wave = tile row, lane l31 = output column: the operand starts at column 4 * l31 of patch row 4 * wave + ky
  const unsigned xa = lds0 + (unsigned)((wave * P::S) * P::RP + 4 * l31) * 2u;
  const unsigned wa = lds0 + (unsigned)P::X_B + (unsigned)(half * P::BN + l31) * 16u;

  // piece i = 256 k + tid = (row r = i / 33 of the 57 (channel, patch row) pairs, column group p = i % 33): image columns
  // gx0 + 4 p .. + 3.  What does not depend on the tile is computed once: the piece's offset relative to the patch origin and its
  // (row, column) for the bounds test
  int p_off[P::NIT], p_yy[P::NIT], p_x4[P::NIT];
#pragma unroll
  for (int k = 0; k < P::NIT; ++k) {
    const int i = tid + 256 * k;
    const int r = i / (P::RP / 4), p = i - r * (P::RP / 4);
    const int c = r / P::PH, yy = r - c * P::PH;
    p_off[k] = ((c * a.Hin + yy) * a.Win + 4 * p) * 4;
    p_yy[k] = i < P::NPIECE ? yy : -(1 << 20);                    // pieces past the patch: never inside the image
    p_x4[k] = 4 * p;
  }
  // TWO tiles ahead: a tile's MFMAs last ~1 us, an HBM round trip longer -- the pieces of tile T + 2 are requested while tile T
  // computes and split while tile T + 1 does (two register sets that swap roles, the tile loop is unrolled by two)
  stemb3_f32x4 va[P::NIT], vb[P::NIT];
  auto fetch = [&](int T, stemb3_f32x4* v) {
    const int b = T / a.n_sp, sp = T - b * a.n_sp;
    const int gy0 = (sp / tiles_x) * P::TH * P::S - P::PAD, gx0 = (sp % tiles_x) * P::TW * P::S - P::PAD - 1;
    const __amdgpu_buffer_rsrc_t xrs = buf_rsrc(a.in + (long long)b * conv_in_bstride(a), (long long)3 * a.Hin * a.Win * 4);
    const int base = (gy0 * a.Win + gx0) * 4;
#pragma unroll
    for (int k = 0; k < P::NIT; ++k) {
      const int gy = gy0 + p_yy[k], gx = gx0 + p_x4[k];
      const bool ok = gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
      const unsigned vo = ok ? (unsigned)(base + p_off[k]) : kOob;
      v[k] = __builtin_bit_cast(stemb3_f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, 0, 0));
    }
  };
  // the pieces in registers -> three bf16 planes in LDS (8 bytes per piece and plane)
  auto stage = [&](const stemb3_f32x4* v) {
#pragma unroll
    for (int k = 0; k < P::NIT; ++k) {
      if (k < P::NIT - 1 || tid < P::NPIECE - 256 * (P::NIT - 1)) {
        unsigned a0, a1, a2, b0, b1, b2;
        split3_bf16_pair((f32x2v){v[k][0], v[k][1]}, a0, a1, a2);
        split3_bf16_pair((f32x2v){v[k][2], v[k][3]}, b0, b1, b2);
        char* d = Xb + (size_t)(tid + 256 * k) * 8;                   // (r * 132 + 4 p) * 2 bytes = i * 8
        *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
        *reinterpret_cast<uint2*>(d + P::PLANE_B) = make_uint2(a1, b1);
        *reinterpret_cast<uint2*>(d + 2 * P::PLANE_B) = make_uint2(a2, b2);
      }
    }
  };
  // bias of this lane's sixteen output channels, and the activation as a clamp (exact: max(x, -inf) = x)
  float bias_r[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bias_r[r] = a.bias ? __shfl(bv, (r & 3) + 8 * (r >> 2) + 4 * half, 64) : 0.0f;
  const float act_lo = a.act == ACT_NONE ? -__builtin_huge_valf() : 0.0f;
  const float act_hi = a.act == ACT_RELU6 ? 6.0f : __builtin_huge_valf();
  if (T0 < T1) fetch(T0, va);
  if (T0 + 1 < T1) fetch(T0 + 1, vb);

  auto tile = [&](int T, stemb3_f32x4* vcur) {
    const int b = T / a.n_sp, sp = T - b * a.n_sp;
    const int oy0 = (sp / tiles_x) * P::TH, ox0 = (sp % tiles_x) * P::TW;
    stage(vcur);                                                      // waits for this tile's pieces (the compiler's vmcnt)
    if (T == T0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the weights' LDS-DMA has landed
    __syncthreads();
    if (T + 2 < T1) fetch(T + 2, vcur);                               // this register set is free again: two tiles ahead

    // 11 k-steps: lanes 0-31 multiply pair q = 2 s, lanes 32-63 pair 2 s + 1 ((channel, tap row) = (q / 7, q % 7); pair 21: zero weights)
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    struct Ops {
      bf16x8 A[3], B[3];
    };
    auto load = [&](Ops& o, auto sc_) {
      constexpr int s_ = decltype(sc_)::value;
      constexpr int q0_ = 2 * s_, q1_ = (2 * s_ + 1 < 21) ? 2 * s_ + 1 : 20;
      constexpr unsigned o0_ = (unsigned)(((q0_ / 7) * P::PH + q0_ % 7) * P::RP) * 2u;
      constexpr unsigned o1_ = (unsigned)(((q1_ / 7) * P::PH + q1_ % 7) * P::RP) * 2u;
      const unsigned xad = xa + (half ? o1_ : o0_);
      lds_read_b128<(s_ * 2 * P::BN * 16)>(o.A[0], wa);
      lds_read_b128<(P::WPLANE_B + s_ * 2 * P::BN * 16)>(o.A[1], wa);
      lds_read_b128<(2 * P::WPLANE_B + s_ * 2 * P::BN * 16)>(o.A[2], wa);
      lds_read2_b64(o.B[0], xad);
      lds_read2_b64(o.B[1], xad + (unsigned)P::PLANE_B);
      lds_read2_b64(o.B[2], xad + 2u * (unsigned)P::PLANE_B);
    };
    // the operands of step s + 1 are requested before the MFMAs of step s (two register sets, exact lgkmcnt: six reads newer)
    Ops O0, O1;
    load(O0, std::integral_constant<int, 0>{});
    static_for<0, P::NSTEP>([&](auto sc_) {
      constexpr int s_ = decltype(sc_)::value;
      Ops& o = (s_ & 1) ? O1 : O0;
      Ops& n = (s_ & 1) ? O0 : O1;
      if constexpr (s_ + 1 < P::NSTEP) {
        load(n, std::integral_constant<int, s_ + 1>{});
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(o.A[0]), "+v"(o.A[1]), "+v"(o.A[2]), "+v"(o.B[0]), "+v"(o.B[1]), "+v"(o.B[2]));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o.A[0]), "+v"(o.A[1]), "+v"(o.A[2]), "+v"(o.B[0]), "+v"(o.B[1]), "+v"(o.B[2]));
      }
      // smallest plane products first (a-plane, b-plane): 11 20 02 10 01 00
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.A[1], o.B[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.A[2], o.B[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.A[0], o.B[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.A[1], o.B[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.A[0], o.B[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o.A[0], o.B[0], acc, 0, 0, 0);
    });

    // epilogue from the accumulator registers: bias (+ folded BN), activation, unconditional buffer stores (conv_stem_s4.h)
    const __amdgpu_buffer_rsrc_t ors = buf_rsrc(a.out + ((long long)b * a.out_ctot + a.out_coff) * HWo, (long long)a.Cout * HWo * 4);
    const int gy = oy0 + wave, gx = ox0 + l31;
    const unsigned voff = (gy < a.Hout && gx < a.Wout) ? (unsigned)(gy * a.Wout + gx) * 4u + (unsigned)(4 * half) * hw4 : kOob;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rr = (r & 3) + 8 * (r >> 2);
      const float o_ = fminf(fmaxf(acc[r] + bias_r[r], act_lo), act_hi);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o_), ors, voff, (unsigned)(co_base + rr) * hw4, 0);
    }
    __syncthreads();      // every wave is done reading the patch: the next tile's planes may overwrite it
  };
  for (int T = T0; T < T1; T += 2) {
    tile(T, va);
    if (T + 1 < T1) tile(T + 1, vb);
  }
}

inline KernelEntry entry_stem_s4_b3() { return KernelEntry{conv_stem_s4_b3_kernel, StemB3::LDS_BYTES, 256}; }

}  // namespace
}  // namespace fdt
