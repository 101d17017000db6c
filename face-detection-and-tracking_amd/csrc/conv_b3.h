// conv_b3.h -- 1x1 / stride-1 convolution with SPLIT-bf16 products on the bf16 matrix pipe (class CONV_1x1_S1_B3, round 5).
//
// Why.  v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate (64 FLOP/clk/SIMD); v_mfma_f32_32x32x16_bf16
// does 16x the work per cycle.  An f32 value is EXACTLY the sum of three bf16 values x = x0 + x1 + x2 (8 + 8 + 8 significant
// bits, each the round-to-nearest bf16 of the remainder), so a product x*y is the sum of nine plane products, and the six
// largest -- x0y0, x0y1, x1y0, x0y2, x2y0, x1y1 -- leave out terms of at most 3 * 2^-24 relative: the size of ONE f32
// rounding.  Each plane product is exact in the bf16 MFMA (8 x 8 significant bits) and is accumulated in f32 like the f32
// MFMA accumulates.  Measured (tools/microbench/bf16x3_mfma.hip, docs/EXPERIMENTS.md R5-4): a [32 x 1024] x [1024 x 32] product
// has 5.5e-7 relative RMS error against f64 in this form, 6.3e-7 with the f32 MFMA; six bf16 MFMAs (192 cycles) replace eight
// f32 MFMAs (512 cycles) per 16 input channels.  The result is NOT bit-identical to the f32 classes (another summation order
// inside the instruction); it passes the same tolerances (tests/test_gpu_conv.py: test_split_bf16_1x1*, the stage and end-to-end
// tests with the plans that use it).
//
// How.  GEMM view and tile shapes of conv_kernel.h (weights = A operand, 128-pixel tile = B operand, four waves, LDS ring by
// LDS-DMA with counted vmcnt), one stage = 16 input channels = ONE bf16 MFMA k-step:
//   * weights are split into their three planes on the HOST (tile_weights) and stored per stage as
//     [plane][k-half][BN couts][8 k] bf16 -- a lane's A operand of a plane is one ds_read_b128;
//   * the activations arrive as f32 [16 channels][128 pixels] exactly like in the f32 class; a lane reads the eight channels
//     of its pixel (8 x ds_read_b32), splits them in registers (v_cvt_pk_bf16_f32 + two subtractions per value) -- vector-ALU
//     work that CO-EXECUTES with the bf16 MFMAs of the SIMD's other waves (SQ_VALU_MFMA_COEXEC_CYCLES = 50 % of the busy
//     cycles in the microbenchmark; it is 0 for f32 MFMAs, R4-4) -- and feeds three B planes;
//   * per (cout tile, pixel tile): six MFMAs, smallest plane products first;
//   * epilogue: the f32 class's (bias, fused bilinear x2 upsample-add, residual, ReLU / ReLU6, second destination, split-K slabs).
// Needs Win % 4 == 0 (16-byte activation staging).  Channels past Cin read as zeros (buffer bounds), weights are zero there.
// S = 2 (class CONV_1x1_S2_B3): the same kernel with a dword gather of every second pixel in the staging (Wout % 4 == 0).
// KS = 3 (class CONV_3x3_S2_B3: 3x3 / stride 2 / padding 1, pyramid.py:99 in the first block of layer2-4): nine stages per 16-channel
// group, one per tap -- the gather of tap (ky, kx) is the centre tap's plus a constant, pixels in the padding carry the out-of-range
// offset (LDS-DMA writes zeros); split-K cuts between channel groups.
#pragma once
#include "conv_kernel.h"

namespace fdt {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <class T, int S = 1>
struct LayoutB3 {
  static constexpr int KC = 16;
  static constexpr int XSZ = KC * T::BM;                       // floats: [16][BM]
  static constexpr int NXV = S == 1 ? XSZ / 1024 : XSZ / 256;  // LDS-DMA instructions per wave and stage: dwordx4 (stride 1), dword
  static constexpr int WSZ = 24 * T::BN;                       // floats holding [3][2][BN][8] bf16
  static constexpr int WSZP = (WSZ + 1023) / 1024 * 1024;
  static constexpr int NW = WSZP / 1024;
  static constexpr int STAGE = XSZ + WSZP;
  static constexpr int LOADS = NXV + NW;
  static constexpr int EROW = T::BM + 4;
  static constexpr int EPI = T::WN * 32 * EROW;
  static constexpr int RING = T::NBUF * STAGE;
  static constexpr size_t LDS_BYTES = (size_t)(RING > EPI ? RING : EPI) * sizeof(float);
  static_assert(XSZ % 1024 == 0, "whole dwordx4 LDS-DMA rounds");
  static_assert(T::NBUF == 3 || T::NBUF == 4, "ring of three or four stages");
};

// x = p0 + p1 + p2 exactly (each conversion rounds to nearest even; the remainders are exact in f32)
__device__ __forceinline__ void split3_bf16(float x, __bf16& p0, __bf16& p1, __bf16& p2) {
  p0 = (__bf16)x;
  const float r1 = x - (float)p0;
  p1 = (__bf16)r1;
  const float r2 = r1 - (float)p1;
  p2 = (__bf16)r2;
}

// The same split on a PAIR of values, with the packed instructions (v_cvt_pk_bf16_f32, v_pk_add_f32): nine vector-ALU operations
// per pair where the compiler's own code for two scalar splits takes thirteen.  P0 / P1 / P2: the pair's three planes as packed
// bf16 pairs (low half = x.x).
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(f32x2v x) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2v));
}
__device__ __forceinline__ f32x2v unpack_bf16_pair(unsigned p) {
  f32x2v f;
  f.x = __builtin_bit_cast(float, p << 16);
  f.y = __builtin_bit_cast(float, p & 0xffff0000u);
  return f;
}
__device__ __forceinline__ void split3_bf16_pair(f32x2v x, unsigned& P0, unsigned& P1, unsigned& P2) {
  P0 = cvt_pk_bf16(x);
  const f32x2v r1 = x - unpack_bf16_pair(P0);
  P1 = cvt_pk_bf16(r1);
  const f32x2v r2 = r1 - unpack_bf16_pair(P1);
  P2 = cvt_pk_bf16(r2);
}

template <int OFF>
__device__ __forceinline__ void lds_read_b128(bf16x8& v, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

template <class T, int S = 1, int KS = 1>
__global__ __launch_bounds__(256, (T::MI * T::NI >= 4) ? 2 : 3) void conv_b3_kernel(const ConvArgs a) {
  using L = LayoutB3<T, S>;
  constexpr int KK = KS * KS;                  // LDS stages per 16-channel group: one per tap (3x3 / stride 2: class CONV_3x3_S2_B3)
  static_assert(KS == 1 || (KS == 3 && S == 2), "1x1 (stride 1, 2) and 3x3 / stride 2 / padding 1");
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / T::WN, wn = wave % T::WN;
  const int half = lane >> 5, l31 = lane & 31;

  const int tiles_x = (a.Wout + T::TW - 1) / T::TW;
  FDT_BLOCK_MAP(a, tile_id, n_tile);
  const int oy0 = (tile_id / tiles_x) * T::TH;
  const int ox0 = (tile_id % tiles_x) * T::TW;
  const int b = blockIdx.z / a.ksplit;
  const int ks = blockIdx.z - b * a.ksplit;

  const int HWin = a.Hin * a.Win;
  const int HWout = a.Hout * a.Wout;
  const float* in_b = a.in + (long long)b * conv_in_bstride(a);
  const int nstages = (a.Cin + L::KC - 1) / L::KC;
  const float* w_t = a.w + (long long)n_tile * nstages * KK * L::WSZP;
  const int s_begin = (int)((long long)nstages * ks / a.ksplit) * KK;       // split-K cuts between channel groups
  const int s_end = (int)((long long)nstages * (ks + 1) / a.ksplit) * KK;

  // staging plan.  Stride 1: float4 v = 256 * k + tid covers 4 consecutive pixels of one tile row of one channel of the stage;
  // stride 2 (the bottleneck's downsample branch): dword v = 256 * k + tid is ONE pixel, every second one of every second row
  const __amdgpu_buffer_rsrc_t xrs = buf_rsrc(in_b, (long long)a.Cin * HWin * 4);
  const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(w_t, 0x7fffffffll);
  unsigned xoff[L::NXV];
  unsigned xfl[KS == 3 ? L::NXV : 1];          // 3x3: which taps of this pixel fall into the padding (bits: 1 top, 2 bottom, 4 left, 8 right, 16 no pixel)
#pragma unroll
  for (int k = 0; k < L::NXV; ++k) {
    const int v = tid + 256 * k;
    if constexpr (S == 1) {
      const int c = v / (T::BM / 4);
      const int p = (v - c * (T::BM / 4)) * 4;
      const int gy = oy0 + p / T::TW, gx = ox0 + p % T::TW;
      const bool ok = gy < a.Hin && gx < a.Win;
      xoff[k] = ok ? (unsigned)(c * HWin + gy * a.Win + gx) * 4u : kOob;
    } else {
      const int c = v / T::BM;
      const int p = v - c * T::BM;
      const int oy = oy0 + p / T::TW, ox = ox0 + p % T::TW;
      const bool ok = oy < a.Hout && ox < a.Wout;
      if constexpr (KS == 1) {
        xoff[k] = ok ? (unsigned)(c * HWin + oy * S * a.Win + ox * S) * 4u : kOob;
      } else {      // the CENTRE tap's pixel (2 oy, 2 ox): always inside the image for a valid output pixel (padding 1)
        xoff[k] = (unsigned)(c * HWin + oy * S * a.Win + ox * S) * 4u;
        xfl[k] = (oy == 0 ? 1u : 0u) | (oy * S + 1 >= a.Hin ? 2u : 0u) | (ox == 0 ? 4u : 0u) | (ox * S + 1 >= a.Win ? 8u : 0u) | (ok ? 0u : 16u);
      }
    }
  }
#define FDT_B3_STAGE(s_, buf_)                                                                            \
  {                                                                                                       \
    const int cg_ = (s_) / KK, tap_ = (s_) - cg_ * KK;         /* 16-channel group, tap (1x1: the stage itself, 0) */ \
    const unsigned xso_ = (unsigned)(cg_ * L::KC) * (unsigned)HWin * 4u;                                  \
    if constexpr (S == 1) {                                                                               \
      float* X_ = smem + (buf_) * L::STAGE + wave * 256;                                                  \
      _Pragma("unroll") for (int k = 0; k < L::NXV; ++k) bglds16(xrs, X_ + 1024 * k, xoff[k], xso_);      \
    } else if constexpr (KS == 1) {                                                                       \
      float* X_ = smem + (buf_) * L::STAGE + wave * 64;                                                   \
      _Pragma("unroll") for (int k = 0; k < L::NXV; ++k) bglds4(xrs, X_ + 256 * k, xoff[k], xso_);        \
    } else {                                                                                              \
      const int ky_ = tap_ / 3, kx_ = tap_ - 3 * ky_;                                                     \
      const int dl_ = ((ky_ - 1) * a.Win + (kx_ - 1)) * 4;                                                \
      const unsigned tm_ = 16u | (ky_ == 0 ? 1u : 0u) | (ky_ == 2 ? 2u : 0u) | (kx_ == 0 ? 4u : 0u) | (kx_ == 2 ? 8u : 0u); \
      float* X_ = smem + (buf_) * L::STAGE + wave * 64;                                                   \
      _Pragma("unroll") for (int k = 0; k < L::NXV; ++k)                                                  \
          bglds4(xrs, X_ + 256 * k, (xfl[k] & tm_) ? kOob : xoff[k] + (unsigned)dl_, xso_);               \
    }                                                                                                     \
    const unsigned wso_ = (unsigned)((s_) * L::WSZP) * 4u;                                                \
    float* W_ = smem + (buf_) * L::STAGE + L::XSZ + wave * 256;                                           \
    _Pragma("unroll") for (int k = 0; k < L::NW; ++k) bglds16(wrs, W_ + 1024 * k, (unsigned)tid * 16u, wso_ + 4096u * k); \
  }

  // per-lane LDS read byte offsets inside a stage
  unsigned xo[T::MI], wo[T::NI];
#pragma unroll
  for (int i = 0; i < T::MI; ++i) xo[i] = (unsigned)((half * 8) * T::BM + wm * (T::MI * 32) + i * 32 + l31) * 4u;
#pragma unroll
  for (int j = 0; j < T::NI; ++j) wo[j] = (unsigned)L::XSZ * 4u + (unsigned)(half * T::BN + wn * (T::NI * 32) + j * 32 + l31) * 16u;

  f32x16 acc[T::NI][T::MI];
#pragma unroll
  for (int j = 0; j < T::NI; ++j)
#pragma unroll
    for (int i = 0; i < T::MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.0f;

  const int nst = s_end - s_begin;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;

  // One stage's operands of this wave: A planes straight from LDS, the eight f32 channels of the lane's pixels
  struct Ops {
    bf16x8 A[T::NI][3];
    float xv[T::MI][8];
    bf16x8 Bp[T::MI][3];
  };
  auto issue_reads = [&](Ops& o, int buf) {
    const unsigned sb = lds0 + (unsigned)(buf * L::STAGE) * 4u;
#pragma unroll
    for (int j = 0; j < T::NI; ++j) {
      const unsigned ad = sb + wo[j];      // [plane][half][BN][8 k] bf16: plane stride 2 * BN * 16 bytes
      lds_read_b128<0>(o.A[j][0], ad);
      lds_read_b128<2 * T::BN * 16>(o.A[j][1], ad);
      lds_read_b128<4 * T::BN * 16>(o.A[j][2], ad);
    }
#pragma unroll
    for (int i = 0; i < T::MI; ++i) {
      const unsigned ad = sb + xo[i];      // channel k = 8 * half + q of this lane's pixel
      lds_read_b32<0 * T::BM * 4>(o.xv[i][0], ad);
      lds_read_b32<1 * T::BM * 4>(o.xv[i][1], ad);
      lds_read_b32<2 * T::BM * 4>(o.xv[i][2], ad);
      lds_read_b32<3 * T::BM * 4>(o.xv[i][3], ad);
      lds_read_b32<4 * T::BM * 4>(o.xv[i][4], ad);
      lds_read_b32<5 * T::BM * 4>(o.xv[i][5], ad);
      lds_read_b32<6 * T::BM * 4>(o.xv[i][6], ad);
      lds_read_b32<7 * T::BM * 4>(o.xv[i][7], ad);
    }
  };
  auto wait_reads = [&](Ops& o) {          // every read issued so far has landed (the asm ties the registers to the wait)
#pragma unroll
    for (int i = 0; i < T::MI; ++i)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(o.xv[i][0]), "+v"(o.xv[i][1]), "+v"(o.xv[i][2]), "+v"(o.xv[i][3]), "+v"(o.xv[i][4]), "+v"(o.xv[i][5]),
                     "+v"(o.xv[i][6]), "+v"(o.xv[i][7]));
#pragma unroll
    for (int j = 0; j < T::NI; ++j) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o.A[j][0]), "+v"(o.A[j][1]), "+v"(o.A[j][2]));
  };
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  auto split_pair = [&](Ops& o, int i, int qp) {       // channels 2 qp, 2 qp + 1 of pixel tile i
    unsigned P0, P1, P2;
    split3_bf16_pair((f32x2v){o.xv[i][2 * qp], o.xv[i][2 * qp + 1]}, P0, P1, P2);
    u32x4v b0 = __builtin_bit_cast(u32x4v, o.Bp[i][0]), b1 = __builtin_bit_cast(u32x4v, o.Bp[i][1]),
           b2 = __builtin_bit_cast(u32x4v, o.Bp[i][2]);
    b0[qp] = P0; b1[qp] = P1; b2[qp] = P2;
    o.Bp[i][0] = __builtin_bit_cast(bf16x8, b0);
    o.Bp[i][1] = __builtin_bit_cast(bf16x8, b1);
    o.Bp[i][2] = __builtin_bit_cast(bf16x8, b2);
  };
  // The six plane products of every (cout tile, pixel tile) of stage `c`, smallest first (a-plane, b-plane): 11 20 02 10 01 00;
  // the reads of stage `n` were issued in front of this call: they are waited for after the first third of the MFMAs and the
  // f32 -> 3 x bf16 split of `n` is spread over the rest -- vector-ALU work in the shadow of this wave's own bf16 MFMAs.
  auto mfmas = [&](Ops& c, Ops& n, bool have_next) {
    constexpr int pa[6] = {1, 2, 0, 1, 0, 0}, pb[6] = {1, 0, 2, 0, 1, 0};
    constexpr int NM = T::NI * T::MI * 6, NS = T::MI * 4;      // NS: channel PAIRS to split
    constexpr int FIRST = NM / 3;                          // MFMAs in front of the wait
    constexpr int PER = (NS + (NM - FIRST) - 1) / (NM - FIRST);   // values split per MFMA behind it
    static_for<0, NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int t = m / 6, p = m % 6, j = t / T::MI, i = t % T::MI;
      acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c.A[j][pa[p]], c.Bp[i][pb[p]], acc[j][i], 0, 0, 0);
      if constexpr (m + 1 >= FIRST) {
        if (have_next) {
          if constexpr (m + 1 == FIRST) wait_reads(n);
          static_for<0, PER>([&](auto ec) {
            constexpr int idx = (m + 1 - FIRST) * PER + decltype(ec)::value;      // compile-time: no dynamic register indexing
            if constexpr (idx < NS) split_pair(n, idx / 4, idx % 4);
          });
        }
      }
    });
  };

  // ring of D = NBUF stages: buffer of stage s = s % D.  Prologue: stages 0 .. D-2 requested, stage 0 waited for (the younger ones
  // stay in flight: exact vmcnt), read and split.
  constexpr int D = T::NBUF;
  auto wait_leaving = [&](int stages) {        // every LDS-DMA group but the `stages` youngest has landed
    if (stages >= 2 && D >= 4)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L::LOADS) : "memory");
    else if (stages >= 1)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::LOADS) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
#pragma unroll
  for (int p = 0; p < D - 1; ++p)
    if (p < nst) FDT_B3_STAGE(s_begin + p, p);
  Ops O[2];
  if (nst > 0) {
    wait_leaving(min(nst, D - 1) - 1);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (D - 1 < nst) FDT_B3_STAGE(s_begin + D - 1, D - 1);
    issue_reads(O[0], 0);
    wait_reads(O[0]);
#pragma unroll
    for (int i = 0; i < T::MI; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) split_pair(O[0], i, q);
  }
  // steady state, two stages per trip (the operand sets swap roles): at the top of step `it` the operands of stage `it` sit in
  // registers; stage it + 1 is waited for (stages it + 2 .. it + D - 1 stay in flight), the barrier also says every wave has
  // finished READING stage it -- so stage it + D may be requested into its buffer --, stage it + 1 is read, and the MFMAs of
  // stage `it` run.
  int buf_next = 1;                                        // buffer of stage it + 1
  auto step = [&](int it, Ops& c, Ops& n) {
    const bool have_next = it + 1 < nst;
    if (have_next) {
      wait_leaving(min(nst - it - 2, D - 2));
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (it + D < nst) FDT_B3_STAGE(s_begin + it + D, (buf_next + D - 1) % D);
      issue_reads(n, buf_next);
    }
    mfmas(c, n, have_next);
    buf_next = buf_next == D - 1 ? 0 : buf_next + 1;
  };
  for (int it = 0; it < nst; it += 2) {
    step(it, O[0], O[1]);
    if (it + 1 < nst) step(it + 1, O[1], O[0]);
  }
#undef FDT_B3_STAGE

  // ---- epilogue: conv_kernel.h's, operation for operation (Wout % 4 == 0 is a precondition of this class) -------------------
  float* E = smem;
  float* dst_b;
  const bool raw = a.ws != nullptr;
  if (raw)
    dst_b = a.ws + ((long long)(b * a.ksplit + ks) * a.Cout) * HWout;
  else
    dst_b = a.out + ((long long)b * a.out_ctot + a.out_coff) * HWout;
  const float* res_b = (!raw && a.res) ? a.res + ((long long)b * a.res_ctot + a.res_coff) * HWout : nullptr;
  // Waves 4 x 1 on a 4 x 32 tile: a wave's accumulator register is 32 consecutive pixels of ONE output row and one cout (lanes 32-63:
  // four couts further) -- two whole 128-byte segments per store.  The plain epilogue then runs from the registers like conv_1x1p.h's:
  // no LDS transpose, no barriers (the short-reduction layers spend as long in the LDS epilogue as in their four to eight stages),
  // the residual of cout tile j + 1 in flight under the stores of tile j.  Same arithmetic: acc + bias, + residual, activation.
  if constexpr (T::WM == 4 && T::WN == 1 && T::TW % 32 == 0) {
    if (!raw && !a.up && !a.out2 && (long long)(a.Cout + 8) * HWout * 4 < (1ll << 31)) {      // kOob = 2^31 must stay out of range
      const unsigned hw4 = (unsigned)HWout * 4u;
      const int gy = oy0 + (wm * 32) / T::TW, gx = ox0 + (wm * 32) % T::TW + l31;      // the wave's 32 pixels: one row segment
      const unsigned voff = (gy < a.Hout && gx < a.Wout) ? (unsigned)(gy * a.Wout + gx) * 4u + (unsigned)(4 * half) * hw4 : kOob;
      const __amdgpu_buffer_rsrc_t ors = buf_rsrc(dst_b, (long long)a.Cout * HWout * 4);          // couts past Cout fall off the end
      const __amdgpu_buffer_rsrc_t rrs = buf_rsrc(res_b, res_b ? (long long)a.Cout * HWout * 4 : 0);
      const __amdgpu_buffer_rsrc_t brs = buf_rsrc(a.bias, a.bias ? (long long)a.Cout * 4 : 0);
      const int co0 = n_tile * T::BN;
      float rv[2][16];
      auto load_res = [&](int j, float* d) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          d[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, voff, (unsigned)(co0 + j * 32 + (r & 3) + 8 * (r >> 2)) * hw4, 0));
      };
      if (res_b) load_res(0, rv[0]);
      static_for<0, T::NI>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const float bvj = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (unsigned)(co0 + j * 32 + l31) * 4u, 0, 0));
        if constexpr (j + 1 < T::NI)
          if (res_b) load_res(j + 1, rv[(j + 1) & 1]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2);
          float v = acc[j][0][r] + (a.bias ? __shfl(bvj, rr + 4 * half, 64) : 0.0f);
          if (res_b) v += rv[j & 1][r];
          if (a.act == ACT_RELU) v = fmaxf(v, 0.0f);
          else if (a.act == ACT_RELU6) v = fminf(fmaxf(v, 0.0f), 6.0f);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ors, voff, (unsigned)(co0 + j * 32 + rr) * hw4, 0);
        }
      });
      return;
    }
  }
  constexpr int ROWS = T::WN * 32;
  constexpr int C4 = T::BM / 4;
  constexpr int PER = (ROWS * C4 + 255) / 256;
#pragma unroll
  for (int j = 0; j < T::NI; ++j) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < T::MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        E[row * L::EROW + wm * (T::MI * 32) + i * 32 + l31] = acc[j][i][r];
      }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int idx = tid + 256 * q;
      if (idx < ROWS * C4) {
        const int row = idx / C4, c4 = idx - row * C4;
        const int p = c4 * 4;
        const int oy = oy0 + p / T::TW, ox = ox0 + p % T::TW;
        const int co = n_tile * T::BN + (row >> 5) * (T::NI * 32) + j * 32 + (row & 31);
        if (oy < a.Hout && ox < a.Wout && co < a.Cout) {
          float4 v = *reinterpret_cast<const float4*>(E + row * L::EROW + p);
          const long long off = (long long)co * HWout + (long long)oy * a.Wout + ox;
          if (!raw) {
            if (a.bias) {
              const float bv = a.bias[co];
              v.x += bv; v.y += bv; v.z += bv; v.w += bv;
            }
            if (a.up) {   // ContextTexture: + bilinear x2 of the coarser map, before the residual like the reduce pass
              float u4[4] = {v.x, v.y, v.z, v.w};
              add_upsampled_x2<4>(a.up + ((long long)b * a.Cout + co) * a.up_h * a.up_w, a.up_h, a.up_w, oy, ox, u4);
              v = make_float4(u4[0], u4[1], u4[2], u4[3]);
            }
            if (res_b) {
              const float4 rv = *reinterpret_cast<const float4*>(res_b + off);
              v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
            }
            if (a.act == ACT_RELU) {
              v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            } else if (a.act == ACT_RELU6) {
              v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f);
              v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
            }
          }
          float* dst = dst_b + off;
          if (a.out2 && co >= a.out2_from)
            dst = a.out2 + ((long long)b * a.out2_ctot + a.out2_coff + (co - a.out2_from)) * HWout + (long long)oy * a.Wout + ox;
          slab_store4(dst, v.x, v.y, v.z, v.w, raw && a.sk_count);
        }
      }
    }
  }
  if (a.sk_count)
    splitk_combine_tile<256>(a, b, tile_id + a.n_sp * n_tile, n_tile * T::BN, T::BN, oy0, ox0, T::TH, T::TW, (unsigned*)smem);
}

using TB3_128x128 = Tile<8, 16, 128, 2, 2, 3>;
using TB3_128x64 = Tile<8, 16, 64, 2, 2, 3>;
using TB3_128x128W = Tile<4, 32, 128, 2, 2, 3>;
using TB3_128x64W = Tile<4, 32, 64, 2, 2, 3>;
// waves 4 x 1: every wave owns 32 of the tile's pixels and ALL its couts, so each activation is split by exactly one wave
// (36 vector-ALU operations per stage and wave in place of 72, R5-11) at the price of every wave reading the whole weight stage
using TB3_128x128W4 = Tile<4, 32, 128, 4, 1, 3>;
using TB3_128x64W4 = Tile<4, 32, 64, 4, 1, 3>;
// long rows: 2 x 64 and 1 x 128 pixels (256 / 512 contiguous bytes per channel row in the staging and in the stores)
using TB3_R2_128 = Tile<2, 64, 128, 4, 1, 3>;
using TB3_R2_64 = Tile<2, 64, 64, 4, 1, 3>;
using TB3_R1_128 = Tile<1, 128, 128, 4, 1, 3>;
using TB3_R1_64 = Tile<1, 128, 64, 4, 1, 3>;
// (a ring of FOUR stages -- Tile<8, 16, BN, 4, 1, 4>, the kernel takes NBUF = 4 -- measured 3-20 % slower than these on every
// backbone shape, docs/EXPERIMENTS.md R5-11: the class is not bound by LDS-DMA latency; not instantiated)

template <class T, int S = 1, int KS = 1>
KernelEntry entry_b3() {
  return KernelEntry{conv_b3_kernel<T, S, KS>, LayoutB3<T, S>::LDS_BYTES, 256};
}

}  // namespace
}  // namespace fdt
