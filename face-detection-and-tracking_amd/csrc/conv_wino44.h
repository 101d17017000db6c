// conv_wino44.h -- Winograd F(4x4,3x3) variant of the 3x3 / stride 1 / pad 1 convolution on the f32 matrix cores:
// 36 multiplies per 4x4 output tile instead of 144 (4x fewer MFMA FLOPs than the direct form, 1.78x fewer than the
// F(2x2,3x3) kernels of conv_wino.h), exact-arithmetic equivalent of nn.Conv2d(k=3, s=1, p=1) up to the f32 rounding of the
// transforms (measured 2.7e-6 relative RMS per layer against 4.3e-7 for F(2x2,3x3): the interpolation points 0, +-1, +-2
// put factors up to 8 into A^T and 5 into B^T).
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        (per 4x4 output tile, per output channel; 6x6 = 36 positions)
//
// Why the F(2x2) design does not carry over, and what this one does instead:
//  * 36 positions x (64 couts x 32 tiles) are 72 accumulator tiles of 32x32 -- 9 per wave with eight waves (144 registers,
//    two waves per SIMD).  A wave therefore owns a GROUP OF POSITIONS (nine of the 36) for half of the couts, not a row of the
//    grid, and cannot form its own B operands from the raw window: the input transform runs once per (tile, channel) --
//    six waves each produce one ROW of B^T d B for all 32 tiles x 2 channels of a k-step (lane = (channel, tile), exactly the
//    MFMA B-operand layout) and write it to a V buffer in LDS; all eight waves then read A (weights U) and B (V) from LDS.
//  * Everything is pipelined by k-step (two input channels = one MFMA k-depth) in SUPER-STEPS of two k-steps between two
//    barriers: at the barrier the operands of k-step s sit in registers, the weights U(s+1), U(s+2) and the raw patches
//    R(s+3), R(s+4) have landed (LDS-DMA rings of four slots each, one vmcnt(0) per super-step = two k-steps of flight) and
//    V(s+1), V(s+2) are written; each half runs its nine MFMAs while it reads a window, transforms it into V two k-steps
//    ahead, prefetches the next operands and issues the LDS-DMA of two / four k-steps ahead (super_step below).
//  * The matrix pipe and the vector ALU of a SIMD do not co-execute on this chip (SQ_VALU_MFMA_COEXEC_CYCLES = 0), so the
//    loop carries no vector address arithmetic: every LDS offset is an immediate (rings of four slots, loop unrolled by
//    four k-steps), both operand streams are buffer loads (a per-lane offset that never changes + a scalar offset per
//    k-step; padding, out-of-image pieces and the channel past an odd Cin are out-of-range lanes: zeros), the transform is
//    packed (v_pk_fma_f32).  Dilation 2 (W44T<true, 2>): tiles = parity sub-lattices of 8x8-pixel cells, row pitch 44
//    floats so that the dword-pair window reads hit 32 different banks.
//  * The operand stream is 1.9x the F(2x2) kernel's per MFMA cycle (18 KB of weights + 6 KB of patch per 1152 matrix-pipe
//    cycles of a SIMD pair) and a k-step moves ~100 KB through LDS: tools/microbench/ldsdma_feed.hip puts feed + MFMAs at
//    0.70-0.74 us per k-step and CU, the kernel runs 0.61 above its fixed part (docs/EXPERIMENTS.md R3-1).
//  * Output transform: a wave folds its nine positions into two 4-vectors per accumulator element (the rows of M A it
//    touches), the four position groups meet through LDS (two rounds of 128 KB), and every wave finishes whole 4x4 tiles of
//    a quarter of the channels: bias / residual / ReLU and 16-byte row stores.
#pragma once
#include <type_traits>

#include "conv_wino.h"

namespace fdt {
namespace {

template <int OFF>
__device__ __forceinline__ void w44_read_b128(f32x4& v, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536 && OFF % 16 == 0, "ds_read_b128 offset");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void w44_read_b64(f32x2& v, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536 && OFF % 8 == 0, "ds_read_b64 offset");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int O0, int O1>
__device__ __forceinline__ void w44_write2st64_b32(unsigned addr, float x, float y) {   // offsets in units of 64 dwords
  static_assert(O0 >= 0 && O0 < 256 && O1 >= 0 && O1 < 256, "ds_write2st64 offsets");
  asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(addr), "v"(x), "v"(y), "n"(O0), "n"(O1) : "memory");
}

// Packed f32 (two columns per instruction), written as asm because the compiler scalarises these into v_fma_f32 pairs: in a kernel
// whose matrix pipe and vector ALU do not co-execute, a VALU instruction saved is four matrix cycles gained.
#ifndef FDT_W44_NOPK
__device__ __forceinline__ f32x2 pk_fma(f32x2 x, f32x2 y, f32x2 z) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(x), "v"(y), "v"(z));
  return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 x, f32x2 y) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y));
  return d;
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 x, f32x2 y) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(x), "v"(y));
  return d;
}
#else
// A/B build (tools/experiments/w44_pk_ab.sh, make EXTRA=-DFDT_W44_NOPK): the SAME transform, every packed instruction replaced
// by its two scalar halves (v_fma_f32 / v_add_f32 / v_sub_f32, written as asm so that the compiler cannot re-pack them) and
// nothing else changed -- MI355X_MICROARCH.md prices a v_pk_fma_f32 beside MFMAs ~22 cycles above two v_fma_f32.  Same bits.
__device__ __forceinline__ f32x2 pk_fma(f32x2 x, f32x2 y, f32x2 z) {
  float d0, d1;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d0) : "v"(x[0]), "v"(y[0]), "v"(z[0]));
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d1) : "v"(x[1]), "v"(y[1]), "v"(z[1]));
  return f32x2{d0, d1};
}
__device__ __forceinline__ f32x2 pk_add(f32x2 x, f32x2 y) {
  float d0, d1;
  asm("v_add_f32 %0, %1, %2" : "=v"(d0) : "v"(x[0]), "v"(y[0]));
  asm("v_add_f32 %0, %1, %2" : "=v"(d1) : "v"(x[1]), "v"(y[1]));
  return f32x2{d0, d1};
}
__device__ __forceinline__ f32x2 pk_sub(f32x2 x, f32x2 y) {
  float d0, d1;
  asm("v_sub_f32 %0, %1, %2" : "=v"(d0) : "v"(x[0]), "v"(y[0]));
  asm("v_sub_f32 %0, %1, %2" : "=v"(d1) : "v"(x[1]), "v"(y[1]));
  return f32x2{d0, d1};
}
#endif

// VEC: Win % 4 == 0 -- the patch is staged as 16-byte pieces of the aligned superset [ox0 - 4, ox0 + 36) of its columns (ONE
// LDS-DMA instruction per wave and k-step instead of three dword ones); the odd-width variant stages it dword by dword.
template <bool VEC_, int D_ = 1>
struct W44T {
  static constexpr bool VEC = VEC_;
  static constexpr int D = D_;                           // dilation (= padding): 1, or 2 for the SSH context convs (pyramid.py:36,38)
  static constexpr int TTH = 4, TTW = 8;                 // 4 x 8 tiles of 4x4 outputs (D = 2: 2 x 4 cells of 8x8 pixels = 4 parity tiles each)
  static constexpr int TH = 4 * TTH, TW = 4 * TTW;       // 16 x 32 output pixels per workgroup
  static constexpr int BN = 64;                          // output channels per workgroup
  static constexpr int PH = TH + 2 * D, PWU = TW + 2 * D;   // staged patch: 18 rows of 34 pixels (D = 2: 20 x 36) ...
  // ... at a row pitch of 40 (ten 16-byte pieces) / 36 floats.  D = 2: 44 -- an eleventh, never-fetched piece per row: the lanes
  // of a window read are (px, py, cx, cy) = +1, +PW, +8, +8 PW floats apart, which at PW = 40 puts cy = 0 / 1 on the same bank
  // (8 x 40 = 5 x 64) and (cx = 3, py = 1) on (0, 0)'s: 55 % of that kernel's LDS cycles were conflicts; at 44 all 32 differ
  static constexpr int PW = VEC ? (D == 2 ? 44 : 40) : 36;
  static constexpr int PIECES_ROW = 10;                  // fetched pieces per row
  // VEC: piece j of a row holds columns ox0 - 4 + 4 j ..; the ring slot starts 4 bytes into its 16-byte unit, so that (D = 1)
  // the window column 0 (= image column ox0 - 1 + 4 tx) sits at float 4 + 4 tx of the row: 16-byte aligned ds_read_b128.
  // D = 2 windows are read as dword pairs two columns apart (no alignment to keep): column 0 = image column ox0 - 2 -> float 3
  static constexpr int XSHIFT = VEC ? 1 : 0, XWIN = VEC ? XSHIFT + 4 - D : 0;
  static constexpr int XPLANE = PH * PW;                 // 720 (800) / 648 floats per channel
  static constexpr int XSZ = 2 * XPLANE;                 // two channels per k-step
  static constexpr int XPIECES = XSZ / 4;                // VEC: 360 (400) pieces of 16 bytes = lanes of waves 0..5 (D = 2: 440 with the spare eleventh piece of each row, + 56 lanes of wave 6)
  static constexpr int XSZP = VEC ? (D == 1 ? 1600 : 1776) : 1536;   // ring slot
  static constexpr int WSZ = 2 * 36 * BN;                // 4608 floats of transformed weights per k-step: 2 x (8 waves x 1 KB) + 2 KB
  static constexpr int VSZ = 36 * 64;                    // transformed input of a k-step: [position][channel][tile]
  static constexpr int U_SLOTS = 4, R_SLOTS = 4, V_SLOTS = 4;   // four each: the loop is unrolled by four, every ring offset an immediate
  static constexpr int U0 = 0, R0 = U_SLOTS * WSZ, V0 = R0 + R_SLOTS * XSZP, RING = V0 + V_SLOTS * VSZ;
  static constexpr int EXCH = 2 * 4 * 8 * 8 * 64;        // epilogue exchange: [cout half][group][8 regs][8 values][lane]
  static constexpr size_t LDS_BYTES = (size_t)(RING > EXCH ? RING : EXCH) * sizeof(float);
  // LDS-DMA instructions per wave per k-step.  VEC: two 1 KB pieces of the weights + (waves 0..5) the patch piece or (waves
  // 6, 7) the last 2 KB of the weights (+ D = 2, wave 6: the patch pieces 384..399); else 2 + 1 dword of weights + 3 dwords of patch
  static constexpr int LOADS = VEC ? 3 : 6;
  static_assert(D == 1 || (D == 2 && VEC), "the dilated form exists for Win % 4 == 0 only");
  static_assert(WSZ == 2 * 2048 + 512 && (!VEC || (XPIECES <= 7 * 64 && XSHIFT + XPIECES * 4 <= XSZP && XSHIFT + 6 * 64 * 4 <= XSZP)),
                "staging plan");
  static_assert((R0 % 4) == 0 && (XSZP % 4) == 0 && (V0 % 4) == 0, "16-byte aligned LDS regions");
};
using W44 = W44T<true>;
using W44odd = W44T<false>;
using W44D2 = W44T<true, 2>;

#ifdef FDT_W44_STAMPS   // diagnostic build (tools/experiments/w44_stamps.sh): where does a workgroup's time go?  Constant-rate clock
__device__ long long g_w44_time[8];   // (100 MHz), summed over workgroups: [0] prologue [1] main loop [2] epilogue round 0 [3] round 1 [4] workgroups
                                      // [5] SHADER-clock cycles (s_memtime) of the main loop: [5] / [1] x 100 MHz = the clock the CUs held
#define W44_STAMP(i) if (threadIdx.x == 0) { const long long c_ = wall_clock64(); atomicAdd((unsigned long long*)&g_w44_time[i], (unsigned long long)(c_ - w44_t_)); w44_t_ = c_; \
    if (i == 0) w44_c_ = clock64(); if (i == 1) atomicAdd((unsigned long long*)&g_w44_time[5], (unsigned long long)(clock64() - w44_c_)); }
#else
#define W44_STAMP(i)
#endif

template <class T>
__global__ __launch_bounds__(512, 2) void conv_wino44_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef FDT_W44_STAMPS
  long long w44_t_ = wall_clock64(), w44_c_ = 0;
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pg = wave >> 1, h = wave & 1;                  // MFMA role: positions 9 pg .. 9 pg + 8, couts 32 h .. 32 h + 31
  const int half = lane >> 5, l31 = lane & 31;

  const int tiles_x = (a.Wout + T::TW - 1) / T::TW;
  FDT_BLOCK_MAP(a, sp_tile, n_tile);
  const int oy0 = (sp_tile / tiles_x) * T::TH;
  const int ox0 = (sp_tile % tiles_x) * T::TW;
  const int b = blockIdx.z / a.ksplit;
  const int ks = blockIdx.z - b * a.ksplit;

  const int HW = a.Hin * a.Win;                            // stride 1, pad 1: Hout == Hin
  const float* in_b = a.in + (long long)b * conv_in_bstride(a);
  const int nstages = (a.Cin + 1) / 2;
  const float* w_t = a.w + (long long)n_tile * nstages * T::WSZ;
  const int s_begin = (int)((long long)nstages * ks / a.ksplit);
  const int s_end = (int)((long long)nstages * (ks + 1) / a.ksplit);
  const int nst = s_end - s_begin;

  const float* zpad = g_zero_pad;
  asm volatile("" : "+s"(zpad));
  // ---- staging plan (the same for every k-step) -----------------------------------------------------------------------------
  // The matrix pipe and the vector ALU of a SIMD do not co-execute on this chip (SQ_VALU_MFMA_COEXEC_CYCLES = 0 for every conv
  // kernel here): each VALU instruction in the main loop is four cycles taken from the MFMAs.  So the loop carries NO address
  // arithmetic in vector registers: LDS offsets are immediates (rings of four slots, loop unrolled by four), and (VEC) both
  // operand streams are BUFFER loads -- a per-lane byte offset that never changes plus a scalar offset advanced per k-step; lanes
  // that stage padding / out-of-image / spare pieces carry an out-of-range offset and receive zeros from the bounds check (which
  // also zeroes the channel past an odd Cin).
  __amdgpu_buffer_rsrc_t wrs, xrs;
  unsigned xvo = 0x80000000u;                              // patch piece of this lane: byte offset in the image, or out of range
  const unsigned wvo = (unsigned)lane * 16u;
  int goff[3];
  unsigned okmask = 0;
  if constexpr (T::VEC) {
    wrs = __builtin_amdgcn_make_buffer_rsrc((void*)w_t, 0, 0x7fffffff, 0x00020000);
    xrs = __builtin_amdgcn_make_buffer_rsrc((void*)in_b, 0, a.Cin * HW * 4, 0x00020000);
    const int e = tid;                                     // waves 0..5: lane e < 360 owns the 16-byte piece (channel c, row yy, piece j)
    const int c = e / (T::XPLANE / 4);
    const int r = e - c * (T::XPLANE / 4);
    const int yy = r / (T::PW / 4), j = r - yy * (T::PW / 4);
    const int gy = oy0 - T::D + yy, gx = ox0 - 4 + 4 * j;
    if (e < 384 && e < T::XPIECES && j < T::PIECES_ROW && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win)
      xvo = (unsigned)(c * HW + gy * a.Win + gx) * 4u;
    if (T::D == 2 && wave == 6) {                          // pieces 384 .. 439: 56 lanes of wave 6, a fourth instruction
      const int e2 = 384 + lane;
      const int c2 = e2 / (T::XPLANE / 4);
      const int r2 = e2 - c2 * (T::XPLANE / 4);
      const int yy2 = r2 / (T::PW / 4), j2 = r2 - yy2 * (T::PW / 4);
      const int gy2 = oy0 - T::D + yy2, gx2 = ox0 - 4 + 4 * j2;
      if (e2 < T::XPIECES && j2 < T::PIECES_ROW && gy2 >= 0 && gy2 < a.Hin && gx2 >= 0 && gx2 < a.Win)
        xvo = (unsigned)(c2 * HW + gy2 * a.Win + gx2) * 4u;
    }
    // The second channel of the last k-step does not exist when Cin is odd: its offset (scalar + per-lane) is >= num_records,
    // and the bounds check of a raw buffer load on gfx950 includes the scalar offset (tools/microbench/buffer_oob_probe.hip,
    // profiles/r03/buffer_oob_probe.txt; pinned by tests/test_gpu_conv.py::test_channels_past_cin_read_as_zero): zeros.
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int e = tid + 512 * k;
      const int c = e / T::XPLANE;
      const int r = e - c * T::XPLANE;
      const int yy = r / T::PW, xx = r - yy * T::PW;
      const int gy = oy0 - 1 + yy, gx = ox0 - 1 + xx;
      const bool ok = (e < T::XSZ) && xx < T::PWU && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
      goff[k] = ok ? (c * HW + gy * a.Win + gx) : 0;
      if (ok) okmask |= (1u << k) | ((unsigned)c << (4 + k));   // bit k: in image; bit 4 + k: which of the two channels
    }
  }

  // LDS-DMA of one k-step into compile-time ring slots: weights U(su) -> slot US, raw patch R(sr) -> slot RS.  k-steps past the
  // end of this workgroup's share are clamped to its last one (valid addresses, the same instruction count; never consumed).
  auto issue_u = [&](auto tail_c, auto us_c, int su) {
    constexpr bool TAILW = decltype(tail_c)::value;     // waves 6, 7 also fetch the last 2 KB of the weights
    constexpr int US = decltype(us_c)::value;
    const int suc = su < nst ? su : nst - 1;
    float* U_ = smem + T::U0 + US * T::WSZ;
    if constexpr (T::VEC) {
      const unsigned ub = (unsigned)((s_begin + suc) * T::WSZ) * 4u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lptr_t)(U_ + wave * 256), 16, wvo, ub + (unsigned)wave * 1024u, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lptr_t)(U_ + 2048 + wave * 256), 16, wvo, ub + 8192u + (unsigned)wave * 1024u, 0, 0);
      if constexpr (TAILW)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lptr_t)(U_ + 4096 + (wave - 6) * 256), 16, wvo,
                                                 ub + 16384u + (unsigned)(wave - 6) * 1024u, 0, 0);
    } else {
      const float* usrc = w_t + (long long)(s_begin + suc) * T::WSZ;
      glds16(usrc + wave * 256 + lane * 4, U_ + wave * 256);
      glds16(usrc + 2048 + wave * 256 + lane * 4, U_ + 2048 + wave * 256);
      glds4(usrc + 4096 + wave * 64 + lane, U_ + 4096 + wave * 64);
    }
  };
  auto issue_r = [&](auto tail_c, auto extra_c, auto rs_c, int sr) {
    constexpr bool TAILW = decltype(tail_c)::value;     // waves 6, 7 stage no patch pieces (but wave 6 the last 56 at D = 2)
    constexpr int RS = decltype(rs_c)::value;
    const int src_ = sr < nst ? sr : nst - 1;
    if constexpr (T::VEC) {
      const unsigned xb = (unsigned)(s_begin + src_) * 2u * (unsigned)HW * 4u;
      if constexpr (!TAILW) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lptr_t)(smem + T::R0 + RS * T::XSZP + T::XSHIFT + wave * 256), 16, xvo, xb, 0, 0);
      } else if constexpr (decltype(extra_c)::value) {
        if (lane < T::XPIECES - 384)   // ONLY those lanes: an LDS-DMA lane writes its 16 bytes wherever it points -- the rest would land in the next ring slot
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lptr_t)(smem + T::R0 + RS * T::XSZP + T::XSHIFT + 6 * 256), 16, xvo, xb, 0, 0);
      }
    } else {
      const int c0 = (s_begin + src_) * 2;
      const float* src = in_b + (long long)c0 * HW;
      const int crem = a.Cin - c0;
      float* R_ = smem + T::R0 + RS * T::XSZP + wave * 64;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const bool ok = ((okmask >> k) & 1u) && (int)((okmask >> (4 + k)) & 1u) < crem;
        glds4(ok ? src + goff[k] : zpad, R_ + 512 * k);
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  // transform role: lane = (channel = half, tile = l31); the tile's 6x6 window starts at (4 ty, 4 tx) of the staged patch
  // D = 1: tile (ty, tx) of the 4 x 8 grid, outputs (4 ty + u, 4 tx + v);  D = 2: cell (cy, cx) of the 2 x 4 grid of 8x8-pixel
  // cells, parity (py, px): outputs (8 cy + py + 2 u, 8 cx + px + 2 v), window = every second pixel from (8 cy + py, 8 cx + px)
  const int ty = T::D == 1 ? (l31 >> 3) : (l31 >> 4), tx = T::D == 1 ? (l31 & 7) : ((l31 >> 2) & 3);
  const int py = (l31 >> 1) & 1, px = l31 & 1;
  const int oyl = T::D == 1 ? 4 * ty : 8 * ty + py, oxl = T::D == 1 ? 4 * tx : 8 * tx + px;     // first output pixel of the tile
  const unsigned xbase = lds0 + (unsigned)(T::R0 + T::XWIN + half * T::XPLANE + oyl * T::PW + oxl) * 4u;
  const unsigned vwbase = lds0 + (unsigned)(T::V0 + lane) * 4u;
  // MFMA role
  const unsigned abase = lds0 + (unsigned)(T::U0 + half * 36 * T::BN + pg * 9 * T::BN + h * 32 + l31) * 4u;
  const unsigned bbase = lds0 + (unsigned)(T::V0 + pg * 9 * 64 + lane) * 4u;

  struct Ops {
    f32x2 a[4], b[4];
    float a8, b8;
  };
  // operands of one k-step from compile-time ring slots: every offset is an immediate
  auto load_ops = [&](Ops& o, auto us_c, auto vs_c) {
    constexpr int UO = decltype(us_c)::value * (T::WSZ / 64), VO = decltype(vs_c)::value * (T::VSZ / 64);   // units of 64 dwords
    lds_read2st64_b32<UO + 0, UO + 1>(o.a[0], abase);
    lds_read2st64_b32<VO + 0, VO + 1>(o.b[0], bbase);
    lds_read2st64_b32<UO + 2, UO + 3>(o.a[1], abase);
    lds_read2st64_b32<VO + 2, VO + 3>(o.b[1], bbase);
    lds_read2st64_b32<UO + 4, UO + 5>(o.a[2], abase);
    lds_read2st64_b32<VO + 4, VO + 5>(o.b[2], bbase);
    lds_read2st64_b32<UO + 6, UO + 7>(o.a[3], abase);
    lds_read2st64_b32<VO + 6, VO + 7>(o.b[3], bbase);
    lds_read_b32<(UO + 8) * 256>(o.a8, abase);
    lds_read_b32<(VO + 8) * 256>(o.b8, bbase);
  };
  static_assert(3 * (T::WSZ / 64) + 8 < 256 && (3 * (T::WSZ / 64) + 8) * 256 < 65536, "ds offset fields");
  auto wait_ops = [&](Ops& o, auto newer_c) {
    constexpr int N_ = decltype(newer_c)::value;
    asm volatile("s_waitcnt lgkmcnt(%10)"
                 : "+v"(o.a[0]), "+v"(o.a[1]), "+v"(o.a[2]), "+v"(o.a[3]), "+v"(o.b[0]), "+v"(o.b[1]), "+v"(o.b[2]), "+v"(o.b[3]),
                   "+v"(o.a8), "+v"(o.b8)
                 : "n"(N_));
  };
  auto mfma_q = [&](const Ops& o, auto qc) {
    constexpr int q = decltype(qc)::value;
    if constexpr (q < 8)
      acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[q >> 1][q & 1], o.b[q >> 1][q & 1], acc[q], 0, 0, 0);
    else
      acc[8] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a8, o.b8, acc[8], 0, 0, 0);
  };

  // One row of V = B^T d B for this lane's (channel, tile):  B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0;
  //                                                                0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
  struct Raw {
    f32x4 lo[4];      // D = 1: columns 0..3 of the window rows this role needs
    f32x2 hi[4];      //        columns 4, 5
    f32x2 pr[4][3];   // D = 2: column pairs (0,1) (2,3) (4,5), read as dword pairs two image columns apart
  };
  const f32x2 k4 = {4.0f, 4.0f}, km5 = {-5.0f, -5.0f}, km4 = {-4.0f, -4.0f}, k2 = {2.0f, 2.0f}, km2 = {-2.0f, -2.0f};

  auto main_loop = [&](auto role_c) {
    constexpr int ROLE = decltype(role_c)::value;                  // 0..5: row of B^T d this wave produces; 6: none
    constexpr int R0_ = (ROLE == 0) ? 0 : 1;                       // first window row needed
    constexpr int RSTEP = (ROLE == 0 || ROLE == 5) ? 2 : 1;        // rows 0,2,4 / 1,2,3,4 / 1,3,5
    constexpr int NROW = (ROLE == 0 || ROLE == 5) ? 3 : 4;
    constexpr int NRAW = (ROLE < 6) ? (T::D == 2 ? 3 : 2) * NROW : 0;
    auto raw_reads = [&](Raw& w, auto rs_c) {
      constexpr int SO = decltype(rs_c)::value * T::XSZP * 4;       // byte offset of the ring slot
      if constexpr (ROLE < 6 && T::D == 2) {
        // window row i = patch row 2 i, column c = patch column 2 c: ds_read2_b32 (8-bit dword offsets, two columns = 4 dwords
        // apart per pair) from a per-k-step base = ring slot + first row (one v_add; a second one for the rows beyond 255 dwords)
        constexpr int ROWD = 2 * T::PW;                             // dwords between window rows
        const unsigned xa = xbase + (unsigned)(SO + R0_ * ROWD * 4);
        const unsigned xb2 = xa + (unsigned)(2 * RSTEP * ROWD * 4);
        static_assert(RSTEP * ROWD + 10 < 256, "ds_read2_b32 offsets");
#pragma unroll
        for (int i = 0; i < NROW; ++i) {
          const unsigned ad = i < 2 ? xa : xb2;
          if (i == 0 || i == 2) {
            lds_read2_b32<0, 2>(w.pr[i][0], ad);
            lds_read2_b32<4, 6>(w.pr[i][1], ad);
            lds_read2_b32<8, 10>(w.pr[i][2], ad);
          } else {
            lds_read2_b32<RSTEP * ROWD + 0, RSTEP * ROWD + 2>(w.pr[i][0], ad);
            lds_read2_b32<RSTEP * ROWD + 4, RSTEP * ROWD + 6>(w.pr[i][1], ad);
            lds_read2_b32<RSTEP * ROWD + 8, RSTEP * ROWD + 10>(w.pr[i][2], ad);
          }
        }
      } else if constexpr (ROLE < 6) {
        w44_read_b128<SO + (R0_ + 0 * RSTEP) * T::PW * 4>(w.lo[0], xbase);
        w44_read_b64<SO + ((R0_ + 0 * RSTEP) * T::PW + 4) * 4>(w.hi[0], xbase);
        w44_read_b128<SO + (R0_ + 1 * RSTEP) * T::PW * 4>(w.lo[1], xbase);
        w44_read_b64<SO + ((R0_ + 1 * RSTEP) * T::PW + 4) * 4>(w.hi[1], xbase);
        w44_read_b128<SO + (R0_ + 2 * RSTEP) * T::PW * 4>(w.lo[2], xbase);
        w44_read_b64<SO + ((R0_ + 2 * RSTEP) * T::PW + 4) * 4>(w.hi[2], xbase);
        if constexpr (NROW == 4) {
          w44_read_b128<SO + (R0_ + 3 * RSTEP) * T::PW * 4>(w.lo[3], xbase);
          w44_read_b64<SO + ((R0_ + 3 * RSTEP) * T::PW + 4) * 4>(w.hi[3], xbase);
        }
      }
    };
    auto wait_raw = [&](Raw& w, auto newer_c) {
      constexpr int N_ = decltype(newer_c)::value;
      if constexpr (ROLE < 6 && T::D == 2) {
        if constexpr (NROW == 4)
          asm volatile("s_waitcnt lgkmcnt(%12)"
                       : "+v"(w.pr[0][0]), "+v"(w.pr[0][1]), "+v"(w.pr[0][2]), "+v"(w.pr[1][0]), "+v"(w.pr[1][1]), "+v"(w.pr[1][2]),
                         "+v"(w.pr[2][0]), "+v"(w.pr[2][1]), "+v"(w.pr[2][2]), "+v"(w.pr[3][0]), "+v"(w.pr[3][1]), "+v"(w.pr[3][2])
                       : "n"(N_));
        else
          asm volatile("s_waitcnt lgkmcnt(%9)"
                       : "+v"(w.pr[0][0]), "+v"(w.pr[0][1]), "+v"(w.pr[0][2]), "+v"(w.pr[1][0]), "+v"(w.pr[1][1]), "+v"(w.pr[1][2]),
                         "+v"(w.pr[2][0]), "+v"(w.pr[2][1]), "+v"(w.pr[2][2])
                       : "n"(N_));
      } else if constexpr (ROLE < 6) {
        if constexpr (NROW == 4)
          asm volatile("s_waitcnt lgkmcnt(%8)"
                       : "+v"(w.lo[0]), "+v"(w.lo[1]), "+v"(w.lo[2]), "+v"(w.lo[3]), "+v"(w.hi[0]), "+v"(w.hi[1]), "+v"(w.hi[2]),
                         "+v"(w.hi[3])
                       : "n"(N_));
        else
          asm volatile("s_waitcnt lgkmcnt(%6)"
                       : "+v"(w.lo[0]), "+v"(w.lo[1]), "+v"(w.lo[2]), "+v"(w.hi[0]), "+v"(w.hi[1]), "+v"(w.hi[2])
                       : "n"(N_));
      }
    };
    // The row transform on PACKED f32 (v_pk_fma_f32 / v_pk_add_f32: two columns per instruction).  t_pairs: T[k] = the pair of
    // columns (2k, 2k + 1) of row ROLE of B^T d;  v_all: v = B^T t for all six columns.
    auto t_pairs = [&](const Raw& w, f32x2 (&Tp)[3]) {
      if constexpr (ROLE < 6) {
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
          f32x2 d[4];
#pragma unroll
          for (int i = 0; i < NROW; ++i) {
            if constexpr (T::D == 2) d[i] = w.pr[i][pr];
            else d[i] = pr == 0 ? __builtin_shufflevector(w.lo[i], w.lo[i], 0, 1) : pr == 1 ? __builtin_shufflevector(w.lo[i], w.lo[i], 2, 3) : w.hi[i];
          }
          if constexpr (ROLE == 0 || ROLE == 5)
            Tp[pr] = pk_fma(k4, d[0], pk_fma(km5, d[1], d[2]));              // 4 d0 - 5 d2 + d4   /   4 d1 - 5 d3 + d5
          else if constexpr (ROLE == 1)
            Tp[pr] = pk_add(pk_fma(km4, d[1], d[3]), pk_fma(km4, d[0], d[2]));   // (d4 - 4 d2) + (d3 - 4 d1)
          else if constexpr (ROLE == 2)
            Tp[pr] = pk_sub(pk_fma(km4, d[1], d[3]), pk_fma(km4, d[0], d[2]));
          else if constexpr (ROLE == 3)
            Tp[pr] = pk_fma(k2, pk_sub(d[2], d[0]), pk_sub(d[3], d[1]));     // (d4 - d2) + 2 (d3 - d1)
          else
            Tp[pr] = pk_fma(km2, pk_sub(d[2], d[0]), pk_sub(d[3], d[1]));
        }
      }
    };
    auto v_all = [&](const f32x2 (&Tp)[3], float (&v)[6]) {          // Tp = (t0,t1), (t2,t3), (t4,t5)
      if constexpr (ROLE < 6) {
        const f32x2 v05 = pk_fma(k4, Tp[0], pk_fma(km5, Tp[1], Tp[2]));   // (v0, v5)
        const f32x2 pq = pk_fma(km4, Tp[1], Tp[2]);                       // .x = t4 - 4 t2
        const float q = fmaf(-4.0f, Tp[0][1], Tp[1][1]);                  // t3 - 4 t1
        const f32x2 cd = pk_sub(Tp[2], Tp[1]);                            // .x = t4 - t2
        const float d = Tp[1][1] - Tp[0][1];                              // t3 - t1
        v[0] = v05[0];
        v[1] = pq[0] + q;
        v[2] = pq[0] - q;
        v[3] = fmaf(2.0f, d, cd[0]);
        v[4] = fmaf(-2.0f, d, cd[0]);
        v[5] = v05[1];
      }
    };
    auto v_store = [&](const float (&v)[6], auto vs_c) {
      constexpr int VO = decltype(vs_c)::value * (T::VSZ / 64) + ROLE * 6;
      if constexpr (ROLE < 6) {
        w44_write2st64_b32<VO + 0, VO + 1>(vwbase, v[0], v[1]);
        w44_write2st64_b32<VO + 2, VO + 3>(vwbase, v[2], v[3]);
        w44_write2st64_b32<VO + 4, VO + 5>(vwbase, v[4], v[5]);
      }
    };
    auto transform_store = [&](const Raw& w, auto vs_c) {
      f32x2 Tp[3];
      float v[6];
      t_pairs(w, Tp);
      v_all(Tp, v);
      v_store(v, vs_c);
    };
    using TAILc = std::bool_constant<ROLE >= 6>;          // waves 6, 7
    using EXTRAc = std::bool_constant<ROLE == 6 && T::D == 2>;
    using N0 = std::integral_constant<int, 0>;
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;

    // ---- prologue: U(0..2), R(0..4) landed; V(0..2) written; operands of k-step 0 in registers
    issue_u(TAILc{}, S0{}, 0);
    issue_u(TAILc{}, S1{}, 1);
    issue_u(TAILc{}, S2{}, 2);
    issue_r(TAILc{}, EXTRAc{}, S0{}, 0);
    issue_r(TAILc{}, EXTRAc{}, S1{}, 1);
    issue_r(TAILc{}, EXTRAc{}, S2{}, 2);
    issue_r(TAILc{}, EXTRAc{}, S3{}, 3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    {
      Raw w0, w1;
      raw_reads(w0, S0{});
      raw_reads(w1, S1{});
      wait_raw(w0, std::integral_constant<int, NRAW>{});
      transform_store(w0, S0{});
      wait_raw(w1, std::integral_constant<int, (ROLE < 6 ? 3 : 0)>{});
      transform_store(w1, S1{});
      raw_reads(w0, S2{});
      wait_raw(w0, N0{});
      transform_store(w0, S2{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();          // V(0..2) visible; every wave is done with R(0..2)
    __builtin_amdgcn_sched_barrier(0);
    issue_r(TAILc{}, EXTRAc{}, S0{}, 4);   // R(4) into the slot of R(0)
    Ops X, Y;
    load_ops(X, S0{}, S0{});
    wait_ops(X, N0{});

#ifndef FDT_W44_EXP
#define FDT_W44_EXP 0     // timing experiments (tools/experiments/w44_variants.sh), a bit mask: 1 no input transform, 2 no barrier,
#endif                    // 4 no LDS-DMA in the loop (32: no patch DMA only, 64: no weight DMA only), 8 no MFMA, 16 no operand reads --
                          // results are wrong for every value but 0
    auto mf = [&](Ops& cur, auto qc) {
      if (!(FDT_W44_EXP & 8)) mfma_q(cur, qc);
      __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>;
    using I7 = std::integral_constant<int, 7>;
    using I8 = std::integral_constant<int, 8>;
    // ---- super-step: the k-steps s = 2 S (ring slots M) and s + 1 (slots M + 1) between TWO barriers ---------------------
    // At its barrier: the operands of k-step s sit in `cur`; U(s+1), U(s+2), R(s+3), R(s+4) have landed (all LDS-DMA of the
    // previous super-step: vmcnt(0), one super-step = two k-steps of flight); V(s+1), V(s+2) are written.  First half: MFMAs
    // of s, window of s+3 -> V(s+3), operands of s+1 -> `nxt`, LDS-DMA of U(s+3) / R(s+5); second half: MFMAs of s+1, window of
    // s+4 -> V(s+4), operands of s+2 -> `cur`, LDS-DMA of U(s+4) / R(s+6).  The first two MFMAs stand in front of the barrier
    // (their operands are registers), four MFMAs between a window read and its wait, the operand prefetch late.
    auto super_step = [&](auto m_c, Ops& cur, Ops& nxt, int s) {
      constexpr int M = decltype(m_c)::value;
      using M0 = std::integral_constant<int, M>;
      using M1 = std::integral_constant<int, (M + 1) & 3>;
      using M2 = std::integral_constant<int, (M + 2) & 3>;
      using M3 = std::integral_constant<int, (M + 3) & 3>;
      mf(cur, I0{});
      mf(cur, I1{});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(FDT_W44_EXP & 2)) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      Raw w;
      f32x2 Tp[3];
      float v[6];
      if (!(FDT_W44_EXP & 1)) raw_reads(w, M3{});
      __builtin_amdgcn_sched_barrier(0);
      mf(cur, I2{});
      if (!(FDT_W44_EXP & 4)) {
        if (!(FDT_W44_EXP & 64)) issue_u(TAILc{}, M3{}, s + 3);
        if (!(FDT_W44_EXP & 32)) issue_r(TAILc{}, EXTRAc{}, M1{}, s + 5);
      }
      __builtin_amdgcn_sched_barrier(0);
      mf(cur, I3{});
      mf(cur, I4{});
      if (!(FDT_W44_EXP & 1)) {
        wait_raw(w, N0{});
        t_pairs(w, Tp);
      }
      __builtin_amdgcn_sched_barrier(0);
      mf(cur, I5{});
      if (!(FDT_W44_EXP & 1)) v_all(Tp, v);
      __builtin_amdgcn_sched_barrier(0);
      mf(cur, I6{});
      if (!(FDT_W44_EXP & 1)) v_store(v, M3{});
      if (!(FDT_W44_EXP & 16)) load_ops(nxt, M1{}, M1{});
      __builtin_amdgcn_sched_barrier(0);
      mf(cur, I7{});
      mf(cur, I8{});
      if (!(FDT_W44_EXP & 16)) wait_ops(nxt, N0{});
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // ---- second half: k-step s + 1
      if (!(FDT_W44_EXP & 1)) raw_reads(w, M0{});
      __builtin_amdgcn_sched_barrier(0);
      mf(nxt, I0{});
      mf(nxt, I1{});
      mf(nxt, I2{});
      if (!(FDT_W44_EXP & 4)) {
        if (!(FDT_W44_EXP & 64)) issue_u(TAILc{}, M0{}, s + 4);
        if (!(FDT_W44_EXP & 32)) issue_r(TAILc{}, EXTRAc{}, M2{}, s + 6);
      }
      __builtin_amdgcn_sched_barrier(0);
      mf(nxt, I3{});
      mf(nxt, I4{});
      if (!(FDT_W44_EXP & 1)) {
        wait_raw(w, N0{});
        t_pairs(w, Tp);
      }
      __builtin_amdgcn_sched_barrier(0);
      mf(nxt, I5{});
      if (!(FDT_W44_EXP & 1)) v_all(Tp, v);
      __builtin_amdgcn_sched_barrier(0);
      mf(nxt, I6{});
      if (!(FDT_W44_EXP & 1)) v_store(v, M0{});
      if (!(FDT_W44_EXP & 16)) load_ops(cur, M2{}, M2{});
      __builtin_amdgcn_sched_barrier(0);
      mf(nxt, I7{});
      mf(nxt, I8{});
      if (!(FDT_W44_EXP & 16)) wait_ops(cur, N0{});                       // also: this wave's V writes are done
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    W44_STAMP(0)
    int s = 0;
    for (; s + 3 < nst; s += 4) {
      super_step(S0{}, X, Y, s);
      super_step(S2{}, X, Y, s + 2);
    }
    if (s + 1 < nst) {
      super_step(S0{}, X, Y, s);
      s += 2;
    }
    if (s < nst) {                          // an odd last k-step: its operands are in X, nothing left to prefetch
      mf(X, I0{}); mf(X, I1{}); mf(X, I2{}); mf(X, I3{}); mf(X, I4{}); mf(X, I5{}); mf(X, I6{}); mf(X, I7{}); mf(X, I8{});
    }
  };
  switch (wave) {
    case 0: main_loop(std::integral_constant<int, 0>{}); break;
    case 1: main_loop(std::integral_constant<int, 1>{}); break;
    case 2: main_loop(std::integral_constant<int, 2>{}); break;
    case 3: main_loop(std::integral_constant<int, 3>{}); break;
    case 4: main_loop(std::integral_constant<int, 4>{}); break;
    case 5: main_loop(std::integral_constant<int, 5>{}); break;
    case 6: main_loop(std::integral_constant<int, 6>{}); break;
    default: main_loop(std::integral_constant<int, 7>{}); break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the (clamped) LDS-DMA of the k-steps past the end must not land in the exchange buffer
  W44_STAMP(1)

  // ---- output transform Y = A^T M A,  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1] -----------------
  // This wave holds positions 9 pg .. 9 pg + 8 of M, i.e. (pg even) row 3 pg / 2 whole + columns 0..2 of the next row, or
  // (pg odd) columns 3..5 of row (3 pg - 1) / 2 + the next row whole.  Per accumulator element it forms the two 4-vectors
  // (M A)[row] it contributes to (vA: first row touched, vB: second), and the groups exchange them through LDS.
  auto full_row = [](float m0, float m1, float m2, float m3, float m4, float m5, float (&r)[4]) {
    const float s1 = m1 + m2, d1 = m1 - m2, s2 = m3 + m4, d2 = m3 - m4;
    r[0] = m0 + s1 + s2;
    r[1] = fmaf(2.0f, d2, d1);
    r[2] = fmaf(4.0f, s2, s1);
    r[3] = fmaf(8.0f, d2, d1) + m5;
  };
  auto lo_row = [](float m0, float m1, float m2, float (&r)[4]) {     // columns 0..2
    r[0] = m0 + m1 + m2;
    r[1] = m1 - m2;
    r[2] = m1 + m2;
    r[3] = m1 - m2;
  };
  auto hi_row = [](float m3, float m4, float m5, float (&r)[4]) {     // columns 3..5
    const float s2 = m3 + m4, d2 = m3 - m4;
    r[0] = s2;
    r[1] = 2.0f * d2;
    r[2] = 4.0f * s2;
    r[3] = fmaf(8.0f, d2, m5);
  };
  const int HWo = a.Hout * a.Wout;
  const bool raw = a.ws != nullptr;
  const bool wt = raw && a.sk_count;   // slabs of an in-kernel combine are stored write-through (conv.h)
  float* dst_b = raw ? a.ws + ((long long)(b * a.ksplit + ks) * a.Cout) * HWo
                     : a.out + ((long long)b * a.out_ctot + a.out_coff) * HWo;
  const float* res_b = (!raw && a.res) ? a.res + ((long long)b * a.res_ctot + a.res_coff) * HWo : nullptr;
  const int oy = oy0 + oyl, ox = ox0 + oxl;
  const bool vec4 = (T::D == 1) && (a.Wout % 4 == 0);      // D = 2: the tile's pixels are two apart, scalar stores
  float* E = smem;
  for (int round = 0; round < 2; ++round) {
    __syncthreads();                       // ring (round 0) / previous round's exchange data is dead
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      float vA[4], vB[4];
      // (static indexing of acc: both rounds are spelled out through the ternaries on a compile-time rr)
      const float m0 = round ? acc[0][8 + rr] : acc[0][rr], m1 = round ? acc[1][8 + rr] : acc[1][rr];
      const float m2 = round ? acc[2][8 + rr] : acc[2][rr], m3 = round ? acc[3][8 + rr] : acc[3][rr];
      const float m4 = round ? acc[4][8 + rr] : acc[4][rr], m5 = round ? acc[5][8 + rr] : acc[5][rr];
      const float m6 = round ? acc[6][8 + rr] : acc[6][rr], m7 = round ? acc[7][8 + rr] : acc[7][rr];
      const float m8 = round ? acc[8][8 + rr] : acc[8][rr];
      if (pg & 1) {
        hi_row(m0, m1, m2, vA);
        full_row(m3, m4, m5, m6, m7, m8, vB);
      } else {
        full_row(m0, m1, m2, m3, m4, m5, vA);
        lo_row(m6, m7, m8, vB);
      }
      float* e = E + (((h * 4 + pg) * 8 + rr) * 8) * 64 + lane;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        e[v * 64] = vA[v];
        e[(4 + v) * 64] = vB[v];
      }
    }
    __syncthreads();
    // this wave finishes registers r = 8 round + 2 pg + {0, 1} of its cout half: whole 4x4 tiles
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rr = 2 * pg + j;
      const int r = 8 * round + rr;
      float R[6][4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        auto at = [&](int src, int val) { return E[(((h * 4 + src) * 8 + rr) * 8 + val) * 64 + lane]; };
        R[0][v] = at(0, v);
        R[1][v] = at(0, 4 + v) + at(1, v);
        R[2][v] = at(1, 4 + v);
        R[3][v] = at(2, v);
        R[4][v] = at(2, 4 + v) + at(3, v);
        R[5][v] = at(3, 4 + v);
      }
      const int co = n_tile * T::BN + h * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co >= a.Cout || ox >= a.Wout) continue;
      const float bv = (!raw && a.bias) ? a.bias[co] : 0.0f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (oy + T::D * u >= a.Hout) continue;
        float y[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const float s1 = R[1][v] + R[2][v], d1 = R[1][v] - R[2][v], s2 = R[3][v] + R[4][v], d2 = R[3][v] - R[4][v];
          y[v] = u == 0 ? R[0][v] + s1 + s2 : u == 1 ? fmaf(2.0f, d2, d1) : u == 2 ? fmaf(4.0f, s2, s1) : fmaf(8.0f, d2, d1) + R[5][v];
        }
        const long long off = (long long)co * HWo + (long long)(oy + T::D * u) * a.Wout + ox;
        if constexpr (T::D == 2) {
          // The lanes px = 0 / 1 of a cell row hold alternating pixels (8 cx + px + 2 v): one exchange between neighbouring
          // lanes (DPP quad permutation) gives every lane four CONSECUTIVE pixels -- the even lane 8 cx .. + 3, the odd lane
          // 8 cx + 4 .. + 7 -- and the row leaves as 16-byte stores like in the undilated form (Wout % 4 == 0 here).
          const float s0 = px ? y[0] : y[2], s1 = px ? y[1] : y[3];
          const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, true));
          const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s1), 0xB1, 0xF, 0xF, true));
          float4 o = px ? make_float4(r0, y[2], r1, y[3]) : make_float4(y[0], r0, y[1], r1);
          const int xs = ox - px + 4 * px;                  // 8 cx (+ 4 for the odd lane)
          const long long off4 = (long long)co * HWo + (long long)(oy + 2 * u) * a.Wout + xs;
          if (xs < a.Wout) {
            if (!raw) {
              o.x += bv; o.y += bv; o.z += bv; o.w += bv;
              if (res_b) {
                const float4 rv = *reinterpret_cast<const float4*>(res_b + off4);
                o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
              }
              if (a.act == ACT_RELU) {
                o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
              } else if (a.act == ACT_RELU6) {
                o.x = fminf(fmaxf(o.x, 0.f), 6.f); o.y = fminf(fmaxf(o.y, 0.f), 6.f);
                o.z = fminf(fmaxf(o.z, 0.f), 6.f); o.w = fminf(fmaxf(o.w, 0.f), 6.f);
              }
            }
            slab_store4(dst_b + off4, o.x, o.y, o.z, o.w, wt);
          }
        } else if (vec4) {
          float4 o = make_float4(y[0], y[1], y[2], y[3]);
          if (!raw) {
            o.x += bv; o.y += bv; o.z += bv; o.w += bv;
            if (res_b) {
              const float4 rv = *reinterpret_cast<const float4*>(res_b + off);
              o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
            }
            if (a.act == ACT_RELU) {
              o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
            } else if (a.act == ACT_RELU6) {
              o.x = fminf(fmaxf(o.x, 0.f), 6.f); o.y = fminf(fmaxf(o.y, 0.f), 6.f);
              o.z = fminf(fmaxf(o.z, 0.f), 6.f); o.w = fminf(fmaxf(o.w, 0.f), 6.f);
            }
          }
          slab_store4(dst_b + off, o.x, o.y, o.z, o.w, wt);
        } else {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            if (ox + T::D * v >= a.Wout) continue;
            float o = y[v];
            if (!raw) {
              o += bv;
              if (res_b) o += res_b[off + T::D * v];
              if (a.act == ACT_RELU) o = fmaxf(o, 0.f);
              else if (a.act == ACT_RELU6) o = fminf(fmaxf(o, 0.f), 6.f);
            }
            slab_store1(dst_b + off + T::D * v, o, wt);
          }
        }
      }
    }
    W44_STAMP(2 + round)
  }
  if (wt) splitk_combine_tile<512>(a, b, sp_tile + a.n_sp * n_tile, n_tile * T::BN, T::BN, oy0, ox0, T::TH, T::TW, (unsigned*)smem);
#ifdef FDT_W44_STAMPS
  if (threadIdx.x == 0) atomicAdd((unsigned long long*)&g_w44_time[4], 1ull);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Twelve-wave form (three waves per SIMD).  The eight-wave kernel above measured as the SUM of its matrix time and its
// non-matrix time per k-step (tools/experiments/w44_variants.sh: MFMAs alone 166 us, everything else alone 152 us, together
// 288 us on 256 -> 256 @ 256x256): a wave that has issued an MFMA does not get to its other work while the pipe is busy, and
// with two waves per SIMD nobody else does either.  Here a wave owns HALF A ROW of the 6x6 position grid (three positions) for
// all 64 couts of the workgroup -- six accumulator tiles, 96 registers, so three waves fit a SIMD -- and
//  * forms its own three B operands in registers from the raw window (lane = (channel, tile) is the MFMA B layout): the V
//    buffer, its LDS writes and the B-operand reads are gone; a B operand feeds two MFMAs (both cout halves);
//  * every wave does the same work per k-step: 6-8 window reads, ~20 VALU of transform, 4 weight reads, 2 LDS-DMA
//    instructions, 6 MFMAs -- no idle roles, no two-wave imbalance per SIMD;
//  * one barrier per k-step as before: at the barrier of k-step s the weights U(s+1) and the patch R(s+1) have landed (rings
//    of three slots, issued two k-steps ahead), A(s) and B(s) sit in registers; k-step s runs its six MFMAs while it reads
//    the window of s+1, transforms it into B(s+1) and prefetches A(s+1).
template <class T>
__global__ __launch_bounds__(768, 3) void conv_wino44b_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(T::VEC, "the twelve-wave form stages the patch as 16-byte pieces (Win % 4 == 0)");
  constexpr int U0 = 0, US = 3, R0 = US * T::WSZ, RS = 3;                   // rings of three slots
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                // 0..11
  const int half = lane >> 5, l31 = lane & 31;

  const int tiles_x = (a.Wout + T::TW - 1) / T::TW;
  FDT_BLOCK_MAP(a, sp_tile, n_tile);
  const int oy0 = (sp_tile / tiles_x) * T::TH;
  const int ox0 = (sp_tile % tiles_x) * T::TW;
  const int b = blockIdx.z / a.ksplit;
  const int ks = blockIdx.z - b * a.ksplit;

  const int HW = a.Hin * a.Win;
  const float* in_b = a.in + (long long)b * conv_in_bstride(a);
  const int nstages = (a.Cin + 1) / 2;
  const float* w_t = a.w + (long long)n_tile * nstages * T::WSZ;
  const int s_begin = (int)((long long)nstages * ks / a.ksplit);
  const int s_end = (int)((long long)nstages * (ks + 1) / a.ksplit);
  const int nst = s_end - s_begin;

  const float* zpad = g_zero_pad;
  asm volatile("" : "+s"(zpad));
  // patch staging: waves 6..11, lane e = 64 (wave - 6) + lane < 360 owns one 16-byte piece (see the eight-wave kernel)
  const float* rp = zpad;
  unsigned rstep = 0;
  bool rch1 = false;
  if (wave >= 6) {
    const int e = tid - 384;
    const int c = e / (T::XPLANE / 4);
    const int r = e - c * (T::XPLANE / 4);
    const int yy = r / (T::PW / 4), j = r - yy * (T::PW / 4);
    const int gy = oy0 - 1 + yy, gx = ox0 - 4 + 4 * j;
    if (e < T::XPIECES && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) {
      rp = in_b + (long long)(2 * s_begin + c) * HW + gy * a.Win + gx;
      rstep = 2u * (unsigned)HW * 4u;
      rch1 = c == 1;
    }
  }
  const bool cin_odd = (a.Cin & 1) != 0;
  // LDS-DMA of k-step st (clamped to the last one of this workgroup's share) into ring slot `slot`: 18 KB of weights as 18
  // pieces of 1 KB (waves 0..11 one each, waves 0..5 a second one) + the patch (waves 6..11): two instructions per wave
  auto issue = [&](auto hi_c, int st, int slot) {
    constexpr bool HI = decltype(hi_c)::value;                              // waves 6..11
    const int sc = st < nst ? st : nst - 1;
    const float* usrc = w_t + (long long)(s_begin + sc) * T::WSZ;
    float* U_ = smem + U0 + slot * T::WSZ;
    glds16(usrc + wave * 256 + lane * 4, U_ + wave * 256);
    if constexpr (!HI) {
      glds16(usrc + (12 + wave) * 256 + lane * 4, U_ + (12 + wave) * 256);
    } else {
      const float* p = (const float*)((const char*)rp + (unsigned long long)sc * rstep);
      if (cin_odd && rch1 && s_begin + sc == nstages - 1) p = zpad;
      glds16(p, smem + R0 + slot * T::XSZP + T::XSHIFT + (wave - 6) * 256);
    }
  };

  f32x16 acc[3][2];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][hh][r] = 0.0f;

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  const int ty = l31 >> 3, tx = l31 & 7;
  const unsigned xbase = lds0 + (unsigned)(R0 + T::XWIN + half * T::XPLANE + 4 * ty * T::PW + 4 * tx) * 4u;
  const int prow = wave >> 1, jh = wave & 1;                                 // row of the position grid, column half
  const unsigned abase = lds0 + (unsigned)(U0 + half * 36 * T::BN + (prow * 6 + jh * 3) * T::BN + l31) * 4u;

  struct AOps {
    f32x2 p01[2];     // positions 0, 1 of the half row, per cout half
    float p2[2];      // position 2
  };
  auto load_a = [&](AOps& o, int slot) {
    const unsigned aa = abase + (unsigned)(slot * T::WSZ) * 4u;
    lds_read2st64_b32<0, 1>(o.p01[0], aa);
    lds_read_b32<2 * 256>(o.p2[0], aa);
    lds_read2st64_b32<0, 1>(o.p01[1], aa + 128u);
    lds_read_b32<2 * 256>(o.p2[1], aa + 128u);
  };
  constexpr int NA = 4;
  auto wait_a = [&](AOps& o, auto newer_c) {
    constexpr int N_ = decltype(newer_c)::value;
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(o.p01[0]), "+v"(o.p01[1]), "+v"(o.p2[0]), "+v"(o.p2[1]) : "n"(N_));
  };
  struct Raw {
    f32x4 lo[4];
    f32x2 hi[4];
  };

  auto main_loop = [&](auto role_c) {
    constexpr int W_ = decltype(role_c)::value;                    // wave index 0..11
    constexpr int ROW = W_ >> 1, JH = W_ & 1;
    constexpr bool HI = W_ >= 6;
    constexpr int R0_ = (ROW == 0) ? 0 : 1;
    constexpr int RSTEP = (ROW == 0 || ROW == 5) ? 2 : 1;
    constexpr int NROW = (ROW == 0 || ROW == 5) ? 3 : 4;
    constexpr int NRAW = 2 * NROW;
    auto raw_reads = [&](Raw& w, int slot) {
      const unsigned xa = xbase + (unsigned)(slot * T::XSZP) * 4u;
      w44_read_b128<(R0_ + 0 * RSTEP) * T::PW * 4>(w.lo[0], xa);
      w44_read_b64<((R0_ + 0 * RSTEP) * T::PW + 4) * 4>(w.hi[0], xa);
      w44_read_b128<(R0_ + 1 * RSTEP) * T::PW * 4>(w.lo[1], xa);
      w44_read_b64<((R0_ + 1 * RSTEP) * T::PW + 4) * 4>(w.hi[1], xa);
      w44_read_b128<(R0_ + 2 * RSTEP) * T::PW * 4>(w.lo[2], xa);
      w44_read_b64<((R0_ + 2 * RSTEP) * T::PW + 4) * 4>(w.hi[2], xa);
      if constexpr (NROW == 4) {
        w44_read_b128<(R0_ + 3 * RSTEP) * T::PW * 4>(w.lo[3], xa);
        w44_read_b64<((R0_ + 3 * RSTEP) * T::PW + 4) * 4>(w.hi[3], xa);
      }
    };
    auto wait_raw = [&](Raw& w, auto newer_c) {
      constexpr int N_ = decltype(newer_c)::value;
      if constexpr (NROW == 4)
        asm volatile("s_waitcnt lgkmcnt(%8)"
                     : "+v"(w.lo[0]), "+v"(w.lo[1]), "+v"(w.lo[2]), "+v"(w.lo[3]), "+v"(w.hi[0]), "+v"(w.hi[1]), "+v"(w.hi[2]),
                       "+v"(w.hi[3])
                     : "n"(N_));
      else
        asm volatile("s_waitcnt lgkmcnt(%6)"
                     : "+v"(w.lo[0]), "+v"(w.lo[1]), "+v"(w.lo[2]), "+v"(w.hi[0]), "+v"(w.hi[1]), "+v"(w.hi[2])
                     : "n"(N_));
    };
    // t[c] = (B^T d)[ROW][c] for the five columns the half row needs (0..4 or 1..5), then its three values of B^T t
    auto t_col = [&](const Raw& w, float (&t)[6], auto cc) {
      constexpr int c = decltype(cc)::value;
      float d[4];
#pragma unroll
      for (int i = 0; i < NROW; ++i) d[i] = c < 4 ? w.lo[i][c < 4 ? c : 0] : w.hi[i][c < 4 ? 0 : c - 4];
      if constexpr (ROW == 0 || ROW == 5)
        t[c] = fmaf(4.0f, d[0], fmaf(-5.0f, d[1], d[2]));
      else if constexpr (ROW == 1)
        t[c] = fmaf(-4.0f, d[1], d[3]) + fmaf(-4.0f, d[0], d[2]);
      else if constexpr (ROW == 2)
        t[c] = fmaf(-4.0f, d[1], d[3]) - fmaf(-4.0f, d[0], d[2]);
      else if constexpr (ROW == 3)
        t[c] = fmaf(2.0f, d[2] - d[0], d[3] - d[1]);
      else
        t[c] = fmaf(-2.0f, d[2] - d[0], d[3] - d[1]);
    };
    auto transform = [&](const Raw& w, float (&v)[3]) {
      float t[6];
      if constexpr (JH == 0) {
        t_col(w, t, std::integral_constant<int, 0>{});
        t_col(w, t, std::integral_constant<int, 1>{});
        t_col(w, t, std::integral_constant<int, 2>{});
        t_col(w, t, std::integral_constant<int, 3>{});
        t_col(w, t, std::integral_constant<int, 4>{});
        const float p = fmaf(-4.0f, t[2], t[4]), q = fmaf(-4.0f, t[1], t[3]);
        v[0] = fmaf(4.0f, t[0], fmaf(-5.0f, t[2], t[4]));
        v[1] = p + q;
        v[2] = p - q;
      } else {
        t_col(w, t, std::integral_constant<int, 1>{});
        t_col(w, t, std::integral_constant<int, 2>{});
        t_col(w, t, std::integral_constant<int, 3>{});
        t_col(w, t, std::integral_constant<int, 4>{});
        t_col(w, t, std::integral_constant<int, 5>{});
        const float c = t[4] - t[2], d = t[3] - t[1];
        v[0] = fmaf(2.0f, d, c);
        v[1] = fmaf(-2.0f, d, c);
        v[2] = fmaf(4.0f, t[1], fmaf(-5.0f, t[3], t[5]));
      }
    };
    using N0 = std::integral_constant<int, 0>;
    using HIc = std::bool_constant<HI>;
    auto mf = [&](const AOps& A_, const float (&B_)[3], auto jc, auto hc) {
      constexpr int j = decltype(jc)::value, hh = decltype(hc)::value;
      const float av = j < 2 ? A_.p01[hh][j < 2 ? j : 0] : A_.p2[hh];
      acc[j][hh] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, B_[j], acc[j][hh], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    // ---- prologue: U(0), R(0), U(1), R(1) landed; B(0), A(0) in registers; U(2), R(2) in flight
    issue(HIc{}, 0, 0);
    issue(HIc{}, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    AOps A0, A1;
    float B0[3], B1[3];
    {
      Raw w;
      raw_reads(w, 0);
      load_a(A0, 0);
      wait_raw(w, std::integral_constant<int, NA>{});
      transform(w, B0);
      wait_a(A0, N0{});
    }
    issue(HIc{}, 2, 2);

    // ---- k-step s: MFMAs on (Ac, Bc); window of s+1 -> Bn; weights of s+1 -> An; LDS-DMA of s+3
    auto step = [&](AOps& Ac, float (&Bc)[3], AOps& An, float (&Bn)[3], int s) {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");       // all but the previous k-step's two instructions have landed
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      Raw w;
      raw_reads(w, (s + 1) % 3);
      load_a(An, (s + 1) % 3);
      __builtin_amdgcn_sched_barrier(0);
      mf(Ac, Bc, I0{}, I0{});
      issue(HIc{}, s + 3, s % 3);
      __builtin_amdgcn_sched_barrier(0);
      mf(Ac, Bc, I0{}, I1{});
      mf(Ac, Bc, I1{}, I0{});
      wait_raw(w, std::integral_constant<int, NA>{});
      transform(w, Bn);
      __builtin_amdgcn_sched_barrier(0);
      mf(Ac, Bc, I1{}, I1{});
      mf(Ac, Bc, I2{}, I0{});
      mf(Ac, Bc, I2{}, I1{});
      wait_a(An, N0{});
    };
    int s = 0;
    for (; s + 1 < nst; s += 2) {
      step(A0, B0, A1, B1, s);
      step(A1, B1, A0, B0, s + 1);
    }
    if (s < nst) step(A0, B0, A1, B1, s);
  };
  switch (wave) {
    case 0: main_loop(std::integral_constant<int, 0>{}); break;
    case 1: main_loop(std::integral_constant<int, 1>{}); break;
    case 2: main_loop(std::integral_constant<int, 2>{}); break;
    case 3: main_loop(std::integral_constant<int, 3>{}); break;
    case 4: main_loop(std::integral_constant<int, 4>{}); break;
    case 5: main_loop(std::integral_constant<int, 5>{}); break;
    case 6: main_loop(std::integral_constant<int, 6>{}); break;
    case 7: main_loop(std::integral_constant<int, 7>{}); break;
    case 8: main_loop(std::integral_constant<int, 8>{}); break;
    case 9: main_loop(std::integral_constant<int, 9>{}); break;
    case 10: main_loop(std::integral_constant<int, 10>{}); break;
    default: main_loop(std::integral_constant<int, 11>{}); break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- output transform: this wave holds M[prow][3 jh .. 3 jh + 2] for both cout halves.  Per accumulator element it folds
  // them into the 4-vector (M A)[prow] restricted to its columns; the twelve waves exchange those through LDS (four rounds of
  // four registers, 96 KB each) and waves 0..7 finish one (cout half, register) each per round: whole 4x4 tiles.
  const int HWo = a.Hout * a.Wout;
  const bool raw = a.ws != nullptr;
  const bool wt = raw && a.sk_count;   // slabs of an in-kernel combine are stored write-through (conv.h)
  float* dst_b = raw ? a.ws + ((long long)(b * a.ksplit + ks) * a.Cout) * HWo
                     : a.out + ((long long)b * a.out_ctot + a.out_coff) * HWo;
  const float* res_b = (!raw && a.res) ? a.res + ((long long)b * a.res_ctot + a.res_coff) * HWo : nullptr;
  const int oy = oy0 + 4 * ty, ox = ox0 + 4 * tx;
  float* E = smem;                                            // [cout half][4 regs][12 waves][4 values][64 lanes]
  for (int round = 0; round < 4; ++round) {
    __syncthreads();
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        float m[3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
          m[j] = round == 0 ? acc[j][hh][rr] : round == 1 ? acc[j][hh][4 + rr] : round == 2 ? acc[j][hh][8 + rr] : acc[j][hh][12 + rr];
        float v4[4];
        if (jh == 0) {                        // columns 0..2 of A^T: [1 1 1; 0 1 -1; 0 1 1; 0 1 -1]
          v4[0] = m[0] + m[1] + m[2];
          v4[1] = m[1] - m[2];
          v4[2] = m[1] + m[2];
          v4[3] = m[1] - m[2];
        } else {                              // columns 3..5: [1 1 0; 2 -2 0; 4 4 0; 8 -8 1]
          const float s2 = m[0] + m[1], d2 = m[0] - m[1];
          v4[0] = s2;
          v4[1] = 2.0f * d2;
          v4[2] = 4.0f * s2;
          v4[3] = fmaf(8.0f, d2, m[2]);
        }
        float* e = E + (((hh * 4 + rr) * 12 + wave) * 4) * 64 + lane;
#pragma unroll
        for (int v = 0; v < 4; ++v) e[v * 64] = v4[v];
      }
    __syncthreads();
    if (wave < 8) {
      const int hh = wave >> 2, rr = wave & 3;
      const int r = 4 * round + rr;
      float R[6][4];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v)
          R[i][v] = E[(((hh * 4 + rr) * 12 + 2 * i) * 4 + v) * 64 + lane] + E[(((hh * 4 + rr) * 12 + 2 * i + 1) * 4 + v) * 64 + lane];
      const int co = n_tile * T::BN + hh * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co < a.Cout && ox < a.Wout) {
        const float bv = (!raw && a.bias) ? a.bias[co] : 0.0f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (oy + u >= a.Hout) continue;
          float y[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const float s1 = R[1][v] + R[2][v], d1 = R[1][v] - R[2][v], s2 = R[3][v] + R[4][v], d2 = R[3][v] - R[4][v];
            y[v] = u == 0 ? R[0][v] + s1 + s2 : u == 1 ? fmaf(2.0f, d2, d1) : u == 2 ? fmaf(4.0f, s2, s1) : fmaf(8.0f, d2, d1) + R[5][v];
          }
          const long long off = (long long)co * HWo + (long long)(oy + u) * a.Wout + ox;
          float4 o = make_float4(y[0], y[1], y[2], y[3]);       // Wout % 4 == 0 (VEC): the tile row is whole and 16-byte aligned
          if (!raw) {
            o.x += bv; o.y += bv; o.z += bv; o.w += bv;
            if (res_b) {
              const float4 rv = *reinterpret_cast<const float4*>(res_b + off);
              o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
            }
            if (a.act == ACT_RELU) {
              o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
            } else if (a.act == ACT_RELU6) {
              o.x = fminf(fmaxf(o.x, 0.f), 6.f); o.y = fminf(fmaxf(o.y, 0.f), 6.f);
              o.z = fminf(fmaxf(o.z, 0.f), 6.f); o.w = fminf(fmaxf(o.w, 0.f), 6.f);
            }
          }
          slab_store4(dst_b + off, o.x, o.y, o.z, o.w, wt);
        }
      }
    }
  }
  if (wt) splitk_combine_tile<768>(a, b, sp_tile + a.n_sp * n_tile, n_tile * T::BN, T::BN, oy0, ox0, T::TH, T::TW, (unsigned*)smem);
}

inline KernelEntry wino44_entry() {
  static_assert(W44::LDS_BYTES >= W44odd::LDS_BYTES, "the entry carries one dynamic-LDS size for both width classes");
  return KernelEntry{conv_wino44_kernel<W44>, W44::LDS_BYTES, 512, conv_wino44_kernel<W44odd>};
}
// dilation 2 (the SSH context convs): eight-wave form only, Win % 4 == 0 only (launch_conv refuses other widths)
inline KernelEntry wino44d2_entry() {
  return KernelEntry{conv_wino44_kernel<W44D2>, W44D2::LDS_BYTES, 512};
}
// twelve waves; odd widths (Win % 4 != 0) fall back to the eight-wave dword-staging kernel -- its own LDS size and block size
// differ, so the entry carries the larger LDS request and launch_conv picks threads per variant (KernelEntry::threads_odd)
inline KernelEntry wino44b_entry() {
  constexpr size_t ring = (size_t)(3 * W44::WSZ + 3 * W44::XSZP) * sizeof(float);
  constexpr size_t exch = (size_t)2 * 4 * 12 * 4 * 64 * sizeof(float);
  constexpr size_t lds = ring > exch ? ring : exch;
  KernelEntry e{conv_wino44b_kernel<W44>, lds > W44::LDS_BYTES ? lds : W44::LDS_BYTES, 768, conv_wino44_kernel<W44odd>};
  e.threads_odd = 512;
  return e;
}

}  // namespace

void conv_fill_wino44(void* row);
void conv_fill_wino44_d2(void* row);

}  // namespace fdt
