// conv_wino.h -- Winograd F(2x2,3x3) variant of the 3x3 / stride 1 / pad 1 convolution on the f32 matrix
// cores: 16 multiplies per 2x2 output block instead of 36 (2.25x fewer MFMA FLOPs), exact-arithmetic
// equivalent of nn.Conv2d(k=3, s=1, p=1) up to f32 rounding of the transforms.
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A          (per 2x2 output block, per output channel)
//
//  * The 16 element-wise products over input channels are 16 independent GEMMs
//    D_t[cout][block] = sum_c U_t[cout][c] * V_t[c][block], t = 0..15, run on v_mfma_f32_32x32x2_f32
//    exactly like conv_kernel: weights (pre-transformed on the host, U = G g G^T) are the A operand,
//    image blocks the B operand.
//  * Input transform on the fly: the raw (2*TTH+2) x (2*TTW+2) patch of KC channels is staged once per
//    stage by LDS-DMA (same zero-word trick for padding); each lane (= one 2x2 output block) reads its
//    4x4 window with 8 ds_read2_b32 and forms the 16 V_t values with 32 VALU adds -- the MFMA B operands.
//    No transformed copy of the activations ever exists in memory.
//  * All 16 accumulators of a wave's 32-cout x 32-block tile stay in registers (256 VGPRs, one wave per
//    SIMD: this kernel trades occupancy for 2.25x less matrix work; 16 independent MFMAs per k-pair give
//    the in-wave parallelism that other waves would otherwise provide).
//  * Output transform in registers (all 16 D_t of one (cout, block) sit in ONE lane): 24 adds, then
//    bias / residual / ReLU and one 8-byte store per row of the 2x2 block.
#pragma once
#include <type_traits>

#include "conv_kernel.h"

namespace fdt {
namespace {


// ---- LDS operand reads with hand-counted waits --------------------------------------------------------
// The compiler's waitcnt model treats every LDS-DMA load (global_load ... lds) as a FLAT access that stays
// pending until a vmcnt(0) it can see; while one is pending, each lgkmcnt wait it inserts is a full
// lgkmcnt(0) -- which would drain the operand prefetch the moment it is issued.  The main loop of the
// 8-wave kernel therefore issues its ds_reads itself and waits with an exact count (LDS returns in order).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int O0, int O1>
__device__ __forceinline__ void lds_read2_b64(f32x4& v, unsigned addr) {      // offsets in units of 8 bytes
  asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1));
}
template <int O0, int O1>
__device__ __forceinline__ void lds_read2_b32(f32x2& v, unsigned addr) {      // offsets in dwords
  asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1));
}
template <int O0, int O1>
__device__ __forceinline__ void lds_read2st64_b32(f32x2& v, unsigned addr) {  // offsets in units of 64 dwords
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(O0), "n"(O1));
}

template <int TTH_, int TTW_, int WM_, int WN_, int KC_, int NBUF_, int D_ = 1>
struct WinoTile {
  static constexpr int TTH = TTH_, TTW = TTW_, WM = WM_, WN = WN_, KC = KC_, NBUF = NBUF_;
  static constexpr int D = D_;                 // dilation (= padding): 1, or 2 for the SSH context convs
  static constexpr int BMT = TTH * TTW;        // 2x2 output blocks per workgroup
  static constexpr int BN = WN * 32;           // output channels per workgroup
  static constexpr int PH = 2 * TTH + 2 * D, PW = 2 * TTW + 2 * D;
  static constexpr int XPLANE = PH * PW;
  static constexpr int XSZ = KC * XPLANE;
  static constexpr int XSZP = (XSZ + 511) / 512 * 512;   // whole LDS-DMA wave-instructions for 4 or 8 waves
  static constexpr int WSZ = KC * 16 * BN;
  static constexpr int WSZP = (WSZ + 1023) / 1024 * 1024;
  static constexpr int STAGE = XSZP + WSZP;
  static constexpr int NX = XSZP / 256, NW = WSZP / 1024;
  static constexpr int LOADS = NX + NW;
  static constexpr size_t LDS_BYTES = (size_t)NBUF * STAGE * sizeof(float);
  static constexpr bool FITS = LDS_BYTES <= 160 * 1024;
  static_assert(WM * WN == 4 && WM * 32 == BMT, "4 waves, 32 blocks per wave row");
  static_assert(KC % 2 == 0 && NX <= 32 && LOADS < 64, "staging limits");
  static_assert((PW % 2) == 0 && (XPLANE % 2) == 0, "8-byte aligned window reads");
};

template <class T>
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / T::WN, wn = wave % T::WN;
  const int half = lane >> 5, l31 = lane & 31;

  constexpr int TH = 2 * T::TTH, TW = 2 * T::TTW;          // output pixels per workgroup
  const int tiles_x = (a.Wout + TW - 1) / TW;
  FDT_BLOCK_MAP(a, sp_tile, n_tile);
  const int oy0 = (sp_tile / tiles_x) * TH;
  const int ox0 = (sp_tile % tiles_x) * TW;
  const int b = blockIdx.z / a.ksplit;
  const int ks = blockIdx.z - b * a.ksplit;

  const int HW = a.Hin * a.Win;                             // stride 1, pad 1: Hout == Hin
  const float* in_b = a.in + (long long)b * conv_in_bstride(a);
  const int nstages = (a.Cin + T::KC - 1) / T::KC;
  const float* w_t = a.w + (long long)n_tile * nstages * T::WSZP;
  const int s_begin = (int)((long long)nstages * ks / a.ksplit);
  const int s_end = (int)((long long)nstages * (ks + 1) / a.ksplit);

  // staging plan: element e = 256 k + tid of the patch, fetched through a buffer descriptor at byte offset xoff[k] relative to
  // the stage's first channel (kOob for padding / out-of-image: zeros from the bounds check, like the channels past Cin)
  const __amdgpu_buffer_rsrc_t xrs = buf_rsrc(in_b, (long long)a.Cin * HW * 4);
  const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(w_t, 0x7fffffffll);
  unsigned xoff[T::NX];
#pragma unroll
  for (int k = 0; k < T::NX; ++k) {
    int e = tid + 256 * k;
    int c = e / T::XPLANE;
    int r = e - c * T::XPLANE;
    int yy = r / T::PW, xx = r - yy * T::PW;
    int gy = oy0 - 1 + yy, gx = ox0 - 1 + xx;
    bool ok = (e < T::XSZ) && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
    xoff[k] = ok ? (unsigned)(c * HW + gy * a.Win + gx) * 4u : kOob;
  }

#define FDT_WSTAGE(s_, buf_)                                                                  \
  {                                                                                           \
    const unsigned xso_ = (unsigned)((s_) * T::KC) * (unsigned)HW * 4u;                       \
    float* X_ = smem + (buf_) * T::STAGE + wave * 64;                                         \
    _Pragma("unroll") for (int k = 0; k < T::NX; ++k) bglds4(xrs, X_ + 256 * k, xoff[k], xso_);   \
    const unsigned wso_ = (unsigned)((s_) * T::WSZP) * 4u;                                    \
    float* W_ = smem + (buf_) * T::STAGE + T::XSZP + wave * 256;                              \
    _Pragma("unroll") for (int k = 0; k < T::NW; ++k) bglds16(wrs, W_ + 1024 * k, (unsigned)tid * 16u, wso_ + 4096u * k); \
  }

  // this lane's 2x2 output block inside the workgroup patch
  const int q = wm * 32 + l31;
  const int ty = q / T::TTW, tx = q % T::TTW;
  const int xo = half * T::XPLANE + (2 * ty) * T::PW + 2 * tx;          // top-left of the 4x4 window
  const int wo = T::XSZP + half * 16 * T::BN + wn * 32 + l31;

  f32x16 acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  const int nst = s_end - s_begin;
#pragma unroll
  for (int p = 0; p < T::NBUF - 1; ++p)
    if (p < nst) FDT_WSTAGE(s_begin + p, p);
  int cur = 0, nxt = T::NBUF - 1;
  for (int it = 0; it < nst; ++it) {
    if (T::NBUF >= 3 && it + 1 < nst)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T::LOADS) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (it + T::NBUF - 1 < nst) FDT_WSTAGE(s_begin + it + T::NBUF - 1, nxt);
    const float* S = smem + cur * T::STAGE;
#pragma unroll
    for (int cp = 0; cp < T::KC / 2; ++cp) {
      // raw 4x4 window d[i][j] of channel 2*cp + half
      float d[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)      // plain float reads (see conv_wino2_kernel: no float2-typed LDS reads)
#pragma unroll
        for (int j = 0; j < 4; ++j) d[i][j] = S[xo + (2 * cp) * T::XPLANE + i * T::PW + j];
      // V = B^T d B,  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
      float t_[4][4], v[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t_[0][j] = d[0][j] - d[2][j];
        t_[1][j] = d[1][j] + d[2][j];
        t_[2][j] = d[2][j] - d[1][j];
        t_[3][j] = d[1][j] - d[3][j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i][0] = t_[i][0] - t_[i][2];
        v[i][1] = t_[i][1] + t_[i][2];
        v[i][2] = t_[i][2] - t_[i][1];
        v[i][3] = t_[i][1] - t_[i][3];
      }
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float u = S[wo + ((2 * cp) * 16 + t) * T::BN];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(u, v[t >> 2][t & 3], acc[t], 0, 0, 0);
      }
    }
    cur = (cur + 1 == T::NBUF) ? 0 : cur + 1;
    nxt = (nxt + 1 == T::NBUF) ? 0 : nxt + 1;
  }
#undef FDT_WSTAGE

  // ---- output transform + epilogue: Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1] -----------------------------
  const int HWo = a.Hout * a.Wout;
  const int oy = oy0 + 2 * ty, ox = ox0 + 2 * tx;
  const bool raw = a.ws != nullptr;
  const bool wt = raw && a.sk_count;   // slabs of an in-kernel combine are stored write-through (conv.h)
  float* dst_b = raw ? a.ws + ((long long)(b * a.ksplit + ks) * a.Cout) * HWo
                     : a.out + ((long long)b * a.out_ctot + a.out_coff) * HWo;
  const float* res_b = (!raw && a.res) ? a.res + ((long long)b * a.res_ctot + a.res_coff) * HWo : nullptr;
  const bool row0 = oy < a.Hout, row1 = oy + 1 < a.Hout;
  const bool col0 = ox < a.Wout, col1 = ox + 1 < a.Wout;
  const bool vec2 = (a.Wout % 2 == 0);          // ox is even -> both columns in range and 8-byte aligned
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = n_tile * T::BN + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    float s0[4], s1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s0[j] = acc[0 * 4 + j][r] + acc[1 * 4 + j][r] + acc[2 * 4 + j][r];
      s1[j] = acc[1 * 4 + j][r] - acc[2 * 4 + j][r] - acc[3 * 4 + j][r];
    }
    float y00 = s0[0] + s0[1] + s0[2], y01 = s0[1] - s0[2] - s0[3];
    float y10 = s1[0] + s1[1] + s1[2], y11 = s1[1] - s1[2] - s1[3];
    if (co < a.Cout && col0) {
      const long long base = (long long)co * HWo + (long long)oy * a.Wout + ox;
      if (!raw) {
        const float bv = a.bias ? a.bias[co] : 0.0f;
        y00 += bv; y01 += bv; y10 += bv; y11 += bv;
        if (res_b) {
          if (row0) { y00 += res_b[base]; if (col1) y01 += res_b[base + 1]; }
          if (row1) { y10 += res_b[base + a.Wout]; if (col1) y11 += res_b[base + a.Wout + 1]; }
        }
        if (a.act == ACT_RELU) {
          y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f);
        } else if (a.act == ACT_RELU6) {
          y00 = fminf(fmaxf(y00, 0.f), 6.f); y01 = fminf(fmaxf(y01, 0.f), 6.f);
          y10 = fminf(fmaxf(y10, 0.f), 6.f); y11 = fminf(fmaxf(y11, 0.f), 6.f);
        }
      }
      if (vec2) {
        if (row0) slab_store2(dst_b + base, y00, y01, wt);
        if (row1) slab_store2(dst_b + base + a.Wout, y10, y11, wt);
      } else {
        if (row0) { slab_store1(dst_b + base, y00, wt); if (col1) slab_store1(dst_b + base + 1, y01, wt); }
        if (row1) { slab_store1(dst_b + base + a.Wout, y10, wt); if (col1) slab_store1(dst_b + base + a.Wout + 1, y11, wt); }
      }
    }
  }
  if (wt) splitk_combine_tile<256>(a, b, sp_tile + a.n_sp * n_tile, n_tile * T::BN, T::BN, oy0, ox0, TH, TW, (unsigned*)smem);
}

// ---------------------------------------------------------------------------------------------------------
// 8-wave form: the 16 transform positions are split between two waves that share one SIMD -- waves 0-3
// own rows 0,1 of the 4x4 position grid, waves 4-7 rows 2,3 -- so each wave carries 8 accumulators
// (128 AGPRs) and TWO waves per SIMD hide each other's LDS / LDS-DMA / barrier latency.  Same workgroup
// tile, same LDS stage and DMA traffic as the 4-wave form; the two halves of the (linear) output
// transform meet through LDS once, in the epilogue.
template <class T>
__global__ __launch_bounds__(512, 2) void conv_wino2_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(T::XSZP % 512 == 0 && T::WSZP % 2048 == 0, "8-wave staging granularity");
  constexpr int NX2 = T::XSZP / 512, NW2 = T::WSZP / 2048, LOADS2 = NX2 + NW2;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int th = wave >> 2;                                  // which half of the position grid
  const int w3 = wave & 3;
  const int wm = w3 / T::WN, wn = w3 % T::WN;
  const int half = lane >> 5, l31 = lane & 31;

  constexpr int TH = 2 * T::TTH, TW = 2 * T::TTW;
  const int tiles_x = (a.Wout + TW - 1) / TW;
  FDT_BLOCK_MAP(a, sp_tile, n_tile);
  const int oy0 = (sp_tile / tiles_x) * TH;
  const int ox0 = (sp_tile % tiles_x) * TW;
  const int b = blockIdx.z / a.ksplit;
  const int ks = blockIdx.z - b * a.ksplit;

  const int HW = a.Hin * a.Win;
  const float* in_b = a.in + (long long)b * conv_in_bstride(a);
  const int nstages = (a.Cin + T::KC - 1) / T::KC;
  const float* w_t = a.w + (long long)n_tile * nstages * T::WSZP;
  const int s_begin = (int)((long long)nstages * ks / a.ksplit);
  const int s_end = (int)((long long)nstages * (ks + 1) / a.ksplit);

  // staging plan: element e = 512 k + tid of the patch, fetched through a buffer descriptor at byte offset xoff[k] relative to
  // the stage's first channel (kOob for padding / out-of-image: zeros from the bounds check, like the channels past Cin)
  const __amdgpu_buffer_rsrc_t xrs = buf_rsrc(in_b, (long long)a.Cin * HW * 4);
  const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(w_t, 0x7fffffffll);
  unsigned xoff[NX2];
#pragma unroll
  for (int k = 0; k < NX2; ++k) {
    int e = tid + 512 * k;
    int c = e / T::XPLANE;
    int r = e - c * T::XPLANE;
    int yy = r / T::PW, xx = r - yy * T::PW;
    int gy = oy0 - T::D + yy, gx = ox0 - T::D + xx;
    bool ok = (e < T::XSZ) && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
    xoff[k] = ok ? (unsigned)(c * HW + gy * a.Win + gx) * 4u : kOob;
  }

#define FDT_W2STAGE(s_, buf_)                                                                  \
  {                                                                                           \
    const unsigned xso_ = (unsigned)((s_) * T::KC) * (unsigned)HW * 4u;                       \
    float* X_ = smem + (buf_) * T::STAGE + wave * 64;                                         \
    _Pragma("unroll") for (int k = 0; k < NX2; ++k) bglds4(xrs, X_ + 512 * k, xoff[k], xso_);   \
    const unsigned wso_ = (unsigned)((s_) * T::WSZP) * 4u;                                    \
    float* W_ = smem + (buf_) * T::STAGE + T::XSZP + wave * 256;                              \
    _Pragma("unroll") for (int k = 0; k < NW2; ++k) bglds16(wrs, W_ + 2048 * k, (unsigned)tid * 16u, wso_ + 8192u * k); \
  }

  // this lane's 2x2 output block {(oyl, oxl) + D * (0|1, 0|1)} inside the workgroup patch.  D = 1: blocks
  // tile the patch; D = 2: the four parity sub-lattices of every 4x4 cell are four blocks.
  const int q = wm * 32 + l31;
  int oyl, oxl;
  if (T::D == 1) {
    oyl = 2 * (q / T::TTW);
    oxl = 2 * (q % T::TTW);
  } else {
    constexpr int CX = (2 * T::TTW) / 4;
    const int cell = q >> 2, par = q & 3;
    oyl = 4 * (cell / CX) + (par >> 1);
    oxl = 4 * (cell % CX) + (par & 1);
  }
  const int xo = half * T::XPLANE + oyl * T::PW + oxl;                 // top-left of the 4x4 (stride D) window
  const int wo = T::XSZP + half * 16 * T::BN + th * 8 * T::BN + wn * 32 + l31;

  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  const int nst = s_end - s_begin;
#pragma unroll
  for (int p = 0; p < T::NBUF - 1; ++p)
    if (p < nst) FDT_W2STAGE(s_begin + p, p);
  // The main loop is instantiated once per half of the position grid and selected by ONE wave-uniform
  // branch: with `th` a compile-time constant the loop body is straight-line code (no per-row branches
  // or selects), so the scheduler can hoist the next channel pair's LDS reads over the current MFMAs.
  // Both instances execute the same s_barrier sequence.
  auto main_loop = [&](auto th_c) {
    constexpr int TH_ = decltype(th_c)::value;
    // LDS operands of one channel pair: rows TH_ .. TH_+2 of the 4x4 (stride D) window and the 8 weights.
    // Plain float reads on purpose: behind a float2-typed LDS read the compiler's waitcnt pass puts a full
    // s_waitcnt vmcnt(0) (it assumes the read may alias the LDS-DMA stores in flight), which serialises
    // the ring.
    auto lds_operands = [&](const float* S, int cp, float (&d)[3][4], float (&u)[8]) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          d[i][j] = S[xo + (2 * cp) * T::XPLANE + ((i + TH_) * T::D) * T::PW + j * T::D];
#pragma unroll
      for (int t = 0; t < 8; ++t) u[t] = S[wo + ((2 * cp) * 16 + t) * T::BN];
    };
    auto mac = [&](const float (&d)[3][4], const float (&u)[8]) {
      // the two rows of B^T d this wave needs: TH_ = 0: (d0 - d2, d1 + d2); TH_ = 1: (d2 - d1, d1 - d3)
      float ra[4], rb[4], v[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ra[j] = TH_ ? (d[1][j] - d[0][j]) : (d[0][j] - d[2][j]);
        rb[j] = TH_ ? (d[0][j] - d[2][j]) : (d[1][j] + d[2][j]);
      }
      v[0][0] = ra[0] - ra[2]; v[0][1] = ra[1] + ra[2]; v[0][2] = ra[2] - ra[1]; v[0][3] = ra[1] - ra[3];
      v[1][0] = rb[0] - rb[2]; v[1][1] = rb[1] + rb[2]; v[1][2] = rb[2] - rb[1]; v[1][3] = rb[1] - rb[3];
#pragma unroll
      for (int t = 0; t < 8; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[t], v[t >> 2][t & 3], acc[t], 0, 0, 0);
    };
    if constexpr (T::NBUF >= 3 && (T::KC / 2) % 2 == 0) {
      // Ring of three with the workgroup barrier in the MIDDLE of a stage: at the barrier of stage `it` every
      // wave has finished stage it-1 (its buffer is free for the DMA of stage it+2) and has waited for its
      // share of stage it+1, so the hand-over it -> it+1 needs no synchronisation at all and the operand
      // prefetch (one channel pair ahead, register sets A/B alternating) runs straight across it.
      struct Ops {
        f32x4 d1[3];          // D = 1: one window row each (ds_read2_b64)
        f32x2 d2[3][2];       // D = 2: a row is four dwords 8 bytes apart (2 x ds_read2_b32)
        f32x2 u[4];           // weights of positions (2k, 2k+1)
      };
      constexpr int NLD = (T::D == 1 ? 3 : 6) + 4;           // ds instructions per operand set
      constexpr int ROWB = T::D * T::PW * 4;                  // bytes between window rows
      static_assert(ROWB % 8 == 0 && (3 * ROWB) / 8 + 1 < 256 && 3 * (ROWB / 4) + 6 < 256, "ds offset fields");
      static_assert(T::BN % 64 == 0 || 7 * T::BN < 256, "weight offsets fit the ds offset fields");
      const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
      const unsigned xb = lds0 + (unsigned)xo * 4u, wb = lds0 + (unsigned)wo * 4u;
      auto load = [&](Ops& o, int buf, int cp) {
        const unsigned xa = xb + (unsigned)(buf * T::STAGE + 2 * cp * T::XPLANE) * 4u;
        const unsigned wa = wb + (unsigned)(buf * T::STAGE + 2 * cp * 16 * T::BN) * 4u;
        if constexpr (T::D == 1) {
          lds_read2_b64<(TH_ + 0) * ROWB / 8, (TH_ + 0) * ROWB / 8 + 1>(o.d1[0], xa);
          lds_read2_b64<(TH_ + 1) * ROWB / 8, (TH_ + 1) * ROWB / 8 + 1>(o.d1[1], xa);
          lds_read2_b64<(TH_ + 2) * ROWB / 8, (TH_ + 2) * ROWB / 8 + 1>(o.d1[2], xa);
        } else {
          lds_read2_b32<(TH_ + 0) * ROWB / 4, (TH_ + 0) * ROWB / 4 + 2>(o.d2[0][0], xa);
          lds_read2_b32<(TH_ + 0) * ROWB / 4 + 4, (TH_ + 0) * ROWB / 4 + 6>(o.d2[0][1], xa);
          lds_read2_b32<(TH_ + 1) * ROWB / 4, (TH_ + 1) * ROWB / 4 + 2>(o.d2[1][0], xa);
          lds_read2_b32<(TH_ + 1) * ROWB / 4 + 4, (TH_ + 1) * ROWB / 4 + 6>(o.d2[1][1], xa);
          lds_read2_b32<(TH_ + 2) * ROWB / 4, (TH_ + 2) * ROWB / 4 + 2>(o.d2[2][0], xa);
          lds_read2_b32<(TH_ + 2) * ROWB / 4 + 4, (TH_ + 2) * ROWB / 4 + 6>(o.d2[2][1], xa);
        }
        if constexpr (T::BN % 64 == 0) {
          constexpr int Q = T::BN / 64;
          lds_read2st64_b32<0 * Q, 1 * Q>(o.u[0], wa);
          lds_read2st64_b32<2 * Q, 3 * Q>(o.u[1], wa);
          lds_read2st64_b32<4 * Q, 5 * Q>(o.u[2], wa);
          lds_read2st64_b32<6 * Q, 7 * Q>(o.u[3], wa);
        } else {
          lds_read2_b32<0 * T::BN, 1 * T::BN>(o.u[0], wa);
          lds_read2_b32<2 * T::BN, 3 * T::BN>(o.u[1], wa);
          lds_read2_b32<4 * T::BN, 5 * T::BN>(o.u[2], wa);
          lds_read2_b32<6 * T::BN, 7 * T::BN>(o.u[3], wa);
        }
      };
      // wait until at most `newer` ds instructions issued after o's are outstanding; o's registers pass through
      // the asm so that no use can be scheduled above it
      auto wait_for = [&](Ops& o, auto newer_c) {
        constexpr int N_ = decltype(newer_c)::value;
        if constexpr (T::D == 1)
          asm volatile("s_waitcnt lgkmcnt(%7)"
                       : "+v"(o.d1[0]), "+v"(o.d1[1]), "+v"(o.d1[2]), "+v"(o.u[0]), "+v"(o.u[1]), "+v"(o.u[2]),
                         "+v"(o.u[3])
                       : "n"(N_));
        else
          asm volatile("s_waitcnt lgkmcnt(%10)"
                       : "+v"(o.d2[0][0]), "+v"(o.d2[0][1]), "+v"(o.d2[1][0]), "+v"(o.d2[1][1]), "+v"(o.d2[2][0]),
                         "+v"(o.d2[2][1]), "+v"(o.u[0]), "+v"(o.u[1]), "+v"(o.u[2]), "+v"(o.u[3])
                       : "n"(N_));
      };
      auto mac_ops = [&](const Ops& o) {
        float d[3][4], u[8];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          if constexpr (T::D == 1) {
            d[i][0] = o.d1[i][0]; d[i][1] = o.d1[i][1]; d[i][2] = o.d1[i][2]; d[i][3] = o.d1[i][3];
          } else {
            d[i][0] = o.d2[i][0][0]; d[i][1] = o.d2[i][0][1]; d[i][2] = o.d2[i][1][0]; d[i][3] = o.d2[i][1][1];
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { u[2 * k] = o.u[k][0]; u[2 * k + 1] = o.u[k][1]; }
        mac(d, u);
      };
      using N0 = std::integral_constant<int, 0>;
      using NL = std::integral_constant<int, NLD>;

      if (nst > 1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS2) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      Ops A, B;
      int cur = 0;
      load(A, 0, 0);
      for (int it = 0; it < nst; ++it) {
        const int nb = (cur + 1 == 3) ? 0 : cur + 1;
#pragma unroll
        for (int cp = 0; cp < T::KC / 2; cp += 2) {
          if (cp == 2 || T::KC / 2 == 2) {
            if (it + 1 < nst) {
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              __builtin_amdgcn_s_barrier();
              __builtin_amdgcn_sched_barrier(0);
              const int fb = (nb + 1 == 3) ? 0 : nb + 1;
              if (it + 2 < nst) FDT_W2STAGE(s_begin + it + 2, fb);
            }
          }
          load(B, cur, cp + 1);
          wait_for(A, NL{});
          mac_ops(A);
          // One straight-line path on purpose (after the last stage this reads a stale buffer and the result
          // is dropped): a second path would meet this one in a PHI, and the copies that resolves into may
          // be placed in front of the wait, i.e. read registers whose ds_read has not landed yet.
          if (cp + 2 < T::KC / 2) load(A, cur, cp + 2);
          else load(A, nb, 0);
          wait_for(B, NL{});
          mac_ops(B);
        }
        cur = nb;
      }
      wait_for(A, N0{});      // drain: A's registers must not be reused while its reads are in flight
    } else {
      int cur = 0, nxt = T::NBUF - 1;
      for (int it = 0; it < nst; ++it) {
        if (T::NBUF >= 3 && it + 1 < nst)
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS2) : "memory");
        else
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (it + T::NBUF - 1 < nst) FDT_W2STAGE(s_begin + it + T::NBUF - 1, nxt);
        const float* S = smem + cur * T::STAGE;
#pragma unroll
        for (int cp = 0; cp < T::KC / 2; ++cp) {
          float d[3][4], u[8];
          lds_operands(S, cp, d, u);
          mac(d, u);
        }
        cur = (cur + 1 == T::NBUF) ? 0 : cur + 1;
        nxt = (nxt + 1 == T::NBUF) ? 0 : nxt + 1;
      }
    }
  };
  if (th) main_loop(std::integral_constant<int, 1>{});
  else main_loop(std::integral_constant<int, 0>{});
#undef FDT_W2STAGE

  // ---- output transform: each half contributes linearly; halves meet through LDS ------------------------
  //   th 0 (rows 0,1): s0 = M0 + M1, s1 = M1          th 1 (rows 2,3): s0 = M2, s1 = -M2 - M3
  __syncthreads();                       // ring is dead; reuse it as the exchange buffer
  float* E = smem + (long long)w3 * (16 * 4 * 64);
  const int HWo = a.Hout * a.Wout;
  const int oy = oy0 + oyl, ox = ox0 + oxl;
  const bool raw = a.ws != nullptr;
  const bool wt = raw && a.sk_count;   // slabs of an in-kernel combine are stored write-through (conv.h)
  float* dst_b = raw ? a.ws + ((long long)(b * a.ksplit + ks) * a.Cout) * HWo
                     : a.out + ((long long)b * a.out_ctot + a.out_coff) * HWo;
  const float* res_b = (!raw && a.res) ? a.res + ((long long)b * a.res_ctot + a.res_coff) * HWo : nullptr;
  const bool row0 = oy < a.Hout, row1 = oy + T::D < a.Hout;
  const bool col0 = ox < a.Wout, col1 = ox + T::D < a.Wout;
  const bool vec2 = (T::D == 1) && (a.Wout % 2 == 0);
  float yp[16][4];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float s0[4], s1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s0[j] = th ? acc[j][r] : (acc[j][r] + acc[4 + j][r]);
      s1[j] = th ? (-acc[j][r] - acc[4 + j][r]) : acc[4 + j][r];
    }
    yp[r][0] = s0[0] + s0[1] + s0[2];
    yp[r][1] = s0[1] - s0[2] - s0[3];
    yp[r][2] = s1[0] + s1[1] + s1[2];
    yp[r][3] = s1[1] - s1[2] - s1[3];
  }
  if (th) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int k = 0; k < 4; ++k) E[(r * 4 + k) * 64 + lane] = yp[r][k];
  }
  __syncthreads();
  // The waves of the second half have handed their part over.  Without an in-kernel combine they are done; with one they stay:
  // splitk_combine_tile<512> strides its elements over all 512 threads of the workgroup (round 5: they used to return here, and
  // the last-arriving workgroup then finished only every other 256-element run of its tile -- unnoticed while the parity test's
  // output buffer happened to be reallocated over the two-pass result it is compared with)
  if (th && !wt) return;
  if (!th)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = n_tile * T::BN + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    float y00 = yp[r][0] + E[(r * 4 + 0) * 64 + lane], y01 = yp[r][1] + E[(r * 4 + 1) * 64 + lane];
    float y10 = yp[r][2] + E[(r * 4 + 2) * 64 + lane], y11 = yp[r][3] + E[(r * 4 + 3) * 64 + lane];
    if (co < a.Cout && col0) {
      const long long base = (long long)co * HWo + (long long)oy * a.Wout + ox;
      if (!raw) {
        const float bv = a.bias ? a.bias[co] : 0.0f;
        y00 += bv; y01 += bv; y10 += bv; y11 += bv;
        if (res_b) {
          if (row0) { y00 += res_b[base]; if (col1) y01 += res_b[base + T::D]; }
          if (row1) { y10 += res_b[base + T::D * a.Wout]; if (col1) y11 += res_b[base + T::D * a.Wout + T::D]; }
        }
        if (a.act == ACT_RELU) {
          y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f);
        } else if (a.act == ACT_RELU6) {
          y00 = fminf(fmaxf(y00, 0.f), 6.f); y01 = fminf(fmaxf(y01, 0.f), 6.f);
          y10 = fminf(fmaxf(y10, 0.f), 6.f); y11 = fminf(fmaxf(y11, 0.f), 6.f);
        }
      }
      if (vec2) {
        if (row0) slab_store2(dst_b + base, y00, y01, wt);
        if (row1) slab_store2(dst_b + base + a.Wout, y10, y11, wt);
      } else {
        if (row0) { slab_store1(dst_b + base, y00, wt); if (col1) slab_store1(dst_b + base + T::D, y01, wt); }
        if (row1) { slab_store1(dst_b + base + T::D * a.Wout, y10, wt); if (col1) slab_store1(dst_b + base + T::D * a.Wout + T::D, y11, wt); }
      }
    }
  }
  if (wt) splitk_combine_tile<512>(a, b, sp_tile + a.n_sp * n_tile, n_tile * T::BN, T::BN, oy0, ox0, TH, TW, (unsigned*)smem);
}

// ---------------------------------------------------------------------------------------------------------
// Quarter-split form of the 8-wave kernel: the four ROWS of the 4x4 position grid go to four waves, and every wave
// covers all 64 output channels of the workgroup (two 32-row MFMA tiles) for its 32 blocks.  Same 8 accumulators,
// but each transformed input value now feeds two MFMAs, a wave needs only two window rows and one of the four row
// transforms: half the transform VALU and a third fewer LDS operand bytes per MFMA than the half-split form (the
// MFMA pipe is what both are bound by; tools/microbench/mfma_mix.hip shows what the operand traffic around a 64-cycle
// f32 MFMA costs).  The output transform is finished through LDS by all four quarters, each storing a quarter of the
// rows.  WM = WN = 2 tiles only (64 blocks x 64 channels).
template <class T>
__global__ __launch_bounds__(512, 2) void conv_wino4_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  static_assert(T::WM == 2 && T::WN == 2 && T::NBUF >= 3 && (T::KC / 2) % 2 == 0, "quarter split: 64x64 tiles, ring of 3");
  static_assert(T::XSZP % 512 == 0 && T::WSZP % 2048 == 0, "8-wave staging granularity");
  constexpr int NX2 = T::XSZP / 512, NW2 = T::WSZP / 2048, LOADS2 = NX2 + NW2;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = wave >> 1;                                   // row of the position grid
  const int wm = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;

  constexpr int TH = 2 * T::TTH, TW = 2 * T::TTW;
  const int tiles_x = (a.Wout + TW - 1) / TW;
  FDT_BLOCK_MAP(a, sp_tile, n_tile);
  const int oy0 = (sp_tile / tiles_x) * TH;
  const int ox0 = (sp_tile % tiles_x) * TW;
  const int b = blockIdx.z / a.ksplit;
  const int ks = blockIdx.z - b * a.ksplit;

  const int HW = a.Hin * a.Win;
  const float* in_b = a.in + (long long)b * conv_in_bstride(a);
  const int nstages = (a.Cin + T::KC - 1) / T::KC;
  const float* w_t = a.w + (long long)n_tile * nstages * T::WSZP;
  const int s_begin = (int)((long long)nstages * ks / a.ksplit);
  const int s_end = (int)((long long)nstages * (ks + 1) / a.ksplit);

  // staging plan: element e = 512 k + tid of the patch, fetched through a buffer descriptor at byte offset xoff[k] relative to
  // the stage's first channel (kOob for padding / out-of-image: zeros from the bounds check, like the channels past Cin)
  const __amdgpu_buffer_rsrc_t xrs = buf_rsrc(in_b, (long long)a.Cin * HW * 4);
  const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(w_t, 0x7fffffffll);
  unsigned xoff[NX2];
#pragma unroll
  for (int k = 0; k < NX2; ++k) {
    int e = tid + 512 * k;
    int c = e / T::XPLANE;
    int r = e - c * T::XPLANE;
    int yy = r / T::PW, xx = r - yy * T::PW;
    int gy = oy0 - T::D + yy, gx = ox0 - T::D + xx;
    bool ok = (e < T::XSZ) && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
    xoff[k] = ok ? (unsigned)(c * HW + gy * a.Win + gx) * 4u : kOob;
  }

#define FDT_W4STAGE(s_, buf_)                                                                  \
  {                                                                                           \
    const unsigned xso_ = (unsigned)((s_) * T::KC) * (unsigned)HW * 4u;                       \
    float* X_ = smem + (buf_) * T::STAGE + wave * 64;                                         \
    _Pragma("unroll") for (int k = 0; k < NX2; ++k) bglds4(xrs, X_ + 512 * k, xoff[k], xso_);   \
    const unsigned wso_ = (unsigned)((s_) * T::WSZP) * 4u;                                    \
    float* W_ = smem + (buf_) * T::STAGE + T::XSZP + wave * 256;                              \
    _Pragma("unroll") for (int k = 0; k < NW2; ++k) bglds16(wrs, W_ + 2048 * k, (unsigned)tid * 16u, wso_ + 8192u * k); \
  }

  const int qb = wm * 32 + l31;
  int oyl, oxl;
  if (T::D == 1) {
    oyl = 2 * (qb / T::TTW);
    oxl = 2 * (qb % T::TTW);
  } else {
    constexpr int CX = (2 * T::TTW) / 4;
    const int cell = qb >> 2, par = qb & 3;
    oyl = 4 * (cell / CX) + (par >> 1);
    oxl = 4 * (cell % CX) + (par & 1);
  }
  const int xo = half * T::XPLANE + oyl * T::PW + oxl;
  const int wo = T::XSZP + half * 16 * T::BN + q * 4 * T::BN + l31;

  f32x16 acc[8];                                               // [position column j][channel tile mt] = j * 2 + mt
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  const int nst = s_end - s_begin;
#pragma unroll
  for (int p = 0; p < 2; ++p)
    if (p < nst) FDT_W4STAGE(s_begin + p, p);

  auto main_loop = [&](auto q_c) {
    constexpr int Q_ = decltype(q_c)::value;
    // window rows this quarter needs: row Q_ of B^T d is  q0: d0 - d2,  q1: d1 + d2,  q2: d2 - d1,  q3: d1 - d3
    constexpr int RA = (Q_ == 0) ? 0 : 1, RB = (Q_ == 3) ? 3 : 2;
    struct Ops {
      f32x4 d1[2];          // D = 1: window rows RA, RB
      f32x2 d2[2][2];       // D = 2
      f32x2 u[2][2];        // weights [channel tile][position pair (0,1) / (2,3)]
    };
    constexpr int NLD = (T::D == 1 ? 2 : 4) + 4;
    constexpr int ROWB = T::D * T::PW * 4;
    static_assert(ROWB % 8 == 0 && (3 * ROWB) / 8 + 1 < 256 && 3 * (ROWB / 4) + 6 < 256 && T::BN == 64, "ds offset fields");
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    const unsigned xb = lds0 + (unsigned)xo * 4u, wb = lds0 + (unsigned)wo * 4u;
    auto load = [&](Ops& o, int buf, int cp) {
      const unsigned xa = xb + (unsigned)(buf * T::STAGE + 2 * cp * T::XPLANE) * 4u;
      const unsigned wa = wb + (unsigned)(buf * T::STAGE + 2 * cp * 16 * T::BN) * 4u;
      if constexpr (T::D == 1) {
        lds_read2_b64<RA * ROWB / 8, RA * ROWB / 8 + 1>(o.d1[0], xa);
        lds_read2_b64<RB * ROWB / 8, RB * ROWB / 8 + 1>(o.d1[1], xa);
      } else {
        lds_read2_b32<RA * ROWB / 4, RA * ROWB / 4 + 2>(o.d2[0][0], xa);
        lds_read2_b32<RA * ROWB / 4 + 4, RA * ROWB / 4 + 6>(o.d2[0][1], xa);
        lds_read2_b32<RB * ROWB / 4, RB * ROWB / 4 + 2>(o.d2[1][0], xa);
        lds_read2_b32<RB * ROWB / 4 + 4, RB * ROWB / 4 + 6>(o.d2[1][1], xa);
      }
      lds_read2st64_b32<0, 1>(o.u[0][0], wa);
      lds_read2st64_b32<2, 3>(o.u[0][1], wa);
      lds_read2st64_b32<0, 1>(o.u[1][0], wa + 128u);          // channel tile 1: +32 floats
      lds_read2st64_b32<2, 3>(o.u[1][1], wa + 128u);
    };
    auto wait_for = [&](Ops& o, auto newer_c) {
      constexpr int N_ = decltype(newer_c)::value;
      if constexpr (T::D == 1)
        asm volatile("s_waitcnt lgkmcnt(%6)"
                     : "+v"(o.d1[0]), "+v"(o.d1[1]), "+v"(o.u[0][0]), "+v"(o.u[0][1]), "+v"(o.u[1][0]), "+v"(o.u[1][1])
                     : "n"(N_));
      else
        asm volatile("s_waitcnt lgkmcnt(%8)"
                     : "+v"(o.d2[0][0]), "+v"(o.d2[0][1]), "+v"(o.d2[1][0]), "+v"(o.d2[1][1]), "+v"(o.u[0][0]),
                       "+v"(o.u[0][1]), "+v"(o.u[1][0]), "+v"(o.u[1][1])
                     : "n"(N_));
    };
    auto mac_ops = [&](const Ops& o) {
      float da[4], db[4], t[4], v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (T::D == 1) {
          da[j] = o.d1[0][j];
          db[j] = o.d1[1][j];
        } else {
          da[j] = o.d2[0][j >> 1][j & 1];
          db[j] = o.d2[1][j >> 1][j & 1];
        }
        t[j] = (Q_ == 1) ? (da[j] + db[j]) : (Q_ == 2) ? (db[j] - da[j]) : (da[j] - db[j]);
      }
      v[0] = t[0] - t[2]; v[1] = t[1] + t[2]; v[2] = t[2] - t[1]; v[3] = t[1] - t[3];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          acc[j * 2 + mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.u[mt][j >> 1][j & 1], v[j], acc[j * 2 + mt], 0, 0, 0);
    };
    using N0 = std::integral_constant<int, 0>;
    using NL = std::integral_constant<int, NLD>;

    if (nst > 1)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS2) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    Ops A, B;
    int cur = 0;
    load(A, 0, 0);
    for (int it = 0; it < nst; ++it) {
      const int nb = (cur + 1 == 3) ? 0 : cur + 1;
#pragma unroll
      for (int cp = 0; cp < T::KC / 2; cp += 2) {
        if (cp == 2 || T::KC / 2 == 2) {
          if (it + 1 < nst) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const int fb = (nb + 1 == 3) ? 0 : nb + 1;
            if (it + 2 < nst) FDT_W4STAGE(s_begin + it + 2, fb);
          }
        }
        load(B, cur, cp + 1);
        wait_for(A, NL{});
        mac_ops(A);
        if (cp + 2 < T::KC / 2) load(A, cur, cp + 2);
        else load(A, nb, 0);
        wait_for(B, NL{});
        mac_ops(B);
      }
      cur = nb;
    }
    wait_for(A, N0{});
  };
  switch (q) {
    case 0: main_loop(std::integral_constant<int, 0>{}); break;
    case 1: main_loop(std::integral_constant<int, 1>{}); break;
    case 2: main_loop(std::integral_constant<int, 2>{}); break;
    default: main_loop(std::integral_constant<int, 3>{}); break;
  }
#undef FDT_W4STAGE

  // ---- output transform Y = A^T M A: this wave holds row q of M for (channel tile mt, register r) --------------
  //   (t0, t1) = (m0 + m1 + m2, m1 - m2 - m3);  y0* = t(q0) + t(q1) + t(q2),  y1* = t(q1) - t(q2) - t(q3)
  __syncthreads();                       // ring is dead; reuse it as the exchange buffer [wave][pair][2][lane]
  float* E = smem;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0 * 2 + mt][r], m1 = acc[1 * 2 + mt][r], m2 = acc[2 * 2 + mt][r], m3 = acc[3 * 2 + mt][r];
      const int p = mt * 16 + r;
      E[((wave * 32 + p) * 2 + 0) * 64 + lane] = m0 + m1 + m2;
      E[((wave * 32 + p) * 2 + 1) * 64 + lane] = m1 - m2 - m3;
    }
  __syncthreads();
  const int HWo = a.Hout * a.Wout;
  const int oy = oy0 + oyl, ox = ox0 + oxl;
  const bool raw = a.ws != nullptr;
  const bool wt = raw && a.sk_count;   // slabs of an in-kernel combine are stored write-through (conv.h)
  float* dst_b = raw ? a.ws + ((long long)(b * a.ksplit + ks) * a.Cout) * HWo
                     : a.out + ((long long)b * a.out_ctot + a.out_coff) * HWo;
  const float* res_b = (!raw && a.res) ? a.res + ((long long)b * a.res_ctot + a.res_coff) * HWo : nullptr;
  const bool row0 = oy < a.Hout, row1 = oy + T::D < a.Hout;
  const bool col0 = ox < a.Wout, col1 = ox + T::D < a.Wout;
  const bool vec2 = (T::D == 1) && (a.Wout % 2 == 0);
  // this wave finishes the pairs with (r & 3) == q: a quarter of the rows, every quarter stores
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = rr * 4 + q;
      const int p = mt * 16 + r;
      float t0[4], t1[4];
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        t0[qq] = E[(((qq * 2 + wm) * 32 + p) * 2 + 0) * 64 + lane];
        t1[qq] = E[(((qq * 2 + wm) * 32 + p) * 2 + 1) * 64 + lane];
      }
      float y00 = t0[0] + t0[1] + t0[2], y01 = t1[0] + t1[1] + t1[2];
      float y10 = t0[1] - t0[2] - t0[3], y11 = t1[1] - t1[2] - t1[3];
      const int co = n_tile * T::BN + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co < a.Cout && col0) {
        const long long base = (long long)co * HWo + (long long)oy * a.Wout + ox;
        if (!raw) {
          const float bv = a.bias ? a.bias[co] : 0.0f;
          y00 += bv; y01 += bv; y10 += bv; y11 += bv;
          if (res_b) {
            if (row0) { y00 += res_b[base]; if (col1) y01 += res_b[base + T::D]; }
            if (row1) { y10 += res_b[base + T::D * a.Wout]; if (col1) y11 += res_b[base + T::D * a.Wout + T::D]; }
          }
          if (a.act == ACT_RELU) {
            y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f);
          } else if (a.act == ACT_RELU6) {
            y00 = fminf(fmaxf(y00, 0.f), 6.f); y01 = fminf(fmaxf(y01, 0.f), 6.f);
            y10 = fminf(fmaxf(y10, 0.f), 6.f); y11 = fminf(fmaxf(y11, 0.f), 6.f);
          }
        }
        if (vec2) {
          if (row0) slab_store2(dst_b + base, y00, y01, wt);
          if (row1) slab_store2(dst_b + base + a.Wout, y10, y11, wt);
        } else {
          if (row0) { slab_store1(dst_b + base, y00, wt); if (col1) slab_store1(dst_b + base + T::D, y01, wt); }
          if (row1) { slab_store1(dst_b + base + T::D * a.Wout, y10, wt); if (col1) slab_store1(dst_b + base + T::D * a.Wout + T::D, y11, wt); }
        }
      }
    }
  if (wt) splitk_combine_tile<512>(a, b, sp_tile + a.n_sp * n_tile, n_tile * T::BN, T::BN, oy0, ox0, TH, TW, (unsigned*)smem);
}

//                           TTH TTW WM WN KC NBUF      patch (px)  couts
using W_64x64    = WinoTile<8, 8, 2, 2, 8, 2>;     //  16 x 16     64
using W_64x64R3  = WinoTile<8, 8, 2, 2, 8, 3>;
using W_128x32   = WinoTile<8, 16, 4, 1, 8, 2>;    //  16 x 32     32
using W_128x32R3 = WinoTile<8, 16, 4, 1, 8, 3>;
using W_32x128   = WinoTile<4, 8, 1, 4, 8, 2>;     //   8 x 16    128
using W_64x64W   = WinoTile<4, 16, 2, 2, 8, 3>;    //   8 x 32     64  (wide rows)
// dilation 2 (pad 2): SSH conv2 / conv2_2 (pyramid.py:36,38)
using WD2_64x64    = WinoTile<8, 8, 2, 2, 8, 2, 2>;
using WD2_64x64R3  = WinoTile<8, 8, 2, 2, 8, 3, 2>;
using WD2_128x32R3 = WinoTile<8, 16, 4, 1, 8, 3, 2>;
using WD2_64x64W   = WinoTile<4, 16, 2, 2, 8, 3, 2>;

template <class T>
KernelEntry wino_entry() {
  return KernelEntry{conv_wino_kernel<T>, T::LDS_BYTES, 256};
}
template <class T>
KernelEntry wino4_entry() {
  constexpr size_t ex = 8 * 32 * 2 * 64 * sizeof(float);   // epilogue exchange buffer: every wave's (t0, t1) pairs
  return KernelEntry{conv_wino4_kernel<T>, T::LDS_BYTES > ex ? T::LDS_BYTES : ex, 512};
}
template <class T>
KernelEntry wino2_entry() {
  constexpr size_t ex = 4 * 16 * 4 * 64 * sizeof(float);   // epilogue exchange buffer
  return KernelEntry{conv_wino2_kernel<T>, T::LDS_BYTES > ex ? T::LDS_BYTES : ex, 512};
}

}  // namespace

void conv_fill_wino(void* row);
void conv_fill_wino_d2(void* row);

}  // namespace fdt
