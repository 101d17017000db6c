// Fused head of a MobileNetV2 InvertedResidual block: 1x1 expand + BN + ReLU6 followed by the depthwise 3x3 (stride 1 | 2,
// pad 1) + BN + ReLU6 (reference pyramid_mb2_try3.py:96-114: conv[0..5] of `self.conv`), in ONE kernel.
//
// Why: at 512^2 / 256^2 the expanded tensor is the whole cost of the block -- features.2 of try3 at batch 8 writes and
// re-reads 805 MB of it for 134 MB of input and 201 MB of output (both launches run at 3.6-4.7 TB/s, i.e. at the HBM
// roof).  Here the expanded activations never leave the CU:
//   * a workgroup owns a TH x 32 output tile of one image; the (TH-1)*S+3 rows x 31*S+3 columns of the Cin-channel input
//     under it are staged once in LDS;
//   * per chunk of 32 hidden channels the expand is a [32 x Cin] x [Cin x positions] GEMM on v_mfma_f32_32x32x2_f32
//     (the chunk's weights transposed into LDS as the A operand, input positions as the B operand straight from the patch;
//     ascending-k fmaf chain, + bias, ReLU6 == the stand-alone conv kernel's arithmetic), written to a second LDS tile --
//     ZERO where the position lies outside the image, because the depthwise conv pads the EXPANDED map;
//   * the depthwise 3x3 reads that tile (taps in the stand-alone kernel's order, + bias, ReLU6) and stores 16-byte
//     strips of the output.
// 50-80 KB of LDS per workgroup -> two or three workgroups per CU, so one's GEMM phase overlaps the other's stencil
// phase.
// Round 4 (PROJ): the 1x1 project conv + BN (+ the residual of a stride-1, inp == oup block; pyramid_mb2_try3.py:115-134) runs in
// the same kernel for oup <= 32: the depthwise output of a chunk stays in LDS (it overwrites the chunk's rows of the expanded
// tile, which its own wave has just consumed) and feeds a [oup x 32] x [32 x pixels] GEMM on the f32 MFMA whose accumulator
// lives across the chunks -- same k pairing and order as the stand-alone 1x1 kernel, so the block's output is bit-identical;
// the residual is read from the staged input patch.  The hidden tensor (6x the input) then never reaches HBM at all, and the
// block is one launch instead of two.
#include <atomic>

#include "common.h"
#include "ops.h"

#ifndef FDT_IR_EXP
#define FDT_IR_EXP 0   // tuning experiments (tools/experiments/ir_variants.sh): 1 no GEMM, 2 no depthwise phase, 3 no hs writes, 4 no staging
#endif

namespace fdt {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ float g_ir_zero[4];   // source of every out-of-image / padding element of the staged patch (zero-initialised)

template <int S>
struct IrTile {
  static constexpr int TH = (S == 1) ? 4 : 2;              // output rows per workgroup
  static constexpr int TW = 32;                            // output columns
  static constexpr int PH = (TH - 1) * S + 3;              // input rows under the tile
  static constexpr int PWR = (TW - 1) * S + 3;             // input columns under the tile
  static constexpr int PWP = (PWR + 3) / 4 * 4;            // row pitch in LDS
  static constexpr int NPOS = PH * PWP;
  static constexpr int NPOSP = (NPOS + 31) / 32 * 32;      // whole MFMA column tiles
  static constexpr int NT = NPOSP / 32;
};

// grid: (tiles_x * tiles_y, 1, B); 256 threads; dynamic LDS: (Cin + 32) * NPOSP + 32 * Cin floats.
// KS = Cin / 2 when known at compile time (the GEMM loop is then fully unrolled: all operand reads of a column tile are in
// flight before its first MFMA), 0 = runtime loop.
template <int S, int KS, bool PROJ = false>
__global__ __launch_bounds__(256, (S == 1 && KS > 0 && KS <= 12 && !PROJ) ? 3 : 1) void expand_dw_kernel(const float* __restrict__ x, int Cin, int H, int W,
                                                        const float* __restrict__ w1, const float* __restrict__ b1,
                                                        const float* __restrict__ wdw, const float* __restrict__ bdw,
                                                        int hid, float* __restrict__ out, int Ho, int Wo,
                                                        const float* __restrict__ wp = nullptr, const float* __restrict__ bp = nullptr,
                                                        int oup = 0, int residual = 0, int total_tiles = 0, int tiles_per_wg = 1) {
  using T = IrTile<S>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                                   // [Cin][NPOSP]
  float* hs = smem + (size_t)Cin * T::NPOSP;          // [32][NPOSP]
  float* ws = hs + 32 * T::NPOSP;                     // [Cin][32]: this chunk's expand weights, k-major (A operand)
  float* wps = ws + 32 * Cin;                         // PROJ: [2][32 k][32 oup] project weights of the chunk, double buffered
  constexpr int NPX = T::TH * T::TW;                  // output pixels of the tile: 128 (stride 1) / 64 (stride 2)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int tiles_x = (Wo + T::TW - 1) / T::TW, tiles_y = (Ho + T::TH - 1) / T::TH;
  const int n_sp = tiles_x * tiles_y;
  // PERSISTENT (round 4): the workgroup walks the (image, tile) pairs ti0 .. ti1 - 1 (launch_expand_dw sizes the grid for the
  // workgroups a CU holds).  When Cin is a template constant and the rows are 16-byte aligned (W % 4 == 0) the patch of the NEXT
  // tile is fetched as 16-byte pieces into registers right after this tile's patch has become visible, and written to LDS when
  // the tile is done: the staging that one-tile workgroups exposed (features.2: 279 -> 210 us without it) runs under the chunks.
  // The barriers of the loop wait for LDS only (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() would also wait for vmcnt(0),
  // i.e. for the prefetch.
  const int ti0 = blockIdx.x * tiles_per_wg, ti1 = min(ti0 + tiles_per_wg, total_tiles);
  constexpr int NQ = (T::PWR + 3 + 3) / 4;              // 16-byte pieces per patch row, from the aligned column ox0 * S - 4
  constexpr bool VEC = KS > 0 && !PROJ;              // (the whole-block form keeps its registers for the project accumulator)
  constexpr int NPC = VEC ? 2 * KS * T::PH * NQ : 1, NITP = (NPC + 255) / 256;
  const bool vec = VEC && (W & 3) == 0;
  f32x4 pv[NITP];
  auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto fetch = [&](int ti) {
    const int b_ = ti / n_sp, sp_ = ti - b_ * n_sp;
    const int gy0_ = (sp_ / tiles_x) * T::TH * S - 1, gxa_ = (sp_ % tiles_x) * T::TW * S - 4;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long long)b_ * Cin * H * W), 0,
                                                                          (int)min((long long)Cin * H * W * 4, 0x7fffffffll), 0x00020000);
#pragma unroll
    for (int k = 0; k < NITP; ++k) {
      const int i = tid + 256 * k;
      const int r = i / NQ, q = i - r * NQ;
      const int c = r / T::PH, py = r - c * T::PH;
      const int gy = gy0_ + py, gx = gxa_ + 4 * q;
      const bool ok = i < NPC && gy >= 0 && gy < H && gx >= 0 && gx < W;       // W % 4 == 0: a piece is wholly inside or outside
      const unsigned vo = ok ? (unsigned)((c * H + gy) * W + gx) * 4u : 0x80000000u;
      pv[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, 0, 0));
    }
  };
  if (vec && ti0 < ti1) fetch(ti0);
  if (VEC) {      // the pitch / tile padding of the patch is never written by the pieces: zero it once (it only feeds positions that
                  // the expand phase zeroes again, but uninitialised LDS may hold NaNs)
    for (int e = tid; e < Cin * T::NPOSP; e += 256) xs[e] = 0.0f;
    lds_barrier();
  }

  for (int ti = ti0; ti < ti1; ++ti) {
  const int b = ti / n_sp, sp = ti - b * n_sp;
  const int oy0 = (sp / tiles_x) * T::TH, ox0 = (sp % tiles_x) * T::TW;
  const int gy0 = oy0 * S - 1, gx0 = ox0 * S - 1;
  const float* xb = x + (long long)b * Cin * H * W;
  f32x16 pacc;                                        // PROJ: out[oup rows][32 pixels of this wave's tile row], across the chunks
#pragma unroll
  for (int r = 0; r < 16; ++r) pacc[r] = 0.0f;

  if (vec) {
    // piece (row r, column group q) holds the patch columns 4 q - 3 .. 4 q of row r (the patch starts at image column ox0 * S - 1)
#pragma unroll
    for (int k = 0; k < NITP; ++k) {
      const int i = tid + 256 * k;
      if (i < NPC) {
        const int r = i / NQ, q = i - r * NQ;
        const int c = r / T::PH, py = r - c * T::PH;
        float* d = xs + c * T::NPOSP + py * T::PWP + 4 * q - 3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int px = 4 * q - 3 + e;
          if (px >= 0 && px < T::PWR) d[e] = pv[k][e];
        }
      }
    }
  } else {
    // ---- stage the input patch by LDS-DMA (nothing passes through VGPRs, all loads in flight together); out-of-image and
    // pitch-padding elements read a zero word.  Element e = 256 * k + tid lands at float e: wave-uniform base + lane * 4.
    const float* zpad = g_ir_zero;
    const int total = Cin * T::NPOSP;
    for (int e0 = 0; e0 < (FDT_IR_EXP == 4 ? 0 : total); e0 += 256) {
      const int e = e0 + tid;
      const int c = e / T::NPOSP, p = e - c * T::NPOSP;
      const int py = p / T::PWP, px = p - py * T::PWP;
      const int gy = gy0 + py, gx = gx0 + px;
      const bool ok = e < total && p < T::NPOS && px < T::PWR && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const float* src = ok ? xb + ((long long)c * H + gy) * W + gx : zpad;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(xs + e0 + wave * 64), 4, 0, 0);
    }
  }
  const int nchunks = (hid + 31) / 32;
  const int ksteps = KS ? KS : (Cin >> 1);

  for (int ch = 0; ch < nchunks; ++ch) {
    // this chunk's expand biases (the 16 accumulator rows of the lane) and the thread's depthwise taps: requested now,
    // consumed after the barrier / the GEMM phase
    float br[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int hc = ch * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      br[r] = hc < hid ? b1[hc] : 0.0f;
    }
    const int cl = tid >> 3, cg = tid & 7;            // depthwise phase: 32 channels x 8 column groups
    const int hcd = ch * 32 + cl;
    const int hcc = hcd < hid ? hcd : hid - 1;
    float k9[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k9[i] = wdw[(long long)hcc * 9 + i];
    const float bb = bdw[hcc];
    // this chunk's 32 x Cin expand weights -> LDS, transposed so that a wave reads one k-row per operand (staging all
    // chunks at once costs a workgroup per CU on the 144-channel blocks: measured slower)
    for (int e = tid; e < 32 * Cin; e += 256) {
      const int k = e >> 5, rr = e & 31;
      const int hw = ch * 32 + rr;
      ws[e] = hw < hid ? w1[(long long)hw * Cin + k] : 0.0f;
    }
    if constexpr (PROJ) {   // wps[k][o] = Wp[o][chunk's hidden channel k]; zeros past oup / hid (the padded rows then add 0)
      float* wq = wps + (ch & 1) * 1024;
      for (int e = tid; e < 1024; e += 256) {
        const int k = e >> 5, o = e & 31;
        const int hw = ch * 32 + k;
        wq[e] = (o < oup && hw < hid) ? wp[(long long)o * hid + hw] : 0.0f;
      }
    }
    if (!vec && ch == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA of the patch has landed
    lds_barrier();
    if (vec && ch == 0 && ti + 1 < ti1) fetch(ti + 1);   // the next tile's patch: in flight under this tile's chunks
    // ---- expand: hs[32][positions] = ReLU6(W1[chunk] . xs + b1), column tiles dealt round-robin to the four waves
    // A wave owns the column tiles wave, wave + 4, ...: their accumulation chains are independent, so they are advanced
    // together (one dependent MFMA chain alone leaves the matrix pipe idle 3/4 of the time at one wave per SIMD); the A
    // operand (weights) is shared by all of them.
    {
      constexpr int JT = (T::NT + 3) / 4;
      f32x16 acc[JT];
#pragma unroll
      for (int t = 0; t < JT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
      const float* acol = ws + half * 32 + l31;
      const float* bcol = xs + wave * 32 + l31 + (size_t)half * T::NPOSP;      // tile t: + t * 128 positions
      if constexpr (FDT_IR_EXP == 1) {
      } else if constexpr (KS > 0) {
        float av[KS], bw[JT][KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          av[s] = acol[s * 64];
#pragma unroll
          for (int t = 0; t < JT; ++t) bw[t][s] = bcol[(size_t)s * 2 * T::NPOSP + t * 128];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int t = 0; t < JT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bw[t][s], acc[t], 0, 0, 0);
      } else {
        for (int s = 0; s < ksteps; ++s) {
          const float a_ = acol[s * 64];
#pragma unroll
          for (int t = 0; t < JT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, bcol[(size_t)s * 2 * T::NPOSP + t * 128], acc[t], 0, 0, 0);
        }
      }
#pragma unroll
      for (int t = 0; t < JT; ++t) {
        const int j = wave + 4 * t;
        if (j >= T::NT) continue;
        const int p = j * 32 + l31;
        const int py = p / T::PWP, px = p - py * T::PWP;
        const int gy = gy0 + py, gx = gx0 + px;
        const bool inside = p < T::NPOS && px < T::PWR && gy >= 0 && gy < H && gx >= 0 && gx < W;
        // three passes over the 16 rows (independent instructions back to back) instead of one dependent
        // add -> clamp -> select -> store chain per row, which an in-order SIMD with two waves cannot hide
        float vv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[r] = acc[t][r] + br[r];
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[r] = fminf(fmaxf(vv[r], 0.0f), 6.0f);
        if (__ballot(!inside) != 0ull) {        // border tile: the depthwise conv zero-pads the EXPANDED map
#pragma unroll
          for (int r = 0; r < 16; ++r) vv[r] = inside ? vv[r] : 0.0f;
        }
        float* hp = hs + 4 * half * T::NPOSP + p;
#pragma unroll
        for (int r = 0; r < (FDT_IR_EXP == 3 ? 1 : 16); ++r) hp[(size_t)((r & 3) + 8 * (r >> 2)) * T::NPOSP] = vv[r];
      }
    }
    lds_barrier();

    // ---- depthwise 3x3: thread = (channel of the chunk, strip of 4 output columns), all TH rows of the tile
    {
      const int hc = hcd;
      if (hc < hid && FDT_IR_EXP != 2) {
        const float* hrow = hs + (size_t)cl * T::NPOSP + cg * 4 * S;
        float acc[T::TH][4];
#pragma unroll
        for (int o = 0; o < T::TH; ++o)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[o][q] = 0.0f;
        constexpr int NV = 3 * S + 3;                 // input columns per strip: S = 1: 6, S = 2: 9
#pragma unroll
        for (int r = 0; r < T::PH; ++r) {
          float v[NV];
          {   // 16-byte aligned strip start (cg * 4 * S floats, pitches are multiples of 4): two or three vector reads
            const float4 q0 = *reinterpret_cast<const float4*>(hrow + r * T::PWP);
            v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w;
            if (S == 1) {
              const float2 q1 = *reinterpret_cast<const float2*>(hrow + r * T::PWP + 4);
              v[4] = q1.x; v[5] = q1.y;
            } else {
              const float4 q1 = *reinterpret_cast<const float4*>(hrow + r * T::PWP + 4);
              v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
              v[NV - 1] = hrow[r * T::PWP + 8];
            }
          }
#pragma unroll
          for (int o = 0; o < T::TH; ++o) {
            const int dy = r - o * S;
            if (dy < 0 || dy > 2) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[o][q] = fmaf(v[q * S + dx], k9[dy * 3 + dx], acc[o][q]);
          }
        }
        const int ox = ox0 + cg * 4;
        float* ob = out + ((long long)b * hid + hc) * Ho * Wo;
        if constexpr (PROJ) {
          // the chunk's depthwise output stays in LDS as the project GEMM's B operand, in the first NPX floats of this
          // channel's row of the expanded tile: the eight threads of a channel sit in one wave and have all read their
          // strips above (LDS operations of a wave complete in order), no other wave touches this row
          __builtin_amdgcn_wave_barrier();
          float* dsr = hs + (size_t)cl * T::NPOSP + cg * 4;
#pragma unroll
          for (int o = 0; o < T::TH; ++o) {
            float4 y;
            y.x = fminf(fmaxf(acc[o][0] + bb, 0.0f), 6.0f); y.y = fminf(fmaxf(acc[o][1] + bb, 0.0f), 6.0f);
            y.z = fminf(fmaxf(acc[o][2] + bb, 0.0f), 6.0f); y.w = fminf(fmaxf(acc[o][3] + bb, 0.0f), 6.0f);
            *reinterpret_cast<float4*>(dsr + o * T::TW) = y;
          }
        } else
#pragma unroll
        for (int o = 0; o < T::TH; ++o) {
          const int oy = oy0 + o;
          if (oy >= Ho) continue;
          float y[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) y[q] = fminf(fmaxf(acc[o][q] + bb, 0.0f), 6.0f);
          if (ox + 3 < Wo && (Wo & 3) == 0) {
            *reinterpret_cast<float4*>(ob + (long long)oy * Wo + ox) = make_float4(y[0], y[1], y[2], y[3]);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (ox + q < Wo) ob[(long long)oy * Wo + ox + q] = y[q];
          }
        }
      }
    }
    lds_barrier();
    if constexpr (PROJ) {
      // ---- project: pacc[oup][pixels] += Wp[:, chunk] . dw[chunk][pixels]; wave w owns tile row w (32 pixels); channel
      // pairs ascending = the k order of the stand-alone 1x1 kernel.  (The padded hidden channels of a last, partly filled
      // chunk hold ReLU6(0 + 0) = 0 from the expand phase and zero weights here.)
      if (wave < T::TH) {
        const float* acol = wps + (ch & 1) * 1024 + half * 32 + l31;
        const float* bcol = hs + (size_t)half * T::NPOSP + wave * T::TW + l31;
        float av[16], bw[16];
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
          av[s2] = acol[s2 * 64];
          bw[s2] = bcol[(size_t)s2 * 2 * T::NPOSP];
        }
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], bw[s2], pacc, 0, 0, 0);
      }
      // no barrier here: the next chunk's expand writes hs only behind its own barrier, which every wave reaches after this
      // phase; the project weights are double buffered
    }
  }
  if constexpr (PROJ) {
    // ---- epilogue of the block: + folded BN bias, + residual (the block's own input, from the staged patch), no activation
    // (pyramid_mb2_try3.py:115-117: linear bottleneck); lane = pixel l31 of tile row `wave`, register r = output channel
    if (wave < T::TH) {
      const int oy = oy0 + wave, ox = ox0 + l31;
      if (oy < Ho && ox < Wo) {
        float* ob = out + (long long)b * oup * Ho * Wo + (long long)oy * Wo + ox;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int o = (r & 3) + 8 * (r >> 2) + 4 * half;
          if (o < oup) {
            float v = pacc[r] + bp[o];
            if (residual) v += xs[(size_t)o * T::NPOSP + (wave + 1) * T::PWP + l31 + 1];   // stride 1, inp == oup: x[o][oy][ox]
            ob[(long long)o * Ho * Wo] = v;
          }
        }
      }
    }
  }
  lds_barrier();      // every wave is done with xs (the residual) and hs: the next tile's patch may overwrite them
  }   // tiles of this workgroup
}

}  // namespace

size_t expand_dw_lds_bytes(int Cin, int stride, int hid) {
  const int npp = stride == 1 ? IrTile<1>::NPOSP : IrTile<2>::NPOSP;
  (void)hid;
  return ((size_t)(Cin + 32) * npp + 32 * (size_t)Cin) * sizeof(float);
}
size_t ir_block_lds_bytes(int Cin, int stride, int hid) { return expand_dw_lds_bytes(Cin, stride, hid) + 2 * 1024 * sizeof(float); }

// tiles a persistent workgroup walks: the grid fills every CU with the workgroups its LDS holds (at most four: 16 waves), and all of
// them get the same number of tiles (+-1 at the end)
static int ir_tiles_per_wg(long long total, size_t lds, int dev) {
  int cus = 256;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const int resident = (int)std::min<size_t>(4, std::max<size_t>(1, (160 * 1024) / lds));
#ifdef FDT_EXPERIMENTS   // tools/experiments/ir_persistent.sh: 0 = one tile per workgroup (the round-3 launch)
  if (const char* e = getenv("FDT_IR_TILES")) { const int v = atoi(e); if (v > 0) return v; }
#endif
  return (int)std::max<long long>(1, (total + (long long)cus * resident - 1) / ((long long)cus * resident));
}

int launch_expand_dw(const float* x, int B, int Cin, int H, int W, const float* w1, const float* b1, const float* wdw,
                     const float* bdw, int hid, int stride, float* out, int Ho, int Wo, hipStream_t st, int dev) {
  FDT_REQUIRE(stride == 1 || stride == 2, FDT_ERR_ARG, "expand_dw: stride %d", stride);
  FDT_REQUIRE(Cin >= 2 && (Cin & 1) == 0 && hid >= 1 && B >= 1 && B <= 65535, FDT_ERR_ARG, "expand_dw: bad channel counts");
  FDT_REQUIRE(Ho == (H - 1) / stride + 1 && Wo == (W - 1) / stride + 1, FDT_ERR_ARG, "expand_dw: output size mismatch");
  const size_t lds = expand_dw_lds_bytes(Cin, stride, hid);
  FDT_REQUIRE(lds <= 160 * 1024, FDT_ERR_ARG, "expand_dw: %d -> %d channels do not fit LDS", Cin, hid);
  const int th = stride == 1 ? IrTile<1>::TH : IrTile<2>::TH;
  if (dev < 0) FDT_HIP(hipGetDevice(&dev));
  const long long total = (long long)ceil_div(Wo, 32) * ceil_div(Ho, th) * B;
  FDT_REQUIRE(total <= 0x7fffffffll, FDT_ERR_ARG, "expand_dw: grid too large");
  const int tpw = ir_tiles_per_wg(total, lds, dev);
  dim3 grid((unsigned)((total + tpw - 1) / tpw));
  typedef void (*Kern)(const float*, int, int, int, const float*, const float*, const float*, const float*, int, float*, int,
                       int, const float*, const float*, int, int, int, int);
  const int ks = Cin >> 1;
  Kern fn = nullptr;
  if (stride == 1)
    fn = ks == 8 ? expand_dw_kernel<1, 8> : ks == 12 ? expand_dw_kernel<1, 12> : ks == 16 ? expand_dw_kernel<1, 16>
                                                                                        : expand_dw_kernel<1, 0>;
  else
    fn = ks == 8 ? expand_dw_kernel<2, 8> : ks == 12 ? expand_dw_kernel<2, 12> : ks == 16 ? expand_dw_kernel<2, 16>
                                                                                        : expand_dw_kernel<2, 0>;
  // per-(function, device) attribute; set on every launch of a not-yet-seen pair (idempotent, a race sets it twice)
  static std::atomic<unsigned long long> seen[16];
  if (dev < 0) FDT_HIP(hipGetDevice(&dev));
  const int slot = (stride - 1) * 4 + (ks == 8 ? 0 : ks == 12 ? 1 : ks == 16 ? 2 : 3);
  if (dev < 16 && !((seen[dev].load(std::memory_order_acquire) >> slot) & 1ull)) {
    FDT_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    seen[dev].fetch_or(1ull << slot, std::memory_order_release);
  } else if (dev >= 16) {
    FDT_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, x, Cin, H, W, w1, b1, wdw, bdw, hid, out, Ho, Wo, (const float*)nullptr,
                     (const float*)nullptr, 0, 0, (int)total, tpw);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

// The whole InvertedResidual block (expand + depthwise + project (+ residual)) as one launch; oup <= 32.
int launch_ir_block(const float* x, int B, int Cin, int H, int W, const float* w1, const float* b1, const float* wdw,
                    const float* bdw, int hid, int stride, const float* wp, const float* bp, int oup, int residual, float* out,
                    int Ho, int Wo, hipStream_t st, int dev) {
  FDT_REQUIRE(stride == 1 || stride == 2, FDT_ERR_ARG, "ir_block: stride %d", stride);
  FDT_REQUIRE(Cin >= 2 && (Cin & 1) == 0 && hid >= 1 && B >= 1 && B <= 65535 && oup >= 1 && oup <= 32 && wp && bp, FDT_ERR_ARG,
              "ir_block: bad channel counts (the fused project handles up to 32 output channels)");
  FDT_REQUIRE(!residual || (stride == 1 && oup == Cin), FDT_ERR_ARG, "ir_block: a residual block has stride 1 and inp == oup");
  FDT_REQUIRE(Ho == (H - 1) / stride + 1 && Wo == (W - 1) / stride + 1, FDT_ERR_ARG, "ir_block: output size mismatch");
  const size_t lds = ir_block_lds_bytes(Cin, stride, hid);
  FDT_REQUIRE(lds <= 160 * 1024, FDT_ERR_ARG, "ir_block: %d -> %d channels do not fit LDS", Cin, hid);
  const int th = stride == 1 ? IrTile<1>::TH : IrTile<2>::TH;
  if (dev < 0) FDT_HIP(hipGetDevice(&dev));
  const long long total = (long long)ceil_div(Wo, 32) * ceil_div(Ho, th) * B;
  FDT_REQUIRE(total <= 0x7fffffffll, FDT_ERR_ARG, "ir_block: grid too large");
  const int tpw = ir_tiles_per_wg(total, lds, dev);
  dim3 grid((unsigned)((total + tpw - 1) / tpw));
  typedef void (*Kern)(const float*, int, int, int, const float*, const float*, const float*, const float*, int, float*, int,
                       int, const float*, const float*, int, int, int, int);
  const int ks = Cin >> 1;
  Kern fn = nullptr;
  if (stride == 1)
    fn = ks == 8 ? expand_dw_kernel<1, 8, true> : ks == 12 ? expand_dw_kernel<1, 12, true> : ks == 16 ? expand_dw_kernel<1, 16, true>
                                                                                                    : expand_dw_kernel<1, 0, true>;
  else
    fn = ks == 8 ? expand_dw_kernel<2, 8, true> : ks == 12 ? expand_dw_kernel<2, 12, true> : ks == 16 ? expand_dw_kernel<2, 16, true>
                                                                                                    : expand_dw_kernel<2, 0, true>;
  static std::atomic<unsigned long long> seen[16];
  if (dev < 0) FDT_HIP(hipGetDevice(&dev));
  const int slot = (stride - 1) * 4 + (ks == 8 ? 0 : ks == 12 ? 1 : ks == 16 ? 2 : 3);
  if (dev >= 16 || !((seen[dev].load(std::memory_order_acquire) >> slot) & 1ull)) {
    FDT_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (dev < 16) seen[dev].fetch_or(1ull << slot, std::memory_order_release);
  }
  hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, x, Cin, H, W, w1, b1, wdw, bdw, hid, out, Ho, Wo, wp, bp, oup, residual, (int)total, tpw);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

}  // namespace fdt

// Stand-alone op (host pointers): the whole InvertedResidual of pyramid_mb2_try3.py:96-134 with expand_ratio != 1 and its
// BatchNorms folded: y = BN(conv1x1_project(ReLU6(BN(dw3x3(ReLU6(BN(conv1x1_expand(x)))))))) (+ x for a stride-1, inp == oup block).
extern "C" int fdt_ir_block(const float* x, int B, int Cin, int H, int W, const float* w1, const float* b1, const float* wdw,
                            const float* bdw, int hid, int stride, const float* wp, const float* bp, int oup, int residual,
                            float* out) {
  const hipStream_t st = fdt::thread_stream();   // never the legacy stream (common.h)
  using namespace fdt;
  FDT_REQUIRE(x && w1 && b1 && wdw && bdw && wp && bp && out && B >= 1 && H >= 1 && W >= 1 && oup >= 1, FDT_ERR_ARG, "fdt_ir_block: bad argument");
  FDT_REQUIRE(stride == 1 || stride == 2, FDT_ERR_ARG, "fdt_ir_block: stride must be 1 or 2");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  FDT_REQUIRE(st, FDT_ERR_HIP, "%s: could not create the calling thread's private stream", __func__);
  DevBuf dx, dw1, db1, dwd, dbd, dwp, dbp, dout;
  const size_t nx = (size_t)B * Cin * H * W, no = (size_t)B * oup * Ho * Wo;
  FDT_TRY(dx.alloc(nx * 4)); FDT_TRY(dw1.alloc((size_t)hid * Cin * 4)); FDT_TRY(db1.alloc((size_t)hid * 4));
  FDT_TRY(dwd.alloc((size_t)hid * 36)); FDT_TRY(dbd.alloc((size_t)hid * 4)); FDT_TRY(dwp.alloc((size_t)oup * hid * 4));
  FDT_TRY(dbp.alloc((size_t)oup * 4)); FDT_TRY(dout.alloc(no * 4));
  FDT_HIP(copy_sync(dx.p, x, nx * 4, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(dw1.p, w1, (size_t)hid * Cin * 4, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(db1.p, b1, (size_t)hid * 4, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(dwd.p, wdw, (size_t)hid * 36, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(dbd.p, bdw, (size_t)hid * 4, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(dwp.p, wp, (size_t)oup * hid * 4, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(dbp.p, bp, (size_t)oup * 4, hipMemcpyHostToDevice, st));
  FDT_TRY(launch_ir_block(dx.as<float>(), B, Cin, H, W, dw1.as<float>(), db1.as<float>(), dwd.as<float>(), dbd.as<float>(), hid,
                          stride, dwp.as<float>(), dbp.as<float>(), oup, residual, dout.as<float>(), Ho, Wo, st, -1));
  FDT_HIP(copy_sync(out, dout.p, no * 4, hipMemcpyDeviceToHost, st));
  return FDT_OK;
}

// ---------------------------------------------------------------------------------------------------
// Stand-alone op (host pointers): ReLU6(BN(dw3x3(ReLU6(BN(conv1x1(x)))))) with the BatchNorms already folded into
// (w1, b1) and (wdw, bdw) -- what the fused kernel computes for conv[0..5] of an InvertedResidual with expand_ratio != 1
// (pyramid_mb2_try3.py:96-114).  The parity tests drive this entry point on odd sizes.
extern "C" int fdt_expand_dw(const float* x, int B, int Cin, int H, int W, const float* w1, const float* b1,
                             const float* wdw, const float* bdw, int hid, int stride, float* out) {
  const hipStream_t st = fdt::thread_stream();   // never the legacy stream (common.h)
  using namespace fdt;
  FDT_REQUIRE(x && w1 && b1 && wdw && bdw && out && B >= 1 && H >= 1 && W >= 1, FDT_ERR_ARG, "fdt_expand_dw: bad argument");
  FDT_REQUIRE(stride == 1 || stride == 2, FDT_ERR_ARG, "fdt_expand_dw: stride must be 1 or 2");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  FDT_REQUIRE(st, FDT_ERR_HIP, "%s: could not create the calling thread's private stream", __func__);
  DevBuf dx, dw1, db1, dwd, dbd, dout;
  const size_t nx = (size_t)B * Cin * H * W, no = (size_t)B * hid * Ho * Wo;
  FDT_TRY(dx.alloc(nx * 4)); FDT_TRY(dw1.alloc((size_t)hid * Cin * 4)); FDT_TRY(db1.alloc((size_t)hid * 4));
  FDT_TRY(dwd.alloc((size_t)hid * 36)); FDT_TRY(dbd.alloc((size_t)hid * 4)); FDT_TRY(dout.alloc(no * 4));
  FDT_HIP(copy_sync(dx.p, x, nx * 4, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(dw1.p, w1, (size_t)hid * Cin * 4, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(db1.p, b1, (size_t)hid * 4, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(dwd.p, wdw, (size_t)hid * 36, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(dbd.p, bdw, (size_t)hid * 4, hipMemcpyHostToDevice, st));
  FDT_TRY(launch_expand_dw(dx.as<float>(), B, Cin, H, W, dw1.as<float>(), db1.as<float>(), dwd.as<float>(),
                           dbd.as<float>(), hid, stride, dout.as<float>(), Ho, Wo, st, -1));
  FDT_HIP(copy_sync(out, dout.p, no * 4, hipMemcpyDeviceToHost, st));
  return FDT_OK;
}

// Tuning hook (not part of include/fdt.h): time the fused kernel on zero-filled device buffers with HIP events.
extern "C" int fdt_debug_expand_dw_bench(int B, int Cin, int H, int W, int hid, int stride, int iters, float* ms_out) {
  const hipStream_t st = fdt::thread_stream();   // never the legacy stream (common.h)
  using namespace fdt;
  FDT_REQUIRE(ms_out && iters >= 1 && (stride == 1 || stride == 2), FDT_ERR_ARG, "fdt_debug_expand_dw_bench: bad argument");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  FDT_REQUIRE(st, FDT_ERR_HIP, "%s: could not create the calling thread's private stream", __func__);
  DevBuf dx, dw1, db1, dwd, dbd, dout;
  const size_t nx = (size_t)B * Cin * H * W, no = (size_t)B * hid * Ho * Wo;
  FDT_TRY(dx.alloc(nx * 4)); FDT_TRY(dw1.alloc((size_t)hid * Cin * 4)); FDT_TRY(db1.alloc((size_t)hid * 4));
  FDT_TRY(dwd.alloc((size_t)hid * 36)); FDT_TRY(dbd.alloc((size_t)hid * 4)); FDT_TRY(dout.alloc(no * 4));
  FDT_HIP(hipMemsetAsync(dx.p, 0, nx * 4, st)); FDT_HIP(hipMemsetAsync(dw1.p, 0, (size_t)hid * Cin * 4, st));
  FDT_HIP(hipMemsetAsync(db1.p, 0, (size_t)hid * 4, st)); FDT_HIP(hipMemsetAsync(dwd.p, 0, (size_t)hid * 36, st));
  FDT_HIP(hipMemsetAsync(dbd.p, 0, (size_t)hid * 4, st));
  hipEvent_t e0, e1;
  FDT_HIP(hipEventCreate(&e0)); FDT_HIP(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i)
    FDT_TRY(launch_expand_dw(dx.as<float>(), B, Cin, H, W, dw1.as<float>(), db1.as<float>(), dwd.as<float>(),
                             dbd.as<float>(), hid, stride, dout.as<float>(), Ho, Wo, st, -1));
  FDT_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i)
    FDT_TRY(launch_expand_dw(dx.as<float>(), B, Cin, H, W, dw1.as<float>(), db1.as<float>(), dwd.as<float>(),
                             dbd.as<float>(), hid, stride, dout.as<float>(), Ho, Wo, st, -1));
  FDT_HIP(hipEventRecord(e1, st));
  FDT_HIP(hipEventSynchronize(e1));
  float ms = 0;
  FDT_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return FDT_OK;
}
