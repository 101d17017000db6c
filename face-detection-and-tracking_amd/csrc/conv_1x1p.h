// conv_1x1p.h -- the 1x1 / stride-1 convolution as a PERSISTENT-TILE kernel (round 4; kernel class CONV_1x1_S1_P).
//
// Why: the bottleneck 1x1 layers of the backbone (pyramid.py:97-103) spend as long writing their output tile (and reading
// the residual) as in their short reduction, and in the one-tile-per-workgroup form every workgroup of the (one-round)
// grid reaches its epilogue at the same time: the matrix pipe idles while 30-70 MB of stores drain, then the launch ends
// (PMC, profiles/r03: 30-42 % matrix-pipe busy).  Here a workgroup walks `tiles_per_wg` consecutive output tiles and the
// LDS ring simply keeps running across them:
//   * the first ring stages of tile t+1 are requested (LDS-DMA) during the last stages of tile t, in front of its epilogue;
//   * the epilogue works from the accumulator registers -- no LDS transpose, so the ring is never "dead": every lane stores
//     dwords with one buffer store per accumulator register (a 32x32 accumulator register is two whole 128-byte row
//     segments), the residual arrives the same way one stage ahead of its use;
//   * stores are fire-and-forget: the MFMAs of tile t+1 start while the stores of tile t drain.
// Everything that goes to memory is an UNCONDITIONAL buffer instruction (pixels outside the image carry an out-of-range
// offset, channels past Cout fall off the end of the descriptor), so the number of vector-memory instructions between two
// points of the program is a compile-time constant and every ring stage is retired with an exact s_waitcnt vmcnt(N) that
// leaves the younger stages -- and the stores behind them -- in flight.
// Same operand layouts, same MFMA order (channel pairs ascending) and the same epilogue arithmetic (acc + bias, + residual,
// activation) as conv_kernel<G_1x1_S1, .>: bit-identical outputs.
#pragma once
#include "conv_kernel.h"

namespace fdt {
namespace {

template <int KC_, int BN_, int NBUF_, int RES_>
struct P1x1 {
  static constexpr int KC = KC_, BN = BN_, NBUF = NBUF_;
  static constexpr int RESIDENT = RES_;                      // workgroups per CU the launch is sized for (= waves per SIMD)
  static constexpr int TH = 4, TW = 32, BM = 128, WM = 2, WN = 2;
  static constexpr int MI = BM / (WM * 32), NI = BN / (WN * 32);
  static constexpr int XSZ = KC * BM, WSZ = KC * BN, WSZP = (WSZ + 1023) / 1024 * 1024;
  static constexpr int STAGE = XSZ + WSZP;                   // floats per ring slot
  static constexpr int NXV = XSZ / 1024, NW = WSZP / 1024, LOADS = NXV + NW;   // LDS-DMA wave-instructions per stage
  static constexpr int NACC = MI * NI * 16;                  // accumulator registers = stores (and residual loads) per tile
  static constexpr size_t LDS_BYTES = (size_t)NBUF * STAGE * sizeof(float);
  static_assert(XSZ % 1024 == 0, "the pixel tile is staged with whole dwordx4 wave-instructions");
  static_assert(NBUF == 3, "ring of three (the host checks nst >= NBUF - 1 = 2)");
  // (vmcnt is 6 bits: wait_vm clamps a count above 63, which is safe -- it only waits for a few more of the younger operations)
};

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 63 ? N : 63) : "memory");
}

template <class P>
__global__ __launch_bounds__(256, P::RESIDENT) void conv1x1p_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / P::WN, wn = wave % P::WN;
  const int half = lane >> 5, l31 = lane & 31;

  const int HW = a.Hin * a.Win;                 // 1x1 stride 1: the output map is the input map
  const int tiles_x = (a.Wout + P::TW - 1) / P::TW;
  const int nst = (a.Cin + P::KC - 1) / P::KC;
  const int total = a.B * a.n_sp * a.n_ct;      // tiles, flattened (image, spatial tile, channel tile): channel tile fastest, so
  const int u_begin = (int)blockIdx.x * a.tiles_per_wg;   // that consecutive tiles of a workgroup re-read one pixel tile from L2
  const int u_end = min(u_begin + a.tiles_per_wg, total);
  if (u_begin >= u_end) return;

  const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(a.w, 0x7fffffffll);
  const __amdgpu_buffer_rsrc_t brs = buf_rsrc(a.bias, a.bias ? (long long)a.Cout * 4 : 0);
  const unsigned hw4 = (unsigned)HW * 4u;

  // ---- tile contexts: `c*` = the tile being computed / stored, `n*` = the tile the ring has run ahead into ------------------
  __amdgpu_buffer_rsrc_t cxrs, nxrs;
  unsigned cxoff[P::NXV], nxoff[P::NXV];
  int c_nt, c_b, c_oy0, c_ox0, n_nt = 0, n_b = 0, n_oy0 = 0, n_ox0 = 0;
  auto decode = [&](int u, __amdgpu_buffer_rsrc_t& xrs, unsigned* xoff, int& nt, int& b, int& oy0, int& ox0) {
    nt = u % a.n_ct;
    const int q = u / a.n_ct;
    const int sp = q % a.n_sp;
    b = q / a.n_sp;
    oy0 = (sp / tiles_x) * P::TH;
    ox0 = (sp % tiles_x) * P::TW;
    xrs = buf_rsrc(a.in + (long long)b * conv_in_bstride(a), (long long)a.Cin * HW * 4);   // channels past Cin read as zeros
#pragma unroll
    for (int k = 0; k < P::NXV; ++k) {
      // float4 v = 256*k + tid covers 4 consecutive pixels of one tile row of one channel of the stage
      const int v = tid + 256 * k;
      const int c = v / (P::BM / 4);
      const int p = (v - c * (P::BM / 4)) * 4;
      const int gy = oy0 + p / P::TW, gx = ox0 + p % P::TW;
      const bool ok = gy < a.Hin && gx < a.Win;             // Win % 4 == 0 (host check): a piece is inside or outside as a whole
      xoff[k] = ok ? (unsigned)(c * HW + gy * a.Win + gx) * 4u : kOob;
    }
  };
  auto issue = [&](const __amdgpu_buffer_rsrc_t xrs, const unsigned* xoff, int nt, int s, int slot) {
    const unsigned xso = (unsigned)(s * P::KC) * hw4;
    float* X = smem + slot * P::STAGE + wave * 256;
#pragma unroll
    for (int k = 0; k < P::NXV; ++k) bglds16(xrs, X + 1024 * k, xoff[k], xso);
    const unsigned wso = (unsigned)((nt * nst + s) * P::WSZP) * 4u;
    float* W = smem + slot * P::STAGE + P::XSZ + wave * 256;
#pragma unroll
    for (int k = 0; k < P::NW; ++k) bglds16(wrs, W + 1024 * k, (unsigned)tid * 16u, wso + 4096u * k);
  };

  decode(u_begin, cxrs, cxoff, c_nt, c_b, c_oy0, c_ox0);
  nxrs = cxrs;
#pragma unroll
  for (int k = 0; k < P::NXV; ++k) nxoff[k] = cxoff[k];

  // ---- the ring's issue cursor: (tile i_u, stage i_s), NBUF - 1 stages ahead of the stage being computed, i.e. in the tile
  // being computed or in the next one (nst >= NBUF - 1, host check); the next tile is decoded when its first stage is issued
  int i_u = u_begin, i_s = 0, i_slot = 0;
  int issued = 0, consumed = 0, issued_at_epi = 0;  // stage groups, flattened over the workgroup's tiles
  auto issue_one = [&](int u_cur) {
    if (i_u >= u_end) return;
    if (i_u == u_cur) {
      issue(cxrs, cxoff, c_nt, i_s, i_slot);
    } else {
      if (i_s == 0) decode(i_u, nxrs, nxoff, n_nt, n_b, n_oy0, n_ox0);
      issue(nxrs, nxoff, n_nt, i_s, i_slot);
    }
    ++issued;
    i_slot = (i_slot + 1 == P::NBUF) ? 0 : i_slot + 1;
    if (++i_s == nst) {
      i_s = 0;
      ++i_u;
    }
  };
#pragma unroll
  for (int p = 0; p < P::NBUF - 1; ++p) issue_one(u_begin);

  // ---- per-lane LDS operand addresses (relative to a slot) ---------------------------------------------------------------
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  unsigned xo[P::MI], wo[P::NI];
#pragma unroll
  for (int i = 0; i < P::MI; ++i) xo[i] = (unsigned)(half * P::BM + wm * (P::MI * 32) + i * 32 + l31) * 4u;
#pragma unroll
  for (int j = 0; j < P::NI; ++j) wo[j] = (unsigned)(P::XSZ + half * P::BN + wn * (P::NI * 32) + j * 32 + l31) * 4u;

  int c_slot = 0;
  for (int u = u_begin; u < u_end; ++u) {
    f32x16 acc[P::NI][P::MI];
#pragma unroll
    for (int j = 0; j < P::NI; ++j)
#pragma unroll
      for (int i = 0; i < P::MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.0f;

    // epilogue addressing of this tile (registers live across the main loop: MI + a few)
    unsigned voff[P::MI];
#pragma unroll
    for (int i = 0; i < P::MI; ++i) {
      const int gy = c_oy0 + wm * P::MI + i, gx = c_ox0 + l31;
      voff[i] = (gy < a.Hout && gx < a.Wout) ? (unsigned)(gy * a.Wout + gx) * 4u + (unsigned)(4 * half) * hw4 : kOob;
    }
    const int co_base = c_nt * P::BN + wn * (P::NI * 32);
    const __amdgpu_buffer_rsrc_t rrs =
        buf_rsrc(a.res ? a.res + ((long long)c_b * a.res_ctot + a.res_coff) * HW : nullptr, a.res ? (long long)a.Cout * HW * 4 : 0);
    float resv[P::NI][P::MI][16];
    float bv[P::NI];

    for (int s = 0; s < nst; ++s) {
      // ---- retire ring stage `consumed`: exactly the younger stage groups (and, for the groups requested in front of the
      // previous tile's epilogue, its stores) stay in flight
      {
        const int ahead = issued - consumed - 1;
        const bool stores_younger = consumed < issued_at_epi;
        if (stores_younger) {
          if (ahead >= 2 && P::NBUF >= 4) wait_vm<2 * P::LOADS + P::NACC>();
          else if (ahead >= 1) wait_vm<P::LOADS + P::NACC>();
          else wait_vm<P::NACC>();
        } else {
          if (ahead >= 2 && P::NBUF >= 4) wait_vm<2 * P::LOADS>();
          else if (ahead >= 1) wait_vm<P::LOADS>();
          else wait_vm<0>();
        }
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      issue_one(u);
      if (s == nst - 1) {
        // bias and residual of THIS tile: requested one stage ahead of the epilogue (unconditional: out-of-range lanes read 0)
#pragma unroll
        for (int j = 0; j < P::NI; ++j)
          bv[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (unsigned)(co_base + j * 32 + l31) * 4u, 0, 0));
        if (a.res) {
#pragma unroll
          for (int j = 0; j < P::NI; ++j)
#pragma unroll
            for (int i = 0; i < P::MI; ++i)
#pragma unroll
              for (int r = 0; r < 16; ++r)
                resv[j][i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rrs, voff[i], (unsigned)(co_base + j * 32 + (r & 3) + 8 * (r >> 2)) * hw4, 0));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- MFMAs of the stage: operands of step c + 1 requested before the MFMAs of step c (two register sets), every
      // set waited for with an exact lgkmcnt (hand-issued reads, see conv_kernel.h)
      {
        constexpr int NSTEP = P::KC / 2;
        constexpr int NLD = P::NI + P::MI;
        const unsigned sb = lds0 + (unsigned)(c_slot * P::STAGE) * 4u;
        unsigned wa[P::NI], xa[P::MI];
#pragma unroll
        for (int j = 0; j < P::NI; ++j) wa[j] = sb + wo[j];
#pragma unroll
        for (int i = 0; i < P::MI; ++i) xa[i] = sb + xo[i];
        struct Ops {
          float r[NLD];       // [0, NI): weights, [NI, NLD): pixels
        };
        auto load = [&](Ops& o, auto sc) {
          constexpr int cp_ = decltype(sc)::value;
#pragma unroll
          for (int j = 0; j < P::NI; ++j) lds_read_b32<(2 * cp_) * P::BN * 4>(o.r[j], wa[j]);
#pragma unroll
          for (int i = 0; i < P::MI; ++i) lds_read_b32<(2 * cp_) * P::BM * 4>(o.r[P::NI + i], xa[i]);
        };
        auto wait_for = [&](Ops& o, auto newer_c) {
          constexpr int N_ = decltype(newer_c)::value;
          if constexpr (NLD == 3)
            asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(o.r[0]), "+v"(o.r[1]), "+v"(o.r[2]) : "n"(N_));
          else
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(o.r[0]), "+v"(o.r[1]), "+v"(o.r[2]), "+v"(o.r[3]) : "n"(N_));
        };
        static_assert(NLD == 3 || NLD == 4, "operand sets of 3 or 4 registers");
        Ops A, B;
        load(A, std::integral_constant<int, 0>{});
        static_for<0, NSTEP>([&](auto sc) {
          constexpr int s_ = decltype(sc)::value;
          Ops& o = (s_ & 1) ? B : A;
          Ops& n = (s_ & 1) ? A : B;
          if constexpr (s_ + 1 < NSTEP) {
            load(n, std::integral_constant<int, s_ + 1>{});
            wait_for(o, std::integral_constant<int, NLD>{});
          } else {
            wait_for(o, std::integral_constant<int, 0>{});
          }
#pragma unroll
          for (int j = 0; j < P::NI; ++j)
#pragma unroll
            for (int i = 0; i < P::MI; ++i)
              acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.r[j], o.r[P::NI + i], acc[j][i], 0, 0, 0);
        });
      }
      ++consumed;
      c_slot = (c_slot + 1 == P::NBUF) ? 0 : c_slot + 1;
    }

    // ---- epilogue from the accumulator registers: lanes 0-31 of register r are 32 consecutive pixels of output channel
    // co_base + j*32 + (r&3) + 8*(r>>2), lanes 32-63 the same pixels four channels further -- two 128-byte segments per store
    {
      const __amdgpu_buffer_rsrc_t ors =
          buf_rsrc(a.out + ((long long)c_b * a.out_ctot + a.out_coff) * HW, (long long)a.Cout * HW * 4);
#pragma unroll
      for (int j = 0; j < P::NI; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2);
          float bias_r = 0.0f;
          if (a.bias) bias_r = __shfl(bv[j], rr + 4 * half, 64);
#pragma unroll
          for (int i = 0; i < P::MI; ++i) {
            float v = acc[j][i][r] + bias_r;
            if (a.res) v += resv[j][i][r];
            if (a.act == ACT_RELU) v = fmaxf(v, 0.0f);
            else if (a.act == ACT_RELU6) v = fminf(fmaxf(v, 0.0f), 6.0f);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ors, voff[i], (unsigned)(co_base + j * 32 + rr) * hw4, 0);
          }
        }
      }
      issued_at_epi = issued;
    }
    // the ring's `n*` tile becomes the tile being computed
    cxrs = nxrs;
#pragma unroll
    for (int k = 0; k < P::NXV; ++k) cxoff[k] = nxoff[k];
    c_nt = n_nt; c_b = n_b; c_oy0 = n_oy0; c_ox0 = n_ox0;
  }
}

template <class P>
KernelEntry entry_p() {
  return KernelEntry{conv1x1p_kernel<P>, P::LDS_BYTES, 256};
}

//                      KC   BN  NBUF resident
using P_K16_N64  = P1x1<16,  64, 3, 4>;
using P_K16_N128 = P1x1<16, 128, 3, 2>;
using P_K32_N64  = P1x1<32,  64, 3, 2>;

}  // namespace
void conv_fill_1x1_p(void* row_k16, void* row_k32);
int conv_1x1p_resident(ConvTile t);   // workgroups per CU a launch is sized for
}  // namespace fdt
