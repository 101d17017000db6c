// conv_1x1p_b3.h -- the split-bf16 1x1 / stride-1 convolution (conv_b3.h) as a PERSISTENT-TILE kernel (class CONV_1x1_S1_PB3,
// round 5): conv_1x1p.h's schedule around conv_b3.h's arithmetic.
//
// Why.  The 1x1 layers of layer1 / layer2 (pyramid.py:97-103 at 256^2 / 128^2) move 170-600 MB per four-frame launch for a few
// GFLOP: on the bf16 pipe their reduction is four to thirty-two short stages, and in the one-tile-per-workgroup form (conv_b3.h)
// every workgroup pays its ring prologue and drains its stores before the next one starts -- 3.0-4.3 TB/s.  Here a workgroup walks
// `tiles_per_wg` consecutive output tiles (channel tile fastest: its pixel tile stays in L2) and the LDS ring keeps running across
// them: the first stages of tile t + 1 are requested during the last stages of tile t, the operands of its first stage are read and
// split under the last MFMAs of tile t, the epilogue runs from the accumulator registers (stores fire-and-forget) while those
// requests are in flight.
//   * arithmetic, operand layouts, weight packing: conv_b3.h (three bf16 planes per f32 operand, six plane products smallest first,
//     f32 accumulate), waves 4 x 1 (a wave = one tile row of 32 pixels x all couts of the tile); same bits as class 21;
//   * every vector-memory instruction is unconditional (out-of-image pixels carry an out-of-range offset, channels past Cout fall
//     off the end of the descriptor), so the number of them between two points of the program is a compile-time constant and a ring
//     stage is retired with an exact s_waitcnt vmcnt(N) that leaves the younger stage -- and the stores behind it -- in flight.
// Needs Win % 4 == 0, at least three stages (Cin > 32: the ring runs three stages ahead of the MFMAs and may not enter the tile after
// the next), no split-K, no fused upsample-add, no second destination, < 2 GB per image.
#pragma once
#include "conv_1x1p.h"
#include "conv_b3.h"

namespace fdt {
namespace {

template <int BN_, int RES_>
struct PB3 {
  static constexpr int KC = 16, BN = BN_, NBUF = 3, RESIDENT = RES_;
  static constexpr int TH = 4, TW = 32, BM = 128;
  static constexpr int NI = BN / 32;                         // a wave: 32 pixels (tile row `wave`) x NI cout tiles
  static constexpr int XSZ = KC * BM, WSZ = 24 * BN, WSZP = (WSZ + 1023) / 1024 * 1024;
  static constexpr int STAGE = XSZ + WSZP;
  static constexpr int NXV = XSZ / 1024, NW = WSZP / 1024, LOADS = NXV + NW;
  static constexpr int NACC = NI * 16;                       // stores (and residual loads) per tile and wave
  static constexpr size_t LDS_BYTES = (size_t)NBUF * STAGE * sizeof(float);
};

template <class P>
__global__ __launch_bounds__(256, P::RESIDENT) void conv1x1p_b3_kernel(const ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  const int HW = a.Hin * a.Win;
  const int tiles_x = (a.Wout + P::TW - 1) / P::TW;
  const int nst = (a.Cin + P::KC - 1) / P::KC;
  const int total = a.B * a.n_sp * a.n_ct;
  const int u_begin = (int)blockIdx.x * a.tiles_per_wg;
  const int u_end = min(u_begin + a.tiles_per_wg, total);
  if (u_begin >= u_end) return;

  const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(a.w, 0x7fffffffll);
  const __amdgpu_buffer_rsrc_t brs = buf_rsrc(a.bias, a.bias ? (long long)a.Cout * 4 : 0);
  const unsigned hw4 = (unsigned)HW * 4u;

  // ---- tile contexts (conv_1x1p.h): `c*` the tile being computed / stored, `n*` the tile the ring has run ahead into -----------
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
    xrs = buf_rsrc(a.in + (long long)b * conv_in_bstride(a), (long long)a.Cin * HW * 4);
#pragma unroll
    for (int k = 0; k < P::NXV; ++k) {
      const int v = tid + 256 * k;
      const int c = v / (P::BM / 4);
      const int p = (v - c * (P::BM / 4)) * 4;
      const int gy = oy0 + p / P::TW, gx = ox0 + p % P::TW;
      const bool ok = gy < a.Hin && gx < a.Win;
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

  // the ring's issue cursor: NBUF - 1 stages ahead of the stage whose operands are being READ (one more than the one computed)
  int i_u = u_begin, i_s = 0, i_slot = 0;
  int u_cur = u_begin;                       // tile of the `c*` context
  auto issue_one = [&]() {
    if (i_u >= u_end) return;
    if (i_u == u_cur) {
      issue(cxrs, cxoff, c_nt, i_s, i_slot);
    } else {
      if (i_s == 0) decode(i_u, nxrs, nxoff, n_nt, n_b, n_oy0, n_ox0);
      issue(nxrs, nxoff, n_nt, i_s, i_slot);
    }
    i_slot = (i_slot + 1 == P::NBUF) ? 0 : i_slot + 1;
    if (++i_s == nst) {
      i_s = 0;
      ++i_u;
    }
  };

  // ---- operands of one stage (conv_b3.h) ---------------------------------------------------------------------------------------
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  const unsigned xo = (unsigned)((half * 8) * P::BM + wave * 32 + l31) * 4u;
  unsigned wo[P::NI];
#pragma unroll
  for (int j = 0; j < P::NI; ++j) wo[j] = (unsigned)P::XSZ * 4u + (unsigned)(half * P::BN + j * 32 + l31) * 16u;
  struct Ops {
    bf16x8 A[P::NI][3];
    float xv[8];
    bf16x8 Bp[3];
  };
  auto issue_reads = [&](Ops& o, int slot) {
    const unsigned sb = lds0 + (unsigned)(slot * P::STAGE) * 4u;
#pragma unroll
    for (int j = 0; j < P::NI; ++j) {
      const unsigned ad = sb + wo[j];
      lds_read_b128<0>(o.A[j][0], ad);
      lds_read_b128<2 * P::BN * 16>(o.A[j][1], ad);
      lds_read_b128<4 * P::BN * 16>(o.A[j][2], ad);
    }
    const unsigned ad = sb + xo;
    lds_read_b32<0 * P::BM * 4>(o.xv[0], ad);
    lds_read_b32<1 * P::BM * 4>(o.xv[1], ad);
    lds_read_b32<2 * P::BM * 4>(o.xv[2], ad);
    lds_read_b32<3 * P::BM * 4>(o.xv[3], ad);
    lds_read_b32<4 * P::BM * 4>(o.xv[4], ad);
    lds_read_b32<5 * P::BM * 4>(o.xv[5], ad);
    lds_read_b32<6 * P::BM * 4>(o.xv[6], ad);
    lds_read_b32<7 * P::BM * 4>(o.xv[7], ad);
  };
  auto wait_reads = [&](Ops& o) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(o.xv[0]), "+v"(o.xv[1]), "+v"(o.xv[2]), "+v"(o.xv[3]), "+v"(o.xv[4]), "+v"(o.xv[5]), "+v"(o.xv[6]), "+v"(o.xv[7]));
#pragma unroll
    for (int j = 0; j < P::NI; ++j) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o.A[j][0]), "+v"(o.A[j][1]), "+v"(o.A[j][2]));
  };
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  auto split_pair = [&](Ops& o, int qp) {
    unsigned P0, P1, P2;
    split3_bf16_pair((f32x2v){o.xv[2 * qp], o.xv[2 * qp + 1]}, P0, P1, P2);
    u32x4v b0 = __builtin_bit_cast(u32x4v, o.Bp[0]), b1 = __builtin_bit_cast(u32x4v, o.Bp[1]), b2 = __builtin_bit_cast(u32x4v, o.Bp[2]);
    b0[qp] = P0; b1[qp] = P1; b2[qp] = P2;
    o.Bp[0] = __builtin_bit_cast(bf16x8, b0);
    o.Bp[1] = __builtin_bit_cast(bf16x8, b1);
    o.Bp[2] = __builtin_bit_cast(bf16x8, b2);
  };
  f32x16 acc[P::NI];
  auto zero_acc = [&]() {
#pragma unroll
    for (int j = 0; j < P::NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  };
  // six plane products per cout tile, smallest first; the reads of the next stage are waited for after the first third of the
  // MFMAs and its split is spread over the rest (conv_b3.h)
  auto mfmas = [&](Ops& c, Ops& n, bool have_next) {
    constexpr int pa[6] = {1, 2, 0, 1, 0, 0}, pb[6] = {1, 0, 2, 0, 1, 0};
    constexpr int NM = P::NI * 6, FIRST = NM / 3;
    constexpr int PER = (4 + (NM - FIRST) - 1) / (NM - FIRST);
    static_for<0, NM>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      constexpr int j = m / 6, p = m % 6;
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c.A[j][pa[p]], c.Bp[pb[p]], acc[j], 0, 0, 0);
      if constexpr (m + 1 >= FIRST) {
        if (have_next) {
          if constexpr (m + 1 == FIRST) wait_reads(n);
          static_for<0, PER>([&](auto ec) {
            constexpr int idx = (m + 1 - FIRST) * PER + decltype(ec)::value;
            if constexpr (idx < 4) split_pair(n, idx);
          });
        }
      }
    });
  };

  // ---- epilogue of the `c*` tile from the accumulator registers (conv_b3.h's register epilogue) ----------------------------------
  const bool has_res = a.res != nullptr;
  auto epilogue = [&]() {
    const int gy = c_oy0 + wave, gx = c_ox0 + l31;
    const unsigned voff = (gy < a.Hout && gx < a.Wout) ? (unsigned)(gy * a.Wout + gx) * 4u + (unsigned)(4 * half) * hw4 : kOob;
    const __amdgpu_buffer_rsrc_t ors = buf_rsrc(a.out + ((long long)c_b * a.out_ctot + a.out_coff) * HW, (long long)a.Cout * HW * 4);
    const __amdgpu_buffer_rsrc_t rrs =
        buf_rsrc(has_res ? a.res + ((long long)c_b * a.res_ctot + a.res_coff) * HW : nullptr, has_res ? (long long)a.Cout * HW * 4 : 0);
    const int co0 = c_nt * P::BN;
    // (one residual buffer: the registers of a second one would spill at this class's occupancy; the loads of cout tile j are
    // requested behind the stores of tile j - 1 and the other resident workgroups' MFMAs cover their latency)
    float rv[16];
    static_for<0, P::NI>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const float bvj = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(brs, (unsigned)(co0 + j * 32 + l31) * 4u, 0, 0));
      if (has_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, voff, (unsigned)(co0 + j * 32 + (r & 3) + 8 * (r >> 2)) * hw4, 0));
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2);
        float v = acc[j][r] + (a.bias ? __shfl(bvj, rr + 4 * half, 64) : 0.0f);
        if (has_res) v += rv[r];
        if (a.act == ACT_RELU) v = fmaxf(v, 0.0f);
        else if (a.act == ACT_RELU6) v = fminf(fmaxf(v, 0.0f), 6.0f);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ors, voff, (unsigned)(co0 + j * 32 + rr) * hw4, 0);
      }
    });
  };
  // vector-memory instructions of one epilogue, per wave: bias loads + stores (+ residual loads)
  constexpr int EPI_PLAIN = P::NI + P::NACC, EPI_RES = P::NI + 2 * P::NACC;

  // ---- prologue: stages 0 and 1 requested, stage 0 waited for, read and split; stage 2 requested ------------------------------
  const int total_stages = (u_end - u_begin) * nst;             // >= 3 (host check: nst >= 3)
  issue_one();
  issue_one();
  wait_vm<P::LOADS>();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  issue_one();
  Ops O[2];
  issue_reads(O[0], 0);
  wait_reads(O[0]);
#pragma unroll
  for (int q = 0; q < 4; ++q) split_pair(O[0], q);
  zero_acc();

  // ---- the flat stage loop, two stages per trip (the operand sets swap roles).  At the top of step g the operands of stage g sit
  // in registers; stage g + 1 is waited for (stage g + 2 and, behind a tile boundary, that tile's stores stay in flight), the barrier
  // also says every wave has finished READING stage g, so stage g + 3 may be requested into its slot; stage g + 1 is read; the MFMAs
  // of stage g run; behind a tile's last stage its epilogue, then the `n*` context becomes the `c*` one.
  int s = 0;                                 // stage of the `c*` tile that step g computes
  int slot_next = 1;                         // ring slot of stage g + 1
  int epi_age = 3;                           // steps since the last epilogue was issued (>= 2: its instructions are older than stage g + 1's requests)
  auto step = [&](int g, Ops& c, Ops& n) {
    const bool have_next = g + 1 < total_stages;
    if (have_next) {
      // younger than the requests of stage g + 1: those of stage g + 2 (if it exists) and an epilogue issued one or two steps ago
      const bool more = g + 2 < total_stages;
      const bool epi = epi_age < 2;
      if (epi) {
        if (has_res) { if (more) wait_vm<P::LOADS + EPI_RES>(); else wait_vm<EPI_RES>(); }
        else { if (more) wait_vm<P::LOADS + EPI_PLAIN>(); else wait_vm<EPI_PLAIN>(); }
      } else {
        if (more) wait_vm<P::LOADS>(); else wait_vm<0>();
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      issue_one();
      issue_reads(n, slot_next);
    }
    mfmas(c, n, have_next);
    slot_next = slot_next == P::NBUF - 1 ? 0 : slot_next + 1;
    ++epi_age;
    if (++s == nst) {
      epilogue();
      epi_age = 0;
      s = 0;
      zero_acc();
      ++u_cur;
      cxrs = nxrs;
#pragma unroll
      for (int k = 0; k < P::NXV; ++k) cxoff[k] = nxoff[k];
      c_nt = n_nt; c_b = n_b; c_oy0 = n_oy0; c_ox0 = n_ox0;
    }
  };
  for (int g = 0; g < total_stages; g += 2) {
    step(g, O[0], O[1]);
    if (g + 1 < total_stages) step(g + 1, O[1], O[0]);
  }
}

template <class P>
KernelEntry entry_pb3() {
  return KernelEntry{conv1x1p_b3_kernel<P>, P::LDS_BYTES, 256};
}

using PB3_N64 = PB3<64, 3>;
using PB3_N128 = PB3<128, 2>;

}  // namespace
void conv_fill_1x1_pb3(void* row);
}  // namespace fdt
