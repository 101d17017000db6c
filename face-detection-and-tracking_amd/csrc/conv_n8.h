// conv_n8.h -- 3x3 / stride 1 / pad 1 convolution for layers with a HANDFUL of output channels: the loc + conf heads
// (4 + 4 channels, pyramid.py:291-306, pyramid_mb2_try3.py:323-339) on the f32 vector ALU instead of the matrix cores.
//
// Why not MFMA: the 32x32x2 tile pads 8 output channels to 32 (25% useful) and the 16x16x4 tile to 16 (50%); both cost
// more matrix-pipe time than the layer has arithmetic (the best MFMA variant, 8-wave Winograd, runs these layers at an
// effective 46 TFLOP/s).  The vector ALU has no padding; its sustained f32 rate on gfx950 is power-capped at ~110 TFLOP/s
// (tools/microbench/valu_peak.hip: the clock drops as waves are added), and the layer also reads 36 FLOP per input byte,
// so it sits near both roofs.
//   * thread = 2 x 4 output pixels x 8 output channels (64 accumulators): an input value feeds 8 FMAs, the 8 weights of a
//     (channel, tap) are ONE pair of wave-uniform 16-byte LDS reads shared by the thread's 8 pixels: 1152 FMAs for 30 LDS
//     reads per input channel;
//   * workgroup = 32 x 64 output pixels of one image; each of its four WAVES owns 8 output rows and stages its OWN 10-row
//     patch (and its own copy of the stage's 72 weights) by LDS-DMA into a private double buffer: no workgroup barrier
//     anywhere, a wave waits only for its own loads (counted vmcnt), so the waves of a SIMD drift apart and cover each
//     other's waits.  The 2 halo rows shared by neighbouring waves are fetched twice (L2 hits);
//   * the patch is staged as 16-byte pieces from the 16-byte aligned superset [ox0-4, ox0+68) of the columns (Win % 4 == 0;
//     else dword pieces): 4 LDS-DMA instructions per wave per input channel;
//   * KC = 1 input channel per stage; weights pre-tiled by tile_weights(): [n_tile][Cin][9][8];
//   * epilogue as in conv_kernel: raw partial sums to the split-K workspace, or + bias, + residual, activation.
// Accumulation order per output: channel, tap row, tap column (ascending) -- an fmaf chain like every other kernel class;
// parity with the oracle is at the conv tolerance (tests/test_gpu_conv.py).
#pragma once
#include "conv_kernel.h"

#ifndef FDT_N8_EXP
#define FDT_N8_EXP 0   // tuning experiments (tools/experiments/n8_variants.sh): 1 no LDS-DMA after the first stages, 2 one tap of nine
#endif

namespace fdt {
namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OFF>
__device__ __forceinline__ void lds_read_b128(f32x4& v, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536 && OFF % 16 == 0, "ds_read_b128 offset");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

struct N8 {
  static constexpr int TH = 32, TW = 64, KC = 1, BN = 8;
  static constexpr int WR = 8;                            // output rows per wave
  static constexpr int PR = WR + 2, PITCH = TW + 8;       // patch rows per wave; columns ox0-4 .. ox0+67
  static constexpr int XSZ = PR * PITCH;                  // 720 floats = 180 16-byte pieces
  static constexpr int WSZ = 9 * BN;                      // 72 floats = 18 pieces, staged right behind the patch
  static constexpr int NV = (XSZ + WSZ + 255) / 256;      // 4 dwordx4 LDS-DMA instructions per wave per stage
  static constexpr int ND = (XSZ + WSZ + 63) / 64;        // 13 dword instructions (Win % 4 != 0)
  static constexpr int BUF = 1024;                        // floats per wave per buffer (>= NV * 256, >= ND * 64)
  static constexpr size_t LDS_BYTES = 4 * 2 * (size_t)BUF * sizeof(float);
  static_assert(NV * 256 <= BUF && ND * 64 <= BUF, "buffer");
};

// VEC: Win % 4 == 0 (16-byte pieces); the dword variant is a separate instantiation so that its index arithmetic does not
// cost the fast one registers (a spill reload inside the loop carries a vmcnt(0) that serialises the LDS-DMA pipeline).
template <bool VEC>
__global__ __launch_bounds__(256, 4) void conv_n8_kernel(const ConvArgs a) {
  using L = N8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (a.Wout + L::TW - 1) / L::TW;
  FDT_BLOCK_MAP(a, tile_id, n_tile);
  const int oy0 = (tile_id / tiles_x) * L::TH + wave * L::WR;      // first output row of this wave
  const int ox0 = (tile_id % tiles_x) * L::TW;
  const int b = blockIdx.z / a.ksplit;
  const int ks = blockIdx.z - b * a.ksplit;
  const int HW = a.Hin * a.Win;                          // == Hout * Wout (stride 1, pad 1)
  const float* in_b = a.in + (long long)b * conv_in_bstride(a);
  const float* w_t = a.w + (long long)n_tile * a.Cin * L::WSZ;
  const int s_begin = (int)((long long)a.Cin * ks / a.ksplit);
  const int s_end = (int)((long long)a.Cin * (ks + 1) / a.ksplit);
  float* mybuf = smem + wave * 2 * L::BUF;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)mybuf;

  // staging plan of the lane (the same for every input channel): piece e = 64 * k + lane of [patch | weights]; padding,
  // out-of-image and beyond-the-end pieces read zeros
  const float* zpad = g_zero_pad;
  asm volatile("" : "+s"(zpad));
  constexpr bool vec = VEC;
  // vector path: byte offset of the lane's piece from the base its class selects (0: channel plane, 1: the stage's
  // weights, 2: the zero word).  32-bit on purpose: 64-bit per-lane offsets do not fit the register budget of four
  // waves per SIMD, and a spill reloaded in the loop carries a vmcnt(0) that serialises the LDS-DMA pipeline.
  unsigned offb[L::NV];
  int cls[L::NV];
#pragma unroll
  for (int k = 0; k < L::NV; ++k) {
    const int e = 64 * k + lane;                         // 16-byte piece
    const int r = e / (L::PITCH / 4), q = e - r * (L::PITCH / 4);
    const int gy = oy0 - 1 + r, gx = ox0 - 4 + 4 * q;
    offb[k] = 0;
    cls[k] = 2;
    if (e < L::XSZ / 4) {
      if (gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) {
        offb[k] = (unsigned)(gy * a.Win + gx) * 4u;
        cls[k] = 0;
      }
    } else if (e < (L::XSZ + L::WSZ) / 4) {
      offb[k] = (unsigned)(e - L::XSZ / 4) * 16u;
      cls[k] = 1;
    }
  }
  auto stage = [&](int c, int buf) {
    const float* src = in_b + (long long)c * HW;
    const float* wsrc = w_t + (long long)c * L::WSZ;
    float* dst = mybuf + buf * L::BUF;
    if constexpr (VEC) {
#pragma unroll
      for (int k = 0; k < L::NV; ++k) {
        const unsigned long long base = cls[k] == 0 ? (unsigned long long)src
                                        : cls[k] == 1 ? (unsigned long long)wsrc : (unsigned long long)zpad;
        glds16(reinterpret_cast<const float*>(base + offb[k]), dst + 256 * k);
      }
    } else {
#pragma unroll 1
      for (int k = 0; k < L::ND; ++k) {
        const int e = 64 * k + lane;                     // float
        const int r = e / L::PITCH, x = e - r * L::PITCH;
        const int gy = oy0 - 1 + r, gx = ox0 - 4 + x;
        const float* p = zpad;
        if (e < L::XSZ) {
          if (gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) p = src + gy * a.Win + gx;
        } else if (e < L::XSZ + L::WSZ) {
          p = wsrc + (e - L::XSZ);
        }
        glds4(p, dst + 64 * k);
      }
    }
  };

  // lane = output rows 2*ty, 2*ty+1 of the wave's 8, columns 4*tx .. 4*tx+3 of the tile
  const int ty = lane >> 4, tx = lane & 15;
  f32x2 acc[2][4][4];   // [row][pixel][channel pair]
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[o][q][j] = f32x2{0.0f, 0.0f};

  const int nst = s_end - s_begin;
  if (nst > 0) stage(s_begin, 0);
  for (int it = 0; it < nst; ++it) {
    // the buffer stage it + 1 overwrites was last read by THIS wave in iteration it - 1: those reads have returned (their
    // values fed FMAs that precede this point in program order)
    __builtin_amdgcn_sched_barrier(0);
    if (it + 1 < nst && (FDT_N8_EXP != 1 || it == 0)) {
      stage(s_begin + it + 1, (it + 1) & 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VEC ? L::NV : L::ND) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    // Operand reads are issued by hand with exact lgkmcnt waits: the compiler books the outstanding LDS-DMA loads as
    // accesses its own LDS reads may alias and would put s_waitcnt vmcnt(0) in front of them -- i.e. wait for the NEXT
    // stage's loads before computing this one (measured: no overlap at all).  Weights run one tap ahead in two
    // register sets; tools/check_async_lds.py lints the ISA for a use of a register whose read is still outstanding.
    const unsigned xaddr = lds0 + (unsigned)(((it & 1) * L::BUF + (2 * ty) * L::PITCH + 4 * tx) * 4);
    const unsigned waddr = lds0 + (unsigned)(((it & 1) * L::BUF + L::XSZ) * 4);
    // The strip's own 4 pixels are one 16-byte read; its left / right neighbour pixels are the neighbour LANE's outer strip
    // values (DPP row shift inside the 16 lanes of a tile row) -- only the two halo columns of the tile row come from LDS, as
    // one dword per lane at an address the 16 lanes share (broadcast, conflict-free).  Per-lane dword reads of columns 4tx+3 /
    // 4tx+8 would touch only the 16 banks = 3 (mod 4): a 4-way conflict (measured: 38 % of the LDS-active cycles).
    const unsigned haddr = lds0 + (unsigned)(((it & 1) * L::BUF + (2 * ty) * L::PITCH) * 4);
    float xl[4], xr[4];
    f32x4 xm[4], wa[2], wb[2];
    static_for<0, 4>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      lds_read_b32<(r * L::PITCH + 3) * 4>(xl[r], haddr);
      lds_read_b128<(r * L::PITCH + 4) * 4>(xm[r], xaddr);
      lds_read_b32<(r * L::PITCH + L::TW + 4) * 4>(xr[r], haddr);
    });
    lds_read_b128<0>(wa[0], waddr);
    lds_read_b128<16>(wb[0], waddr);
    static_for<0, (FDT_N8_EXP == 2 ? 1 : 9)>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int dy = t / 3, dx = t % 3, set = t % 2;
      if constexpr (t + 1 < 9) {
        lds_read_b128<(t + 1) * 32>(wa[(t + 1) % 2], waddr);
        lds_read_b128<(t + 1) * 32 + 16>(wb[(t + 1) % 2], waddr);
      }
      constexpr int after = t < 8 ? 2 : 0;                     // reads issued behind this tap's weights
      if constexpr (t == 0)
        asm volatile("s_waitcnt lgkmcnt(%14)"
                     : "+v"(wa[0]), "+v"(wb[0]), "+v"(xm[0]), "+v"(xm[1]), "+v"(xm[2]), "+v"(xm[3]), "+v"(xl[0]), "+v"(xl[1]),
                       "+v"(xl[2]), "+v"(xl[3]), "+v"(xr[0]), "+v"(xr[1]), "+v"(xr[2]), "+v"(xr[3])
                     : "n"(after));
      else
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(wa[set]), "+v"(wb[set]) : "n"(after));
      if constexpr (t == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {   // lane 0 / 15 of a row keep the halo word (bound_ctrl off), the others take the neighbour's
          // (copies first: __builtin_bit_cast of a vector-element lvalue reads element 0 with this compiler)
          const float last = xm[r][3], first = xm[r][0];
          xl[r] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(xl[r]), __float_as_int(last), 0x111, 0xf, 0xf, false));
          xr[r] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(xr[r]), __float_as_int(first), 0x101, 0xf, 0xf, false));
        }
      }
      const f32x2 wp[4] = {f32x2{wa[set][0], wa[set][1]}, f32x2{wa[set][2], wa[set][3]}, f32x2{wb[set][0], wb[set][1]},
                           f32x2{wb[set][2], wb[set][3]}};
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int col = q + dx;                              // 0 .. 5 of the 6-wide strip
          const float x = col == 0 ? xl[o + dy] : col == 5 ? xr[o + dy] : xm[o + dy][col - 1];
          const f32x2 xx = f32x2{x, x};
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[o][q][j] = __builtin_elementwise_fma(xx, wp[j], acc[o][q][j]);
        }
    });
    asm volatile("" ::: "memory");
  }

  // ---- epilogue ----
  const int ox = ox0 + 4 * tx;
  const bool raw = a.ws != nullptr;
  float* dst_b = raw ? a.ws + ((long long)(b * a.ksplit + ks) * a.Cout) * HW
                     : a.out + ((long long)b * a.out_ctot + a.out_coff) * HW;
  const float* res_b = (!raw && a.res) ? a.res + ((long long)b * a.res_ctot + a.res_coff) * HW : nullptr;
  const bool vst = vec && ox + 3 < a.Wout;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int co = n_tile * L::BN + 2 * j + h;
      if (co >= a.Cout) continue;
      const float bv = (!raw && a.bias) ? a.bias[co] : 0.0f;
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        const int oy = oy0 + 2 * ty + o;
        if (oy >= a.Hout) continue;
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = acc[o][q][j][h];
        const long long off = (long long)co * HW + (long long)oy * a.Wout + ox;
        if (!raw) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v[q] += bv;
            if (res_b && ox + q < a.Wout) v[q] += res_b[off + q];
            if (a.act == ACT_RELU) v[q] = fmaxf(v[q], 0.0f);
            else if (a.act == ACT_RELU6) v[q] = fminf(fmaxf(v[q], 0.0f), 6.0f);
          }
        }
        if (vst) {
          *reinterpret_cast<float4*>(dst_b + off) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (ox + q < a.Wout) dst_b[off + q] = v[q];
        }
      }
    }
}

inline KernelEntry n8_entry() {
  KernelEntry e;
  e.fn = conv_n8_kernel<true>;
  e.fn_odd = conv_n8_kernel<false>;
  e.lds = N8::LDS_BYTES;
  e.threads = 256;
  return e;
}

}  // namespace
}  // namespace fdt
