// conv_kernel.h -- the implicit-GEMM convolution kernel template (see conv.hip for the design notes).
// Included by conv.hip (table, launch) and by one conv_inst_*.hip per convolution class so that the
// ~70 instantiations compile in parallel.
#pragma once
#include <type_traits>

#include "common.h"
#include "conv.h"

namespace fdt {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KH_, int KW_, int S_, int D_, int PAD_, int KC_>
struct Geom {
  static constexpr int KH = KH_, KW = KW_, S = S_, D = D_, PAD = PAD_, KC = KC_;
  static constexpr int TAPS = KH * KW;
  // 1x1 strided convs stage only the pixels they use (patch sampling stride = conv stride)
  static constexpr int PS = (KH == 1 && KW == 1) ? S : 1;
  static constexpr int LS = S / PS;  // lane-to-lane stride inside the staged patch
};

template <int TH_, int TW_, int BN_, int WM_, int WN_, int NBUF_>
struct Tile {
  static constexpr int TH = TH_, TW = TW_, BM = TH_ * TW_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int NBUF = NBUF_;   // LDS stage buffers: 2 = double buffer, 3 = ring, one stage further ahead
  static constexpr int MI = BM / (WM * 32), NI = BN / (WN * 32);
  // workgroups per CU we want resident (second __launch_bounds__ argument = waves per SIMD)
  static constexpr int MIN_WAVES = (MI * NI >= 4) ? 3 : 4;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(MI * WM * 32 == BM && NI * WN * 32 == BN, "tile/wave mismatch");
};

template <class G, class T>
struct Layout {
  static constexpr int PH = (T::TH - 1) * G::LS + (G::KH - 1) * G::D + 1;
  static constexpr int PW = (T::TW - 1) * G::LS + (G::KW - 1) * G::D + 1;
  static constexpr int XPLANE = PH * PW;
  static constexpr int XSZ = G::KC * XPLANE;
  static constexpr int XSZP = (XSZ + 255) / 256 * 256;   // whole LDS-DMA wave-instructions (64 x 4 B)
  static constexpr int WSZ = G::KC * G::TAPS * T::BN;     // floats per weight stage
  static constexpr int WSZP = (WSZ + 1023) / 1024 * 1024; // whole LDS-DMA wave-instructions (64 x 16 B)
  static constexpr int STAGE = XSZP + WSZP;              // floats per LDS buffer
  static constexpr int NX = XSZP / 256;                  // dword LDS-DMA instructions per wave per stage
  static constexpr int NW = WSZP / 1024;                 // dwordx4 LDS-DMA instructions per wave per stage
  // epilogue transpose tile: one NI slice at a time, [WN*32 couts][BM pixels (+4 pad)] floats
  static constexpr int EROW = T::BM + 4;
  static constexpr int EPI = T::WN * 32 * EROW;
  static constexpr int RING = T::NBUF * STAGE;
  static constexpr size_t LDS_BYTES = (size_t)(RING > EPI ? RING : EPI) * sizeof(float);
  // 1x1 stride-1: the patch IS the output tile, rows are 16-byte multiples -> stage X with dwordx4 DMA
  static constexpr bool VECX = (G::KH == 1 && G::KW == 1 && G::S == 1 && (XSZ % 1024) == 0);
  static constexpr int NXV = XSZ / 1024;
  static constexpr int LOADS = NX + NW;                  // LDS-DMA wave-instructions per stage
  static_assert((T::NBUF >= 4 ? 2 : 1) * LOADS < 64, "vmcnt is 6 bits");
  static_assert(G::KC % 2 == 0, "k-pairs are two input channels at one tap");
  static_assert(NX <= 32, "per-lane staging offsets live in registers");
};

__device__ float g_zero_pad[4];   // source of every padded / out-of-image element (zero-initialised)

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA: each lane's 4 / 16 bytes go global -> LDS without touching VGPRs.  The LDS address is the
// wave-uniform `l` + lane * size; the global address is per lane.
__device__ __forceinline__ void glds4(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 4, 0, 0);
}
__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

// LDS-DMA through a buffer descriptor: lane address = base + scalar offset + per-lane offset, bounds-checked against num_records
// INCLUDING the scalar offset on gfx950 (tools/microbench/buffer_oob_probe.hip) -- an out-of-range lane stages zeros.  The conv
// kernels use it so that a stage's addresses are ONE scalar add (the channel / stage offset) on per-lane offsets that never
// change: padding and out-of-image elements carry kOob, the channels past Cin fall off the end of the descriptor.  (The matrix
// pipe and the vector ALU of a SIMD do not co-execute here -- SQ_VALU_MFMA_COEXEC_CYCLES = 0 -- so the per-load pointer selects
// and 64-bit adds of the global_load form were paid in MFMA time.)
constexpr unsigned kOob = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* p, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes > 0x7fffffffll ? 0x7fffffff : (int)bytes, 0x00020000);
}
__device__ __forceinline__ void bglds4(__amdgpu_buffer_rsrc_t r, float* l, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)l, 4, voff, soff, 0, 0);
}
__device__ __forceinline__ void bglds16(__amdgpu_buffer_rsrc_t r, float* l, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lptr_t)l, 16, voff, soff, 0, 0);
}

// Compile-time loop: f(std::integral_constant<int, I>{}) for I = I0 .. N-1 (the step index feeds "n" asm constraints).
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// One LDS dword per lane, issued by hand (see the main loop): byte offset OFF from the lane's byte address.
template <int OFF>
__device__ __forceinline__ void lds_read_b32(float& v, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

template <class G, class T>
__global__ __launch_bounds__(256, T::MIN_WAVES) void conv_kernel(const ConvArgs a) {
  using L = Layout<G, T>;
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
  const int nstages = (a.Cin + G::KC - 1) / G::KC;
  const float* w_t = a.w + (long long)n_tile * nstages * L::WSZP;
  // this workgroup's share of the reduction (split-K over input-channel stages)
  const int s_begin = (int)((long long)nstages * ks / a.ksplit);
  const int s_end = (int)((long long)nstages * (ks + 1) / a.ksplit);

  // ---- per-lane staging plan (invariant over the stages) --------------------------------------------
  // Element e = 256*k + tid of the [KC][PH][PW] patch is fetched by lane (tid & 63) of wave (tid >> 6)
  // with its k-th LDS-DMA instruction: a buffer load at byte offset xoff[k] (relative to the stage's first
  // channel) -- or kOob for padding / out-of-image elements, which the descriptor's bounds check turns into
  // zeros, like the channels past Cin.  The loads are unconditional and carry no per-stage vector arithmetic.
  const __amdgpu_buffer_rsrc_t xrs = buf_rsrc(in_b, (long long)a.Cin * HWin * 4);
  const __amdgpu_buffer_rsrc_t wrs = buf_rsrc(w_t, 0x7fffffffll);
  unsigned xoff[L::NX];
  const bool vecx = L::VECX && (a.Win % 4 == 0);
  if (L::VECX && vecx) {
    // float4 v = 256*k + tid covers 4 consecutive pixels of one tile row of one channel
#pragma unroll
    for (int k = 0; k < L::NXV; ++k) {
      int v = tid + 256 * k;
      int c = v / (T::BM / 4);
      int p = (v - c * (T::BM / 4)) * 4;
      int gy = oy0 + p / T::TW, gx = ox0 + p % T::TW;
      bool ok = gy < a.Hin && gx < a.Win;
      xoff[k] = ok ? (unsigned)(c * HWin + gy * a.Win + gx) * 4u : kOob;
    }
  } else
#pragma unroll
  for (int k = 0; k < L::NX; ++k) {
    int e = tid + 256 * k;
    int c = e / L::XPLANE;
    int r = e - c * L::XPLANE;
    int yy = r / L::PW, xx = r - yy * L::PW;
    int gy = oy0 * G::S - G::PAD + yy * G::PS;
    int gx = ox0 * G::S - G::PAD + xx * G::PS;
    bool ok = (e < L::XSZ) && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
    xoff[k] = ok ? (unsigned)(c * HWin + gy * a.Win + gx) * 4u : kOob;
  }

#define FDT_STAGE(s_, buf_)                                                                 \
  {                                                                                         \
    const unsigned xso_ = (unsigned)((s_) * G::KC) * (unsigned)HWin * 4u;                   \
    if (L::VECX && vecx) {                                                                  \
      float* X_ = smem + (buf_) * L::STAGE + wave * 256;                                    \
      _Pragma("unroll") for (int k = 0; k < L::NXV; ++k) bglds16(xrs, X_ + 1024 * k, xoff[k], xso_); \
    } else {                                                                                \
      float* X_ = smem + (buf_) * L::STAGE + wave * 64;                                     \
      _Pragma("unroll") for (int k = 0; k < L::NX; ++k) bglds4(xrs, X_ + 256 * k, xoff[k], xso_);    \
    }                                                                                       \
    const unsigned wso_ = (unsigned)((s_) * L::WSZP) * 4u;                                  \
    float* W_ = smem + (buf_) * L::STAGE + L::XSZP + wave * 256;                            \
    _Pragma("unroll") for (int k = 0; k < L::NW; ++k) bglds16(wrs, W_ + 1024 * k, (unsigned)tid * 16u, wso_ + 4096u * k); \
  }

  // ---- per-lane LDS read offsets ------------------------------------------------------------------
  int xo[T::MI], wo[T::NI];
#pragma unroll
  for (int i = 0; i < T::MI; ++i) {
    int p = wm * (T::MI * 32) + i * 32 + l31;
    int py = p / T::TW, px = p % T::TW;
    xo[i] = half * L::XPLANE + py * G::LS * L::PW + px * G::LS;
  }
#pragma unroll
  for (int j = 0; j < T::NI; ++j)
    wo[j] = L::XSZP + half * G::TAPS * T::BN + wn * (T::NI * 32) + j * 32 + l31;

  f32x16 acc[T::NI][T::MI];
#pragma unroll
  for (int j = 0; j < T::NI; ++j)
#pragma unroll
    for (int i = 0; i < T::MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.0f;

  // ---- main loop: LDS ring of NBUF stages ---------------------------------------------------------------
  // Stage it+NBUF-1 is issued (LDS-DMA, nothing in VGPRs) right after the barrier that retires the buffer
  // it overwrites; stage `it` is waited for with a COUNTED vmcnt (the newer stages stay in flight) and a
  // raw s_barrier -- __syncthreads() would drain vmcnt(0) and serialise the ring.
  const int nst = s_end - s_begin;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
#pragma unroll
  for (int p = 0; p < T::NBUF - 1; ++p)
    if (p < nst) FDT_STAGE(s_begin + p, p);
  int cur = 0, nxt = T::NBUF - 1;
  for (int it = 0; it < nst; ++it) {
    // leave exactly the newer stages' LDS-DMA instructions in flight: min(NBUF - 2, stages left) of them
    // (their instruction count differs between the dword and the dwordx4 staging of X)
    {
      const int ahead = nst - 1 - it;
      const bool vx = L::VECX && vecx;
      if (T::NBUF >= 4 && ahead >= 2) {
        if (vx) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (L::NXV + L::NW)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * L::LOADS) : "memory");
      } else if (T::NBUF >= 3 && ahead >= 1) {
        if (vx) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::NXV + L::NW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::LOADS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (it + T::NBUF - 1 < nst) FDT_STAGE(s_begin + it + T::NBUF - 1, nxt);
    // Operands of step s + 1 are requested before the MFMAs of step s are issued (two register sets), and every set is
    // waited for with an EXACT lgkmcnt: the reads are issued through inline asm because the compiler's waitcnt model books
    // each outstanding LDS-DMA load as a FLAT access and would turn its own LDS waits into lgkmcnt(0) right after the
    // reads -- which left the matrix pipe idle for one LDS round trip per step (most of the time of the short-K 1x1
    // layers).  Same step order as the plain loop (taps outer, channel pairs inner), so sums are bit-identical.
    // tools/check_async_lds.py lints the generated ISA for any use of a register of a still-outstanding read.
    {
      constexpr int NSTEP = G::TAPS * (G::KC / 2);
      constexpr int NLD = T::NI + T::MI;
      const unsigned sb = lds0 + (unsigned)(cur * L::STAGE) * 4u;
      unsigned wa[T::NI], xa[T::MI];
#pragma unroll
      for (int j = 0; j < T::NI; ++j) wa[j] = sb + (unsigned)wo[j] * 4u;
#pragma unroll
      for (int i = 0; i < T::MI; ++i) xa[i] = sb + (unsigned)xo[i] * 4u;
      struct Ops {
        float r[NLD];       // [0, NI): weights, [NI, NLD): pixels
      };
      auto load = [&](Ops& o, auto sc) {
        constexpr int s_ = decltype(sc)::value;
        constexpr int t_ = s_ / (G::KC / 2), cp_ = s_ % (G::KC / 2);
        constexpr int kx_ = (2 * cp_) * L::XPLANE + (t_ / G::KW) * G::D * L::PW + (t_ % G::KW) * G::D;
        constexpr int kw_ = ((2 * cp_) * G::TAPS + t_) * T::BN;
#pragma unroll
        for (int j = 0; j < T::NI; ++j) lds_read_b32<kw_ * 4>(o.r[j], wa[j]);
#pragma unroll
        for (int i = 0; i < T::MI; ++i) lds_read_b32<kx_ * 4>(o.r[T::NI + i], xa[i]);
      };
      auto wait_for = [&](Ops& o, auto newer_c) {
        constexpr int N_ = decltype(newer_c)::value;
        if constexpr (NLD == 2)
          asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(o.r[0]), "+v"(o.r[1]) : "n"(N_));
        else if constexpr (NLD == 3)
          asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(o.r[0]), "+v"(o.r[1]), "+v"(o.r[2]) : "n"(N_));
        else
          asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(o.r[0]), "+v"(o.r[1]), "+v"(o.r[2]), "+v"(o.r[3]) : "n"(N_));
      };
      static_assert(NLD >= 2 && NLD <= 4, "operand sets of 2..4 registers");
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
        for (int j = 0; j < T::NI; ++j)
#pragma unroll
          for (int i = 0; i < T::MI; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.r[j], o.r[T::NI + i], acc[j][i], 0, 0, 0);
      });
    }
    cur = (cur + 1 == T::NBUF) ? 0 : cur + 1;
    nxt = (nxt + 1 == T::NBUF) ? 0 : nxt + 1;
  }
#undef FDT_STAGE

  // ---- epilogue -------------------------------------------------------------------------------------
  const int co_base = n_tile * T::BN + wn * (T::NI * 32) + 4 * half;
  if (a.Wout % 4 == 0) {
    // Vector path: transpose each 32-cout slice through LDS so that every lane owns 4 consecutive
    // pixels of one output channel -> 16-byte residual loads and stores (4x fewer VMEM instructions,
    // whole 64/128-byte row segments per 4/8 lanes).  The stage ring is dead by now and is reused.
    float* E = smem;
    float* dst_b;
    long long dst_cstride = HWout;
    const bool raw = a.ws != nullptr;
    if (raw)
      dst_b = a.ws + ((long long)(b * a.ksplit + ks) * a.Cout) * HWout;
    else
      dst_b = a.out + ((long long)b * a.out_ctot + a.out_coff) * HWout;
    const float* res_b = (!raw && a.res) ? a.res + ((long long)b * a.res_ctot + a.res_coff) * HWout : nullptr;
    constexpr int ROWS = T::WN * 32;
    constexpr int C4 = T::BM / 4;
    constexpr int PER = (ROWS * C4 + 255) / 256;
#pragma unroll
    for (int j = 0; j < T::NI; ++j) {
      __syncthreads();   // previous pass consumed / main loop finished reading the ring
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
            const long long off = (long long)co * dst_cstride + (long long)oy * a.Wout + ox;
            if (!raw) {
              if (a.bias) {
                const float bv = a.bias[co];
                v.x += bv; v.y += bv; v.z += bv; v.w += bv;
              }
              if (G::KH == 1 && G::S == 1 && a.up) {   // ContextTexture: + bilinear x2 of the coarser map, before the residual like the reduce pass
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
            if (a.out2 && co >= a.out2_from)      // second destination (conv.h: ConvArgs.out2; never with split-K / residual)
              dst = a.out2 + ((long long)b * a.out2_ctot + a.out2_coff + (co - a.out2_from)) * HWout + (long long)oy * a.Wout + ox;
            slab_store4(dst, v.x, v.y, v.z, v.w, raw && a.sk_count);
          }
        }
      }
    }
    if (a.sk_count)
      splitk_combine_tile<256>(a, b, tile_id + a.n_sp * n_tile, n_tile * T::BN, T::BN, oy0, ox0, T::TH, T::TW, (unsigned*)smem);
    return;
  }
  if (a.ws) {
    // raw partial sums -> workspace [b][ks][Cout][HWout]; splitk_reduce_kernel finishes the layer
    float* ws = a.ws + ((long long)(b * a.ksplit + ks) * a.Cout) * HWout;
#pragma unroll
    for (int i = 0; i < T::MI; ++i) {
      const int p = wm * (T::MI * 32) + i * 32 + l31;
      const int oy = oy0 + p / T::TW, ox = ox0 + p % T::TW;
      const bool pix_ok = oy < a.Hout && ox < a.Wout;
      const int pix = oy * a.Wout + ox;
#pragma unroll
      for (int j = 0; j < T::NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = co_base + j * 32 + (r & 3) + 8 * (r >> 2);
          if (pix_ok && co < a.Cout) ws[(long long)co * HWout + pix] = acc[j][i][r];
        }
    }
    return;
  }
  float* out_b = a.out + ((long long)b * a.out_ctot + a.out_coff) * HWout;
  const float* res_b = a.res ? a.res + ((long long)b * a.res_ctot + a.res_coff) * HWout : nullptr;
#pragma unroll
  for (int i = 0; i < T::MI; ++i) {
    const int p = wm * (T::MI * 32) + i * 32 + l31;
    const int oy = oy0 + p / T::TW, ox = ox0 + p % T::TW;
    const bool pix_ok = oy < a.Hout && ox < a.Wout;
    const int pix = pix_ok ? oy * a.Wout + ox : 0;
#pragma unroll
    for (int j = 0; j < T::NI; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co_base + j * 32 + (r & 3) + 8 * (r >> 2);
        const int coc = co < a.Cout ? co : a.Cout - 1;
        float v = acc[j][i][r];
        if (a.bias) v += a.bias[coc];
        if (G::KH == 1 && G::S == 1 && a.up && pix_ok)
          add_upsampled_x2<1>(a.up + ((long long)b * a.Cout + coc) * a.up_h * a.up_w, a.up_h, a.up_w, oy, ox, &v);
        if (res_b) v += res_b[(long long)coc * HWout + pix];
        if (a.act == ACT_RELU) v = fmaxf(v, 0.0f);
        else if (a.act == ACT_RELU6) v = fminf(fmaxf(v, 0.0f), 6.0f);
        if (pix_ok && co < a.Cout) {
          if (a.out2 && co >= a.out2_from)
            a.out2[((long long)b * a.out2_ctot + a.out2_coff + (co - a.out2_from)) * HWout + pix] = v;
          else
            out_b[(long long)co * HWout + pix] = v;
        }
      }
    }
  }
}


struct KernelEntry {
  void (*fn)(const ConvArgs);
  size_t lds;
  int threads = 256;
  void (*fn_odd)(const ConvArgs) = nullptr;   // variant for Win % 4 != 0, where the class has one (conv_n8.h)
  int threads_odd = 0;                        // its block size where that differs (0: the same)
};

//                     TH  TW  BN  WM WN NBUF
using T_128x128    = Tile<8, 16, 128, 2, 2, 2>;
using T_128x64     = Tile<8, 16, 64, 2, 2, 2>;
using T_128x32     = Tile<8, 16, 32, 4, 1, 2>;
using T_64x64      = Tile<8, 8, 64, 2, 2, 2>;
using T_64x128     = Tile<8, 8, 128, 1, 4, 2>;
using T_128x128W   = Tile<4, 32, 128, 2, 2, 2>;   // "wide": 4 rows x 32 px -- one 128-byte row per half wave
using T_128x64W    = Tile<4, 32, 64, 2, 2, 2>;
using T_128x128R3  = Tile<8, 16, 128, 2, 2, 3>;   // ring of 3 LDS stages
using T_128x64R3   = Tile<8, 16, 64, 2, 2, 3>;
using T_64x64R3    = Tile<8, 8, 64, 2, 2, 3>;
using T_64x128R3   = Tile<8, 8, 128, 1, 4, 3>;
using T_128x128WR3 = Tile<4, 32, 128, 2, 2, 3>;
using T_128x64WR3  = Tile<4, 32, 64, 2, 2, 3>;
using T_128x32R3   = Tile<8, 16, 32, 4, 1, 3>;
using T_128x128R4  = Tile<8, 16, 128, 2, 2, 4>;   // ring of 4 LDS stages
using T_128x64R4   = Tile<8, 16, 64, 2, 2, 4>;
using T_64x64R4    = Tile<8, 8, 64, 2, 2, 4>;
using T_64x128R4   = Tile<8, 8, 128, 1, 4, 4>;

template <class G, class T>
KernelEntry entry() {
  return KernelEntry{conv_kernel<G, T>, Layout<G, T>::LDS_BYTES, 256};
}

template <class G>
void fill_row(KernelEntry* row) {
  row[TILE_128x128] = entry<G, T_128x128>();
  row[TILE_128x64] = entry<G, T_128x64>();
  row[TILE_128x32] = entry<G, T_128x32>();
  row[TILE_64x64] = entry<G, T_64x64>();
  row[TILE_64x128] = entry<G, T_64x128>();
  row[TILE_128x128W] = entry<G, T_128x128W>();
  row[TILE_128x64W] = entry<G, T_128x64W>();
  row[TILE_128x128R3] = entry<G, T_128x128R3>();
  row[TILE_128x64R3] = entry<G, T_128x64R3>();
  row[TILE_64x64R3] = entry<G, T_64x64R3>();
  row[TILE_64x128R3] = entry<G, T_64x128R3>();
  row[TILE_128x128WR3] = entry<G, T_128x128WR3>();
  row[TILE_128x64WR3] = entry<G, T_128x64WR3>();
  row[TILE_128x32R3] = entry<G, T_128x32R3>();
}

using G_1x1_S1 = Geom<1, 1, 1, 1, 0, 16>;
using G_1x1_S2 = Geom<1, 1, 2, 1, 0, 16>;
using G_1x1_S1_K32 = Geom<1, 1, 1, 1, 0, 32>;
using G_1x1_S1_K64 = Geom<1, 1, 1, 1, 0, 64>;
using G_3x3_S1 = Geom<3, 3, 1, 1, 1, 4>;
using G_3x3_S1_D2 = Geom<3, 3, 1, 2, 2, 4>;
using G_3x3_S2 = Geom<3, 3, 2, 1, 1, 4>;
using G_7x7_S2 = Geom<7, 7, 2, 1, 3, 2>;
using G_7x7_S4 = Geom<7, 7, 4, 1, 3, 2>;
using G_7x7_S2_P1 = Geom<7, 7, 2, 1, 1, 2>;
using G_5x5_S2 = Geom<5, 5, 2, 1, 2, 2>;

}  // namespace

// one per conv_inst_*.hip
void conv_fill_1x1_s1(void* row);
void conv_fill_1x1_s2(void* row);
void conv_fill_1x1_s1_deep(void* row_k32, void* row_k64);
void conv_fill_3x3_s1(void* row);
void conv_fill_3x3_s1_d2(void* row);
void conv_fill_3x3_s2(void* row);
void conv_fill_n8(void* row);   // conv_n8.h
void conv_fill_stems(void* row_7x7_s2, void* row_7x7_s4, void* row_5x5_s2, void* row_7x7_s2_p1);
void conv_fill_1x1_b3(void* row);   // conv_b3.h
void conv_fill_1x1_s2_b3(void* row);
void conv_fill_3x3_s2_b3(void* row);
void conv_fill_stem_b3(void* row);  // conv_stem_b3.h
void conv_fill_stem_u8b(void* row_s2, void* row_s4);  // conv_stem_u8b.h
void conv_fill_1x1_pb3(void* row);                    // conv_1x1p_b3.h
int conv_1x1pb3_resident(ConvTile t);

}  // namespace fdt
