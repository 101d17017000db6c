// Implicit-GEMM convolution for gfx950 on the f32-input matrix cores.
//
// Replaces every nn.Conv2d (+ folded eval BatchNorm, + ReLU/ReLU6, + residual add, + the bilinear x2
// upsample-add of ContextTexture) on the reference's forward path:
//   pyramid.py:229-266 (backbone, LFPN, SSH), :291-306 (heads); pyramid_mb2_try3.py:218-340;
//   FACEBOX/networks.py:87-116.
//
// Design (MI355X-first, not a translated CUDA tiling):
//  * GEMM view: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel], k = (tap, cin).  The *weights* are
//    the MFMA A operand (rows) and the *pixels* the B operand (columns), so the 32x32 accumulator has
//    32 consecutive output pixels on lanes 0..31: stores to the NCHW output are 128-byte row segments
//    with no transpose, and the NCHW input is read along its contiguous W axis.
//  * v_mfma_f32_32x32x2_f32: exact f32 (bitwise an fmaf chain), 256 FLOP/clk/CU.  Each operand is ONE
//    dword per lane, so a k-pair costs one ds_read_b32 per 32x32 tile edge -- LDS bandwidth is never
//    the limiter at this rate; what matters is keeping the matrix pipe issuing back-to-back.
//  * No im2col in memory: per stage a workgroup stages the (TH*s + halo) x (TW*s + halo) input patch
//    of KC channels once in LDS and reads the kh*kw taps as shifted views (compile-time immediates),
//    so global->LDS traffic is ~1.4x the input instead of 9x for a 3x3.
//  * Weights are pre-tiled on the host to [n_tile][stage][KC][taps][BN]: each stage is one contiguous,
//    16-byte-vectorised stream.
//  * Register-staged double buffering: stage s+1's global loads are issued before stage s's MFMAs and
//    written to the other LDS buffer after them; one barrier per stage.  256 threads = 4 waves (one per
//    SIMD); 2-3 workgroups per CU hide the rest of the latency.
//  * Fused epilogue: + bias (folded BN), + residual, + bilinear-upsampled coarser map, ReLU/ReLU6,
//    direct write into a channel slice of the destination (kills torch.cat / permute).
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

template <int TH_, int TW_, int BN_, int WM_, int WN_>
struct Tile {
  static constexpr int TH = TH_, TW = TW_, BM = TH_ * TW_, BN = BN_, WM = WM_, WN = WN_;
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
  static constexpr size_t LDS_BYTES = 2 * STAGE * sizeof(float);
  static_assert(G::KC % 2 == 0, "k-pairs are two input channels at one tap");
  static_assert(NX <= 32, "okmask is 32 bits");
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
  const int tile_id = blockIdx.x;
  const int oy0 = (tile_id / tiles_x) * T::TH;
  const int ox0 = (tile_id % tiles_x) * T::TW;
  const int n_tile = blockIdx.y;
  const int b = blockIdx.z / a.ksplit;
  const int ks = blockIdx.z - b * a.ksplit;

  const int HWin = a.Hin * a.Win;
  const int HWout = a.Hout * a.Wout;
  const float* in_b = a.in + (long long)b * a.Cin * HWin;
  const int nstages = (a.Cin + G::KC - 1) / G::KC;
  const float* w_t = a.w + (long long)n_tile * nstages * L::WSZP;
  // this workgroup's share of the reduction (split-K over input-channel stages)
  const int s_begin = (int)((long long)nstages * ks / a.ksplit);
  const int s_end = (int)((long long)nstages * (ks + 1) / a.ksplit);

  // ---- per-lane staging plan (invariant over the stages) --------------------------------------------
  // Element e = 256*k + tid of the [KC][PH][PW] patch is fetched by lane (tid & 63) of wave (tid >> 6)
  // with its k-th LDS-DMA instruction.  Padding / out-of-image elements read g_zero_pad instead, so
  // the loads are unconditional and nothing is predicated per lane.
  int goff[L::NX];
  unsigned okmask = 0;
#pragma unroll
  for (int k = 0; k < L::NX; ++k) {
    int e = tid + 256 * k;
    int c = e / L::XPLANE;
    int r = e - c * L::XPLANE;
    int yy = r / L::PW, xx = r - yy * L::PW;
    int gy = oy0 * G::S - G::PAD + yy * G::PS;
    int gx = ox0 * G::S - G::PAD + xx * G::PS;
    bool ok = (e < L::XSZ) && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
    goff[k] = ok ? (c * HWin + gy * a.Win + gx) : 0;
    if (ok) okmask |= (1u << k);
  }

#define FDT_STAGE(s_, buf_)                                                                 \
  {                                                                                         \
    const int c0_ = (s_) * G::KC;                                                           \
    const float* src_ = in_b + (long long)c0_ * HWin;                                       \
    const int crem_ = a.Cin - c0_;                                                          \
    float* X_ = smem + (buf_) * L::STAGE + wave * 64;                                       \
    _Pragma("unroll") for (int k = 0; k < L::NX; ++k) {                                     \
      const int c_ = (tid + 256 * k) / L::XPLANE;                                           \
      const bool ok_ = ((okmask >> k) & 1u) && c_ < crem_;                                  \
      glds4(ok_ ? src_ + goff[k] : g_zero_pad, X_ + 256 * k);                               \
    }                                                                                       \
    const float* wsrc_ = w_t + (long long)(s_) * L::WSZP + tid * 4;                         \
    float* W_ = smem + (buf_) * L::STAGE + L::XSZP + wave * 256;                            \
    _Pragma("unroll") for (int k = 0; k < L::NW; ++k) glds16(wsrc_ + 1024 * k, W_ + 1024 * k); \
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

  if (s_begin < s_end) FDT_STAGE(s_begin, 0);
  __syncthreads();   // drains the LDS-DMA (vmcnt(0)) and publishes the stage

  for (int s = s_begin; s < s_end; ++s) {
    const int cur = (s - s_begin) & 1;
    // the other buffer was last read in the previous iteration, which ended with a barrier
    if (s + 1 < s_end) FDT_STAGE(s + 1, cur ^ 1);
    const float* S = smem + cur * L::STAGE;
#pragma unroll
    for (int t = 0; t < G::TAPS; ++t) {
#pragma unroll
      for (int cp = 0; cp < G::KC / 2; ++cp) {
        const int kx = (2 * cp) * L::XPLANE + (t / G::KW) * G::D * L::PW + (t % G::KW) * G::D;
        const int kw = ((2 * cp) * G::TAPS + t) * T::BN;
        float av[T::NI], bv[T::MI];
#pragma unroll
        for (int j = 0; j < T::NI; ++j) av[j] = S[wo[j] + kw];
#pragma unroll
        for (int i = 0; i < T::MI; ++i) bv[i] = S[xo[i] + kx];
#pragma unroll
        for (int j = 0; j < T::NI; ++j)
#pragma unroll
          for (int i = 0; i < T::MI; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[i], acc[j][i], 0, 0, 0);
      }
    }
    __syncthreads();   // next stage landed (vmcnt(0)) and everyone is done reading `cur`
  }
#undef FDT_STAGE

  // ---- epilogue -------------------------------------------------------------------------------------
  const int co_base = n_tile * T::BN + wn * (T::NI * 32) + 4 * half;
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
        if (res_b) v += res_b[(long long)coc * HWout + pix];
        if (a.act == ACT_RELU) v = fmaxf(v, 0.0f);
        else if (a.act == ACT_RELU6) v = fminf(fmaxf(v, 0.0f), 6.0f);
        if (pix_ok && co < a.Cout) out_b[(long long)co * HWout + pix] = v;
      }
    }
  }
}

// Second pass of a split-K layer: out = act(sum_ks partial + bias + upsample + residual).  Fixed
// summation order -> bitwise reproducible (no float atomics).  One thread per 4 consecutive pixels.
__global__ void splitk_reduce_kernel(const ConvArgs a) {
  const int HWout = a.Hout * a.Wout;
  const long long total = (long long)a.B * a.Cout * HWout;
  long long t = ((long long)blockIdx.x * blockDim.x + threadIdx.x);
  if (t >= total) return;
  const int pix = (int)(t % HWout);
  const int co = (int)((t / HWout) % a.Cout);
  const int b = (int)(t / ((long long)HWout * a.Cout));
  float v = 0.0f;
  const float* ws = a.ws + ((long long)b * a.ksplit * a.Cout + co) * HWout + pix;
  for (int k = 0; k < a.ksplit; ++k) v += ws[(long long)k * a.Cout * HWout];
  if (a.bias) v += a.bias[co];
  if (a.up) {
    const int oy = pix / a.Wout, ox = pix % a.Wout;
    float sy = fmaxf(0.5f * (oy + 0.5f) - 0.5f, 0.0f);
    float sx = fmaxf(0.5f * (ox + 0.5f) - 0.5f, 0.0f);
    int y0 = (int)sy, x0 = (int)sx;
    y0 = y0 < a.up_h - 1 ? y0 : a.up_h - 1;
    x0 = x0 < a.up_w - 1 ? x0 : a.up_w - 1;
    const int y1 = y0 + (y0 < a.up_h - 1 ? 1 : 0);
    const int x1 = x0 + (x0 < a.up_w - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float* u = a.up + ((long long)b * a.Cout + co) * a.up_h * a.up_w;
    const float top = (1.0f - lx) * u[y0 * a.up_w + x0] + lx * u[y0 * a.up_w + x1];
    const float bot = (1.0f - lx) * u[y1 * a.up_w + x0] + lx * u[y1 * a.up_w + x1];
    v += (1.0f - ly) * top + ly * bot;
  }
  if (a.res) v += a.res[((long long)b * a.res_ctot + a.res_coff + co) * HWout + pix];
  if (a.act == ACT_RELU) v = fmaxf(v, 0.0f);
  else if (a.act == ACT_RELU6) v = fminf(fmaxf(v, 0.0f), 6.0f);
  a.out[((long long)b * a.out_ctot + a.out_coff + co) * HWout + pix] = v;
}

// ---- kernel table ---------------------------------------------------------------------------------------
using G_1x1_S1 = Geom<1, 1, 1, 1, 0, 16>;
using G_1x1_S2 = Geom<1, 1, 2, 1, 0, 16>;
using G_3x3_S1 = Geom<3, 3, 1, 1, 1, 4>;
using G_3x3_S1_D2 = Geom<3, 3, 1, 2, 2, 4>;
using G_3x3_S2 = Geom<3, 3, 2, 1, 1, 4>;
using G_7x7_S2 = Geom<7, 7, 2, 1, 3, 2>;
using G_7x7_S4 = Geom<7, 7, 4, 1, 3, 2>;
using G_5x5_S2 = Geom<5, 5, 2, 1, 2, 2>;

using T_128x128 = Tile<8, 16, 128, 2, 2>;
using T_128x64 = Tile<8, 16, 64, 2, 2>;
using T_128x32 = Tile<8, 16, 32, 4, 1>;
using T_64x64 = Tile<8, 8, 64, 2, 2>;
using T_64x128 = Tile<8, 8, 128, 1, 4>;

struct KernelEntry {
  void (*fn)(const ConvArgs);
  size_t lds;
};

template <class G, class T>
KernelEntry entry() {
  return KernelEntry{conv_kernel<G, T>, Layout<G, T>::LDS_BYTES};
}

template <class G>
void fill_row(KernelEntry* row) {
  row[TILE_128x128] = entry<G, T_128x128>();
  row[TILE_128x64] = entry<G, T_128x64>();
  row[TILE_128x32] = entry<G, T_128x32>();
  row[TILE_64x64] = entry<G, T_64x64>();
  row[TILE_64x128] = entry<G, T_64x128>();
}

struct Table {
  KernelEntry e[CONV_KIND_COUNT][CONV_TILE_COUNT];
  bool attr_set[CONV_KIND_COUNT][CONV_TILE_COUNT];
  Table() {
    memset(e, 0, sizeof(e));
    memset(attr_set, 0, sizeof(attr_set));
    fill_row<G_1x1_S1>(e[CONV_1x1_S1]);
    fill_row<G_1x1_S2>(e[CONV_1x1_S2]);
    fill_row<G_3x3_S1>(e[CONV_3x3_S1]);
    fill_row<G_3x3_S1_D2>(e[CONV_3x3_S1_D2]);
    fill_row<G_3x3_S2>(e[CONV_3x3_S2]);
    // stems / FaceBox: only the tiles their Cout needs
    e[CONV_7x7_S2][TILE_128x64] = entry<G_7x7_S2, T_128x64>();
    e[CONV_7x7_S2][TILE_64x64] = entry<G_7x7_S2, T_64x64>();
    e[CONV_7x7_S4][TILE_128x32] = entry<G_7x7_S4, T_128x32>();
    e[CONV_5x5_S2][TILE_128x64] = entry<G_5x5_S2, T_128x64>();
    e[CONV_5x5_S2][TILE_64x64] = entry<G_5x5_S2, T_64x64>();
  }
};

Table& table() {
  static Table t;
  return t;
}

const ConvGeom kGeoms[CONV_KIND_COUNT] = {
    {1, 1, 1, 1, 0, 16}, {1, 1, 2, 1, 0, 16}, {3, 3, 1, 1, 1, 4}, {3, 3, 1, 2, 2, 4},
    {3, 3, 2, 1, 1, 4},  {7, 7, 2, 1, 3, 2},  {7, 7, 4, 1, 3, 2}, {5, 5, 2, 1, 2, 2},
};
const int kTileDims[CONV_TILE_COUNT][4] = {  // BM, BN, TH, TW
    {128, 128, 8, 16}, {128, 64, 8, 16}, {128, 32, 8, 16}, {64, 64, 8, 8}, {64, 128, 8, 8}};

}  // namespace

ConvGeom conv_geom(ConvKind k) { return kGeoms[k]; }
int tile_bm(ConvTile t) { return kTileDims[t][0]; }
int tile_bn(ConvTile t) { return kTileDims[t][1]; }
int tile_th(ConvTile t) { return kTileDims[t][2]; }
int tile_tw(ConvTile t) { return kTileDims[t][3]; }

bool conv_supported(ConvKind kind, ConvTile tile) { return table().e[kind][tile].fn != nullptr; }

void tile_weights(const float* w, const float* scale, int Cout, int Cin, ConvKind kind, ConvTile tile,
                  std::vector<float>& out) {
  const ConvGeom g = conv_geom(kind);
  const int taps = g.kh * g.kw, BN = tile_bn(tile), KC = g.kc;
  const int n_tiles = (Cout + BN - 1) / BN;
  const int nstages = (Cin + KC - 1) / KC;
  const size_t wszp = ((size_t)KC * taps * BN + 1023) / 1024 * 1024;   // Layout::WSZP
  out.assign((size_t)n_tiles * nstages * wszp, 0.0f);
  for (int co = 0; co < Cout; ++co) {
    const float sc = scale ? scale[co] : 1.0f;
    const int nt = co / BN, n = co % BN;
    for (int ci = 0; ci < Cin; ++ci) {
      const int s = ci / KC, c = ci % KC;
      const float* src = w + ((size_t)co * Cin + ci) * taps;
      float* dst = out.data() + ((size_t)nt * nstages + s) * wszp + ((size_t)c * taps) * BN + n;
      for (int t = 0; t < taps; ++t) dst[(size_t)t * BN] = src[t] * sc;
    }
  }
}

long long conv_ws_floats(const ConvArgs& a) {
  return (a.ksplit > 1 || a.up) ? (long long)a.B * a.ksplit * a.Cout * a.Hout * a.Wout : 0;
}

double conv_flops(const ConvArgs& a, ConvKind kind) {
  const ConvGeom g = conv_geom(kind);
  return 2.0 * a.B * (double)a.Hout * a.Wout * a.Cout * a.Cin * g.kh * g.kw;
}

int launch_conv(ConvKind kind, ConvTile tile, const ConvArgs& a, hipStream_t st) {
  KernelEntry& ke = table().e[kind][tile];
  FDT_REQUIRE(ke.fn, FDT_ERR_ARG, "launch_conv: kernel (kind %d, tile %d) not instantiated", kind, tile);
  const ConvGeom g = conv_geom(kind);
  // shape contract of the kernel: checked on the host before any launch
  const int eh = (a.Hin + 2 * g.pad - g.dil * (g.kh - 1) - 1) / g.stride + 1;
  const int ew = (a.Win + 2 * g.pad - g.dil * (g.kw - 1) - 1) / g.stride + 1;
  FDT_REQUIRE(eh == a.Hout && ew == a.Wout, FDT_ERR_ARG,
              "launch_conv: output %dx%d does not match input %dx%d for kind %d", a.Hout, a.Wout, a.Hin,
              a.Win, kind);
  FDT_REQUIRE(a.B >= 1 && a.Cin >= 1 && a.Cout >= 1 && a.out_coff >= 0 &&
                  a.out_coff + a.Cout <= a.out_ctot && a.in && a.w && a.out,
              FDT_ERR_ARG, "launch_conv: bad channel/pointer arguments");
  if (a.res) FDT_REQUIRE(a.res_coff + a.Cout <= a.res_ctot, FDT_ERR_ARG, "launch_conv: bad residual slice");
  if (a.up) FDT_REQUIRE(a.up_h * 2 >= a.Hout && a.up_w * 2 >= a.Wout && a.up_h >= 1 && a.up_w >= 1,
                        FDT_ERR_ARG, "launch_conv: upsample source too small");
  if (!table().attr_set[kind][tile]) {
    FDT_HIP(hipFuncSetAttribute((const void*)ke.fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)ke.lds));
    table().attr_set[kind][tile] = true;
  }
  const int nstages = ceil_div(a.Cin, g.kc);
  FDT_REQUIRE(a.ksplit >= 1 && a.ksplit <= nstages, FDT_ERR_ARG,
              "launch_conv: bad split-K %d (stages %d)", a.ksplit, nstages);
  // split-K layers and layers with the fused upsample-add finish in splitk_reduce_kernel
  FDT_REQUIRE(!(a.ksplit > 1 || a.up) || a.ws, FDT_ERR_ARG, "launch_conv: workspace required");
  FDT_REQUIRE(!a.ws || a.ksplit > 1 || a.up, FDT_ERR_ARG, "launch_conv: unexpected workspace");
  const int tiles = ceil_div(a.Hout, tile_th(tile)) * ceil_div(a.Wout, tile_tw(tile));
  dim3 grid(tiles, ceil_div(a.Cout, tile_bn(tile)), a.B * a.ksplit);
  FDT_REQUIRE(grid.y <= 65535 && grid.z <= 65535, FDT_ERR_ARG, "launch_conv: grid too large");
  hipLaunchKernelGGL(ke.fn, grid, dim3(256), ke.lds, st, a);
  FDT_LAUNCH_CHECK();
  if (a.ws) {
    const long long total = (long long)a.B * a.Cout * a.Hout * a.Wout;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)ceil_div_ll(total, 256)), dim3(256), 0, st, a);
    FDT_LAUNCH_CHECK();
  }
  return FDT_OK;
}

}  // namespace fdt

// ---------------------------------------------------------------------------------------------------
// Tuning hook (not part of include/fdt.h): time one conv configuration on random data with HIP events.
// Used by tools/conv_bench.py and tools/autotune.py.
extern "C" int fdt_debug_conv_bench(int kind, int tile, int ksplit, int B, int Cin, int Hin, int Win,
                                    int Cout, int has_res, int has_up, int act, int iters, float* ms_out) {
  using namespace fdt;
  FDT_REQUIRE(kind >= 0 && kind < CONV_KIND_COUNT && tile >= 0 && tile < CONV_TILE_COUNT && ms_out && iters >= 1,
              FDT_ERR_ARG, "fdt_debug_conv_bench: bad argument");
  FDT_REQUIRE(conv_supported((ConvKind)kind, (ConvTile)tile), FDT_ERR_ARG, "kernel not instantiated");
  const ConvGeom g = conv_geom((ConvKind)kind);
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.Cin = Cin; a.Hin = Hin; a.Win = Win; a.Cout = Cout;
  a.Hout = (Hin + 2 * g.pad - g.dil * (g.kh - 1) - 1) / g.stride + 1;
  a.Wout = (Win + 2 * g.pad - g.dil * (g.kw - 1) - 1) / g.stride + 1;
  a.out_ctot = Cout; a.res_ctot = Cout; a.act = act; a.ksplit = ksplit;
  std::vector<float> w((size_t)Cout * Cin * g.kh * g.kw), tiled;
  unsigned s = 12345u;
  for (auto& v : w) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  tile_weights(w.data(), nullptr, Cout, Cin, (ConvKind)kind, (ConvTile)tile, tiled);
  DevBuf din, dw, db, dout, dres, dup, dws;
  const size_t n_in = (size_t)B * Cin * Hin * Win, n_out = (size_t)B * Cout * a.Hout * a.Wout;
  FDT_TRY(din.alloc(n_in * 4)); FDT_TRY(dw.alloc(tiled.size() * 4)); FDT_TRY(db.alloc((size_t)Cout * 4));
  FDT_TRY(dout.alloc(n_out * 4));
  std::vector<float> hin(n_in);
  for (auto& v : hin) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  FDT_HIP(hipMemcpy(din.p, hin.data(), n_in * 4, hipMemcpyHostToDevice));
  FDT_HIP(hipMemcpy(dw.p, tiled.data(), tiled.size() * 4, hipMemcpyHostToDevice));
  FDT_HIP(hipMemset(db.p, 0, (size_t)Cout * 4));
  a.in = din.as<float>(); a.w = dw.as<float>(); a.bias = db.as<float>(); a.out = dout.as<float>();
  if (has_res) { FDT_TRY(dres.alloc(n_out * 4)); FDT_HIP(hipMemset(dres.p, 0, n_out * 4)); a.res = dres.as<float>(); }
  if (has_up) {
    a.up_h = (a.Hout + 1) / 2; a.up_w = (a.Wout + 1) / 2;
    FDT_TRY(dup.alloc((size_t)B * Cout * a.up_h * a.up_w * 4));
    FDT_HIP(hipMemset(dup.p, 0, (size_t)B * Cout * a.up_h * a.up_w * 4));
    a.up = dup.as<float>();
  }
  if (ksplit > 1 || has_up) { FDT_TRY(dws.alloc((size_t)conv_ws_floats(a) * 4)); a.ws = dws.as<float>(); }
  hipEvent_t e0, e1;
  FDT_HIP(hipEventCreate(&e0)); FDT_HIP(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) FDT_TRY(launch_conv((ConvKind)kind, (ConvTile)tile, a, 0));
  FDT_HIP(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) FDT_TRY(launch_conv((ConvKind)kind, (ConvTile)tile, a, 0));
  FDT_HIP(hipEventRecord(e1, 0));
  FDT_HIP(hipEventSynchronize(e1));
  float ms = 0;
  FDT_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return FDT_OK;
}
