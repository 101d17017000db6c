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
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(MI * WM * 32 == BM && NI * WN * 32 == BN, "tile/wave mismatch");
};

template <class G, class T>
struct Layout {
  static constexpr int PH = (T::TH - 1) * G::LS + (G::KH - 1) * G::D + 1;
  static constexpr int PW = (T::TW - 1) * G::LS + (G::KW - 1) * G::D + 1;
  static constexpr int XPLANE = PH * PW;
  static constexpr int XSZ = G::KC * XPLANE;
  static constexpr int XSZP = (XSZ + 3) / 4 * 4;
  static constexpr int WSZ = G::KC * G::TAPS * T::BN;
  static constexpr int STAGE = XSZP + WSZ;           // floats per LDS buffer
  static constexpr int NX = (XSZ + 255) / 256;       // x elements staged per thread
  static constexpr int NW4 = (WSZ / 4 + 255) / 256;  // float4 weights staged per thread
  static constexpr size_t LDS_BYTES = 2 * STAGE * sizeof(float);
  static_assert(WSZ % 4 == 0, "weight stage must be float4-able");
  static_assert(G::KC % 2 == 0, "k-pairs are two input channels at one tap");
};

template <class G, class T>
__global__ __launch_bounds__(256) void conv_kernel(const ConvArgs a) {
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
  const int b = blockIdx.z;

  const int HWin = a.Hin * a.Win;
  const int HWout = a.Hout * a.Wout;
  const float* in_b = a.in + (long long)b * a.Cin * HWin;
  const int nstages = (a.Cin + G::KC - 1) / G::KC;
  const float* w_t = a.w + (long long)n_tile * nstages * L::WSZ;

  // ---- per-thread staging plan (invariant over the stages) -----------------------------------------
  int goff[L::NX];   // offset inside the stage's channel block, or -1 if outside the image / patch
  int gch[L::NX];    // channel inside the stage
#pragma unroll
  for (int k = 0; k < L::NX; ++k) {
    int e = tid + 256 * k;
    int c = e / L::XPLANE;
    int r = e - c * L::XPLANE;
    int yy = r / L::PW, xx = r - yy * L::PW;
    int gy = oy0 * G::S - G::PAD + yy * G::PS;
    int gx = ox0 * G::S - G::PAD + xx * G::PS;
    bool ok = (e < L::XSZ) && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
    goff[k] = ok ? (c * HWin + gy * a.Win + gx) : -1;
    gch[k] = c;
  }

  float xr[L::NX];
  float4 wr[L::NW4];

  auto load_stage = [&](int s) {
    const int c0 = s * G::KC;
    const float* src = in_b + (long long)c0 * HWin;
#pragma unroll
    for (int k = 0; k < L::NX; ++k) {
      bool ok = goff[k] >= 0 && (c0 + gch[k]) < a.Cin;
      xr[k] = ok ? src[goff[k]] : 0.0f;
    }
    const float4* wsrc = reinterpret_cast<const float4*>(w_t + (long long)s * L::WSZ);
#pragma unroll
    for (int k = 0; k < L::NW4; ++k) {
      int v = tid + 256 * k;
      if (v < L::WSZ / 4) wr[k] = wsrc[v];
    }
  };
  auto store_stage = [&](int buf) {
    float* X = smem + buf * L::STAGE;
    float4* W4 = reinterpret_cast<float4*>(X + L::XSZP);
#pragma unroll
    for (int k = 0; k < L::NX; ++k) {
      int e = tid + 256 * k;
      if (e < L::XSZ) X[e] = xr[k];
    }
#pragma unroll
    for (int k = 0; k < L::NW4; ++k) {
      int v = tid + 256 * k;
      if (v < L::WSZ / 4) W4[v] = wr[k];
    }
  };

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

  load_stage(0);
  store_stage(0);
  __syncthreads();

  for (int s = 0; s < nstages; ++s) {
    const bool more = (s + 1) < nstages;
    if (more) load_stage(s + 1);
    const float* S = smem + (s & 1) * L::STAGE;
#pragma unroll
    for (int t = 0; t < G::TAPS; ++t) {
#pragma unroll
      for (int cp = 0; cp < G::KC / 2; ++cp) {
        constexpr int dummy = 0;
        (void)dummy;
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
    if (more) store_stage((s + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue -------------------------------------------------------------------------------------
  float* out_b = a.out + ((long long)b * a.out_ctot + a.out_coff) * HWout;
  const float* res_b = a.res ? a.res + ((long long)b * a.res_ctot + a.res_coff) * HWout : nullptr;
  const float* up_b = a.up ? a.up + (long long)b * a.Cout * a.up_h * a.up_w : nullptr;
#pragma unroll
  for (int i = 0; i < T::MI; ++i) {
    const int p = wm * (T::MI * 32) + i * 32 + l31;
    const int oy = oy0 + p / T::TW, ox = ox0 + p % T::TW;
    const bool pix_ok = oy < a.Hout && ox < a.Wout;
    const int pix = oy * a.Wout + ox;
    // bilinear x2, align_corners=False (F.interpolate, pyramid.py:65): src = 0.5*(dst+0.5)-0.5, >= 0
    int y0 = 0, y1 = 0, x0 = 0, x1 = 0;
    float ly = 0.f, lx = 0.f;
    if (up_b) {
      float sy = fmaxf(0.5f * (oy + 0.5f) - 0.5f, 0.0f);
      float sx = fmaxf(0.5f * (ox + 0.5f) - 0.5f, 0.0f);
      y0 = (int)sy;
      x0 = (int)sx;
      y0 = y0 < a.up_h - 1 ? y0 : a.up_h - 1;
      x0 = x0 < a.up_w - 1 ? x0 : a.up_w - 1;
      y1 = y0 + (y0 < a.up_h - 1 ? 1 : 0);
      x1 = x0 + (x0 < a.up_w - 1 ? 1 : 0);
      ly = sy - (float)y0;
      lx = sx - (float)x0;
    }
#pragma unroll
    for (int j = 0; j < T::NI; ++j) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = n_tile * T::BN + wn * (T::NI * 32) + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (pix_ok && co < a.Cout) {
          float v = acc[j][i][r];
          if (a.bias) v += a.bias[co];
          if (up_b) {
            const float* u = up_b + (long long)co * a.up_h * a.up_w;
            float top = (1.0f - lx) * u[y0 * a.up_w + x0] + lx * u[y0 * a.up_w + x1];
            float bot = (1.0f - lx) * u[y1 * a.up_w + x0] + lx * u[y1 * a.up_w + x1];
            v += (1.0f - ly) * top + ly * bot;
          }
          if (res_b) v += res_b[(long long)co * HWout + pix];
          if (a.act == ACT_RELU) v = fmaxf(v, 0.0f);
          else if (a.act == ACT_RELU6) v = fminf(fmaxf(v, 0.0f), 6.0f);
          out_b[(long long)co * HWout + pix] = v;
        }
      }
    }
  }
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
  out.assign((size_t)n_tiles * nstages * KC * taps * BN, 0.0f);
  for (int co = 0; co < Cout; ++co) {
    const float sc = scale ? scale[co] : 1.0f;
    const int nt = co / BN, n = co % BN;
    for (int ci = 0; ci < Cin; ++ci) {
      const int s = ci / KC, c = ci % KC;
      const float* src = w + ((size_t)co * Cin + ci) * taps;
      float* dst = out.data() + ((((size_t)nt * nstages + s) * KC + c) * taps) * BN + n;
      for (int t = 0; t < taps; ++t) dst[(size_t)t * BN] = src[t] * sc;
    }
  }
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
  const int tiles = ceil_div(a.Hout, tile_th(tile)) * ceil_div(a.Wout, tile_tw(tile));
  dim3 grid(tiles, ceil_div(a.Cout, tile_bn(tile)), a.B);
  hipLaunchKernelGGL(ke.fn, grid, dim3(256), ke.lds, st, a);
  FDT_LAUNCH_CHECK();
  return FDT_OK;
}

}  // namespace fdt
