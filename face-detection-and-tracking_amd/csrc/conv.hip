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
//  * Both operands arrive by LDS-DMA (global_load_lds_dword / _dwordx4): nothing is staged in VGPRs.  LDS is a
//    ring of 2-4 stages; stage i is retired with a counted s_waitcnt vmcnt(N) + raw s_barrier so that the
//    newer stages stay in flight.  256 threads = 4 waves (one per SIMD); 2-3 workgroups per CU hide the rest.
//  * Fused epilogue: + bias (folded BN), + residual, + bilinear-upsampled coarser map (in the conv epilogue
//    itself when the layer is not split along K, else in splitk_reduce_kernel), ReLU/ReLU6, direct write into
//    a channel slice of the destination (kills torch.cat / permute).
#include <atomic>

#include "conv_kernel.h"
#include "conv_wino.h"
#include "conv_wino44.h"
#include "conv_1x1p.h"
#include "conv_stem_u8.h"
#include "conv_stem_s4.h"

namespace fdt {
namespace {

// Second pass of a split-K layer: out = act(sum_ks partial + bias + upsample + residual).  Fixed
// summation order -> bitwise reproducible (no float atomics).  VEC = 4: one thread per 4 consecutive
// pixels with 16-byte accesses (Wout % 4 == 0), else one pixel per thread.
template <int VEC>
__global__ void splitk_reduce_kernel(const ConvArgs a) {
  const int HWout = a.Hout * a.Wout;
  const int HWv = HWout / VEC;
  const long long total = (long long)a.B * a.Cout * HWv;
  long long t = ((long long)blockIdx.x * blockDim.x + threadIdx.x);
  if (t >= total) return;
  const int pix = (int)(t % HWv) * VEC;
  const int co = (int)((t / HWv) % a.Cout);
  const int b = (int)(t / ((long long)HWv * a.Cout));
  float v[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) v[e] = 0.0f;
  const float* ws = a.ws + ((long long)b * a.ksplit * a.Cout + co) * HWout + pix;
  const long long kstride = (long long)a.Cout * HWout;
  for (int k = 0; k < a.ksplit; ++k) {
    if (VEC == 4) {
      const float4 p = *reinterpret_cast<const float4*>(ws + k * kstride);
      v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
    } else {
      v[0] += ws[k * kstride];
    }
  }
  const float bv = a.bias ? a.bias[co] : 0.0f;
#pragma unroll
  for (int e = 0; e < VEC; ++e) v[e] += bv;
  if (a.up) {
    // the VEC pixels of a thread sit in one output row (VEC == 4 only when Wout % 4 == 0)
    const int oy = pix / a.Wout;
    add_upsampled_x2<VEC>(a.up + ((long long)b * a.Cout + co) * a.up_h * a.up_w, a.up_h, a.up_w, oy, pix - oy * a.Wout, v);
  }
  const long long ooff = ((long long)b * a.out_ctot + a.out_coff + co) * HWout + pix;
  if (a.res) {
    const float* r = a.res + ((long long)b * a.res_ctot + a.res_coff + co) * HWout + pix;
    if (VEC == 4) {
      const float4 p = *reinterpret_cast<const float4*>(r);
      v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
    } else {
      v[0] += r[0];
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    if (a.act == ACT_RELU) v[e] = fmaxf(v[e], 0.0f);
    else if (a.act == ACT_RELU6) v[e] = fminf(fmaxf(v[e], 0.0f), 6.0f);
  }
  if (VEC == 4)
    *reinterpret_cast<float4*>(a.out + ooff) = make_float4(v[0], v[1], v[2], v[3]);
  else
    a.out[ooff] = v[0];
}

}  // namespace
// Experiment hook (tools/experiments/deletion.sh, FDT_SKIP_OPS=@reduce): the split-K reduce passes are not launched -- the
// ceiling of what an in-kernel combine could gain.  Set by model.hip around a pass; never in production.
thread_local bool exp_skip_reduce = false;
namespace {

struct ReduceGroup {
  ConvArgs a[kReduceGroupMax];
  int blk0[kReduceGroupMax + 1];   // first block of each layer in the grid
  int n;
};
static_assert(sizeof(ReduceGroup) <= 3584, "kernel arguments");

// One thread per 4 consecutive pixels of a layer with Wout % 4 == 0, else per pixel: splitk_reduce_kernel<4 / 1>, layer by
// table lookup.
__global__ void splitk_reduce_group_kernel(const ReduceGroup g) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < kReduceGroupMax; ++i)
    if (i < g.n && (int)blockIdx.x >= g.blk0[i]) l = i;
  const ConvArgs& a = g.a[l];
  const int HWout = a.Hout * a.Wout;
  const bool vec = (a.Wout & 3) == 0;
  const int HWv = vec ? HWout / 4 : HWout;
  const long long total = (long long)a.B * a.Cout * HWv;
  const long long t = (long long)((int)blockIdx.x - g.blk0[l]) * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int pix = (int)(t % HWv) * (vec ? 4 : 1);
  const int co = (int)((t / HWv) % a.Cout);
  const int b = (int)(t / ((long long)HWv * a.Cout));
  float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const float* ws = a.ws + ((long long)b * a.ksplit * a.Cout + co) * HWout + pix;
  const long long kstride = (long long)a.Cout * HWout;
  for (int k = 0; k < a.ksplit; ++k) {
    if (vec) {
      const float4 p = *reinterpret_cast<const float4*>(ws + k * kstride);
      v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
    } else {
      v[0] += ws[k * kstride];
    }
  }
  const float bv = a.bias ? a.bias[co] : 0.0f;
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] += bv;
  if (a.up) {
    const int oy = pix / a.Wout;
    const float* u = a.up + ((long long)b * a.Cout + co) * a.up_h * a.up_w;
    if (vec) add_upsampled_x2<4>(u, a.up_h, a.up_w, oy, pix - oy * a.Wout, v);
    else add_upsampled_x2<1>(u, a.up_h, a.up_w, oy, pix - oy * a.Wout, v);
  }
  const long long ooff = ((long long)b * a.out_ctot + a.out_coff + co) * HWout + pix;
  if (a.res) {
    const float* r = a.res + ((long long)b * a.res_ctot + a.res_coff + co) * HWout + pix;
    if (vec) {
      const float4 p = *reinterpret_cast<const float4*>(r);
      v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
    } else {
      v[0] += r[0];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (a.act == ACT_RELU) v[e] = fmaxf(v[e], 0.0f);
    else if (a.act == ACT_RELU6) v[e] = fminf(fmaxf(v[e], 0.0f), 6.0f);
  }
  if (vec)
    *reinterpret_cast<float4*>(a.out + ooff) = make_float4(v[0], v[1], v[2], v[3]);
  else
    a.out[ooff] = v[0];
}

struct Table {
  KernelEntry e[CONV_KIND_COUNT][CONV_TILE_COUNT];
  // per device: the attribute is per (function, device).  Handles on different host threads race here only to set
  // the same value twice (idempotent); the flag itself is atomic.
  std::atomic<unsigned char> attr_set[16][CONV_KIND_COUNT][CONV_TILE_COUNT];
  Table() {
    memset(e, 0, sizeof(e));
    for (auto& d : attr_set)
      for (auto& k : d)
        for (auto& t : k) t.store(0, std::memory_order_relaxed);
    conv_fill_1x1_s1(e[CONV_1x1_S1]);
    conv_fill_1x1_s2(e[CONV_1x1_S2]);
    conv_fill_3x3_s1(e[CONV_3x3_S1]);
    conv_fill_3x3_s1_d2(e[CONV_3x3_S1_D2]);
    conv_fill_3x3_s2(e[CONV_3x3_S2]);
    conv_fill_stems(e[CONV_7x7_S2], e[CONV_7x7_S4], e[CONV_5x5_S2], e[CONV_7x7_S2_P1]);
    conv_fill_wino(e[CONV_3x3_S1_WINO]);
    conv_fill_wino_d2(e[CONV_3x3_D2_WINO]);
    conv_fill_1x1_s1_deep(e[CONV_1x1_S1_K32], e[CONV_1x1_S1_K64]);
    conv_fill_n8(e[CONV_3x3_S1_N8]);
    conv_fill_wino44(e[CONV_3x3_S1_WINO44]);
    conv_fill_wino44_d2(e[CONV_3x3_D2_WINO44]);
    conv_fill_1x1_p(e[CONV_1x1_S1_P16], e[CONV_1x1_S1_P32]);
    conv_fill_stem_u8(e[CONV_7x7_S2_U8]);
    conv_fill_stem_s4(e[CONV_7x7_S4_K168], e[CONV_7x7_S4_U8]);
    conv_fill_1x1_b3(e[CONV_1x1_S1_B3]);
    conv_fill_stem_b3(e[CONV_7x7_S4_B3]);
    conv_fill_1x1_s2_b3(e[CONV_1x1_S2_B3]);
    conv_fill_stem_u8b(e[CONV_7x7_S2_U8B], e[CONV_7x7_S4_U8B]);
    conv_fill_1x1_pb3(e[CONV_1x1_S1_PB3]);
    conv_fill_3x3_s2_b3(e[CONV_3x3_S2_B3]);
  }
};

Table& table() {
  static Table t;
  return t;
}

const ConvGeom kGeoms[CONV_KIND_COUNT] = {
    {1, 1, 1, 1, 0, 16, 0}, {1, 1, 2, 1, 0, 16, 0}, {3, 3, 1, 1, 1, 4, 0}, {3, 3, 1, 2, 2, 4, 0},
    {3, 3, 2, 1, 1, 4, 0},  {7, 7, 2, 1, 3, 2, 0},  {7, 7, 4, 1, 3, 2, 0}, {5, 5, 2, 1, 2, 2, 0},
    {3, 3, 1, 1, 1, 8, 1},  {3, 3, 1, 2, 2, 8, 1},  {1, 1, 1, 1, 0, 32, 0},  {1, 1, 1, 1, 0, 64, 0},
    {7, 7, 2, 1, 1, 2, 0},  {3, 3, 1, 1, 1, 1, 0},  {3, 3, 1, 1, 1, 2, 2},  {3, 3, 1, 2, 2, 2, 2},
    {1, 1, 1, 1, 0, 16, 0}, {1, 1, 1, 1, 0, 32, 0}, {7, 7, 2, 1, 3, 4, 0},  {7, 7, 4, 1, 3, 3, 0},
    {7, 7, 4, 1, 3, 3, 0},  {1, 1, 1, 1, 0, 16, 0}, {7, 7, 4, 1, 3, 3, 0},  {1, 1, 2, 1, 0, 16, 0},
    {7, 7, 2, 1, 3, 3, 0},  {7, 7, 4, 1, 3, 3, 0},  {1, 1, 1, 1, 0, 16, 0},  {3, 3, 2, 1, 1, 16, 0},
};
const int kTileDims[CONV_TILE_COUNT][4] = {  // BM, BN, TH, TW   (order of enum ConvTile)
    {128, 128, 8, 16}, {128, 64, 8, 16}, {128, 32, 8, 16}, {64, 64, 8, 8},   {64, 128, 8, 8},
    {128, 128, 4, 32}, {128, 64, 4, 32}, {128, 128, 8, 16}, {128, 64, 8, 16}, {64, 64, 8, 8},
    {64, 128, 8, 8},   {128, 128, 4, 32}, {128, 64, 4, 32}, {128, 32, 8, 16},
    // Winograd: BM in pixels = 4 x blocks
    {256, 64, 16, 16}, {256, 64, 16, 16}, {512, 32, 16, 32}, {512, 32, 16, 32}, {128, 128, 8, 16},
    {128, 128, 8, 16}, {256, 64, 8, 32},
    // 8-wave Winograd
    {256, 64, 16, 16}, {256, 64, 16, 16}, {512, 32, 16, 32}, {256, 64, 8, 32},
    // ring of four
    {128, 128, 8, 16}, {128, 64, 8, 16}, {64, 64, 8, 8}, {64, 128, 8, 8},
    // quarter-split Winograd
    {256, 64, 16, 16}, {256, 64, 8, 32},
    // packed-f32 VALU heads
    {2048, 8, 32, 64},
    // Winograd F(4x4,3x3): eight waves, twelve waves
    {512, 64, 16, 32}, {512, 64, 16, 32},
    // persistent-tile 1x1
    {128, 64, 4, 32}, {128, 128, 4, 32},
    // 4x32 px, 32 ch
    {128, 32, 4, 32},
    // long-row tiles: 2x64, 1x128 px
    {128, 128, 2, 64}, {128, 64, 2, 64}, {128, 128, 1, 128}, {128, 64, 1, 128}};

}  // namespace

ConvGeom conv_geom(ConvKind k) { return kGeoms[k]; }
ConvKind conv_base_kind(ConvKind k) {
  switch (k) {
    case CONV_3x3_S1_WINO:
    case CONV_3x3_S1_WINO44:
    case CONV_3x3_S1_N8: return CONV_3x3_S1;
    case CONV_3x3_D2_WINO:
    case CONV_3x3_D2_WINO44: return CONV_3x3_S1_D2;
    case CONV_1x1_S1_K32:
    case CONV_1x1_S1_K64:
    case CONV_1x1_S1_P16:
    case CONV_1x1_S1_P32:
    case CONV_1x1_S1_B3:
    case CONV_1x1_S1_PB3: return CONV_1x1_S1;
    case CONV_1x1_S2_B3: return CONV_1x1_S2;
    case CONV_3x3_S2_B3: return CONV_3x3_S2;
    case CONV_7x7_S2_U8:
    case CONV_7x7_S2_U8B: return CONV_7x7_S2;
    case CONV_7x7_S4_U8:
    case CONV_7x7_S4_K168:
    case CONV_7x7_S4_B3:
    case CONV_7x7_S4_U8B: return CONV_7x7_S4;
    default: return k;
  }
}
bool tile_is_wino(ConvTile t) {
  return (t >= TILE_WINO_64x64 && t <= TILE_WINO8_64x64W) || t == TILE_WINO4_64x64R3 || t == TILE_WINO4_64x64W;
}
bool tile_is_wino44(ConvTile t) { return t == TILE_WINO44_32x64 || t == TILE_WINO44B_32x64; }
bool kind_is_u8b_stem(ConvKind k) { return k == CONV_7x7_S2_U8B || k == CONV_7x7_S4_U8B; }
bool kind_is_persistent(ConvKind k) { return k == CONV_1x1_S1_P16 || k == CONV_1x1_S1_P32 || k == CONV_1x1_S1_PB3; }
bool kind_is_u8_stem(ConvKind k) { return k == CONV_7x7_S2_U8 || k == CONV_7x7_S4_U8 || kind_is_u8b_stem(k); }
static int device_cus(int dev) {
  static std::atomic<int> cus[16];
  int v = cus[dev].load(std::memory_order_relaxed);
  if (v > 0) return v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  cus[dev].store(v, std::memory_order_relaxed);
  return v;
}
int tile_bm(ConvTile t) { return kTileDims[t][0]; }
int tile_bn(ConvTile t) { return kTileDims[t][1]; }
int tile_th(ConvTile t) { return kTileDims[t][2]; }
int tile_tw(ConvTile t) { return kTileDims[t][3]; }

bool conv_supported(ConvKind kind, ConvTile tile) { return table().e[kind][tile].fn != nullptr; }
size_t conv_lds_bytes(ConvKind kind, ConvTile tile) { return table().e[kind][tile].lds; }

// A finite f32 as three bf16 planes, x = p[0] + p[1] + p[2] exactly (each the round-to-nearest-even bf16 of the remainder): the
// host-side split of the weights of every bf16-pipe class (conv_b3.h: split3_bf16 is the device form)
static void split3_bf16_host(float x, unsigned short p[3]) {
  auto rne = [](float v) -> unsigned short {
    unsigned u;
    memcpy(&u, &v, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
  };
  auto widen = [](unsigned short h) -> float {
    const unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
  };
  p[0] = rne(x);
  const float r1 = x - widen(p[0]);
  p[1] = rne(r1);
  p[2] = rne(r1 - widen(p[1]));
}

void tile_weights(const float* w, const float* scale, int Cout, int Cin, ConvKind kind, ConvTile tile,
                  std::vector<float>& out) {
  const ConvGeom g = conv_geom(kind);
  const int ktaps = g.kh * g.kw, taps = g.wino == 2 ? 36 : g.wino ? 16 : ktaps, BN = tile_bn(tile), KC = g.kc;
  const int n_tiles = (Cout + BN - 1) / BN;
  const int nstages = (Cin + KC - 1) / KC;
  // Layout::WSZP (dwordx4 LDS-DMA granularity); the 8-channel VALU kernel stages exactly its 72 weights (N8::WSZ)
  // ... and the F(4x4,3x3) kernel exactly its 2 x 36 x 64 (W44::WSZ: 2 x 8 KB + 2 KB of LDS-DMA per k-step)
  if (kind == CONV_7x7_S4_K168 || kind == CONV_7x7_S4_U8) {
    // conv_stem_s4.h: [channel tile][c][ky][8 columns = taps -1 .. 6][32 couts], the first column zero (StemS4::WSZ = 5376 floats per tile)
    out.assign((size_t)n_tiles * 3 * 7 * 8 * 32, 0.0f);
    if (Cin != 3 || BN != 32) return;      // (conv_shape_supported keeps such a layer off these classes)
    for (int co = 0; co < Cout; ++co)
      for (int ci = 0; ci < 3; ++ci)
        for (int t = 0; t < 49; ++t)
          out[(((size_t)(co / 32) * 3 + ci) * 7 + t / 7) * 256 + (size_t)(t % 7 + 1) * 32 + co % 32] =
              w[((size_t)co * 3 + ci) * 49 + t] * (scale ? scale[co] : 1.0f);
    return;
  }
  if (kind_is_u8b_stem(kind)) {
    // conv_stem_u8b.h: per group of 32 couts [k-step s][plane][lane = 32 h + cout % 32][8 columns = taps -1 .. 6] bf16 -- a lane's
    // A operand of (step, plane) is the 16 bytes it loads into its registers; pair q = 2 s + h = (channel q / 7, tap row q % 7),
    // pair 21 and the first column are zero; a channel tile of BN couts is BN / 32 consecutive groups (StemU8B::WSZ = 8448 floats each)
    const int groups = n_tiles * (BN / 32);
    out.assign((size_t)groups * 8448, 0.0f);
    if (Cin != 3 || BN % 32) return;
    unsigned short* o16 = reinterpret_cast<unsigned short*>(out.data());
    for (int co = 0; co < Cout; ++co)
      for (int q = 0; q < 21; ++q)
        for (int kx = 0; kx < 7; ++kx) {
          const float wv = w[((size_t)co * 3 + q / 7) * 49 + (q % 7) * 7 + kx] * (scale ? scale[co] : 1.0f);
          unsigned short pl[3];
          split3_bf16_host(wv, pl);
          for (int pp = 0; pp < 3; ++pp)
            o16[(size_t)(co / 32) * 16896 + ((((size_t)(q / 2) * 3 + pp) * 2 + q % 2) * 32 + co % 32) * 8 + kx + 1] = pl[pp];
        }
    return;
  }
  if (kind == CONV_7x7_S4_B3) {
    // conv_stem_b3.h: per channel tile [plane][k-step s][k-half h][32 couts][8 columns = taps -1 .. 6] bf16, pair q = 2 s + h =
    // (channel q / 7, tap row q % 7); pair 21 and the first column are zero (StemB3::WSZ = 8448 floats per tile)
    out.assign((size_t)n_tiles * 8448, 0.0f);
    if (Cin != 3 || BN != 32) return;
    unsigned short* o16 = reinterpret_cast<unsigned short*>(out.data());
    for (int co = 0; co < Cout; ++co)
      for (int q = 0; q < 21; ++q)
        for (int kx = 0; kx < 7; ++kx) {
          const float wv = w[((size_t)co * 3 + q / 7) * 49 + (q % 7) * 7 + kx] * (scale ? scale[co] : 1.0f);
          unsigned short pl[3];
          split3_bf16_host(wv, pl);
          for (int pp = 0; pp < 3; ++pp)
            o16[(size_t)(co / 32) * 16896 + ((((size_t)pp * 11 + q / 2) * 2 + q % 2) * 32 + co % 32) * 8 + kx + 1] = pl[pp];
        }
    return;
  }
  if (kind == CONV_3x3_S2_B3) {
    // conv_b3.h, KS = 3: per (channel tile, 16-channel group, tap) [plane][k-half][BN couts][8 k] bf16 -- the 1x1 class's stage with the
    // group's weights of ONE tap; couts past Cout and channels past Cin stay zero
    const size_t wszp_b3 = ((size_t)24 * BN + 1023) / 1024 * 1024;
    out.assign((size_t)n_tiles * nstages * 9 * wszp_b3, 0.0f);
    unsigned short* o16 = reinterpret_cast<unsigned short*>(out.data());
    for (int co = 0; co < Cout; ++co) {
      const float sc = scale ? scale[co] : 1.0f;
      for (int ci = 0; ci < Cin; ++ci)
        for (int t = 0; t < 9; ++t) {
          const float wv = w[((size_t)co * Cin + ci) * 9 + t] * sc;
          unsigned short pl[3];
          split3_bf16_host(wv, pl);
          const size_t st = ((size_t)(co / BN) * nstages + ci / 16) * 9 + t;
          const int kk = ci % 16, h = kk / 8, q = kk % 8;
          for (int pp = 0; pp < 3; ++pp)
            o16[st * wszp_b3 * 2 + (((size_t)pp * 2 + h) * BN + co % BN) * 8 + q] = pl[pp];
        }
    }
    return;
  }
  if (kind == CONV_1x1_S1_B3 || kind == CONV_1x1_S2_B3 || kind == CONV_1x1_S1_PB3) {
    // conv_b3.h: per (channel tile, stage of 16 input channels) [plane][k-half][BN couts][8 k] bf16 -- the three bf16 planes of
    // every BN-folded f32 weight, w = p0 + p1 + p2 exactly (each the round-to-nearest-even bf16 of the remainder) -- padded to
    // whole dwordx4 LDS-DMA rounds (LayoutB3::WSZP floats); couts past Cout and channels past Cin stay zero
    const size_t wszp_b3 = ((size_t)24 * BN + 1023) / 1024 * 1024;
    out.assign((size_t)n_tiles * nstages * wszp_b3, 0.0f);
    unsigned short* o16 = reinterpret_cast<unsigned short*>(out.data());
    for (int co = 0; co < Cout; ++co) {
      const float sc = scale ? scale[co] : 1.0f;
      const int nt = co / BN, n = co % BN;
      for (int ci = 0; ci < Cin; ++ci) {
        const int s_ = ci / 16, k = ci % 16, h = k / 8, i = k % 8;
        const float wv = w[(size_t)co * Cin + ci] * sc;
        unsigned short pl[3];
        split3_bf16_host(wv, pl);
        for (int pp = 0; pp < 3; ++pp)
          o16[(((size_t)nt * nstages + s_) * wszp_b3) * 2 + ((((size_t)pp * 2 + h) * BN + n) * 8 + i)] = pl[pp];
      }
    }
    return;
  }
  const size_t wszp = (BN == 8 || g.wino == 2) ? (size_t)KC * taps * BN : ((size_t)KC * taps * BN + 1023) / 1024 * 1024;
  out.assign((size_t)n_tiles * nstages * wszp, 0.0f);
  for (int co = 0; co < Cout; ++co) {
    const float sc = scale ? scale[co] : 1.0f;
    const int nt = co / BN, n = co % BN;
    for (int ci = 0; ci < Cin; ++ci) {
      const int s = ci / KC, c = ci % KC;
      const float* src = w + ((size_t)co * Cin + ci) * ktaps;
      float* dst = out.data() + ((size_t)nt * nstages + s) * wszp + ((size_t)c * taps) * BN + n;
      if (!g.wino) {
        for (int t = 0; t < taps; ++t) dst[(size_t)t * BN] = src[t] * sc;
      } else if (g.wino == 2) {
        // F(4x4,3x3): U = G g G^T, G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1] (points 0, +-1,
        // +-2, inf), in f64 with one rounding to f32
        static const double G6[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                        {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
        double gk[3][3], tmp[6][3];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) gk[i][j] = (double)src[i * 3 + j] * (double)sc;
        for (int i = 0; i < 6; ++i)
          for (int j = 0; j < 3; ++j) tmp[i][j] = G6[i][0] * gk[0][j] + G6[i][1] * gk[1][j] + G6[i][2] * gk[2][j];
        for (int i = 0; i < 6; ++i)
          for (int j = 0; j < 6; ++j)
            dst[(size_t)(i * 6 + j) * BN] = (float)(tmp[i][0] * G6[j][0] + tmp[i][1] * G6[j][1] + tmp[i][2] * G6[j][2]);
      } else {
        // U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], in f64 with one rounding to f32
        double gk[3][3], tmp[4][3], U[4][4];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) gk[i][j] = (double)src[i * 3 + j] * (double)sc;
        for (int j = 0; j < 3; ++j) {
          tmp[0][j] = gk[0][j];
          tmp[1][j] = 0.5 * (gk[0][j] + gk[1][j] + gk[2][j]);
          tmp[2][j] = 0.5 * (gk[0][j] - gk[1][j] + gk[2][j]);
          tmp[3][j] = gk[2][j];
        }
        for (int i = 0; i < 4; ++i) {
          U[i][0] = tmp[i][0];
          U[i][1] = 0.5 * (tmp[i][0] + tmp[i][1] + tmp[i][2]);
          U[i][2] = 0.5 * (tmp[i][0] - tmp[i][1] + tmp[i][2]);
          U[i][3] = tmp[i][2];
        }
        for (int t = 0; t < 16; ++t) dst[(size_t)t * BN] = (float)U[t >> 2][t & 3];
      }
    }
  }
}

// Split-K layers finish in splitk_reduce_kernel.  A layer with the fused upsample-add and ksplit == 1 adds the
// upsampled map in the conv epilogue itself (no workspace round trip).
long long conv_ws_floats(const ConvArgs& a) {
  return a.ksplit > 1 ? (long long)a.B * a.ksplit * a.Cout * a.Hout * a.Wout : 0;
}

int launch_reduce_group(const ConvArgs* const* layers, int n, hipStream_t st) {
  FDT_REQUIRE(layers && n >= 1, FDT_ERR_ARG, "launch_reduce_group: nothing to reduce");
  if (exp_skip_reduce) return FDT_OK;
  for (int i0 = 0; i0 < n; i0 += kReduceGroupMax) {
    ReduceGroup g;
    memset(&g, 0, sizeof(g));
    g.n = std::min(kReduceGroupMax, n - i0);
    long long blk = 0;
    for (int i = 0; i < g.n; ++i) {
      const ConvArgs& a = *layers[i0 + i];
      FDT_REQUIRE(a.ksplit > 1 && a.ws && a.out && a.defer_reduce == 2, FDT_ERR_ARG, "launch_reduce_group: layer %d has no deferred slabs", i0 + i);
      g.a[i] = a;
      g.blk0[i] = (int)blk;
      const long long hwv = (a.Wout & 3) == 0 ? (long long)a.Hout * a.Wout / 4 : (long long)a.Hout * a.Wout;
      blk += ceil_div_ll((long long)a.B * a.Cout * hwv, 256);
      FDT_REQUIRE(blk <= 0x7fffffffll, FDT_ERR_ARG, "launch_reduce_group: grid too large");
    }
    g.blk0[g.n] = (int)blk;
    hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3((unsigned)blk), dim3(256), 0, st, g);
    FDT_LAUNCH_CHECK();
  }
  return FDT_OK;
}

long long conv_sk_counters(ConvKind kind, ConvTile tile, const ConvArgs& a) {
  (void)kind;
  return (long long)a.B * ceil_div(a.Hout, tile_th(tile)) * ceil_div(a.Wout, tile_tw(tile)) * ceil_div(a.Cout, tile_bn(tile));
}

bool conv_combine_supported(ConvKind kind, ConvTile tile, const ConvArgs& a) {
  // (the 8-channel VALU kernel is not in: its four waves per SIMD leave no registers for the write-through store path)
  return conv_supported(kind, tile) && kind != CONV_3x3_S1_N8 && a.ksplit > 1 && a.ws && a.Wout % 4 == 0 &&
         (long long)a.ksplit * a.Cout * a.Hout * a.Wout * 4 < (1ll << 31);
}

// Shape limits of a kernel class beyond the (kind, tile) table: the autotuner and fdt_conv2d ask before they launch.
// the direct implicit-GEMM classes of conv_kernel.h (one epilogue: the only one that knows ConvArgs.out2)
static bool kind_is_direct(ConvKind k) {
  return k == CONV_1x1_S1 || k == CONV_1x1_S2 || k == CONV_3x3_S1 || k == CONV_3x3_S1_D2 || k == CONV_3x3_S2 || k == CONV_7x7_S2 ||
         k == CONV_7x7_S4 || k == CONV_5x5_S2 || k == CONV_1x1_S1_K32 || k == CONV_1x1_S1_K64 || k == CONV_7x7_S2_P1 ||
         k == CONV_1x1_S1_B3 || k == CONV_1x1_S2_B3 || k == CONV_3x3_S2_B3;       // conv_b3.h carries the same epilogue
}

bool conv_shape_supported(ConvKind kind, ConvTile tile, const ConvArgs& a) {
  if (!conv_supported(kind, tile)) return false;
  if (a.out2 && (!kind_is_direct(kind) || a.ksplit > 1 || a.ws || a.res || a.up || a.sk_count)) return false;
  if (a.in_bstride && (kind_is_u8_stem(kind) || kind == CONV_7x7_S4_K168)) return false;
  if (kind == CONV_3x3_D2_WINO44 && (a.Win & 3)) return false;
  if (kind_is_u8b_stem(kind)) {   // one bf16 plane holds a pixel minus an INTEGER mean exactly; 12-byte groups of four pixels
    for (int c = 0; c < 3; ++c)
      if (a.u8_mean[c] != (float)(int)a.u8_mean[c] || a.u8_mean[c] < 0.0f || a.u8_mean[c] > 255.0f) return false;
    if ((a.Win & 3) || !(a.u8_scale > 0.0f) || (long long)a.Hin * a.Win * 3 >= (1ll << 31)) return false;
  }
  if (kind_is_u8_stem(kind))   // the raw-frame stem: three input channels, one stage, plain epilogue, one channel tile per 32 / 64 couts
    return a.in_u8 != nullptr && a.Cin == 3 && a.ksplit <= 1 && !a.ws && !a.res && !a.up && !a.sk_count &&
           (long long)(a.Cout + 64) * a.Hout * a.Wout * 4 < (1ll << 31);
  if (a.in_u8) return false;   // every other class reads f32 NCHW
  if (kind == CONV_7x7_S4_B3)      // f32 NCHW frames, 16-byte pieces only
    return a.Cin == 3 && (a.Win & 3) == 0 && a.ksplit <= 1 && !a.ws && !a.res && !a.up && !a.sk_count && !a.out2 && !a.in_bstride &&
           (long long)3 * a.Hin * a.Win * 4 < (1ll << 31) && (long long)(a.Cout + 64) * a.Hout * a.Wout * 4 < (1ll << 31);
  if (kind == CONV_7x7_S4_K168)
    return a.Cin == 3 && a.ksplit <= 1 && !a.ws && !a.res && !a.up && !a.sk_count && (long long)3 * a.Hin * a.Win * 4 < (1ll << 31) &&
           (long long)(a.Cout + 64) * a.Hout * a.Wout * 4 < (1ll << 31);
  if (kind == CONV_1x1_S1_B3)      // 16-byte activation staging and the vector epilogue only
    return (a.Win & 3) == 0 && (long long)(a.Cout + 128) * a.Hout * a.Wout * 4 < (1ll << 40);
  if (kind == CONV_1x1_S2_B3 || kind == CONV_3x3_S2_B3)      // the vector epilogue only
    return (a.Wout & 3) == 0 && (long long)(a.Cout + 128) * a.Hout * a.Wout * 4 < (1ll << 40);
  if (kind_is_persistent(kind)) {
    const int nstages = ceil_div(a.Cin, conv_geom(kind).kc);
    const long long hw = (long long)a.Hin * a.Win;
    // (the split-bf16 form runs its ring one stage further ahead -- operands of stage g + 1 are read under the MFMAs of stage g --: three stages)
    return (a.Win & 3) == 0 && a.ksplit <= 1 && !a.up && !a.sk_count && !a.ws && nstages >= (kind == CONV_1x1_S1_PB3 ? 3 : 2) &&
           ((long long)a.Cout + 128) * hw * 4 < (1ll << 31) && (long long)a.Cin * hw * 4 < (1ll << 31);
  }
  return true;
}

double conv_flops(const ConvArgs& a, ConvKind kind) {
  const ConvGeom g = conv_geom(kind);
  return 2.0 * a.B * (double)a.Hout * a.Wout * a.Cout * a.Cin * g.kh * g.kw;
}

int launch_conv(ConvKind kind, ConvTile tile, const ConvArgs& a_in, hipStream_t st, int dev) {
  ConvArgs a = a_in;
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
                  a.out_coff + a.Cout <= a.out_ctot && (a.in || a.in_u8) && a.w && a.out,
              FDT_ERR_ARG, "launch_conv: bad channel/pointer arguments");
  FDT_REQUIRE(!a.out2 || (a.out2_from > 0 && a.out2_from < a.Cout && a.out2_coff >= 0 &&
                          a.out2_coff + (a.Cout - a.out2_from) <= a.out2_ctot && a.out_coff + a.out2_from <= a.out_ctot),
              FDT_ERR_ARG, "launch_conv: bad second destination (out2_from %d of Cout %d)", a.out2_from, a.Cout);
  FDT_REQUIRE(a.in_bstride == 0 || a.in_bstride >= (long long)a.Cin * a.Hin * a.Win, FDT_ERR_ARG,
              "launch_conv: in_bstride %lld smaller than one image of the slice", a.in_bstride);
  FDT_REQUIRE(!(a.out2 || a.in_bstride) || conv_shape_supported(kind, tile, a), FDT_ERR_ARG,
              "launch_conv: kernel class %d does not take a second destination / a channel-slice input for this layer", (int)kind);
  FDT_REQUIRE(kind_is_persistent(kind) || conv_shape_supported(kind, tile, a), FDT_ERR_ARG,
              "launch_conv: kernel class %d is not instantiated for this layer (raw-frame stem needs ConvArgs.in_u8 and Cin = 3; every "
              "other class reads f32)", (int)kind);
  if (a.res) FDT_REQUIRE(a.res_coff + a.Cout <= a.res_ctot, FDT_ERR_ARG, "launch_conv: bad residual slice");
  if (a.up) FDT_REQUIRE(a.up_h * 2 >= a.Hout && a.up_w * 2 >= a.Wout && a.up_h >= 1 && a.up_w >= 1,
                        FDT_ERR_ARG, "launch_conv: upsample source too small");
  // the kernels address one image through a buffer descriptor (32-bit byte offsets, bounds-checked)
  FDT_REQUIRE((long long)a.Cin * a.Hin * a.Win * 4 < (1ll << 31), FDT_ERR_ARG,
              "launch_conv: input image of %d x %d x %d exceeds the 2 GB a buffer descriptor addresses", a.Cin, a.Hin, a.Win);
  FDT_REQUIRE(!(kind == CONV_3x3_D2_WINO44 && (a.Win & 3)), FDT_ERR_ARG,
              "launch_conv: the dilated Winograd F(4x4,3x3) kernel is not instantiated for Win %% 4 != 0");
  if (dev < 0) FDT_HIP(hipGetDevice(&dev));
  FDT_REQUIRE(dev < 16, FDT_ERR_ARG, "launch_conv: device index %d out of range", dev);
  if (!table().attr_set[dev][kind][tile].load(std::memory_order_acquire)) {
    FDT_HIP(hipFuncSetAttribute((const void*)ke.fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)ke.lds));
    if (ke.fn_odd)
      FDT_HIP(hipFuncSetAttribute((const void*)ke.fn_odd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ke.lds));
    table().attr_set[dev][kind][tile].store(1, std::memory_order_release);
  }
  const int nstages = ceil_div(a.Cin, g.kc);
  FDT_REQUIRE(a.ksplit >= 1 && a.ksplit <= nstages, FDT_ERR_ARG,
              "launch_conv: bad split-K %d (stages %d)", a.ksplit, nstages);
  // split-K layers finish in splitk_reduce_kernel (which then also carries the fused upsample-add)
  FDT_REQUIRE(a.ksplit == 1 || a.ws, FDT_ERR_ARG, "launch_conv: workspace required");
  FDT_REQUIRE(!a.ws || a.ksplit > 1, FDT_ERR_ARG, "launch_conv: unexpected workspace");
  // in-kernel combine (conv.h, splitk_combine_tile): 16-byte slab rows, one image's slabs behind one buffer descriptor
  FDT_REQUIRE(!(a.sk_count && a.defer_reduce), FDT_ERR_ARG, "launch_conv: in-kernel combine and deferred reduce exclude each other");
  FDT_REQUIRE(!a.defer_reduce || a.ksplit > 1, FDT_ERR_ARG, "launch_conv: nothing to defer without split-K");
  FDT_REQUIRE(a.defer_reduce != 1 || (!a.res && !a.up && a.act == ACT_NONE), FDT_ERR_ARG,
              "launch_conv: a consumer that sums the slabs itself adds the bias only (the head convs)");
  FDT_REQUIRE(!a.sk_count || conv_combine_supported(kind, tile, a), FDT_ERR_ARG,
              "launch_conv: in-kernel split-K combine needs ksplit > 1, Wout %% 4 == 0 and < 2 GB of slabs per image");
  FDT_REQUIRE(!a.up || conv_base_kind(kind) == CONV_1x1_S1, FDT_ERR_ARG,
              "launch_conv: the fused upsample-add exists for the 1x1 class only (ContextTexture.main_conv)");
  const int tiles = ceil_div(a.Hout, tile_th(tile)) * ceil_div(a.Wout, tile_tw(tile));
  const int n_ct = ceil_div(a.Cout, tile_bn(tile));
  a.n_sp = tiles;
  a.n_ct = n_ct;
  if (kind_is_persistent(kind)) {
    // conv_1x1p.h: a workgroup walks tiles_per_wg consecutive tiles (channel tile fastest).  The grid is sized so that every
    // CU holds its `resident` workgroups at once and all of them get the same number of tiles (+-1 at the end).
    FDT_REQUIRE(conv_shape_supported(kind, tile, a), FDT_ERR_ARG,
                "launch_conv: the persistent 1x1 kernel is not instantiated for this layer (needs Win %% 4 == 0, Cin >= 2 ring "
                "stages, no split-K, no fused upsample-add, < 2 GB per image)");
    const long long total = (long long)a.B * tiles * n_ct;
    FDT_REQUIRE(total <= 0x7fffffffll, FDT_ERR_ARG, "launch_conv: grid too large");
#ifdef FDT_EXPERIMENTS   // tools/experiments/r4_job6.sh: workgroups per CU the grid is sized for
    static const int env_r = getenv("FDT_P1_RESIDENT") ? atoi(getenv("FDT_P1_RESIDENT")) : 0;
#else
    const int env_r = 0;
#endif
    const int resident = env_r > 0 ? env_r : kind == CONV_1x1_S1_PB3 ? conv_1x1pb3_resident(tile) : conv_1x1p_resident(tile);
    const long long slots = (long long)device_cus(dev) * resident;
    const int k = (int)ceil_div_ll(total, slots);
    a.tiles_per_wg = k;
    hipLaunchKernelGGL(ke.fn, dim3((unsigned)ceil_div_ll(total, k)), dim3(ke.threads), ke.lds, st, a);
    FDT_LAUNCH_CHECK();
    return FDT_OK;
  }
  if (kind == CONV_7x7_S4_K168 || kind == CONV_7x7_S4_U8 || kind == CONV_7x7_S4_B3 || kind_is_u8b_stem(kind)) {
    // conv_stem_s4.h / conv_stem_b3.h: persistent over the (image, spatial tile) pairs, three (split-bf16: two) workgroups per
    // CU; grid.y = channel tile
    const long long total = (long long)a.B * tiles;
    FDT_REQUIRE(total <= 0x7fffffffll && n_ct <= 65535, FDT_ERR_ARG, "launch_conv: grid too large");
    const int k = (int)ceil_div_ll(total, (long long)device_cus(dev) * (kind == CONV_7x7_S4_B3 || kind_is_u8b_stem(kind) ? 2 : 3));
    a.tiles_per_wg = k;
    hipLaunchKernelGGL(ke.fn, dim3((unsigned)ceil_div_ll(total, k), (unsigned)n_ct), dim3(ke.threads), ke.lds, st, a);
    FDT_LAUNCH_CHECK();
    return FDT_OK;
  }
  FDT_REQUIRE(a.map_mode >= CONV_MAP_ROWS && a.map_mode < CONV_MAP_COUNT, FDT_ERR_ARG, "launch_conv: bad map mode");
  const long long gx = (a.map_mode == CONV_MAP_XCD_SPATIAL || a.map_mode == CONV_MAP_XCD_REGION)
                           ? (long long)ceil_div(tiles, 8) * 8 * n_ct
                           : a.map_mode == CONV_MAP_XCD_CHANNEL ? (long long)ceil_div(n_ct, 8) * 8 * tiles : (long long)tiles * n_ct;
  FDT_REQUIRE(gx <= 0x7fffffffll && (long long)a.B * a.ksplit <= 65535, FDT_ERR_ARG, "launch_conv: grid too large");
  dim3 grid((unsigned)gx, 1, a.B * a.ksplit);
  const bool odd = ke.fn_odd && (a.Win & 3);
  hipLaunchKernelGGL(odd ? ke.fn_odd : ke.fn, grid, dim3(odd && ke.threads_odd ? ke.threads_odd : ke.threads), ke.lds, st, a);
  FDT_LAUNCH_CHECK();
  if (a.ws && !a.sk_count && !a.defer_reduce && !exp_skip_reduce) {
    const long long total = (long long)a.B * a.Cout * a.Hout * a.Wout;
    if (a.Wout % 4 == 0)
      hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3((unsigned)ceil_div_ll(total / 4, 256)), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3((unsigned)ceil_div_ll(total, 256)), dim3(256), 0, st, a);
    FDT_LAUNCH_CHECK();
  }
  return FDT_OK;
}

}  // namespace fdt

// ---------------------------------------------------------------------------------------------------
// Stand-alone convolution op (host pointers): F.conv2d + bias + optional fused residual / bilinear x2
// upsample-add / activation, i.e. one launch of the kernel the detector graphs are made of.  `tile` and
// `ksplit` select a specific kernel variant (tile < 0 / ksplit <= 0: pick automatically); the parity
// tests sweep every instantiated variant through this entry point.
extern "C" int fdt_conv2d(const float* x, int B, int Cin, int H, int W, const float* w_oihw,
                          const float* bias, int Cout, int ksize, int stride, int pad, int dil,
                          const float* residual, const float* up, int up_h, int up_w, int act, int tile,
                          int ksplit, float* out) {
  const hipStream_t st = fdt::thread_stream();   // never the legacy stream (common.h)
  using namespace fdt;
  FDT_REQUIRE(x && w_oihw && out && B >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1, FDT_ERR_ARG,
              "fdt_conv2d: bad argument");
  int kind = -1;
  for (int k = 0; k < CONV_KIND_COUNT; ++k) {
    const ConvGeom g = conv_geom((ConvKind)k);
    if (conv_base_kind((ConvKind)k) == (ConvKind)k && g.kh == ksize && g.kw == ksize && g.stride == stride &&
        g.pad == pad && g.dil == dil)
      kind = k;
  }
  FDT_REQUIRE(kind >= 0, FDT_ERR_ARG, "fdt_conv2d: no kernel class for k=%d stride=%d pad=%d dil=%d", ksize,
              stride, pad, dil);
  const ConvGeom g = conv_geom((ConvKind)kind);
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.Cin = Cin; a.Hin = H; a.Win = W; a.Cout = Cout;
  a.Hout = (H + 2 * g.pad - g.dil * (g.kh - 1) - 1) / g.stride + 1;
  a.Wout = (W + 2 * g.pad - g.dil * (g.kw - 1) - 1) / g.stride + 1;
  FDT_REQUIRE(a.Hout >= 1 && a.Wout >= 1, FDT_ERR_ARG, "fdt_conv2d: empty output");
  a.out_ctot = Cout; a.res_ctot = Cout; a.act = act;
  if (tile < 0) {
    tile = -1;
    for (int t = 0; t < CONV_TILE_COUNT && tile < 0; ++t)
      if (conv_supported((ConvKind)kind, (ConvTile)t) && (Cout > 64 || tile_bn((ConvTile)t) <= 64)) tile = t;
    for (int t = 0; t < CONV_TILE_COUNT && tile < 0; ++t)
      if (conv_supported((ConvKind)kind, (ConvTile)t)) tile = t;
  }
  if (tile >= 100) {   // explicit implementation variant: tile = variant_kind * 100 + tile
    const int alt = tile / 100;
    tile %= 100;
    FDT_REQUIRE(alt < CONV_KIND_COUNT && conv_base_kind((ConvKind)alt) == (ConvKind)kind, FDT_ERR_ARG,
                "fdt_conv2d: kernel class %d does not implement this convolution", alt);
    kind = alt;
  }
  if (kind == CONV_3x3_S1 && tile_is_wino((ConvTile)tile)) kind = CONV_3x3_S1_WINO;
  if (kind == CONV_3x3_S1 && tile_is_wino44((ConvTile)tile)) kind = CONV_3x3_S1_WINO44;
  if (kind == CONV_3x3_S1_D2 && tile == TILE_WINO44_32x64) kind = CONV_3x3_D2_WINO44;
  if (kind == CONV_3x3_S1_D2 && tile_is_wino((ConvTile)tile)) kind = CONV_3x3_D2_WINO;
  if (kind == CONV_3x3_S1 && tile == TILE_N8_32x64) kind = CONV_3x3_S1_N8;
  if (kind == CONV_1x1_S1 && (tile == TILE_P_128x64 || tile == TILE_P_128x128)) kind = CONV_1x1_S1_P16;
  if (kind == CONV_7x7_S4 && tile == TILE_128x32W) kind = CONV_7x7_S4_K168;   // conv_stem_s4.h (Cin = 3 only: checked at launch)
  FDT_REQUIRE(tile >= 0 && tile < CONV_TILE_COUNT && conv_supported((ConvKind)kind, (ConvTile)tile), FDT_ERR_ARG,
              "fdt_conv2d: kernel (kind %d, tile %d) not instantiated", kind, tile);
  const bool combine = ksplit > 0 && (ksplit & FDT_SPLIT_COMBINE);   // in-kernel combine instead of the reduce pass
  if (combine) ksplit &= FDT_SPLIT_COMBINE - 1;
  a.ksplit = ksplit > 0 ? ksplit : 1;
  std::vector<float> tiled;
  tile_weights(w_oihw, nullptr, Cout, Cin, (ConvKind)kind, (ConvTile)tile, tiled);
  const size_t n_in = (size_t)B * Cin * H * W, n_out = (size_t)B * Cout * a.Hout * a.Wout;
  FDT_REQUIRE(st, FDT_ERR_HIP, "%s: could not create the calling thread's private stream", __func__);
  DevBuf din, dw, db, dout, dres, dup, dws;
  FDT_TRY(din.alloc(n_in * 4)); FDT_TRY(dw.alloc(tiled.size() * 4)); FDT_TRY(dout.alloc(n_out * 4));
  FDT_HIP(copy_sync(din.p, x, n_in * 4, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(dw.p, tiled.data(), tiled.size() * 4, hipMemcpyHostToDevice, st));
  a.in = din.as<float>(); a.w = dw.as<float>(); a.out = dout.as<float>();
  // this entry point is what the parity tests drive: an output element the kernel fails to write must not inherit a plausible
  // value from whatever the freshly allocated buffer held before (a previous call's result, most of the time) -- all-ones bits = NaN
  FDT_HIP(hipMemsetAsync(dout.p, 0xFF, n_out * 4, st));
  if (bias) {
    FDT_TRY(db.alloc((size_t)Cout * 4));
    FDT_HIP(copy_sync(db.p, bias, (size_t)Cout * 4, hipMemcpyHostToDevice, st));
    a.bias = db.as<float>();
  }
  if (residual) {
    FDT_TRY(dres.alloc(n_out * 4));
    FDT_HIP(copy_sync(dres.p, residual, n_out * 4, hipMemcpyHostToDevice, st));
    a.res = dres.as<float>();
  }
  if (up) {
    FDT_REQUIRE(up_h >= 1 && up_w >= 1, FDT_ERR_ARG, "fdt_conv2d: bad upsample source size");
    const size_t n_up = (size_t)B * Cout * up_h * up_w;
    FDT_TRY(dup.alloc(n_up * 4));
    FDT_HIP(copy_sync(dup.p, up, n_up * 4, hipMemcpyHostToDevice, st));
    a.up = dup.as<float>(); a.up_h = up_h; a.up_w = up_w;
  }
  if (a.ksplit > 1) { FDT_TRY(dws.alloc((size_t)conv_ws_floats(a) * 4)); a.ws = dws.as<float>(); }
  DevBuf dcnt;
  if (combine) {
    FDT_REQUIRE(conv_combine_supported((ConvKind)kind, (ConvTile)tile, a), FDT_ERR_ARG,
                "fdt_conv2d: in-kernel split-K combine needs ksplit > 1 and Wout %% 4 == 0");
    const size_t nc = (size_t)conv_sk_counters((ConvKind)kind, (ConvTile)tile, a) * sizeof(unsigned);
    FDT_TRY(dcnt.alloc(nc));
    FDT_HIP(hipMemsetAsync(dcnt.p, 0, nc, st));
    a.sk_count = dcnt.as<unsigned>();
  }
  if (const char* mm = getenv("FDT_CONV_MAP")) a.map_mode = atoi(mm);   // test hook: workgroup map (conv.h CONV_MAP_*)
  FDT_TRY(launch_conv((ConvKind)kind, (ConvTile)tile, a, st));
  FDT_HIP(copy_sync(out, dout.p, n_out * 4, hipMemcpyDeviceToHost, st));
  return FDT_OK;
}

// Host-only query (not part of include/fdt.h; no GPU needed): is (kernel class, tile) instantiated, and which base class (the layer
// geometry it serves) does the class belong to?  tests/test_gpu_timed_plans.py checks every row of every committed plan with it.
extern "C" int fdt_debug_conv_class(int kind, int tile, int* base_kind) {
  if (kind < 0 || kind >= fdt::CONV_KIND_COUNT || tile < 0 || tile >= fdt::CONV_TILE_COUNT) return 0;
  if (base_kind) *base_kind = (int)fdt::conv_base_kind((fdt::ConvKind)kind);
  return fdt::conv_supported((fdt::ConvKind)kind, (fdt::ConvTile)tile) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------
// Tuning hook (not part of include/fdt.h): time one conv configuration on random data with HIP events.
// Used by tools/conv_bench.py and tools/autotune.py.
extern "C" int fdt_debug_conv_bench(int kind, int tile, int ksplit, int B, int Cin, int Hin, int Win,
                                    int Cout, int has_res, int has_up, int act, int iters, float* ms_out) {
  const hipStream_t st = fdt::thread_stream();   // never the legacy stream (common.h)
  using namespace fdt;
  FDT_REQUIRE(kind >= 0 && kind < CONV_KIND_COUNT && tile >= 0 && tile < CONV_TILE_COUNT && ms_out && iters >= 1,
              FDT_ERR_ARG, "fdt_debug_conv_bench: bad argument");
  const bool combine = ksplit > 0 && (ksplit & FDT_SPLIT_COMBINE);   // in-kernel combine instead of the reduce pass
  if (combine) ksplit &= FDT_SPLIT_COMBINE - 1;
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
  FDT_REQUIRE(st, FDT_ERR_HIP, "%s: could not create the calling thread's private stream", __func__);
  DevBuf din, dw, db, dout, dres, dup, dws;
  const size_t n_in = (size_t)B * Cin * Hin * Win, n_out = (size_t)B * Cout * a.Hout * a.Wout;
  FDT_TRY(din.alloc(n_in * 4)); FDT_TRY(dw.alloc(tiled.size() * 4)); FDT_TRY(db.alloc((size_t)Cout * 4));
  FDT_TRY(dout.alloc(n_out * 4));
  std::vector<float> hin(n_in);
  for (auto& v : hin) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
  FDT_HIP(copy_sync(din.p, hin.data(), n_in * 4, hipMemcpyHostToDevice, st));
  FDT_HIP(copy_sync(dw.p, tiled.data(), tiled.size() * 4, hipMemcpyHostToDevice, st));
  FDT_HIP(hipMemsetAsync(db.p, 0, (size_t)Cout * 4, st));
  a.in = din.as<float>(); a.w = dw.as<float>(); a.bias = db.as<float>(); a.out = dout.as<float>();
  if (has_res) { FDT_TRY(dres.alloc(n_out * 4)); FDT_HIP(hipMemsetAsync(dres.p, 0, n_out * 4, st)); a.res = dres.as<float>(); }
  if (has_up) {
    a.up_h = (a.Hout + 1) / 2; a.up_w = (a.Wout + 1) / 2;
    FDT_TRY(dup.alloc((size_t)B * Cout * a.up_h * a.up_w * 4));
    FDT_HIP(hipMemsetAsync(dup.p, 0, (size_t)B * Cout * a.up_h * a.up_w * 4, st));
    a.up = dup.as<float>();
  }
  if (ksplit > 1) { FDT_TRY(dws.alloc((size_t)conv_ws_floats(a) * 4)); a.ws = dws.as<float>(); }
  DevBuf dcnt;
  if (combine) {
    FDT_REQUIRE(conv_combine_supported((ConvKind)kind, (ConvTile)tile, a), FDT_ERR_ARG,
                "fdt_debug_conv_bench: in-kernel split-K combine needs ksplit > 1 and Wout %% 4 == 0");
    const size_t nc = (size_t)conv_sk_counters((ConvKind)kind, (ConvTile)tile, a) * sizeof(unsigned);
    FDT_TRY(dcnt.alloc(nc));
    FDT_HIP(hipMemsetAsync(dcnt.p, 0, nc, st));
    a.sk_count = dcnt.as<unsigned>();
  }
  if (const char* mm = getenv("FDT_CONV_MAP")) a.map_mode = atoi(mm);   // tuning hook: workgroup map (conv.h CONV_MAP_*)
  hipEvent_t e0, e1;
  FDT_HIP(hipEventCreate(&e0)); FDT_HIP(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) FDT_TRY(launch_conv((ConvKind)kind, (ConvTile)tile, a, st));
  FDT_HIP(hipEventRecord(e0, st));
  for (int i = 0; i < iters; ++i) FDT_TRY(launch_conv((ConvKind)kind, (ConvTile)tile, a, st));
  FDT_HIP(hipEventRecord(e1, st));
  FDT_HIP(hipEventSynchronize(e1));
  float ms = 0;
  FDT_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return FDT_OK;
}
