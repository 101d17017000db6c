// Implicit-GEMM convolution on the f32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32): launch
// interface.  See conv.hip for the kernel design.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

namespace fdt {

// Convolution classes the three nets instantiate.
enum ConvKind {
  CONV_1x1_S1 = 0,
  CONV_1x1_S2,
  CONV_3x3_S1,      // pad 1
  CONV_3x3_S1_D2,   // pad 2, dilation 2
  CONV_3x3_S2,      // pad 1
  CONV_7x7_S2,      // pad 3 (Res50 stem)
  CONV_7x7_S4,      // pad 3 (FaceBox conv1)
  CONV_5x5_S2,      // pad 2 (FaceBox conv2)
  CONV_3x3_S1_WINO, // same arithmetic class as CONV_3x3_S1, computed with Winograd F(2x2,3x3)
  CONV_3x3_D2_WINO, // same arithmetic class as CONV_3x3_S1_D2 (dilation 2), Winograd on the parity sub-lattices
  CONV_1x1_S1_K32,  // CONV_1x1_S1 with 32 / 64 input channels per LDS stage (fewer, longer stages for the
  CONV_1x1_S1_K64,  // small-tile, deep-K layers)
  CONV_7x7_S2_P1,   // pad 1 (stem of pyramid_mb2_try4.py:16: conv_bn with a 7x7 kernel, padding left at 1)
  CONV_3x3_S1_N8,   // same arithmetic class as CONV_3x3_S1, 8 output channels per workgroup on the packed-f32 VALU (conv_n8.h)
  CONV_3x3_S1_WINO44,  // same arithmetic class as CONV_3x3_S1, Winograd F(4x4,3x3): 36 taps, two input channels per stage (conv_wino44.h)
  CONV_3x3_D2_WINO44,  // same arithmetic class as CONV_3x3_S1_D2 (dilation 2), Winograd F(4x4,3x3) on the parity sub-lattices
  CONV_1x1_S1_P16,     // CONV_1x1_S1 as a persistent-tile kernel (conv_1x1p.h): a workgroup walks several output tiles, the LDS
  CONV_1x1_S1_P32,     // ring runs across them, register epilogue; 16 / 32 input channels per ring stage
  CONV_7x7_S2_U8,      // CONV_7x7_S2 / _S4 reading the raw uint8 HWC BGR frame: the (float)u8 - mean (/ scale) of the ingest
  CONV_7x7_S4_U8,      // happens in the conv's own staging (conv_stem_u8.h / conv_stem_s4.h); ConvArgs.in_u8 instead of ConvArgs.in
  CONV_7x7_S4_K168,    // CONV_7x7_S4 for Cin = 3 (f32 NCHW input) as 3 x 7 x 8 k-columns instead of 4 x 49, three workgroups per CU (conv_stem_s4.h)
  CONV_1x1_S1_B3,      // CONV_1x1_S1 with split-bf16 products on v_mfma_f32_32x32x16_bf16 (three bf16 planes per operand, six plane
                       // products, f32 accumulate: conv_b3.h).  Same tolerance as the f32 classes, not the same bits.
  CONV_7x7_S4_B3,      // CONV_7x7_S4 for Cin = 3 with split-bf16 products (conv_stem_b3.h): FaceBoxes' conv1 on the bf16 matrix pipe
  CONV_1x1_S2_B3,      // CONV_1x1_S2 with split-bf16 products (conv_b3.h, S = 2): the bottleneck's downsample branch
  CONV_7x7_S2_U8B,     // the raw-uint8 stems on the bf16 matrix pipe (conv_stem_u8b.h): a pixel minus an integer mean is exact in ONE
  CONV_7x7_S4_U8B,     // bf16, the weights carry three planes: three exact plane products per k-step, f32 accumulate.  ConvArgs.in_u8.
  CONV_1x1_S1_PB3,     // CONV_1x1_S1_B3 as a persistent-tile kernel (conv_1x1p_b3.h): conv_1x1p.h's schedule, conv_b3.h's arithmetic and bits
  CONV_3x3_S2_B3,      // CONV_3x3_S2 (padding 1) with split-bf16 products (conv_b3.h, KS = 3): nine tap stages per 16-channel group
  CONV_KIND_COUNT
};

// Output-tile shapes.  BM = output pixels (TH x TW patch), BN = output channels per workgroup.
enum ConvTile {
  TILE_128x128 = 0,  // 8x16 px, 128 ch : the workhorse
  TILE_128x64,       // 8x16 px,  64 ch
  TILE_128x32,       // 8x16 px,  32 ch : narrow heads (Cout = 8) and thin MobileNet layers
  TILE_64x64,        // 8x8  px,  64 ch : small maps
  TILE_64x128,       // 8x8  px, 128 ch : small maps, wide layers
  TILE_128x128W,     // 4x32 px ("wide"): one 128-byte image row per half wave, conflict-free patch reads
  TILE_128x64W,
  TILE_128x128R3,    // R3: ring of three LDS stages (one more stage of LDS-DMA in flight)
  TILE_128x64R3,
  TILE_64x64R3,
  TILE_64x128R3,
  TILE_128x128WR3,
  TILE_128x64WR3,
  TILE_128x32R3,
  // Winograd tiles (only valid with CONV_3x3_S1_WINO): <2x2 blocks> x <couts>
  TILE_WINO_64x64,     // 16x16 px,  64 ch
  TILE_WINO_64x64R3,
  TILE_WINO_128x32,    // 16x32 px,  32 ch
  TILE_WINO_128x32R3,
  TILE_WINO_32x128,    //  8x16 px, 128 ch
  TILE_WINO_32x128R3,
  TILE_WINO_64x64W,    //  8x32 px,  64 ch
  // 8-wave Winograd (two waves per SIMD, 8 accumulators each)
  TILE_WINO8_64x64,
  TILE_WINO8_64x64R3,
  TILE_WINO8_128x32R3,
  TILE_WINO8_64x64W,
  // ring of FOUR LDS stages (two stages of LDS-DMA in flight behind the one being waited for): 1x1 classes,
  // whose short stages are bound by load latency rather than by the matrix cores
  TILE_128x128R4,
  TILE_128x64R4,
  TILE_64x64R4,
  TILE_64x128R4,
  // quarter-split 8-wave Winograd (conv_wino4_kernel): 16x16 px / 8x32 px, 64 ch
  TILE_WINO4_64x64R3,
  TILE_WINO4_64x64W,
  // packed-f32 VALU kernel for narrow heads (only valid with CONV_3x3_S1_N8): 32x64 px, 8 ch
  TILE_N8_32x64,
  // Winograd F(4x4,3x3) (only valid with CONV_3x3_S1_WINO44): 16x32 px (4 x 8 tiles of 4x4), 64 ch, eight waves
  TILE_WINO44_32x64,
  TILE_WINO44B_32x64,  // twelve waves: a wave owns half a row of the position grid and forms its own B operands (no V buffer)
  // persistent-tile 1x1 kernel (only valid with CONV_1x1_S1_P16 / _P32): 4x32 px, 64 / 128 ch
  TILE_P_128x64,
  TILE_P_128x128,
  TILE_128x32W,        // 4x32 px, 32 ch (the u8 stem of FaceBoxes: 24 output channels)
  // long-row tiles of the split-bf16 1x1 class (conv_b3.h, waves 4 x 1): 2x64 / 1x128 px -- 256 / 512 contiguous bytes per channel row
  TILE_R2_128x128, TILE_R2_128x64, TILE_R1_128x128, TILE_R1_128x64,
  CONV_TILE_COUNT
};

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_RELU6 = 2 };

struct ConvGeom {   // static description of one kernel class
  int kh, kw, stride, dil, pad, kc;   // kc = input channels per LDS stage
  int wino;                           // 1: weights are stored Winograd-transformed, F(2x2,3x3): 16 taps; 2: F(4x4,3x3): 36 taps
};
// The kind a layer's arithmetic belongs to (CONV_3x3_S1_WINO -> CONV_3x3_S1).
ConvKind conv_base_kind(ConvKind k);
ConvGeom conv_geom(ConvKind k);
int tile_bm(ConvTile t);
int tile_bn(ConvTile t);
int tile_th(ConvTile t);
int tile_tw(ConvTile t);

struct ConvArgs {
  const float* in;        // [B][Cin][Hin][Win]
  const float* w;         // tiled weights, see tile_weights()
  const float* bias;      // [Cout] (BN folded in) or nullptr
  float* out;             // [B][out_ctot][Hout][Wout]; channels [out_coff, out_coff+Cout) are written
  const float* res;       // residual [B][res_ctot][Hout][Wout] added before the activation, or nullptr
  const float* up;        // [B][Cout][up_h][up_w]: bilinear x2 (align_corners=False) upsample added, or nullptr
  int B, Cin, Hin, Win, Cout, Hout, Wout;
  int out_ctot, out_coff, res_ctot, res_coff, up_h, up_w;
  int act;
  int ksplit;             // >= 1: split the input-channel reduction over this many workgroups
  float* ws;              // split-K workspace, B*ksplit*Cout*Hout*Wout floats (ksplit > 1, or `up` on a Winograd tile)
  int defer_reduce;       // the split-K slabs stay in ws: 1 for a consumer that sums them itself (the grouped head finalize,
                          // ops.h; bias only), 2 for a later launch_reduce_group() (any epilogue)
  unsigned* sk_count;     // non-null: in-kernel split-K combine (splitk_combine_tile below), one zeroed counter per
                          // (image, output tile): conv_sk_counters() of them; null: splitk_reduce_kernel finishes the layer
  // workgroup -> (spatial tile, output-channel tile) map (FDT_BLOCK_MAP below): map_mode is the caller's choice
  // (CONV_MAP_*), n_sp / n_ct are filled in by launch_conv
  int map_mode, n_sp, n_ct;
  const unsigned char* in_u8;   // CONV_7x7_S*_U8 only: [B][Hin][Win][3] raw BGR bytes (instead of `in`); value = ((float)u8 -
  float u8_mean[3];             // u8_mean[c]) / u8_scale, the arithmetic of the ingest kernel (ops.hip: preprocess_kernel)
  float u8_scale;
  int tiles_per_wg;       // persistent-tile kernels (conv_1x1p.h): consecutive output tiles one workgroup walks; filled in by launch_conv
  // Input as a channel slice of a wider tensor: `in` points at the slice's first channel of image 0 and consecutive images are
  // in_bstride floats apart (0 = Cin * Hin * Win, a tensor of its own).  Every class honours it.
  long long in_bstride;
  // Second destination (FaceBoxes' Inception, FACEBOX/networks.py:43-57: the 1x1 branches that read the same x run as ONE
  // launch): output channels [out2_from, Cout) go to out2 [B][out2_ctot][Hout][Wout] at channel out2_coff + (co - out2_from),
  // channels below out2_from to `out` as usual.  Direct classes only (conv_kernel.h), no split-K (conv_shape_supported).
  float* out2;
  int out2_from, out2_ctot, out2_coff;
};
// floats between consecutive images of the input
__host__ __device__ inline long long conv_in_bstride(const ConvArgs& a) {
  return a.in_bstride ? a.in_bstride : (long long)a.Cin * a.Hin * a.Win;
}

enum { CONV_MAP_ROWS = 0, CONV_MAP_XCD_SPATIAL = 1, CONV_MAP_XCD_CHANNEL = 2, CONV_MAP_XCD_REGION = 3, CONV_MAP_COUNT = 4 };

// Workgroup map.  CONV_MAP_ROWS: id = channel_tile * n_sp + spatial_tile (all spatial tiles of one channel tile
// first; consecutive workgroups land on different XCDs).  The two XCD-aware maps use that workgroups are dealt
// round-robin over the 8 XCDs (each with its own 4 MB L2): lane8 = id % 8 (which workgroups share an XCD), q = id / 8,
//   XCD_SPATIAL (activations are the bigger operand): the XCD owns the spatial tiles s = 8*(q / n_ct) + lane8 and walks
//           the output-channel tiles of one spatial tile back to back -> the input patch is fetched into L2 once;
//   XCD_CHANNEL (weights are the bigger operand):    the XCD owns the channel tiles n = 8*(q / n_sp) + lane8 and walks
//           the spatial tiles -> every XCD streams only its 1/8 of the weights.
//   XCD_REGION (round 4): like XCD_SPATIAL, but the XCD owns a CONTIGUOUS run of ceil(n_sp / 8) spatial tiles (s = lane8 *
//           per + q / n_ct) instead of every eighth one -- neighbouring tiles share their halo rows and the 128-byte lines
//           their row segments straddle, and with interleaved ownership each of those lines is fetched from HBM once PER
//           XCD (measured with the TCC's read-request size classes, profiles/r04/conv_hbm_traffic.json: 3.0x the input on
//           the F(4x4) kernel's 160-byte row pieces, 2.0x on 8x16-pixel tiles' 64-byte rows).
// Workgroups decoded outside the tile grid exit at once.  Pure speed: nothing depends on the placement.  Measured on
// the Res50 graph the XCD-aware maps cut the conv kernels' HBM fetches by a quarter but are not faster on every
// layer (the MFMA-bound ones lose a little), so the map is a per-layer choice of the autotuner.
#define FDT_BLOCK_MAP(a_, s_, n_)                                               \
  int s_, n_;                                                                   \
  {                                                                             \
    const int id_ = blockIdx.x, l8_ = id_ & 7, q_ = id_ >> 3;                   \
    if ((a_).map_mode == CONV_MAP_XCD_SPATIAL) {                                \
      n_ = q_ % (a_).n_ct;                                                      \
      s_ = (q_ / (a_).n_ct) * 8 + l8_;                                          \
    } else if ((a_).map_mode == CONV_MAP_XCD_CHANNEL) {                         \
      s_ = q_ % (a_).n_sp;                                                      \
      n_ = (q_ / (a_).n_sp) * 8 + l8_;                                          \
    } else if ((a_).map_mode == CONV_MAP_XCD_REGION) {                          \
      const int per_ = ((a_).n_sp + 7) >> 3;                                    \
      n_ = q_ % (a_).n_ct;                                                      \
      s_ = q_ / (a_).n_ct;                                                      \
      if (s_ >= per_) return;                                                   \
      s_ += l8_ * per_;                                                         \
    } else {                                                                    \
      s_ = id_ % (a_).n_sp;                                                     \
      n_ = id_ / (a_).n_sp;                                                     \
    }                                                                           \
    if (s_ >= (a_).n_sp || n_ >= (a_).n_ct) return;                             \
  }

// Bilinear x2 upsample (align_corners=False, F.interpolate of pyramid.py:65) of the coarser map `u` [up_h][up_w] at the
// 4 consecutive output pixels (oy, ox0 .. ox0+3), added to v[0..3]: the fused `+ up` of ContextTexture.forward
// (pyramid.py:66-68).  One row lookup per call; same operand order in the conv epilogue and in splitk_reduce_kernel.
template <int VEC>
__device__ __forceinline__ void add_upsampled_x2(const float* __restrict__ u, int up_h, int up_w, int oy, int ox0,
                                                 float* v) {
  const float sy = fmaxf(0.5f * (oy + 0.5f) - 0.5f, 0.0f);
  int y0 = (int)sy;
  y0 = y0 < up_h - 1 ? y0 : up_h - 1;
  const int y1 = y0 + (y0 < up_h - 1 ? 1 : 0);
  const float ly = sy - (float)y0;
  const float* u0 = u + y0 * up_w;
  const float* u1 = u + y1 * up_w;
  if (VEC == 4 && (ox0 & 3) == 0) {
    // Four consecutive output columns 4m .. 4m+3 read source columns 2m-1 .. 2m+2 (clamped) with the fixed weights
    // (.25,.75) (.75,.25) (.25,.75) (.75,.25): 3 or 4 loads per source row instead of 8.  Same products and sums per
    // element as the generic form below (at the left edge column 0 has weight 0 on its right neighbour, as there).
    const int m2 = ox0 >> 1;                                  // 2m
    const int xl = m2 > 0 ? m2 - 1 : 0;
    const int xa = m2 < up_w - 1 ? m2 : up_w - 1;             // 2m, clamped (crop: the tile may hang over the map)
    const int xb = m2 + 1 < up_w - 1 ? m2 + 1 : up_w - 1;
    const int xr = m2 + 2 < up_w - 1 ? m2 + 2 : up_w - 1;
    float t0[4], t1[4];
    t0[0] = u0[xl]; t1[0] = u1[xl];
    if ((up_w & 1) == 0 && m2 + 1 < up_w) {                   // 8-byte aligned pair (plane and row sizes are even)
      const float2 p0 = *reinterpret_cast<const float2*>(u0 + m2), p1 = *reinterpret_cast<const float2*>(u1 + m2);
      t0[1] = p0.x; t0[2] = p0.y; t1[1] = p1.x; t1[2] = p1.y;
    } else {
      t0[1] = u0[xa]; t0[2] = u0[xb]; t1[1] = u1[xa]; t1[2] = u1[xb];
    }
    t0[3] = u0[xr]; t1[3] = u1[xr];
    const float l0 = m2 > 0 ? 0.75f : 0.0f;                   // column 0: sx clamps to 0 -> weight 1 on source column 0
    const int ia[4] = {m2 > 0 ? 0 : 1, 1, 1, 2};
    const float lx[4] = {l0, 0.25f, 0.75f, 0.25f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // x1 = x0 + (x0 < up_w - 1): past the last source column both taps are that column (t[] is clamped the same way)
      const float top = (1.0f - lx[e]) * t0[ia[e]] + lx[e] * t0[ia[e] + 1];
      const float bot = (1.0f - lx[e]) * t1[ia[e]] + lx[e] * t1[ia[e] + 1];
      v[e] += (1.0f - ly) * top + ly * bot;
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const float sx = fmaxf(0.5f * (ox0 + e + 0.5f) - 0.5f, 0.0f);
    int x0 = (int)sx;
    x0 = x0 < up_w - 1 ? x0 : up_w - 1;
    const int x1 = x0 + (x0 < up_w - 1 ? 1 : 0);
    const float lx = sx - (float)x0;
    const float top = (1.0f - lx) * u0[x0] + lx * u0[x1];
    const float bot = (1.0f - lx) * u1[x0] + lx * u1[x1];
    v[e] += (1.0f - ly) * top + ly * bot;
  }
}

// ---- in-kernel split-K combine ----------------------------------------------------------------------------------
// The ksplit workgroups of an output tile write their partial slabs to the workspace and take a ticket at the tile's
// counter; the one that arrives LAST sums the slabs in ks order -- the order and operand order of splitk_reduce_kernel,
// so the two paths are bit-identical -- and applies bias / upsample-add / residual / activation.  No workgroup ever
// waits for another one (no spinning: nothing to deadlock on), and the result does not depend on who arrives last.
// Visibility across the XCDs' L2s (MI355X_MICROARCH.md, "Valid forms"): every slab byte is stored `sc1`
// (write-through), every storing wave drains its stores (`s_waitcnt vmcnt(0)`) in front of the workgroup barrier behind
// which ONE lane adds to the counter (agent scope), the last arriver's waves load the slabs only behind a barrier
// that lane joins after its add has returned, and every slab load is a `buffer_load ... sc1`.  That is the guide's
// measured sc1-only hand-off -- measured at ONE workgroup per CU; the conv kernels run two to four.  So the ticket is
// taken with ACQ_REL ordering at agent scope on top of it (round 4): the release half writes back whatever of this
// workgroup's stores an L2 still holds before the add, the acquire half invalidates the adding CU's L1 before the last
// arriver's waves are released to their loads.  The feature is off in every committed plan (docs/EXPERIMENTS.md R3-4:
// not faster); where it is used, it rests on the architectural form, with the sc1 path as the fast case.
// Data-register hazard of the hand-written stores: a store of MORE than 8 bytes reads its data registers up to two wait
// states after issue (hence `s_nop 1` behind the dwordx4 form); the dwordx2 / dword forms read theirs at issue.
// One counter array serves all layers of a handle: correct because a handle's convs run strictly in order on one stream
// (the last arriver of a tile resets its counter before the launch ends, and the next launch that uses it is ordered
// behind this one) -- a caller that runs two convs of one handle concurrently must give them separate counters.
typedef float slab_f4 __attribute__((ext_vector_type(4)));
typedef float slab_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void slab_store4(float* p, float x, float y, float z, float w, bool wt) {
  if (wt) {
    const slab_f4 v = {x, y, z, w};
    // s_nop 1: a store of more than 8 bytes reads its data registers up to two wait states after issue; the compiler's
    // hazard recogniser covers that for the stores it emits itself, not for this one (seen: one lane's .z overwritten)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
  } else {
    *reinterpret_cast<float4*>(p) = make_float4(x, y, z, w);
  }
}
__device__ __forceinline__ void slab_store2(float* p, float x, float y, bool wt) {
  if (wt) {
    const slab_f2 v = {x, y};
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  } else {
    *reinterpret_cast<float2*>(p) = make_float2(x, y);
  }
}
__device__ __forceinline__ void slab_store1(float* p, float x, bool wt) {
  if (wt)
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
  else
    *p = x;
}

// Called by every thread of the workgroup after its slab stores; the tile is output channels [co0, co0 + nco) x rows
// [oy0, oy0 + th) x columns [ox0, ox0 + tw) of image b (clipped to the map here).  Needs Wout % 4 == 0 and ox0, tw
// multiples of 4 (launch_conv checks the first, the tiles guarantee the rest).  `flag`: one LDS word the caller no
// longer needs.  tile_id = s_tile + n_sp * n_tile.
template <int THREADS>
__device__ __forceinline__ void splitk_combine_tile(const ConvArgs& a, int b, int tile_id, int co0, int nco, int oy0,
                                                    int ox0, int th, int tw, unsigned* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's slab stores have reached memory
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned* c = a.sk_count + (long long)b * a.n_sp * a.n_ct + tile_id;
    const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the acquire's L1 invalidate has completed before the barrier below releases the other waves
    const unsigned last = old == (unsigned)a.ksplit - 1u ? 1u : 0u;
    if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return;
  const int HWo = a.Hout * a.Wout;
  const int rows = min(th, a.Hout - oy0), c4 = min(tw, a.Wout - ox0) >> 2, ncov = min(nco, a.Cout - co0);
  const int per_c = rows * c4, total = ncov * per_c;
  const unsigned kstride = (unsigned)a.Cout * (unsigned)HWo * 4u;   // bytes between the slabs of one image
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.ws + (long long)b * a.ksplit * a.Cout * HWo), 0, (int)(kstride * (unsigned)a.ksplit), 0x00020000);
  const float* res_b = a.res ? a.res + ((long long)b * a.res_ctot + a.res_coff) * HWo : nullptr;
  float* out_b = a.out + ((long long)b * a.out_ctot + a.out_coff) * HWo;
  typedef unsigned slab_u4 __attribute__((ext_vector_type(4)));
  for (int e = threadIdx.x; e < total; e += THREADS) {
    const int c = e / per_c, r = e - c * per_c;
    const int y = r / c4, x4 = r - y * c4;
    const int co = co0 + c, oy = oy0 + y, ox = ox0 + 4 * x4;
    const int pix = oy * a.Wout + ox;
    const unsigned off = ((unsigned)co * (unsigned)HWo + (unsigned)pix) * 4u;
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int k = 0;
    for (; k + 4 <= a.ksplit; k += 4) {   // four slabs in flight per lane and element
      slab_u4 p[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) p[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + (unsigned)(k + j) * kstride, 0, 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const slab_f4 f = __builtin_bit_cast(slab_f4, p[j]);
        v[0] += f.x; v[1] += f.y; v[2] += f.z; v[3] += f.w;
      }
    }
    for (; k < a.ksplit; ++k) {
      const slab_f4 f = __builtin_bit_cast(slab_f4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + (unsigned)k * kstride, 0, 16));
      v[0] += f.x; v[1] += f.y; v[2] += f.z; v[3] += f.w;
    }
    const float bv = a.bias ? a.bias[co] : 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] += bv;
    if (a.up) add_upsampled_x2<4>(a.up + ((long long)b * a.Cout + co) * a.up_h * a.up_w, a.up_h, a.up_w, oy, ox, v);
    const long long o = (long long)co * HWo + pix;
    if (res_b) {
      const float4 rv = *reinterpret_cast<const float4*>(res_b + o);
      v[0] += rv.x; v[1] += rv.y; v[2] += rv.z; v[3] += rv.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (a.act == ACT_RELU) v[j] = fmaxf(v[j], 0.0f);
      else if (a.act == ACT_RELU6) v[j] = fminf(fmaxf(v[j], 0.0f), 6.0f);
    }
    *reinterpret_cast<float4*>(out_b + o) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

extern thread_local bool exp_skip_reduce;   // experiment hook, see conv.hip

// The reduce passes of up to kReduceGroupMax split-K layers whose slabs were left in their (distinct) workspaces
// (defer_reduce = 2) in ONE launch: what splitk_reduce_kernel does per layer, same order, same bits.  The caller keeps every
// layer's inputs (slabs, residual, upsample source) untouched and its output unread until this has run.
constexpr int kReduceGroupMax = 16;
int launch_reduce_group(const ConvArgs* const* layers, int n, hipStream_t st);

// Counters an in-kernel split-K combine needs (ConvArgs.sk_count): one per image and output tile.
long long conv_sk_counters(ConvKind kind, ConvTile tile, const ConvArgs& a);
bool conv_combine_supported(ConvKind kind, ConvTile tile, const ConvArgs& a);

// Workspace floats a split-K launch needs.
long long conv_ws_floats(const ConvArgs& a);

// Re-tile OIHW weights for (kind, tile): [Cout_pad/BN][ceil(Cin/KC)][KC][taps][BN], zero padded.
// `scale` (per output channel, may be null) is the folded BN factor.
void tile_weights(const float* w_oihw, const float* scale, int Cout, int Cin, ConvKind kind,
                  ConvTile tile, std::vector<float>& out);

// FLOPs (2*MAC) of one launch, algorithmic (no padding).
double conv_flops(const ConvArgs& a, ConvKind kind);

// `device`: the HIP device `st` belongs to (< 0: ask the runtime); only used to set the per-(function, device)
// dynamic-LDS attribute once.
int launch_conv(ConvKind kind, ConvTile tile, const ConvArgs& a, hipStream_t st, int device = -1);
bool conv_supported(ConvKind kind, ConvTile tile);
bool conv_shape_supported(ConvKind kind, ConvTile tile, const ConvArgs& a);   // ... and this layer's shape / epilogue
size_t conv_lds_bytes(ConvKind kind, ConvTile tile);   // dynamic LDS of the instantiation

}  // namespace fdt
