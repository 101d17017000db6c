// Detector model handle: weight store (reference state-dict keys), eval-BN folding, per-shape execution
// plan, and the forward pass as a flat list of HIP launches on one stream.
//
// Forward graphs restated from:
//   Res50   pyramid.py:218-351  (Bottleneck :97-103, ContextTexture :61-69, SSHContext :41-48)
//   try3    pyramid_mb2_try3.py:218-340 (InvertedResidual :73-134)
//   FaceBox FACEBOX/networks.py:87-116 (Inception :43-57), FACEBOX/multibox_layer.py:28-50
// Dead work the reference computes and throws away in test phase (head_loc/head_conf,
// pyramid.py:312-317) is skipped; its weights are still accepted by set_tensor.
#include <algorithm>
#include <cmath>
#include <map>
#include <memory>
#include <mutex>
#include <tuple>
#include <set>
#include <string>
#include <vector>

#include "common.h"
#include "conv.h"
#include "ops.h"
#include "postproc.h"

using namespace fdt;

namespace {

struct HostT {
  std::vector<float> v;
  std::vector<long long> dims;
};

struct Tensor {
  std::string name;
  int C = 0, H = 0, W = 0;
  float* d = nullptr;
};

enum OpType { OP_CONV, OP_POOL, OP_DW, OP_HEADFIN, OP_MBOXFIN, OP_PAD, OP_EXPDW, OP_DWPROJ };

struct Op {
  OpType type;
  std::string name;
  double flops = 0;
  // conv
  ConvKind kind;
  ConvTile tile;
  ConvArgs ca;
  // pool / dw / finalize
  int in_t = -1, in2_t = -1, out_t = -1, out2_t = -1, stride = 1, crelu = 0, act = 0;
  int ksize = 3, pad = 1, dil = 1;   // depthwise geometry
  const float* w = nullptr;
  const float* bias = nullptr;
  const float* w2 = nullptr;     // OP_EXPDW: depthwise taps / bias (w, bias = the 1x1 expand)
  const float* bias2 = nullptr;
  const float* w3 = nullptr;     // OP_EXPDW with the block's 1x1 project in the same kernel (fused_ir.hip: PROJ): [oup][hid] / [oup]
  const float* bias3 = nullptr;
  int oup = 0, residual = 0;
  int hid = 0;
  int level0 = 0, p_off = 0, anchors = 1;
  bool needs_ws = false;
  bool combine = false;   // split-K partial sums are combined inside the conv kernel (ConvArgs.sk_count)
  bool last_head = false; // OP_HEADFIN: the one that launches the grouped finalize of all levels
  bool head = false;      // OP_CONV: a head conv (its split-K slabs go to the grouped finalize, no reduce pass)
  bool lazy = false;      // OP_CONV: the reduce pass waits, with others, for the first op that needs its result (plan_reduces)
  bool flush_before = false;   // any op: the pending reduce passes run (as one launch) before this op
  long long ws_off = 0;   // lazy: this layer's slabs inside the shared workspace
  // the 7x7 stem on the input tensor also exists as a raw-uint8 kernel (conv_stem_u8.h): launched instead of [ingest kernel +
  // this op] whenever a forward is fed uint8 frames (fdt_model::u8_src)
  bool u8_stem = false;
  ConvKind u8_kind = CONV_7x7_S2_U8;
  ConvTile u8_tile = TILE_128x64W;
  const float* u8_w = nullptr;
  // ... and the 3x3 / 2 stem of the MobileNetV2 detectors as a streaming vector-ALU kernel (stream_ir.hip): u8_w = [27][Cout]
  bool u8_stream = false;
};

struct DevW {
  float* w = nullptr;
  float* bias = nullptr;
};

}  // namespace

// Weights of one net on one GPU: the state dict as loaded, the BN-folded host copies and the tiled device copies per
// (layer, kernel class, tile).  Shared (shared_ptr) between a handle and its fdt_model_clone()s, so that several frames
// in flight cost one weight copy; the mutex covers the caches (handles may live on different host threads).
struct WeightStore {
  std::map<std::string, HostT> sd;
  struct HostW {
    std::vector<float> w, bias;   // BN-folded OIHW weights (+ fused second conv), bias
    int Cout = 0, Cin = 0;
  };
  std::map<std::string, HostW> host_w;   // per conv layer, kept for re-tiling
  std::map<std::string, DevW> wcache;    // key: layer|kind|tile
  std::mutex mu;
  void drop_device() {
    for (auto& kv : wcache) {
      if (kv.second.w) (void)hipFree(kv.second.w);
      if (kv.second.bias) (void)hipFree(kv.second.bias);
    }
    wcache.clear();
    host_w.clear();
  }
  ~WeightStore() { drop_device(); }
};

constexpr size_t kMaxGraphs = 16;   // captured graphs kept per handle
struct GraphKey {   // everything a captured forward bakes in besides the plan itself
  const void* out;
  const void* counts;
  int run_detect;
  float conf_t, nms_t;
  int first_op;        // 1: the stem ran eagerly in front of the graph on the caller's uint8 frame (fused ingest)
  bool operator<(const GraphKey& o) const {
    return std::tie(out, counts, run_detect, conf_t, nms_t, first_op) <
           std::tie(o.out, o.counts, o.run_detect, o.conf_t, o.nms_t, o.first_op);
  }
};

// One ticket of the pipelined host ingest (fdt_model_forward_async): pinned staging for the frame(s) and the Detect
// record, the device-side copies, and the events that tell the host / a consumer stream how far the ticket has got.
struct AsyncSlot {
  void* h_in = nullptr;
  void* d_in = nullptr;
  size_t in_bytes = 0;
  float *d_out = nullptr, *h_out = nullptr;
  int *d_counts = nullptr, *h_counts = nullptr;
  size_t out_floats = 0, n_counts = 0;
  hipEvent_t copied = nullptr, fwd = nullptr, done = nullptr, consumed = nullptr;
  bool busy = false, consumed_pending = false;
  bool released = false;   // retired by fdt_model_release: its forward may still be running when the slot is issued again
  int ticket = -1;
  void release() {
    if (h_in) (void)hipHostFree(h_in);
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (h_out) (void)hipHostFree(h_out);
    if (d_counts) (void)hipFree(d_counts);
    if (h_counts) (void)hipHostFree(h_counts);
    for (hipEvent_t* e : {&copied, &fwd, &done, &consumed})
      if (*e) (void)hipEventDestroy(*e);
    *this = AsyncSlot();
  }
};
constexpr int kAsyncSlots = 2;

struct fdt_model {
  int arch = 0, device = 0;
  hipStream_t stream = nullptr;
  AsyncSlot slots[kAsyncSlots];
  int next_ticket = 0;
  std::shared_ptr<WeightStore> W = std::make_shared<WeightStore>();
  std::set<std::string> expected;      // keys the forward graph reads
  bool finalized = false;

  // detect / priorbox configuration (mutable attributes of the reference module)
  int top_k = 750, nms_top_k = 5000;
  float conf_t = 0.3f, nms_t = 0.5f;
  bool pb_set = false, priors_dirty = true;
  int pb_w = 0, pb_h = 0;
  std::vector<int> pb_stride, pb_box;

  // plan
  int pB = 0, pH = 0, pW = 0;
  bool dry = false;
  std::vector<Tensor> tensors;
  std::vector<Op> ops;
  // the launches of a forward after the ingest kernel (convs ... Detect), captured once per (plan, output buffers,
  // thresholds) and replayed with one hipGraphLaunch: the small-frame configs are bound by launch issue otherwise
  std::map<GraphKey, hipGraphExec_t> graphs;
  std::map<GraphKey, unsigned long long> graph_used;   // last replay of each graph (LRU eviction at kMaxGraphs)
  unsigned long long graph_clock = 0;
  int plan_runs = 0;          // eager forwards since the plan was (re)built; capture starts at the second
  bool use_graph = true;
  bool stem_b3 = true;        // FaceBoxes' stem as split-bf16 products (conv_stem_b3.h); FDT_STEM_B3=0 at create time: the f32 form
  bool stream_ir = true;      // try3 / try4 / try5: the streaming vector-ALU kernels of stream_ir.hip (FDT_STREAM_IR=0 at create time: off)
  int fb_fuse = 2;            // FaceBoxes' Inception: 0 eight launches per block, 1 the three 1x1 branches on x as one, 2 also conv4 | conv6 (Builder::inception)
  struct Hint { int kind, tile, split, map, combine; };   // combine: in-kernel split-K combine (conv.h) instead of the reduce pass
  std::map<std::string, Hint> hints;                        // autotuned (kernel class, tile, split) per layer
  int hB = 0, hH = 0, hW = 0;                               // shape the hints were tuned for
  std::vector<void*> plan_allocs;
  std::vector<std::pair<int, int>> levels;  // (H, W) of each detection source
  int P = 0;
  float *d_loc = nullptr, *d_conf = nullptr, *d_logits = nullptr, *d_priors = nullptr, *d_out = nullptr;
  int* d_counts = nullptr;
  void* d_ws = nullptr;
  DetectPlan dplan;
  // fused ingest (conv_stem_u8.h): the uint8 frames of the forward being enqueued, their mean / scale; whether the last
  // forward ran fused (tensor "input" is then not materialised until somebody asks for it)
  int fuse_stem = 1;   // 0 off; 1 (default): the stride-2 stem (Res50: neutral in time, one launch and 25 MB per frame less); 2: also the
                       // stride-4 stem of FaceBoxes (measured SLOWER there: 230 vs 182 us per batch of 16 -- 19 source pixels staged
                       // per output pixel through byte loads; kept selectable, profiles/r04/fused_ingest_ab.txt)
  const unsigned char* u8_src = nullptr;
  float u8_mean[3] = {0.f, 0.f, 0.f};
  float u8_scale = 1.0f;
  bool last_fused = false;
  const unsigned char* last_u8_src = nullptr;
  unsigned char* d_frames_u8 = nullptr;
  unsigned char* d_src_u8 = nullptr;   // un-resized source frames (host entry point of the resize ingest)
  size_t src_bytes = 0;
  float *d_fb_boxes = nullptr, *d_fb_probs = nullptr;
  double flops_per_frame = 0;
  long long ws_floats = 0;   // split-K / fused-upsample workspace shared by all layers
  float* d_convws = nullptr;
  bool flush_at_end = false;     // reduce passes still pending behind the last op (plan_reduces)
  float* d_headws = nullptr;     // split-K slabs of the head convs, one region per level (they live until the grouped finalize)
  long long headws_floats = 0;
  HeadFinArgs headfin;           // table of the grouped head finalize (ops.h); nlev == 0: no PyramidBox heads
  unsigned* d_skcnt = nullptr;   // tile counters of the in-kernel split-K combine: shared by all layers (every layer leaves them zero)
  long long sk_counters = 0;

  // profiling
  bool profile = false;
  int seg_first = -1, seg_last = -1;   // segment timing (fdt_model_profile_segment): events only around ops [seg_first, seg_last]
  hipEvent_t seg_ev[2] = {nullptr, nullptr};
  long long passes = 0;   // eager or captured passes through run_ops (the FDT_SKIP_OPS hook spares the first)
  std::vector<hipEvent_t> ev;

  ~fdt_model() {
    free_plan();
    for (auto e : ev) (void)hipEventDestroy(e);
    for (auto e : seg_ev)
      if (e) (void)hipEventDestroy(e);
    if (d_src_u8) (void)hipFree(d_src_u8);
    for (auto& sl : slots) sl.release();
    if (stream) (void)hipStreamDestroy(stream);
  }
  void drop_graphs() {
    // A replay may still be executing on a CALLER's stream (fdt_model_forward_dev / the pipeline's per-slot streams), which a
    // synchronise of this handle's own stream does not cover: destroying a hipGraphExec_t under a replay in flight is a
    // use-after-free inside the runtime (an intermittent host-side crash, not a kernel fault).  Plan changes are rare.
    if (!graphs.empty()) (void)fdt::device_sync();
    for (auto& kv : graphs) (void)hipGraphExecDestroy(kv.second);
    graphs.clear();
    graph_used.clear();
    plan_runs = 0;
  }
  void free_plan() {
    drop_graphs();
    for (void* p : plan_allocs) (void)hipFree(p);
    plan_allocs.clear();
    tensors.clear();
    ops.clear();
    levels.clear();
    d_loc = d_conf = d_logits = d_priors = d_out = nullptr;
    d_counts = nullptr;
    d_ws = nullptr;
    d_frames_u8 = nullptr;
    d_fb_boxes = d_fb_probs = nullptr;
    d_convws = nullptr;
    d_skcnt = nullptr;
    sk_counters = 0;
    d_headws = nullptr;
    headws_floats = 0;
    headfin.nlev = 0;
    ws_floats = 0;
    pB = pH = pW = 0;
  }
};

namespace {

// ------------------------------------------------------------------------------------------ builder
struct Builder {
  fdt_model* m;
  int B;
  int rc = FDT_OK;

  int fail(int code) {
    if (rc == FDT_OK) rc = code;
    return -1;
  }

  int new_tensor(const std::string& name, int C, int H, int W) {
    Tensor t;
    t.name = name;
    t.C = C;
    t.H = H;
    t.W = W;
    if (!m->dry) {
      size_t bytes = (size_t)B * C * H * W * sizeof(float);
      if (hipMalloc((void**)&t.d, bytes) != hipSuccess) {
        set_error("hipMalloc(%zu) failed for tensor %s", bytes, name.c_str());
        return fail(FDT_ERR_HIP);
      }
      m->plan_allocs.push_back(t.d);
    }
    m->tensors.push_back(t);
    return (int)m->tensors.size() - 1;
  }

  const HostT* get(const std::string& key) {
    m->expected.insert(key);
    if (m->dry) return nullptr;
    auto it = m->W->sd.find(key);
    if (it == m->W->sd.end()) {
      set_error("missing weight tensor '%s'", key.c_str());
      fail(FDT_ERR_STATE);
      return nullptr;
    }
    return &it->second;
  }

  // Fold eval BatchNorm (eps 1e-5) into per-channel scale / bias.  bias_out = (conv_bias - mean) *
  // scale + beta, scale = gamma / sqrt(var + eps).
  void fold(const std::string& bn, const HostT* conv_bias, int Cout, std::vector<float>& scale,
            std::vector<float>& bias) {
    scale.assign(Cout, 1.0f);
    bias.assign(Cout, 0.0f);
    if (conv_bias)
      for (int c = 0; c < Cout; ++c) bias[c] = conv_bias->v[c];
    if (bn.empty()) return;
    const HostT* g = get(bn + ".weight");
    const HostT* be = get(bn + ".bias");
    const HostT* mu = get(bn + ".running_mean");
    const HostT* var = get(bn + ".running_var");
    if (!g || !be || !mu || !var) return;
    for (int c = 0; c < Cout; ++c) {
      double s = (double)g->v[c] / std::sqrt((double)var->v[c] + 1e-5);
      scale[c] = (float)s;
      bias[c] = (float)(((double)bias[c] - (double)mu->v[c]) * s + (double)be->v[c]);
    }
  }

  // Pick (tile, split-K) for one layer with a small analytic model of the kernel: every wave issues
  // its MFMAs (64 cycles each) on its own SIMD; how well one SIMD stays busy depends on how many
  // waves share it (latency hiding) and on the MFMAs a wave issues per LDS round trip.  Layers with
  // few output tiles and a deep reduction (the 32x32 .. 8x8 maps with 512-2048 channels) are split
  // along K so that >= 2-3 workgroups land on every CU; the partial sums are combined in a fixed
  // order by splitk_reduce_kernel.
  static void choose(ConvKind kind, int Cout, int Cin, int Ho, int Wo, int B, bool needs_ws,
                     ConvTile& best_tile, int& best_split) {
    const ConvGeom g = conv_geom(kind);
    const int nstages = ceil_div(Cin, g.kc);
    const double mfma_per_stage_tile = (double)g.kh * g.kw * (g.kc / 2);
    double best_cost = 1e300;
    best_tile = TILE_128x128;
    best_split = 1;
    for (int t = 0; t < CONV_TILE_COUNT; ++t) {
      if (!conv_supported(kind, (ConvTile)t)) continue;
      const int bm = tile_bm((ConvTile)t), bn = tile_bn((ConvTile)t);
      const int mfma_per_wave = (bm / 32) * (bn / 32) / 4;   // 32x32 tiles per wave (4 waves)
      const int resident = mfma_per_wave >= 4 ? 3 : 4;          // workgroups per CU (launch bounds)
      const double tile_eff = mfma_per_wave >= 4 ? 1.0 : (mfma_per_wave == 2 ? 0.85 : 0.7);
      const long long tiles_m = (long long)ceil_div(Ho, tile_th((ConvTile)t)) * ceil_div(Wo, tile_tw((ConvTile)t));
      const long long base_blocks = tiles_m * ceil_div(Cout, bn) * B;
      for (int split = 1; split <= 64; split *= 2) {
        if (split > 1 && (split > nstages / 2 || nstages < 8)) break;
        const long long blocks = base_blocks * split;
        const double per_cu = (double)blocks / 256.0;
        const double share = std::min((double)resident, std::max(1.0, std::ceil(per_cu)));
        const double occ_eff = share >= 4 ? 0.80 : (share >= 3 ? 0.72 : (share >= 2 ? 0.55 : 0.32));
        const double rounds = std::max(1.0, std::ceil(per_cu));   // workgroups each CU runs in total
        const double stage_cycles = mfma_per_stage_tile * mfma_per_wave * 64.0;
        const double block_cycles = 4000.0 + std::ceil((double)nstages / split) * stage_cycles;
        double cycles = rounds * block_cycles / (occ_eff * tile_eff);
        if (split > 1) {
          // reduce pass: read split partials + write, ~3 TB/s effective, plus a launch
          const double bytes = (double)B * Cout * Ho * Wo * 4.0 * (split + 1);
          cycles += 2.0e9 * (bytes / 3.0e12) + 6000.0;
        }
        if (cycles < best_cost) {
          best_cost = cycles;
          best_tile = (ConvTile)t;
          best_split = split;
        }
      }
    }
  }

  // Upload (cached) tiled weights + bias of a layer for (kind, tile) from the folded host copy.
  static int device_weights(fdt_model* m, const std::string& key, ConvKind kind, ConvTile tile, DevW& out) {
    std::string ck = key + "|" + std::to_string((int)kind) + "|" + std::to_string((int)tile);
    WeightStore& W = *m->W;
    std::lock_guard<std::mutex> lk(W.mu);
    auto it = W.wcache.find(ck);
    if (it != W.wcache.end()) {
      out = it->second;
      return FDT_OK;
    }
    auto hw = W.host_w.find(key);
    FDT_REQUIRE(hw != W.host_w.end(), FDT_ERR_STATE, "no folded weights for layer %s", key.c_str());
    std::vector<float> tiled;
    tile_weights(hw->second.w.data(), nullptr, hw->second.Cout, hw->second.Cin, kind, tile, tiled);
    DevW d;
    FDT_HIP(hipMalloc((void**)&d.w, tiled.size() * 4));
    FDT_HIP(copy_sync(d.w, tiled.data(), tiled.size() * 4, hipMemcpyHostToDevice, m->stream));
    FDT_HIP(hipMalloc((void**)&d.bias, (size_t)hw->second.Cout * 4));
    FDT_HIP(copy_sync(d.bias, hw->second.bias.data(), (size_t)hw->second.Cout * 4, hipMemcpyHostToDevice, m->stream));
    W.wcache[ck] = d;
    out = d;
    return FDT_OK;
  }

  struct ConvOpt {
    std::string bn;           // BatchNorm module to fold ("" = none)
    bool bias = true;         // conv has its own bias tensor
    int act = ACT_NONE;
    int out_t = -1, out_coff = 0;   // write into a channel slice of an existing tensor
    int res_t = -1;           // residual tensor (same shape as the output)
    int up_t = -1;            // coarser map to bilinear-upsample and add
    std::string name2;        // second conv concatenated along Cout (fused loc+conf heads)
    int cout2 = 0;
    int groups = 1;           // grouped 1x1 (pyramid_mobile_try1.py:185-186): run as the block-diagonal dense conv
    // More convolutions of the same geometry in the SAME launch, concatenated along Cout behind the layer itself (FaceBoxes'
    // Inception, FACEBOX/networks.py:43-57).  Each keeps its own bias and BatchNorm fold; a part may read only the input
    // channels [in_off, in_off + in_c) of the launch's input (its weights are zero elsewhere: the zeros add an exact 0.0f to
    // every fmaf chain, like the grouped 1x1 above).  w_in_c > 0 restricts the layer's own weights the same way.
    struct Part {
      std::string name, bn;
      int cout = 0, in_off = 0, in_c = 0;
    };
    std::vector<Part> more;
    int w_in_off = 0, w_in_c = 0;
    // output channels from out2_from on go to tensor out2_t at channel out2_coff (ConvArgs.out2)
    int out2_t = -1, out2_from = 0, out2_coff = 0;
    // the input is the channel slice [in_coff, in_coff + in_c) of tensor in_t (ConvArgs.in_bstride)
    int in_coff = 0, in_c = 0;
  };

  static bool stem_s4_enabled() {
#ifdef FDT_EXPERIMENTS   // A/B against the generic direct kernel (tools/experiments/r4_job27.sh)
    if (const char* e = getenv("FDT_STEM_S4")) return atoi(e) != 0;
#endif
    return true;
  }

  int conv(const std::string& name, int in_t, int Cout, ConvKind kind, const ConvOpt& o) {
    if (rc != FDT_OK) return -1;
    Tensor in = m->tensors[in_t];
    const int in_ctot = in.C;
    if (o.in_c > 0) {                         // a channel slice of the tensor is this launch's input
      if (o.in_coff < 0 || o.in_coff + o.in_c > in.C) {
        set_error("input slice out of range at %s", name.c_str());
        return fail(FDT_ERR_STATE);
      }
      in.C = o.in_c;
      if (!m->dry) in.d += (size_t)o.in_coff * in.H * in.W;
    }
    const ConvGeom g = conv_geom(kind);
    const int Ho = (in.H + 2 * g.pad - g.dil * (g.kh - 1) - 1) / g.stride + 1;
    const int Wo = (in.W + 2 * g.pad - g.dil * (g.kw - 1) - 1) / g.stride + 1;
    int Ctot = Cout + o.cout2;
    for (const auto& pt : o.more) Ctot += pt.cout;
    const bool special = !o.more.empty() || o.out2_t >= 0 || o.in_c > 0 || o.w_in_c > 0;   // direct classes, no split-K
    if (Ho < 1 || Wo < 1) {
      set_error("input too small: layer %s would have a %dx%d output", name.c_str(), Ho, Wo);
      return fail(FDT_ERR_ARG);
    }
    int out_t = o.out_t;
    if (out_t < 0) out_t = new_tensor(name, Ctot, Ho, Wo);
    if (out_t < 0) return -1;
    const HostT* w = get(name + ".weight");
    const HostT* b = o.bias ? get(name + ".bias") : nullptr;
    if (!o.bn.empty())
      for (const char* sfx : {".weight", ".bias", ".running_mean", ".running_var"}) (void)get(o.bn + sfx);
    const HostT* w2 = o.cout2 ? get(o.name2 + ".weight") : nullptr;
    const HostT* b2 = o.cout2 ? get(o.name2 + ".bias") : nullptr;
    for (const auto& pt : o.more) {           // the parts' keys belong to the state dict whether or not weights are loaded yet
      (void)get(pt.name + ".weight");
      (void)get(pt.name + ".bias");
      if (!pt.bn.empty())
        for (const char* sfx : {".weight", ".bias", ".running_mean", ".running_var"}) (void)get(pt.bn + sfx);
    }
    Op op;
    op.type = OP_CONV;
    op.name = name;
    op.kind = kind;
    int ksplit = 1, map_mode = CONV_MAP_ROWS;
    bool combine = false;
    auto hint = m->hints.find(name);
    if (hint != m->hints.end() && m->hB == B && m->hH == m->tensors[0].H && m->hW == m->tensors[0].W &&
        conv_base_kind((ConvKind)hint->second.kind) == kind &&
        (o.out2_t < 0 || ((ConvKind)hint->second.kind == kind || (ConvKind)hint->second.kind == CONV_1x1_S1_K32 ||
                          (ConvKind)hint->second.kind == CONV_1x1_S1_K64 || (ConvKind)hint->second.kind == CONV_1x1_S1_B3)) &&
        ((ConvKind)hint->second.kind != CONV_1x1_S1_B3 || (in.W & 3) == 0) &&
        conv_supported((ConvKind)hint->second.kind, (ConvTile)hint->second.tile) &&
        // the persistent 1x1 class has shape limits of its own (conv.hip: conv_shape_supported); a plan entry that does not
        // fit this layer is ignored like one for another shape
        !(((ConvKind)hint->second.kind == CONV_1x1_S1_P16 || (ConvKind)hint->second.kind == CONV_1x1_S1_P32 ||
           (ConvKind)hint->second.kind == CONV_1x1_S1_PB3) &&
          ((in.W & 3) || o.up_t >= 0 || hint->second.split != 1 || (o.out2_t >= 0 && (ConvKind)hint->second.kind == CONV_1x1_S1_PB3) ||
           ceil_div(in.C, conv_geom((ConvKind)hint->second.kind).kc) < ((ConvKind)hint->second.kind == CONV_1x1_S1_PB3 ? 3 : 2)))) {
      kind = (ConvKind)hint->second.kind;     // e.g. the Winograd implementation of a 3x3/s1 layer
      op.kind = kind;
      op.tile = (ConvTile)hint->second.tile;
      ksplit = std::max(1, std::min(hint->second.split, ceil_div(in.C, conv_geom(kind).kc)));
      map_mode = hint->second.map;
      combine = hint->second.combine != 0;
    } else if (kind == CONV_3x3_S1 && Ctot <= 8 && (long long)B * Ho * Wo >= 32768) {
      // narrow loc/conf heads on the large maps: the vector-ALU kernel (conv_n8.h: 70-75 TFLOP/s against the 46 of the
      // best MFMA variant, which pads 8 output channels to 32), split along K until ~512 workgroups are in the grid
      kind = CONV_3x3_S1_N8;
      op.kind = kind;
      op.tile = TILE_N8_32x64;
      const long long wgs = (long long)B * ceil_div(Ho, tile_th(op.tile)) * ceil_div(Wo, tile_tw(op.tile));
      while (wgs * ksplit < 512 && ksplit * 2 <= std::min(32, in.C / 8)) ksplit *= 2;
    } else if (kind == CONV_7x7_S4 && in.C == 3 && o.groups == 1 && o.res_t < 0 && o.up_t < 0 &&
               conv_supported(CONV_7x7_S4_K168, TILE_128x32W) && stem_s4_enabled()) {
      // FaceBoxes' stem: K = 3 x 7 x 8 instead of 4 x 49, three workgroups per CU (conv_stem_s4.h); with 16-byte rows the
      // split-bf16 form of the same k layout on the bf16 matrix pipe (conv_stem_b3.h; FDT_STEM_B3=0: the f32 form)
      const bool b3_ok = m->stem_b3;
      kind = (b3_ok && (in.W & 3) == 0 && special == false && conv_supported(CONV_7x7_S4_B3, TILE_128x32W)) ? CONV_7x7_S4_B3
                                                                                                               : CONV_7x7_S4_K168;
      op.kind = kind;
      op.tile = TILE_128x32W;
    } else {
      choose(kind, Ctot, in.C, Ho, Wo, B, o.up_t >= 0, op.tile, ksplit);
    }
    if (o.out2_t >= 0) ksplit = 1;            // the reduce passes know one destination
    ConvArgs& a = op.ca;
    memset(&a, 0, sizeof(a));
    a.ksplit = ksplit;
    a.map_mode = map_mode;
    a.B = B;
    a.Cin = in.C;
    a.Hin = in.H;
    a.Win = in.W;
    a.Cout = Ctot;
    a.Hout = Ho;
    a.Wout = Wo;
    a.out_ctot = m->tensors[out_t].C;
    a.out_coff = o.out_coff;
    a.act = o.act;
    if (o.in_c > 0) a.in_bstride = (long long)in_ctot * in.H * in.W;
    if (o.out2_t >= 0) {
      const Tensor& t2 = m->tensors[o.out2_t];
      if (t2.H != Ho || t2.W != Wo || o.out2_from <= 0 || o.out2_from >= Ctot || o.out2_coff + (Ctot - o.out2_from) > t2.C ||
          o.out_coff + o.out2_from > m->tensors[out_t].C || o.res_t >= 0 || o.up_t >= 0) {
        set_error("bad second destination at %s", name.c_str());
        return fail(FDT_ERR_STATE);
      }
      a.out2 = t2.d;
      a.out2_from = o.out2_from;
      a.out2_ctot = t2.C;
      a.out2_coff = o.out2_coff;
      op.out2_t = o.out2_t;
    }
    op.flops = conv_flops(a, kind);
    {                                         // algorithmic FLOPs: the zero blocks of a block-diagonal launch are not work
      double macs = (double)Cout * (o.w_in_c > 0 ? o.w_in_c : in.C) / std::max(1, o.groups) + (double)o.cout2 * in.C;
      for (const auto& pt : o.more) macs += (double)pt.cout * (pt.in_c > 0 ? pt.in_c : in.C);
      op.flops = 2.0 * B * (double)Ho * Wo * g.kh * g.kw * macs;
    }
    op.out_t = out_t;
    op.needs_ws = ksplit > 1;
    if (op.needs_ws) m->ws_floats = std::max(m->ws_floats, conv_ws_floats(a));
    {
      ConvArgs probe = a;
      probe.ws = (float*)16;   // conv_combine_supported only asks whether there is one
#ifdef FDT_EXPERIMENTS   // tools/experiments/force_combine.sh (make EXTRA=-DFDT_EXPERIMENTS): 1 all / 0 none
      static const int force = getenv("FDT_FORCE_COMBINE") ? atoi(getenv("FDT_FORCE_COMBINE")) : -1;
      if (force >= 0) combine = force != 0 && ksplit <= (getenv("FDT_FORCE_COMBINE_MAXS") ? atoi(getenv("FDT_FORCE_COMBINE_MAXS")) : 4096);
#endif
      op.combine = combine && conv_combine_supported(kind, op.tile, probe);
      if (op.combine) m->sk_counters = std::max(m->sk_counters, conv_sk_counters(kind, op.tile, a));
    }
    if (!m->dry) {
      if (rc != FDT_OK) return -1;
      const size_t per = (size_t)in.C * g.kh * g.kw;
      const size_t per_own = (size_t)(o.w_in_c > 0 ? o.w_in_c : in.C) * g.kh * g.kw;      // the layer's own weights: [Cout][w_in_c][kh][kw]
      if (w->v.size() * o.groups != per_own * Cout || (b && (int)b->v.size() != Cout) ||
          (w2 && w2->v.size() != per * o.cout2) || in.C % o.groups || Cout % o.groups || (o.groups > 1 && o.cout2) ||
          ((o.w_in_c > 0 || !o.more.empty()) && (o.groups > 1 || o.cout2)) || o.w_in_off + (o.w_in_c > 0 ? o.w_in_c : in.C) > in.C) {
        set_error("weight shape mismatch for layer %s", name.c_str());
        return fail(FDT_ERR_STATE);
      }
      std::vector<float> scale, bias, wcat;
      fold(o.bn, b, Cout, scale, bias);
      if (rc != FDT_OK) return -1;
      const std::vector<float>* wsrc = &w->v;
      if (o.w_in_c > 0 || !o.more.empty()) {
        // [Ctot][in.C][kh][kw]: the layer's rows, then every part's, each over its own input channels (zeros elsewhere)
        const size_t kk = (size_t)g.kh * g.kw;
        wcat.assign(per * Ctot, 0.0f);
        auto place = [&](const std::vector<float>& src, int co0, int cout, int in_off, int in_c) {
          for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < in_c; ++ci)
              for (size_t k = 0; k < kk; ++k)
                wcat[(size_t)(co0 + co) * per + (size_t)(in_off + ci) * kk + k] = src[((size_t)co * in_c + ci) * kk + k];
        };
        place(w->v, 0, Cout, o.w_in_off, o.w_in_c > 0 ? o.w_in_c : in.C);
        int co0 = Cout;
        for (const auto& pt : o.more) {
          const HostT* pw = get(pt.name + ".weight");
          const HostT* pb = get(pt.name + ".bias");
          const int pin = pt.in_c > 0 ? pt.in_c : in.C;
          if (!pw || !pb || pw->v.size() != (size_t)pt.cout * pin * kk || (int)pb->v.size() != pt.cout || pt.in_off + pin > in.C) {
            set_error("weight shape mismatch for part %s of layer %s", pt.name.c_str(), name.c_str());
            return fail(FDT_ERR_STATE);
          }
          std::vector<float> ps, pbias;
          fold(pt.bn, pb, pt.cout, ps, pbias);
          if (rc != FDT_OK) return -1;
          place(pw->v, co0, pt.cout, pt.in_off, pin);
          scale.insert(scale.end(), ps.begin(), ps.end());
          bias.insert(bias.end(), pbias.begin(), pbias.end());
          co0 += pt.cout;
        }
        wsrc = &wcat;
      }
      if (o.groups > 1) {   // zeros outside the diagonal blocks add exact 0.0f to every fmaf chain
        const size_t cig = in.C / o.groups, cog = Cout / o.groups, kk = (size_t)g.kh * g.kw;
        wcat.assign(per * Cout, 0.0f);
        for (int co = 0; co < Cout; ++co) {
          const size_t gi = co / cog;
          for (size_t ci = 0; ci < cig; ++ci)
            for (size_t k = 0; k < kk; ++k)
              wcat[(size_t)co * per + (gi * cig + ci) * kk + k] = w->v[((size_t)co * cig + ci) * kk + k];
        }
        wsrc = &wcat;
      }
      if (o.cout2) {
        wcat = w->v;
        wcat.insert(wcat.end(), w2->v.begin(), w2->v.end());
        scale.resize(Ctot, 1.0f);
        bias.resize(Ctot, 0.0f);
        for (int c = 0; c < o.cout2; ++c) bias[Cout + c] = b2->v[c];
        wsrc = &wcat;
      }
      bool have_host;
      {
        std::lock_guard<std::mutex> lk(m->W->mu);
        have_host = m->W->host_w.count(name) != 0;
      }
      if (!have_host) {
        WeightStore::HostW hw;
        hw.Cout = Ctot;
        hw.Cin = in.C;
        hw.bias = bias;
        hw.w = *wsrc;
        for (int co = 0; co < Ctot; ++co)
          for (size_t k = 0; k < per; ++k) hw.w[(size_t)co * per + k] *= scale[co];
        std::lock_guard<std::mutex> lk(m->W->mu);
        m->W->host_w[name] = std::move(hw);
      }
      DevW dw;
      int r = device_weights(m, name, kind, op.tile, dw);
      if (r != FDT_OK) return fail(r);
      // the stem on the input tensor also gets its raw-uint8 form (conv_stem_u8.h; Res50: 7x7 / 2 -> 64, FaceBoxes: 7x7 / 4 -> 24)
      if (in_t == 0 && in.C == 3 && o.groups == 1 && !o.cout2 && o.res_t < 0 && o.up_t < 0 &&
          (conv_base_kind(kind) == CONV_7x7_S2 || conv_base_kind(kind) == CONV_7x7_S4)) {
        op.u8_kind = conv_base_kind(kind) == CONV_7x7_S2 ? CONV_7x7_S2_U8 : CONV_7x7_S4_U8;
        op.u8_tile = conv_base_kind(kind) == CONV_7x7_S2 ? TILE_128x64W : TILE_128x32W;
        // ... on the bf16 matrix pipe where the frame width allows its 12-byte groups (conv_stem_u8b.h; FDT_STEM_B3=0: the f32 form)
        const ConvKind u8b = conv_base_kind(kind) == CONV_7x7_S2 ? CONV_7x7_S2_U8B : CONV_7x7_S4_U8B;
        if (m->stem_b3 && (in.W & 3) == 0 && conv_supported(u8b, op.u8_tile)) op.u8_kind = u8b;
        if (conv_supported(op.u8_kind, op.u8_tile)) {
          DevW du;
          r = device_weights(m, name, op.u8_kind, op.u8_tile, du);
          if (r != FDT_OK) return fail(r);
          op.u8_w = du.w;
          op.u8_stem = true;
        }
      }
      // the 3x3 / 2 stem of try3 / try5 (3 -> 32) on raw uint8 frames: a streaming kernel (stream_ir.hip), weights [27][32]
      if (in_t == 0 && in.C == 3 && o.groups == 1 && !o.cout2 && !special && o.res_t < 0 && o.up_t < 0 && m->stream_ir &&
          conv_base_kind(kind) == CONV_3x3_S2 && stem3x3s2_u8_supported(in.H, in.W, Ctot) && out_t >= 0 && o.out_coff == 0 &&
          m->tensors[out_t].C == Ctot) {
        std::lock_guard<std::mutex> lk(m->W->mu);
        const std::string ck = name + "|s3u8";
        auto it = m->W->wcache.find(ck);
        DevW d;
        if (it == m->W->wcache.end()) {
          const WeightStore::HostW& hw = m->W->host_w[name];          // [Cout][3][3][3], BN folded
          std::vector<float> wt((size_t)27 * Ctot);
          for (int co = 0; co < Ctot; ++co)
            for (int k = 0; k < 27; ++k) wt[(size_t)k * Ctot + co] = hw.w[(size_t)co * 27 + k];
          if (hipMalloc((void**)&d.w, wt.size() * 4) != hipSuccess) {
            set_error("hipMalloc failed for %s", ck.c_str());
            return fail(FDT_ERR_HIP);
          }
          (void)copy_sync(d.w, wt.data(), wt.size() * 4, hipMemcpyHostToDevice, m->stream);
          d.bias = nullptr;      // the kernel takes the layer's own bias (ConvArgs.bias); the cache frees every pointer it holds once
          m->W->wcache[ck] = d;
        } else {
          d = it->second;
        }
        op.u8_w = d.w;
        op.u8_stem = true;
        op.u8_stream = true;
      }
      a.in = in.d;
      a.w = dw.w;
      a.bias = dw.bias;
      a.out = m->tensors[out_t].d;
      if (o.res_t >= 0) {
        const Tensor& rt = m->tensors[o.res_t];
        if (rt.H != Ho || rt.W != Wo || rt.C != Ctot) {
          set_error("residual shape mismatch at %s", name.c_str());
          return fail(FDT_ERR_STATE);
        }
        a.res = rt.d;
        a.res_ctot = rt.C;
        a.res_coff = 0;
      }
      if (o.up_t >= 0) {
        const Tensor& ut = m->tensors[o.up_t];
        if (ut.C != Ctot) {
          set_error("upsample channel mismatch at %s", name.c_str());
          return fail(FDT_ERR_STATE);
        }
        a.up = ut.d;
        a.up_h = ut.H;
        a.up_w = ut.W;
      }
    }
    m->ops.push_back(op);
    m->flops_per_frame += op.flops / B;
    return out_t;
  }

  int pool(const std::string& name, int in_t, int stride, int crelu) {
    if (rc != FDT_OK) return -1;
    const Tensor in = m->tensors[in_t];
    const int Ho = (in.H - 1) / stride + 1, Wo = (in.W - 1) / stride + 1;
    int out_t = new_tensor(name, crelu ? 2 * in.C : in.C, Ho, Wo);
    if (out_t < 0) return -1;
    Op op;
    op.type = OP_POOL;
    op.name = name;
    op.in_t = in_t;
    op.out_t = out_t;
    op.stride = stride;
    op.crelu = crelu;
    memset(&op.ca, 0, sizeof(op.ca));
    m->ops.push_back(op);
    return out_t;
  }

  int pad1(const std::string& name, int in_t) {
    if (rc != FDT_OK) return -1;
    const Tensor in = m->tensors[in_t];
    int out_t = new_tensor(name, in.C, in.H + 2, in.W + 2);
    if (out_t < 0) return -1;
    Op op;
    op.type = OP_PAD;
    op.name = name;
    op.in_t = in_t;
    op.out_t = out_t;
    memset(&op.ca, 0, sizeof(op.ca));
    m->ops.push_back(op);
    return out_t;
  }

  // depthwise 3x3 + BN + ReLU6
  int dwconv(const std::string& name, const std::string& bn, int in_t, int stride, int act, int K = 3, int pad = 1,
             int dil = 1, bool has_bias = false) {
    if (rc != FDT_OK) return -1;
    const Tensor in = m->tensors[in_t];
    const int Ho = (in.H + 2 * pad - dil * (K - 1) - 1) / stride + 1;
    const int Wo = (in.W + 2 * pad - dil * (K - 1) - 1) / stride + 1;
    if (Ho < 1 || Wo < 1) {
      set_error("input too small: layer %s would have a %dx%d output", name.c_str(), Ho, Wo);
      return fail(FDT_ERR_ARG);
    }
    int out_t = new_tensor(name, in.C, Ho, Wo);
    if (out_t < 0) return -1;
    const HostT* w = get(name + ".weight");
    const HostT* cb = has_bias ? get(name + ".bias") : nullptr;
    Op op;
    op.type = OP_DW;
    op.name = name;
    op.in_t = in_t;
    op.out_t = out_t;
    op.stride = stride;
    op.act = act;
    op.ksize = K;
    op.pad = pad;
    op.dil = dil;
    op.flops = 2.0 * B * (double)Ho * Wo * in.C * K * K;
    memset(&op.ca, 0, sizeof(op.ca));
    std::vector<float> scale, bias;
    fold(bn, cb, in.C, scale, bias);
    if (!m->dry) {
      if (rc != FDT_OK) return -1;
      if ((int)w->v.size() != in.C * K * K) {
        set_error("depthwise weight shape mismatch for %s", name.c_str());
        return fail(FDT_ERR_STATE);
      }
      std::string ck = name + "|dw";
      std::lock_guard<std::mutex> lk(m->W->mu);
      auto it = m->W->wcache.find(ck);
      DevW d;
      if (it == m->W->wcache.end()) {
        std::vector<float> ws(w->v);
        for (int c = 0; c < in.C; ++c)
          for (int k = 0; k < K * K; ++k) ws[c * K * K + k] *= scale[c];
        if (hipMalloc((void**)&d.w, ws.size() * 4) != hipSuccess ||
            hipMalloc((void**)&d.bias, bias.size() * 4) != hipSuccess) {
          set_error("hipMalloc failed for %s", name.c_str());
          return fail(FDT_ERR_HIP);
        }
        (void)copy_sync(d.w, ws.data(), ws.size() * 4, hipMemcpyHostToDevice, m->stream);
        (void)copy_sync(d.bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice, m->stream);
        m->W->wcache[ck] = d;
      } else {
        d = it->second;
      }
      op.w = d.w;
      op.bias = d.bias;
    }
    m->ops.push_back(op);
    m->flops_per_frame += op.flops / B;
    return out_t;
  }

  // conv[0..5] of an InvertedResidual with expand_ratio != 1 as ONE kernel (fused_ir.hip): 1x1 expand + BN + ReLU6 and the
  // depthwise 3x3 + BN + ReLU6; the expanded tensor never reaches HBM.  pyramid_mb2_try3.py:96-114.
  // oup > 0: the block's 1x1 project conv (conv.6) + BN (conv.7) (+ residual) in the same kernel: the whole block, output [oup]
  int expand_dw(const std::string& p, int in_t, int hid, int stride, int oup = 0, bool residual = false) {
    if (rc != FDT_OK) return -1;
    const Tensor in = m->tensors[in_t];
    const int Ho = (in.H - 1) / stride + 1, Wo = (in.W - 1) / stride + 1;
    const std::string n1 = p + ".conv.0", bn1 = p + ".conv.1", n2 = p + ".conv.3", bn2 = p + ".conv.4", n3 = p + ".conv.6",
                      bn3 = p + ".conv.7";
    int out_t = oup ? new_tensor(n3, oup, Ho, Wo) : new_tensor(n2, hid, Ho, Wo);
    if (out_t < 0) return -1;
    const HostT* w1 = get(n1 + ".weight");
    const HostT* w2 = get(n2 + ".weight");
    const HostT* w3 = oup ? get(n3 + ".weight") : nullptr;
    std::vector<float> s1, b1, s2, b2, s3, b3;
    fold(bn1, nullptr, hid, s1, b1);
    fold(bn2, nullptr, hid, s2, b2);
    if (oup) fold(bn3, nullptr, oup, s3, b3);
    Op op;
    op.type = OP_EXPDW;
    op.name = p + (oup ? ".expand_dw_project" : ".expand_dw");
    op.in_t = in_t;
    op.out_t = out_t;
    op.stride = stride;
    op.hid = hid;
    op.oup = oup;
    op.residual = residual ? 1 : 0;
    op.flops = 2.0 * B * ((double)in.H * in.W * in.C * hid + (double)Ho * Wo * hid * 9 + (double)Ho * Wo * hid * oup);
    memset(&op.ca, 0, sizeof(op.ca));
    if (!m->dry) {
      if (rc != FDT_OK) return -1;
      if ((int)w1->v.size() != hid * in.C || (int)w2->v.size() != hid * 9) {
        set_error("weight shape mismatch for the fused block %s", p.c_str());
        return fail(FDT_ERR_STATE);
      }
      std::lock_guard<std::mutex> lk(m->W->mu);
      DevW d1, d2;
      auto it = m->W->wcache.find(p + "|expdw1");
      if (it == m->W->wcache.end()) {
        std::vector<float> a(w1->v), d(w2->v);
        for (int c = 0; c < hid; ++c) {
          for (int k = 0; k < in.C; ++k) a[(size_t)c * in.C + k] *= s1[c];
          for (int k = 0; k < 9; ++k) d[(size_t)c * 9 + k] *= s2[c];
        }
        if (hipMalloc((void**)&d1.w, a.size() * 4) != hipSuccess || hipMalloc((void**)&d1.bias, (size_t)hid * 4) != hipSuccess ||
            hipMalloc((void**)&d2.w, d.size() * 4) != hipSuccess || hipMalloc((void**)&d2.bias, (size_t)hid * 4) != hipSuccess) {
          set_error("hipMalloc failed for %s", p.c_str());
          return fail(FDT_ERR_HIP);
        }
        (void)copy_sync(d1.w, a.data(), a.size() * 4, hipMemcpyHostToDevice, m->stream);
        (void)copy_sync(d1.bias, b1.data(), (size_t)hid * 4, hipMemcpyHostToDevice, m->stream);
        (void)copy_sync(d2.w, d.data(), d.size() * 4, hipMemcpyHostToDevice, m->stream);
        (void)copy_sync(d2.bias, b2.data(), (size_t)hid * 4, hipMemcpyHostToDevice, m->stream);
        m->W->wcache[p + "|expdw1"] = d1;
        m->W->wcache[p + "|expdw2"] = d2;
      } else {
        d1 = it->second;
        d2 = m->W->wcache[p + "|expdw2"];
      }
      op.w = d1.w;
      op.bias = d1.bias;
      op.w2 = d2.w;
      op.bias2 = d2.bias;
      if (oup) {
        if ((int)w3->v.size() != oup * hid) {
          set_error("weight shape mismatch for the project conv of the fused block %s", p.c_str());
          return fail(FDT_ERR_STATE);
        }
        DevW d3;
        auto it3 = m->W->wcache.find(p + "|expdw3");
        if (it3 == m->W->wcache.end()) {
          std::vector<float> a(w3->v);
          for (int c = 0; c < oup; ++c)
            for (int k = 0; k < hid; ++k) a[(size_t)c * hid + k] *= s3[c];     // the folding conv() applies to a 1x1 (one rounding per weight)
          if (hipMalloc((void**)&d3.w, a.size() * 4) != hipSuccess || hipMalloc((void**)&d3.bias, (size_t)oup * 4) != hipSuccess) {
            set_error("hipMalloc failed for %s", p.c_str());
            return fail(FDT_ERR_HIP);
          }
          (void)copy_sync(d3.w, a.data(), a.size() * 4, hipMemcpyHostToDevice, m->stream);
          (void)copy_sync(d3.bias, b3.data(), (size_t)oup * 4, hipMemcpyHostToDevice, m->stream);
          m->W->wcache[p + "|expdw3"] = d3;
        } else {
          d3 = it3->second;
        }
        op.w3 = d3.w;
        op.bias3 = d3.bias;
      }
    }
    m->ops.push_back(op);
    m->flops_per_frame += op.flops / B;
    return out_t;
  }

  // depthwise 3x3 (stride 1) + BN + ReLU6 and the block's 1x1 project + BN (+ residual) as ONE streaming kernel
  // (stream_ir.hip): conv[i..i+2] and conv[i+3..i+4] of an InvertedResidual, pyramid_mb2_try3.py:84-94,96-134
  int dw_project(const std::string& p, int i, int in_t, int oup, int res_t) {
    if (rc != FDT_OK) return -1;
    const Tensor in = m->tensors[in_t];
    const std::string ndw = p + ".conv." + std::to_string(i), bdw = p + ".conv." + std::to_string(i + 1),
                      npr = p + ".conv." + std::to_string(i + 3), bpr = p + ".conv." + std::to_string(i + 4);
    int out_t = new_tensor(npr, oup, in.H, in.W);
    if (out_t < 0) return -1;
    const HostT* wd = get(ndw + ".weight");
    const HostT* wp = get(npr + ".weight");
    std::vector<float> s1, b1, s2, b2;
    fold(bdw, nullptr, in.C, s1, b1);
    fold(bpr, nullptr, oup, s2, b2);
    Op op;
    op.type = OP_DWPROJ;
    op.name = p + ".dw_project";
    op.in_t = in_t;
    op.in2_t = res_t;
    op.out_t = out_t;
    op.hid = in.C;
    op.oup = oup;
    op.flops = 2.0 * B * (double)in.H * in.W * in.C * (9 + oup);
    memset(&op.ca, 0, sizeof(op.ca));
    if (!m->dry) {
      if (rc != FDT_OK) return -1;
      if ((int)wd->v.size() != in.C * 9 || (int)wp->v.size() != oup * in.C) {
        set_error("weight shape mismatch for the fused depthwise + project of %s", p.c_str());
        return fail(FDT_ERR_STATE);
      }
      std::lock_guard<std::mutex> lk(m->W->mu);
      DevW d1, d2;
      auto it = m->W->wcache.find(p + "|dwproj1");
      if (it == m->W->wcache.end()) {
        std::vector<float> a(wd->v), t((size_t)in.C * oup);
        for (int c = 0; c < in.C; ++c)
          for (int k = 0; k < 9; ++k) a[(size_t)c * 9 + k] *= s1[c];                       // the folding dwconv() applies
        for (int o = 0; o < oup; ++o)
          for (int c = 0; c < in.C; ++c) t[(size_t)c * oup + o] = wp->v[(size_t)o * in.C + c] * s2[o];   // ... and conv(); transposed
        if (hipMalloc((void**)&d1.w, a.size() * 4) != hipSuccess || hipMalloc((void**)&d1.bias, (size_t)in.C * 4) != hipSuccess ||
            hipMalloc((void**)&d2.w, t.size() * 4) != hipSuccess || hipMalloc((void**)&d2.bias, (size_t)oup * 4) != hipSuccess) {
          set_error("hipMalloc failed for %s", p.c_str());
          return fail(FDT_ERR_HIP);
        }
        (void)copy_sync(d1.w, a.data(), a.size() * 4, hipMemcpyHostToDevice, m->stream);
        (void)copy_sync(d1.bias, b1.data(), (size_t)in.C * 4, hipMemcpyHostToDevice, m->stream);
        (void)copy_sync(d2.w, t.data(), t.size() * 4, hipMemcpyHostToDevice, m->stream);
        (void)copy_sync(d2.bias, b2.data(), (size_t)oup * 4, hipMemcpyHostToDevice, m->stream);
        m->W->wcache[p + "|dwproj1"] = d1;
        m->W->wcache[p + "|dwproj2"] = d2;
      } else {
        d1 = it->second;
        d2 = m->W->wcache[p + "|dwproj2"];
      }
      op.w = d1.w;
      op.bias = d1.bias;
      op.w2 = d2.w;
      op.bias2 = d2.bias;
    }
    m->ops.push_back(op);
    m->flops_per_frame += op.flops / B;
    return out_t;
  }

  void mboxfin(int map_t, int anchors) {   // FACEBOX/multibox_layer.py:34-48
    if (rc != FDT_OK) return;
    Op op;
    op.type = OP_MBOXFIN;
    op.name = "multibox_finalize";
    op.in_t = map_t;
    op.anchors = anchors;
    memset(&op.ca, 0, sizeof(op.ca));
    m->levels.push_back({m->tensors[map_t].H * anchors, m->tensors[map_t].W});
    m->ops.push_back(op);
  }

  void headfin(int head_t, int level0) {
    if (rc != FDT_OK) return;
    Op op;
    op.type = OP_HEADFIN;
    op.name = "head_finalize";
    op.in_t = head_t;
    op.level0 = level0;
    memset(&op.ca, 0, sizeof(op.ca));
    m->levels.push_back({m->tensors[head_t].H, m->tensors[head_t].W});
    m->ops.push_back(op);
  }

  // ---------------------------------------------------------------- shared PyramidBox blocks
  int ssh(const std::string& n, int x, int xc) {   // pyramid.py:41-48
    if (rc != FDT_OK) return -1;
    const Tensor xin = m->tensors[x];
    int src = new_tensor(n, 2 * xc, xin.H, xin.W);
    if (src < 0) return -1;
    ConvOpt o;
    o.act = ACT_RELU;
    o.out_t = src;
    o.out_coff = 0;
    conv(n + ".conv1", x, xc, CONV_3x3_S1, o);
    ConvOpt t;
    t.act = ACT_RELU;
    int x2 = conv(n + ".conv2", x, xc / 2, CONV_3x3_S1_D2, t);
    o.out_coff = xc;
    conv(n + ".conv2_1", x2, xc / 2, CONV_3x3_S1, o);
    int x22 = conv(n + ".conv2_2", x2, xc / 2, CONV_3x3_S1_D2, t);
    o.out_coff = xc + xc / 2;
    conv(n + ".conv2_2_1", x22, xc / 2, CONV_3x3_S1, o);
    return src;
  }

  int ct(const std::string& n, int up, int main, int C) {   // pyramid.py:61-69
    ConvOpt o;
    int u = conv(n + ".up_conv", up, C, CONV_1x1_S1, o);
    ConvOpt mo;
    mo.up_t = u;
    return conv(n + ".main_conv", main, C, CONV_1x1_S1, mo);
  }

  void heads(const std::vector<int>& sources) {   // pyramid.py:291-309
    for (size_t i = 0; i < sources.size(); ++i) {
      ConvOpt o;
      o.name2 = "face_conf." + std::to_string(i);
      o.cout2 = 4;
      int hm = conv("face_loc." + std::to_string(i), sources[i], 4, CONV_3x3_S1, o);
      if (hm < 0) return;
      m->tensors[hm].name = "head" + std::to_string(i);
      headfin(hm, i == 0);
    }
  }

  // ---------------------------------------------------------------- Res50   pyramid.py:218-351
  void build_res50(int H, int W) {
    int x = new_tensor("input", 3, H, W);
    ConvOpt st;
    st.bn = "bn1";
    st.bias = false;
    st.act = ACT_RELU;
    int c1 = conv("conv1", x, 64, CONV_7x7_S2, st);
    if (c1 < 0) return;
    m->tensors[c1].name = "stem";
    int h = pool("pool", c1, 2, 0);
    int in_planes = 64;
    const int planes_[4] = {64, 128, 256, 512}, nblk[4] = {3, 4, 6, 3}, strd[4] = {1, 2, 2, 2};
    int feats[4];
    for (int li = 0; li < 4; ++li) {
      for (int bi = 0; bi < nblk[li]; ++bi) {
        const int stv = bi == 0 ? strd[li] : 1;
        const int planes = planes_[li];
        std::string p = "layer" + std::to_string(li + 1) + "." + std::to_string(bi);
        ConvOpt o1;
        o1.bn = p + ".bn1";
        o1.bias = false;
        o1.act = ACT_RELU;
        int a1 = conv(p + ".conv1", h, planes, CONV_1x1_S1, o1);
        ConvOpt o2;
        o2.bn = p + ".bn2";
        o2.bias = false;
        o2.act = ACT_RELU;
        int a2 = conv(p + ".conv2", a1, planes, stv == 2 ? CONV_3x3_S2 : CONV_3x3_S1, o2);
        int sc = h;
        if (stv != 1 || in_planes != planes * 4) {
          ConvOpt od;
          od.bn = p + ".downsample.1";
          od.bias = false;
          sc = conv(p + ".downsample.0", h, planes * 4, stv == 2 ? CONV_1x1_S2 : CONV_1x1_S1, od);
        }
        ConvOpt o3;
        o3.bn = p + ".bn3";
        o3.bias = false;
        o3.act = ACT_RELU;
        o3.res_t = sc;
        h = conv(p + ".conv3", a2, planes * 4, CONV_1x1_S1, o3);
        if (rc != FDT_OK) return;
        in_planes = planes * 4;
      }
      feats[li] = h;
      m->tensors[h].name = "c" + std::to_string(li + 2);
    }
    const int c2 = feats[0], c3 = feats[1], c4 = feats[2], c5 = feats[3];
    ConvOpt a;
    a.act = ACT_RELU;
    a.bn = "layer5.1";
    int c6a = conv("layer5.0", c5, 512, CONV_1x1_S1, a);
    a.bn = "layer5.4";
    int c6 = conv("layer5.3", c6a, 512, CONV_3x3_S2, a);
    a.bn = "layer6.1";
    int c7a = conv("layer6.0", c6, 128, CONV_1x1_S1, a);
    a.bn = "layer6.4";
    int c7 = conv("layer6.3", c7a, 256, CONV_3x3_S2, a);
    if (rc != FDT_OK) return;
    m->tensors[c6].name = "c6";
    m->tensors[c7].name = "c7";
    ConvOpt lin;
    int c5_lat = conv("latlayer_fc", c5, 2048, CONV_1x1_S1, lin);
    int c6_lat = conv("latlayer_c6", c6, 512, CONV_1x1_S1, lin);
    int c7_lat = conv("latlayer_c7", c7, 256, CONV_1x1_S1, lin);
    int c4_f = ct("conv5_ct_py", c5_lat, c4, 1024);
    int c3_f = ct("conv4_ct_py", c4_f, c3, 512);
    int c2_f = ct("conv3_ct_py", c3_f, c2, 256);
    if (rc != FDT_OK) return;
    m->tensors[c4_f].name = "c4_ct";
    m->tensors[c3_f].name = "c3_ct";
    m->tensors[c2_f].name = "c2_ct";
    int c2_s = conv("smooth_c3", c2_f, 256, CONV_3x3_S1, lin);
    int c3_s = conv("smooth_c4", c3_f, 512, CONV_3x3_S1, lin);
    int c4_s = conv("smooth_c5", c4_f, 1024, CONV_3x3_S1, lin);
    if (rc != FDT_OK) return;
    m->tensors[c2_s].name = "c2_smooth";
    m->tensors[c3_s].name = "c3_smooth";
    m->tensors[c4_s].name = "c4_smooth";
    std::vector<int> src;
    src.push_back(ssh("conv2_SSH", c2_s, 256));
    src.push_back(ssh("conv3_SSH", c3_s, 256));
    src.push_back(ssh("conv4_SSH", c4_s, 256));
    src.push_back(ssh("conv5_SSH", c5_lat, 256));
    src.push_back(ssh("conv6_SSH", c6_lat, 256));
    src.push_back(ssh("conv7_SSH", c7_lat, 256));
    if (rc != FDT_OK) return;
    for (size_t i = 0; i < src.size(); ++i) m->tensors[src[i]].name = "src" + std::to_string(i);
    heads(src);
  }

  // ---------------------------------------------------------------- try3   pyramid_mb2_try3.py:218-340
  int inverted_residual(const std::string& p, int x, int inp, int oup, int stride, int t) {
    int h = x;
    int i = 0;
    const int hid = (int)std::lround((double)inp * t);
    // The expanded tensor is t times the input: on the large maps writing and re-reading it IS the block's cost, so the
    // expand and the depthwise conv run as one kernel there (measured on try3 at batch 8: 319 vs 473 us for features.2,
    // 140 vs 189 us for features.4; from 128^2 down the two separate launches are faster).  FDT_FUSE_IR=0|1 forces it.
    const Tensor xin = m->tensors[x];
    bool fuse = t != 1 && (inp & 1) == 0 && (stride == 1 || stride == 2) && (long long)xin.H * xin.W >= 256ll * 256;
    if (const char* e = getenv("FDT_FUSE_IR")) fuse = t != 1 && (inp & 1) == 0 && atoi(e) != 0;
    if (fuse && expand_dw_lds_bytes(inp, stride, hid) > 80 * 1024) fuse = false;   // keep two workgroups per CU
    // round 4: the 1x1 project + BN (+ residual) can run in the same kernel when the block has at most 32 output channels
    // (fused_ir.hip: PROJ; one launch per block, bit-identical).  Measured on try3 at batch 8 it is 2 % SLOWER end to end
    // (2001-2018 vs 2052-2060 frames/s: the kernel is bound by its three barriers per chunk, and the project phase adds a
    // dependent 16-MFMA chain inside them, while the launch it saves was overlapped anyway; docs/EXPERIMENTS.md R4-9) --
    // so it is opt-in: FDT_FUSE_IR=2 (0 none, 1 expand + depthwise wherever it fits, unset: that on the large maps).
    const int fuse_env = getenv("FDT_FUSE_IR") ? atoi(getenv("FDT_FUSE_IR")) : -1;
    if (fuse && oup <= 32 && fuse_env == 2 && ir_block_lds_bytes(inp, stride, hid) <= 80 * 1024)
      return expand_dw(p, h, hid, stride, oup, stride == 1 && inp == oup);
    if (fuse) {
      h = expand_dw(p, h, hid, stride);
      i += 6;
    } else {
      if (t != 1) {
        ConvOpt o;
        o.bias = false;
        o.bn = p + ".conv." + std::to_string(i + 1);
        o.act = ACT_RELU6;
        h = conv(p + ".conv." + std::to_string(i), h, hid, CONV_1x1_S1, o);
        i += 3;
      }
      // On the large maps the depthwise output (as big as its input) is the block's traffic: depthwise + project as one
      // streaming kernel (stream_ir.hip) -- features.1 of try3 at batch 8: 235 us as two launches.  Below 256^2 the grid of
      // 4-pixel strips no longer fills the chip and the two launches stay.
      const Tensor hin = m->tensors[h];
      if (m->stream_ir && stride == 1 && dw_project_supported(hid, hin.H, hin.W, oup) && (long long)hin.H * hin.W >= 256ll * 256)
        return dw_project(p, i, h, oup, inp == oup ? x : -1);
      h = dwconv(p + ".conv." + std::to_string(i), p + ".conv." + std::to_string(i + 1), h, stride, ACT_RELU6);
      i += 3;
    }
    ConvOpt o;
    o.bias = false;
    o.bn = p + ".conv." + std::to_string(i + 1);
    if (stride == 1 && inp == oup) o.res_t = x;   // :131-134
    return conv(p + ".conv." + std::to_string(i), h, oup, CONV_1x1_S1, o);
  }

  // try3 (pyramid_mb2_try3.py), and its two siblings that differ only in the stem and the smooth layers:
  //   try5 (pyramid_mb2_try5.py:184-191): smooth_c2/3/4 = InvertedResidual(c, c, 1, t) -> 3x3, smooth_c6 = 1x1 pad 1
  //   try4 (pyramid_mb2_try4.py:16,184-191): try5 + 7x7/pad-1 stem + smooth_c5 = 1x1 pad 1
  void build_try3(int H, int W, int variant = 3) {
    int x = new_tensor("input", 3, H, W);
    ConvOpt st;
    st.bn = "features.0.1";
    st.bias = false;
    st.act = ACT_RELU6;
    int h = conv("features.0.0", x, 32, variant == 4 ? CONV_7x7_S2_P1 : CONV_3x3_S2, st);
    if (h < 0) return;
    m->tensors[h].name = "stem";
    const int cfgs[7][4] = {{1, 16, 1, 1}, {6, 24, 2, 2}, {6, 32, 3, 2}, {6, 64, 4, 2},
                            {6, 96, 3, 1}, {6, 160, 3, 2}, {6, 320, 1, 1}};
    int inp = 32, idx = 1;
    std::map<int, int> taps;
    for (auto& c : cfgs) {
      for (int i = 0; i < c[2]; ++i) {
        h = inverted_residual("features." + std::to_string(idx), h, inp, c[1], i == 0 ? c[3] : 1, c[0]);
        if (rc != FDT_OK) return;
        taps[idx] = h;
        inp = c[1];
        ++idx;
      }
    }
    int c2 = taps[3], c3 = taps[6], c4 = taps[13], c5 = taps[17];   // :229-236
    int c6 = inverted_residual("layer6", c5, 320, 160, 2, 6);       // :238
    if (rc != FDT_OK) return;
    m->tensors[c2].name = "c2";
    m->tensors[c3].name = "c3";
    m->tensors[c4].name = "c4";
    m->tensors[c5].name = "c5";
    m->tensors[c6].name = "c6";
    ConvOpt lin;
    if (variant == 3) {
      c6 = conv("smooth_c6", c6, 160, CONV_3x3_S1, lin);   // :242-243
    } else {
      c6 = conv("smooth_c6", pad1("smooth_c6.pad", c6), 160, CONV_1x1_S1, lin);   // Conv2d(160,160,1,padding=1)
    }
    if (variant == 4) {
      c5 = conv("smooth_c5", pad1("smooth_c5.pad", c5), 320, CONV_1x1_S1, lin);   // Conv2d(320,320,1,padding=1)
    } else {
      c5 = conv("smooth_c5", c5, 320, CONV_3x3_S1, lin);
    }
    c4 = ct("conv4_ct_py", c5, c4, 96);                  // :245-247
    c3 = ct("conv3_ct_py", c4, c3, 32);
    c2 = ct("conv2_ct_py", c3, c2, 24);
    if (variant == 3) {
      c2 = conv("smooth_c2", c2, 24, CONV_3x3_S1, lin);    // :249-251
      c3 = conv("smooth_c3", c3, 32, CONV_3x3_S1, lin);
      c4 = conv("smooth_c4", c4, 96, CONV_3x3_S1, lin);
    } else {   // nn.Sequential(InvertedResidual(c, c, 1, t), nn.Conv2d(c, c, 3, padding=1))
      c2 = conv("smooth_c2.1", inverted_residual("smooth_c2.0", c2, 24, 24, 1, 4), 24, CONV_3x3_S1, lin);
      c3 = conv("smooth_c3.1", inverted_residual("smooth_c3.0", c3, 32, 32, 1, 4), 32, CONV_3x3_S1, lin);
      c4 = conv("smooth_c4.1", inverted_residual("smooth_c4.0", c4, 96, 96, 1, 2), 96, CONV_3x3_S1, lin);
    }
    if (rc != FDT_OK) return;
    m->tensors[c2].name = "c2_smooth";
    m->tensors[c3].name = "c3_smooth";
    m->tensors[c4].name = "c4_smooth";
    m->tensors[c5].name = "c5_smooth";
    m->tensors[c6].name = "c6_smooth";
    std::vector<int> src;
    src.push_back(ssh("conv2_SSH", c2, 128));
    src.push_back(ssh("conv3_SSH", c3, 128));
    src.push_back(ssh("conv4_SSH", c4, 128));
    src.push_back(ssh("conv5_SSH", c5, 128));
    src.push_back(ssh("conv6_SSH", c6, 128));
    if (rc != FDT_OK) return;
    for (size_t i = 0; i < src.size(); ++i) m->tensors[src[i]].name = "src" + std::to_string(i);
    heads(src);   // zip() truncates to the 5 sources (:288): face_*.5 are dead weights
  }

  // ---------------------------------------------------------------- try1 / try2   pyramid_mobile_try{1,2}.py
  // Mobilenetv2(inp, oup, k, stride, t, padding, dilation, side_way[, bias]) (:103-134): 1x1 expand + BN + ReLU6,
  // depthwise kxk + BN + ReLU6, 1x1 project + BN, optional identity add.
  int mbv2(const std::string& p, int x, int inp, int oup, int k, int stride, int t, int pad, int dil, bool side_way,
           bool dw_bias = false) {
    ConvOpt o1;
    o1.bias = false;
    o1.bn = p + ".bn1";
    o1.act = ACT_RELU6;
    int h = conv(p + ".conv1", x, inp * t, CONV_1x1_S1, o1);
    h = dwconv(p + ".conv2", p + ".bn2", h, stride, ACT_RELU6, k, pad, dil, dw_bias);
    ConvOpt o3;
    o3.bias = false;
    o3.bn = p + ".bn3";
    if (side_way) o3.res_t = x;
    return conv(p + ".conv3", h, oup, CONV_1x1_S1, o3);
  }
  // Mobilenetv1(cin, cout, k, stride, padding, dilation[, bias]) (:84-99): depthwise + BN + ReLU, 1x1 (no bias)
  int mbv1(const std::string& p, int x, int cout, int k, int stride, int pad, int dil, bool dw_bias,
           const ConvOpt& tail) {
    int h = dwconv(p + ".conv1", p + ".bn", x, stride, ACT_RELU, k, pad, dil, dw_bias);
    ConvOpt o = tail;
    o.bias = false;
    return conv(p + ".conv2", h, cout, CONV_1x1_S1, o);
  }

  void build_try12(int H, int W, int variant) {
    int x = new_tensor("input", 3, H, W);
    ConvOpt st;   // c1 = F.relu(self.bn1(self.conv1_my(x)))  (:232)
    st.bn = "bn1";
    st.act = ACT_RELU;
    int c1 = mbv1("conv1_my", x, 64, 7, 2, 3, 1, false, st);
    if (c1 < 0) return;
    m->tensors[c1].name = "stem";
    int h = pool("pool", c1, 2, 0);
    struct Blk { int inp, oup, k, stride, t, pad, dil, side; };
    std::vector<std::vector<Blk>> layers;
    if (variant == 1) {   // pyramid_mobile_try1.py:160-181
      layers = {{{64, 64, 3, 1, 2, 1, 1, 1}, {64, 64, 3, 1, 2, 1, 1, 1}, {64, 256, 3, 1, 2, 1, 1, 0}},
                {{256, 64, 5, 2, 2, 2, 1, 0}, {64, 512, 3, 1, 2, 2, 2, 0}},
                {{512, 256, 5, 2, 2, 2, 1, 0}, {256, 256, 5, 1, 2, 2, 1, 1}, {256, 1024, 3, 1, 2, 2, 2, 0}},
                {{1024, 256, 5, 2, 2, 2, 1, 0}, {256, 2048, 3, 1, 2, 1, 1, 0}}};
    } else {              // pyramid_mobile_try2.py:163-189 (t = 4 by default, 2 in layer3)
      layers = {{{64, 64, 3, 1, 4, 1, 1, 1}, {64, 64, 3, 1, 4, 1, 1, 1}, {64, 64, 3, 1, 4, 1, 1, 1}},
                {{64, 64, 3, 2, 4, 1, 1, 0}, {64, 64, 3, 1, 4, 1, 1, 1}, {64, 64, 3, 1, 4, 1, 1, 1},
                 {64, 128, 3, 1, 4, 1, 1, 0}},
                {{128, 128, 3, 2, 2, 1, 1, 0}, {128, 128, 3, 1, 2, 1, 1, 1}, {128, 128, 3, 1, 2, 1, 1, 1},
                 {128, 128, 3, 1, 2, 1, 1, 1}, {128, 128, 3, 1, 2, 1, 1, 1}, {128, 256, 3, 1, 2, 1, 1, 0}},
                {{256, 256, 3, 2, 4, 1, 1, 0}, {256, 256, 3, 1, 4, 1, 1, 1}, {256, 512, 3, 1, 4, 1, 1, 0}}};
    }
    int feats[4];
    for (int li = 0; li < 4; ++li) {
      for (size_t bi = 0; bi < layers[li].size(); ++bi) {
        const Blk& b = layers[li][bi];
        h = mbv2("layer" + std::to_string(li + 1) + "_my." + std::to_string(bi), h, b.inp, b.oup, b.k, b.stride, b.t,
                 b.pad, b.dil, b.side != 0);
        if (rc != FDT_OK) return;
      }
      feats[li] = h;
    }
    const int t56 = variant == 1 ? 2 : 4;
    const int c5_in = variant == 1 ? 2048 : 512;
    int c6 = mbv2("layer5_my", feats[3], c5_in, 512, 3, 2, t56, 1, 1, false, variant == 2);   // :182 / try2 :191
    int c7 = mbv2("layer6_my", c6, 512, 256, 3, 2, t56, 1, 1, false, variant == 2);
    if (rc != FDT_OK) return;
    if (variant == 2) {   // layerN_adj: Conv2d(c, 4c', 1, bias=False) applied after the whole backbone (try2 :255-258)
      const int adj[4] = {256, 512, 1024, 2048};
      ConvOpt nb;
      nb.bias = false;
      for (int li = 0; li < 4; ++li) feats[li] = conv("layer" + std::to_string(li + 1) + "_adj", feats[li], adj[li],
                                                      CONV_1x1_S1, nb);
      if (rc != FDT_OK) return;
    }
    const int c2 = feats[0], c3 = feats[1], c4 = feats[2], c5 = feats[3];
    m->tensors[c2].name = "c2";
    m->tensors[c3].name = "c3";
    m->tensors[c4].name = "c4";
    m->tensors[c5].name = "c5";
    m->tensors[c6].name = "c6";
    m->tensors[c7].name = "c7";
    ConvOpt lin;
    lin.groups = 4;
    int c5_lat = conv("latlayer_fc_my", c5, 2048, CONV_1x1_S1, lin);   // groups=4 (:185)
    lin.groups = 2;
    int c6_lat = conv("latlayer_c6_my", c6, 512, CONV_1x1_S1, lin);    // groups=2
    lin.groups = 1;
    int c7_lat = conv("latlayer_c7_my", c7, 256, CONV_1x1_S1, lin);
    int c4_f = ct("conv5_ct_py", c5_lat, c4, 1024);
    int c3_f = ct("conv4_ct_py", c4_f, c3, 512);
    int c2_f = ct("conv3_ct_py", c3_f, c2, 256);
    if (rc != FDT_OK) return;
    m->tensors[c4_f].name = "c4_ct";
    m->tensors[c3_f].name = "c3_ct";
    m->tensors[c2_f].name = "c2_ct";
    ConvOpt none;
    int c2_s = mbv1("smooth_c3_my", c2_f, 256, 3, 1, 1, 1, variant == 2, none);
    int c3_s = mbv1("smooth_c4_my", c3_f, 512, 3, 1, 1, 1, variant == 2, none);
    int c4_s = mbv1("smooth_c5_my", c4_f, 1024, 3, 1, 1, 1, variant == 2, none);
    if (rc != FDT_OK) return;
    m->tensors[c2_s].name = "c2_smooth";
    m->tensors[c3_s].name = "c3_smooth";
    m->tensors[c4_s].name = "c4_smooth";
    std::vector<int> src;
    src.push_back(ssh("conv2_SSH", c2_s, 256));
    src.push_back(ssh("conv3_SSH", c3_s, 256));
    src.push_back(ssh("conv4_SSH", c4_s, 256));
    src.push_back(ssh("conv5_SSH", c5_lat, 256));
    src.push_back(ssh("conv6_SSH", c6_lat, 256));
    src.push_back(ssh("conv7_SSH", c7_lat, 256));
    if (rc != FDT_OK) return;
    for (size_t i = 0; i < src.size(); ++i) m->tensors[src[i]].name = "src" + std::to_string(i);
    heads(src);
  }

  // ---------------------------------------------------------------- FaceBox   FACEBOX/networks.py:87-116
  int cbr(const std::string& n, int x, int cout, ConvKind kind, int out_t = -1, int coff = 0) {
    ConvOpt o;   // conv_bn_relu(): Conv2d(bias) -> BatchNorm2d -> ReLU   (networks.py:11-16)
    o.bn = n + ".1";
    o.act = ACT_RELU;
    o.out_t = out_t;
    o.out_coff = coff;
    return conv(n + ".0", x, cout, kind, o);
  }

  // Inception (networks.py:43-57).  The three 1x1 branches that read x itself -- conv1 (-> out[0:32]), conv3 and conv5 (24 channels
  // each, the inputs of the two 3x3 branches) -- are ONE launch with two destinations (ConvArgs.out2): 80 output channels over
  // the same staged input instead of three passes over it; conv4 (on conv3's output, -> out[64:96]) and conv6 (on conv5's) are
  // one block-diagonal 3x3 launch over the 48-channel pair.  Five launches per block instead of eight, the same sums per
  // output channel (a part's weights are zero over the other part's input channels: exact +0.0f).  FDT_FB_FUSE=0 / 1 (create
  // time): the un-fused form / only the 1x1 launch, for A/B runs and the bit-equality test.
  int inception(const std::string& n, int x) {
    if (rc != FDT_OK) return -1;
    const Tensor xin = m->tensors[x];
    int out = new_tensor(n, 128, xin.H, xin.W);
    if (out < 0) return -1;
    const int fuse = m->fb_fuse;
    if (fuse == 0) {
      cbr(n + ".conv1", x, 32, CONV_1x1_S1, out, 0);
      int xp = pool(n + ".pool", x, 1, 0);
      cbr(n + ".conv2", xp, 32, CONV_1x1_S1, out, 32);
      int t3 = cbr(n + ".conv3", x, 24, CONV_1x1_S1);
      cbr(n + ".conv4", t3, 32, CONV_3x3_S1, out, 64);
      int t5 = cbr(n + ".conv5", x, 24, CONV_1x1_S1);
      int t6 = cbr(n + ".conv6", t5, 32, CONV_3x3_S1);
      cbr(n + ".conv7", t6, 32, CONV_3x3_S1, out, 96);
      return out;
    }
    int t35 = new_tensor(n + ".conv3_5", 48, xin.H, xin.W);     // [conv3 | conv5]
    if (t35 < 0) return -1;
    {
      ConvOpt o;                                                  // conv1 | conv3 | conv5 on x
      o.bn = n + ".conv1.1";
      o.act = ACT_RELU;
      o.out_t = out;
      o.out_coff = 0;
      o.more.push_back({n + ".conv3.0", n + ".conv3.1", 24, 0, 0});
      o.more.push_back({n + ".conv5.0", n + ".conv5.1", 24, 0, 0});
      o.out2_t = t35;
      o.out2_from = 32;
      o.out2_coff = 0;
      conv(n + ".conv1.0", x, 32, CONV_1x1_S1, o);
    }
    int xp = pool(n + ".pool", x, 1, 0);
    cbr(n + ".conv2", xp, 32, CONV_1x1_S1, out, 32);
    int t6 = new_tensor(n + ".conv6.0", 32, xin.H, xin.W);
    if (t6 < 0) return -1;
    if (fuse >= 2) {
      ConvOpt o;                                                  // conv4 on t35[0:24] | conv6 on t35[24:48]
      o.bn = n + ".conv4.1";
      o.act = ACT_RELU;
      o.out_t = out;
      o.out_coff = 64;
      o.w_in_off = 0;
      o.w_in_c = 24;
      o.more.push_back({n + ".conv6.0", n + ".conv6.1", 32, 24, 24});
      o.out2_t = t6;
      o.out2_from = 32;
      o.out2_coff = 0;
      conv(n + ".conv4.0", t35, 32, CONV_3x3_S1, o);
    } else {
      ConvOpt o4;
      o4.bn = n + ".conv4.1";
      o4.act = ACT_RELU;
      o4.out_t = out;
      o4.out_coff = 64;
      o4.in_coff = 0;
      o4.in_c = 24;
      conv(n + ".conv4.0", t35, 32, CONV_3x3_S1, o4);
      ConvOpt o6;
      o6.bn = n + ".conv6.1";
      o6.act = ACT_RELU;
      o6.out_t = t6;
      o6.in_coff = 24;
      o6.in_c = 24;
      conv(n + ".conv6.0", t35, 32, CONV_3x3_S1, o6);
    }
    cbr(n + ".conv7", t6, 32, CONV_3x3_S1, out, 96);
    return out;
  }

  void build_facebox(int H, int W) {
    int x = new_tensor("input", 3, H, W);
    ConvOpt o1;
    o1.bn = "bn1";
    int c1 = conv("conv1", x, 24, CONV_7x7_S4, o1);       // :89-90
    int p1 = pool("crelu_pool1", c1, 2, 1);               // :91-93  CReLU + max_pool2d(3,2,1)
    ConvOpt o2;
    o2.bn = "bn2";
    int c2 = conv("conv2", p1, 64, CONV_5x5_S2, o2);      // :94-95
    int p2 = pool("crelu_pool2", c2, 2, 1);               // :96-98
    if (rc != FDT_OK) return;
    int h = inception("inception1", p2);
    h = inception("inception2", h);
    h = inception("inception3", h);                        // :99-101
    if (rc != FDT_OK) return;
    m->tensors[h].name = "hs0";
    int c31 = cbr("conv3_1", h, 128, CONV_1x1_S1);
    int hs1 = cbr("conv3_2", c31, 256, CONV_3x3_S2);      // :105-106
    int c41 = cbr("conv4_1", hs1, 128, CONV_1x1_S1);
    int hs2 = cbr("conv4_2", c41, 256, CONV_3x3_S2);      // :109-110
    if (rc != FDT_OK) return;
    m->tensors[hs1].name = "hs1";
    m->tensors[hs2].name = "hs2";
    const int srcs[3] = {h, hs1, hs2};
    const int anchors[3] = {21, 1, 1};                     // multibox_layer.py:14
    for (int i = 0; i < 3; ++i) {
      ConvOpt o;
      o.name2 = "multilbox.conf_layers." + std::to_string(i);
      o.cout2 = anchors[i] * 2;
      int mp = conv("multilbox.loc_layers." + std::to_string(i), srcs[i], anchors[i] * 4, CONV_3x3_S1, o);
      if (mp < 0) return;
      m->tensors[mp].name = "mbox" + std::to_string(i);
      mboxfin(mp, anchors[i]);
    }
  }
};

bool ignored_key(const fdt_model* m, const std::string& k) {
  auto ends_with = [&](const char* s) {
    size_t n = strlen(s);
    return k.size() >= n && k.compare(k.size() - n, n, s) == 0;
  };
  if (ends_with("num_batches_tracked")) return true;
  if (m->arch != FDT_ARCH_FACEBOX) {
    if (k.rfind("head_loc.", 0) == 0 || k.rfind("head_conf.", 0) == 0) return true;   // pyramid.py:312-317
    if ((m->arch == FDT_ARCH_TRY3 || m->arch == FDT_ARCH_TRY4 || m->arch == FDT_ARCH_TRY5) &&
        (k.rfind("face_loc.5.", 0) == 0 || k.rfind("face_conf.5.", 0) == 0))
      return true;
  }
  return false;
}

int build_graph(fdt_model* m, int B, int H, int W) {
  Builder bld{m, B};
  m->flops_per_frame = 0;
  if (m->arch == FDT_ARCH_RES50)
    bld.build_res50(H, W);
  else if (m->arch == FDT_ARCH_TRY3)
    bld.build_try3(H, W);
  else if (m->arch == FDT_ARCH_TRY4)
    bld.build_try3(H, W, 4);
  else if (m->arch == FDT_ARCH_TRY5)
    bld.build_try3(H, W, 5);
  else if (m->arch == FDT_ARCH_TRY1)
    bld.build_try12(H, W, 1);
  else if (m->arch == FDT_ARCH_TRY2)
    bld.build_try12(H, W, 2);
  else if (m->arch == FDT_ARCH_FACEBOX)
    bld.build_facebox(H, W);
  else {
    set_error("unknown arch %d", m->arch);
    return FDT_ERR_ARG;
  }
  return bld.rc;
}

// priors (net.priorbox(idx, f_w, f_h) per source, pyramid.py:275-283)
int make_priors(fdt_model* m, int H, int W) {
  if (m->arch == FDT_ARCH_FACEBOX) {   // anchors are fixed: DataEncoder.__init__ (encoderl.py:12-48)
    FDT_REQUIRE(m->P == 21824, FDT_ERR_ARG, "FaceBox needs 1024x1024 input (21824 anchors), got %d priors", m->P);
    FDT_TRY(launch_facebox_anchors(m->d_priors, m->stream));
    m->priors_dirty = false;
    return FDT_OK;
  }
  const int nl = (int)m->levels.size();
  std::vector<int> stride = m->pb_stride, box = m->pb_box;
  int pw = m->pb_set ? m->pb_w : W, ph = m->pb_set ? m->pb_h : H;
  if (!m->pb_set) {
    // module default: PriorBoxLayer(size=640, size) (pyramid.py:113) -- callers override it with the
    // frame size; a forward at another size with the default object keeps 640 like the reference.
    pw = ph = 640;
    stride.clear();
    box.clear();
    for (int i = 0; i < nl; ++i) {
      stride.push_back(4 << i);
      box.push_back(16 << i);
    }
  }
  FDT_REQUIRE((int)stride.size() >= nl && (int)box.size() >= nl, FDT_ERR_STATE,
              "priorbox has %d levels, the net has %d sources", (int)stride.size(), nl);
  int off = 0;
  for (int i = 0; i < nl; ++i) {
    FDT_TRY(launch_priorbox(pw, ph, stride[i], box[i], 1, nullptr, 0, m->levels[i].second,
                            m->levels[i].first, m->d_priors + (size_t)off * 4, m->stream));
    off += m->levels[i].first * m->levels[i].second;
  }
  m->priors_dirty = false;
  return FDT_OK;
}

// (Re)build the execution plan for a batch shape.
int setup_heads(fdt_model* m);
int plan_reduces(fdt_model* m, int B);

int make_plan(fdt_model* m, int B, int H, int W) {
  if (m->pB == B && m->pH == H && m->pW == W && !m->ops.empty()) {
    if (m->priors_dirty) {
      FDT_HIP(fdt::device_sync());
      FDT_TRY(make_priors(m, H, W));
    }
    return FDT_OK;
  }
  FDT_HIP(hipStreamSynchronize(m->stream));
  m->free_plan();
  m->dry = false;
  FDT_HIP(hipGetLastError());      // a stale error of an earlier, unchecked runtime call would otherwise surface at the first launch check below
  int rc = build_graph(m, B, H, W);
  if (rc != FDT_OK) {
    m->free_plan();
    return rc;
  }
  FDT_HIP(hipGetLastError());      // ... and one left behind by the plan's own allocations / uploads
  // detection levels -> prior offsets
  int P = 0;
  size_t li = 0;
  for (auto& op : m->ops)
    if (op.type == OP_HEADFIN || op.type == OP_MBOXFIN) {
      op.p_off = P;
      P += m->levels[li].first * m->levels[li].second;
      ++li;
    }
  m->P = P;
  auto dalloc = [&](void** p, size_t bytes) -> int {
    FDT_HIP(hipMalloc(p, bytes));
    m->plan_allocs.push_back(*p);
    return FDT_OK;
  };
  FDT_TRY(dalloc((void**)&m->d_loc, (size_t)B * P * 16));
  FDT_TRY(dalloc((void**)&m->d_conf, (size_t)B * P * 8));
  FDT_TRY(dalloc((void**)&m->d_logits, (size_t)B * P * 8));
  FDT_TRY(dalloc((void**)&m->d_priors, (size_t)P * 16));
  FDT_TRY(dalloc((void**)&m->d_out, (size_t)B * 2 * m->top_k * 5 * 4));
  FDT_TRY(dalloc((void**)&m->d_counts, (size_t)B * 2 * 4));
  FDT_TRY(dalloc((void**)&m->d_frames_u8, (size_t)B * H * W * 3));
  m->dplan = make_detect_plan(B, P, m->arch == FDT_ARCH_FACEBOX ? P : m->nms_top_k);
  if (m->arch == FDT_ARCH_FACEBOX) {
    FDT_TRY(dalloc((void**)&m->d_fb_boxes, (size_t)B * P * 16));
    FDT_TRY(dalloc((void**)&m->d_fb_probs, (size_t)B * P * 4));
  }
  FDT_TRY(dalloc(&m->d_ws, m->dplan.bytes));
  m->ws_floats = 0;   // the split-K workspace is sized and handed out by plan_reduces() below (one allocation per plan)
  if (m->sk_counters) {
    FDT_TRY(dalloc((void**)&m->d_skcnt, (size_t)m->sk_counters * sizeof(unsigned)));
    FDT_HIP(hipMemsetAsync(m->d_skcnt, 0, (size_t)m->sk_counters * sizeof(unsigned), m->stream));
    for (auto& op : m->ops)
      if (op.type == OP_CONV && op.combine) op.ca.sk_count = m->d_skcnt;
  }
  FDT_TRY(setup_heads(m));
  FDT_HIP(hipGetLastError());
  FDT_TRY(plan_reduces(m, B));
  FDT_HIP(hipGetLastError());
  FDT_TRY(make_priors(m, H, W));
  m->pB = B;
  m->pH = H;
  m->pW = W;
  // profiling events (the segment pair survives a re-plan: it is not tied to the op list)
  for (auto e : m->ev) (void)hipEventDestroy(e);
  m->ev.clear();
  m->seg_first = m->seg_last = -1;
  return FDT_OK;
}

// profiling events: ev[i] before op i, ev[n] after the last op, ev[n+1] after Detect, ev[n+2] before the ingest kernel
int ensure_profile_events(fdt_model* m) {
  if (m->ev.size() != m->ops.size() + 3) {
    for (auto e : m->ev) (void)hipEventDestroy(e);
    m->ev.assign(m->ops.size() + 3, nullptr);
    for (auto& e : m->ev) FDT_HIP(hipEventCreate(&e));
  }
  return FDT_OK;
}

// Experiment hook (tools/experiments/deletion.sh), compiled in only with -DFDT_EXPERIMENTS (make EXTRA=-DFDT_EXPERIMENTS; the
// product library has no way to drop ops): FDT_SKIP_OPS="prefix,prefix" leaves the ops whose name starts with one
// of the prefixes out of the launch sequence after the handle's first pass (so that everything downstream, Detect's
// data-dependent NMS included, keeps reading plausible maps) -- what is a layer group worth to the multi-stream step?
// Results are wrong by construction.
#ifdef FDT_EXPERIMENTS
bool op_skipped(const std::string& name) {
  static const std::vector<std::string> prefixes = [] {
    std::vector<std::string> v;
    if (const char* e = getenv("FDT_SKIP_OPS")) {
      std::string s(e), cur;
      for (char c : s + ",") {
        if (c == ',') {
          if (!cur.empty()) v.push_back(cur);
          cur.clear();
        } else {
          cur += c;
        }
      }
    }
    return v;
  }();
  for (const auto& p : prefixes)
    if (name.compare(0, p.size(), p) == 0) return true;
  return false;
}
#else
static inline bool op_skipped(const std::string&) { return false; }
#endif

// The grouped head finalize: every OP_HEADFIN follows its head conv; the LAST one launches one kernel for all levels.  A head
// conv that is split along K leaves its slabs in a region of its own (they must survive the other heads) and skips its reduce
// pass -- the finalize kernel sums them.  Called when the plan is made and again after an autotune (the splits may change).
int setup_heads(fdt_model* m) {
  HeadFinArgs& h = m->headfin;
  memset(&h, 0, sizeof(h));
  long long need = 0;
  for (size_t i = 0; i < m->ops.size(); ++i) {
    if (m->ops[i].type != OP_HEADFIN && m->ops[i].type != OP_MBOXFIN) continue;
    FDT_REQUIRE(i > 0 && m->ops[i - 1].type == OP_CONV && m->ops[i - 1].out_t == m->ops[i].in_t && h.nlev < 8, FDT_ERR_STATE,
                "head finalize without its head conv");
    const Op& c = m->ops[i - 1];
    if (c.ca.ksplit > 1) need += conv_ws_floats(c.ca);
    ++h.nlev;
  }
  if (!h.nlev) return FDT_OK;
  if (need > m->headws_floats) {
    float* p = nullptr;
    FDT_HIP(hipMalloc((void**)&p, (size_t)need * 4));
    m->plan_allocs.push_back(p);
    m->d_headws = p;
    m->headws_floats = need;
  }
  long long off = 0;
  int l = 0, blk = 0;
  for (size_t i = 0; i < m->ops.size(); ++i) {
    if (m->ops[i].type != OP_HEADFIN && m->ops[i].type != OP_MBOXFIN) continue;
    Op& c = m->ops[i - 1];
    const Tensor& t = m->tensors[m->ops[i].in_t];
    // the grouped finalize reads every level with ONE kernel chosen from the first level: all levels must be of that kind,
    // and a head tensor has exactly the channels the kernel strides over (8 = loc + conf of PyramidBox, 6 per anchor of
    // the FaceBoxes multibox layer)
    const int anchors_i = m->ops[i].type == OP_MBOXFIN ? m->ops[i].anchors : 0;
    FDT_REQUIRE(t.C == (anchors_i ? anchors_i * 6 : 8), FDT_ERR_STATE, "head tensor of %s has %d channels, the finalize kernel expects %d",
                c.name.c_str(), t.C, anchors_i ? anchors_i * 6 : 8);
    FDT_REQUIRE(l == 0 || (h.lv[0].anchors != 0) == (anchors_i != 0), FDT_ERR_STATE,
                "head levels of different kinds (PyramidBox max-in-out / FaceBoxes multibox) in one graph");
    HeadLevel& L = h.lv[l++];
    L.HW = t.H * t.W;
    L.level0 = m->ops[i].level0;
    L.p_off = m->ops[i].p_off;
    L.blk0 = blk;
    L.anchors = m->ops[i].type == OP_MBOXFIN ? m->ops[i].anchors : 0;
    blk += ceil_div(L.HW * (L.anchors ? L.anchors : 1), 256);
    L.ksplit = c.ca.ksplit;
    c.combine = false;
    c.head = true;
    c.ca.sk_count = nullptr;
    if (c.ca.ksplit > 1) {
      c.ca.defer_reduce = 1;
      c.ca.ws = m->d_headws + off;
      off += conv_ws_floats(c.ca);
      L.src = c.ca.ws;
      L.bias = c.ca.bias;
    } else {
      c.ca.defer_reduce = 0;
      L.src = t.d;
      L.bias = nullptr;
    }
    m->ops[i].last_head = l == h.nlev;
  }
  h.nblocks = blk;
  h.P = m->P;
  h.loc = m->d_loc;
  h.conf = m->d_conf;
  h.logits = m->d_logits;
  return FDT_OK;
}

// Lazy reduce passes.  A split-K conv leaves its slabs in a region of the workspace of its own and its reduce pass (bias,
// upsample-add, residual, activation, the write into its slice of the output tensor) waits until the first later op that touches
// what it writes or rewrites what it reads; everything pending at that point runs as ONE launch (launch_reduce_group).  A
// dependent launch costs ~4.5 us of the multi-stream step whatever it does (docs/EXPERIMENTS.md R3-4), and the three branch
// outputs of every SSH module are read by nobody before the heads: a Res50 frame at 1024^2 makes 48 -> 3x fewer reduce launches.
// Byte ranges decide, not tensor names: conservative (whole tensors) except for convs writing disjoint channel slices of one
// tensor at batch 1 (the concatenated SSH outputs).  Same kernels' arithmetic, same order: same bits.
struct ByteRange {
  const char* lo;
  const char* hi;
  bool overlaps(const ByteRange& o) const { return lo && o.lo && lo < o.hi && o.lo < hi; }
};
ByteRange tensor_range(const fdt_model* m, int t, int B) {
  if (t < 0) return {nullptr, nullptr};
  const Tensor& x = m->tensors[t];
  return {(const char*)x.d, (const char*)x.d + (size_t)B * x.C * x.H * x.W * 4};
}
struct OpAccess {
  ByteRange rd[3], wr;   // wr of a conv: its channel slice when the batch is 1 (contiguous), else the whole tensor
  ByteRange wr2;         // ConvArgs.out2: the whole second tensor
};
OpAccess op_access(const fdt_model* m, const Op& op, int B) {
  OpAccess x{{{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}}, {nullptr, nullptr}, {nullptr, nullptr}};
  if (op.type == OP_CONV) {
    const ConvArgs& a = op.ca;
    const size_t hwo = (size_t)a.Hout * a.Wout;
    x.rd[0] = {(const char*)a.in, (const char*)a.in + (size_t)a.B * conv_in_bstride(a) * 4};
    if (a.out2) x.wr2 = {(const char*)a.out2, (const char*)a.out2 + (size_t)a.B * a.out2_ctot * hwo * 4};
    if (a.res) x.rd[1] = {(const char*)a.res, (const char*)a.res + (size_t)a.B * a.res_ctot * hwo * 4};
    if (a.up) x.rd[2] = {(const char*)a.up, (const char*)a.up + (size_t)a.B * a.Cout * a.up_h * a.up_w * 4};
    if (a.B == 1)
      x.wr = {(const char*)(a.out + (size_t)a.out_coff * hwo), (const char*)(a.out + (size_t)(a.out_coff + a.Cout) * hwo)};
    else
      x.wr = {(const char*)a.out, (const char*)a.out + (size_t)a.B * a.out_ctot * hwo * 4};
  } else {
    x.rd[0] = tensor_range(m, op.in_t, B);
    x.rd[1] = tensor_range(m, op.in2_t, B);
    x.wr = tensor_range(m, op.out_t, B);
  }
  return x;
}

// The handle's split-K workspace grows by replacement: the superseded buffer is freed (plan time / after an autotune: nothing
// of this handle is in flight, and workspaces are never shared between handles).
int grow_conv_workspace(fdt_model* m, long long need) {
  if (!(need > m->ws_floats || (need && !m->d_convws))) return FDT_OK;
  float* p = nullptr;
  FDT_HIP(hipMalloc((void**)&p, (size_t)need * 4));
  if (m->d_convws) {
    FDT_HIP(fdt::device_sync());
    for (auto it = m->plan_allocs.begin(); it != m->plan_allocs.end(); ++it)
      if (*it == (void*)m->d_convws) {
        m->plan_allocs.erase(it);
        break;
      }
    (void)hipFree(m->d_convws);
  }
  m->plan_allocs.push_back(p);
  m->d_convws = p;
  m->ws_floats = need;
  return FDT_OK;
}

int plan_reduces(fdt_model* m, int B) {
  static const bool off = getenv("FDT_LAZY_REDUCE") && atoi(getenv("FDT_LAZY_REDUCE")) == 0;   // test hook: a reduce pass per layer
  std::vector<OpAccess> pend;          // accesses of the layers whose reduce pass is pending
  long long epoch = 0, need = 0;
  for (auto& op : m->ops) {
    op.flush_before = false;
    op.lazy = false;
    op.ws_off = 0;
    if (op.type == OP_CONV && !op.head) op.ca.defer_reduce = 0;
  }
  for (auto& op : m->ops) {
    const OpAccess x = op_access(m, op, B);
    bool conflict = false;
    for (const auto& p : pend) {
      for (const auto& r : x.rd) conflict |= r.overlaps(p.wr);                    // reads what a pending pass will write
      conflict |= x.wr.overlaps(p.wr) || x.wr2.overlaps(p.wr);                     // writes it
      for (const auto& r : p.rd) conflict |= x.wr.overlaps(r) || x.wr2.overlaps(r); // rewrites what a pending pass will read
    }
    // the heads' finalize and Detect read through tables / other paths: nothing may be pending behind them
    if (op.type == OP_HEADFIN || op.type == OP_MBOXFIN) conflict |= !pend.empty();
    if (conflict) {
      op.flush_before = true;
      pend.clear();
      epoch = 0;
    }
    if (op.type == OP_CONV && op.ca.ksplit > 1 && !op.head && !op.combine && !off) {
      op.lazy = true;
      op.ws_off = epoch;
      op.ca.defer_reduce = 2;
      epoch += conv_ws_floats(op.ca);
      need = std::max(need, epoch);
      OpAccess px = x;
      px.rd[0] = {nullptr, nullptr};   // the pending PASS reads the slabs, the residual and the upsample source, not the conv's input
      pend.push_back(px);
    } else if (op.type == OP_CONV && op.ca.ksplit > 1) {
      need = std::max(need, epoch + conv_ws_floats(op.ca));   // its slabs live behind the pending ones for the length of the op
      op.ws_off = epoch;
    }
  }
  m->flush_at_end = !pend.empty();
  FDT_TRY(grow_conv_workspace(m, need));
  for (auto& op : m->ops)
    if (op.type == OP_CONV && !op.head) op.ca.ws = op.ca.ksplit > 1 ? m->d_convws + op.ws_off : nullptr;
  return FDT_OK;
}

// The raw-frame stem of op (conv_stem_u8.h) on the uint8 frames of the forward being enqueued.
int launch_u8_stem(fdt_model* m, const Op& op, hipStream_t st) {
  if (op.u8_stream)
    return launch_stem3x3s2_u8(m->u8_src, op.ca.B, op.ca.Hin, op.ca.Win, m->u8_mean, op.u8_w, op.ca.bias, op.ca.Cout, op.ca.act,
                               op.ca.out, st);
  ConvArgs a = op.ca;
  a.in = nullptr;
  a.in_u8 = m->u8_src;
  for (int c = 0; c < 3; ++c) a.u8_mean[c] = m->u8_mean[c];
  a.u8_scale = m->u8_scale;
  a.w = op.u8_w;
  a.ksplit = 1;
  a.ws = nullptr;
  a.sk_count = nullptr;
  a.defer_reduce = 0;
  a.map_mode = CONV_MAP_ROWS;
  return launch_conv(op.u8_kind, op.u8_tile, a, st, m->device);
}

int run_ops(fdt_model* m, int B, hipStream_t st, size_t first_op = 0) {
  // per-op events (fdt_model_profile_enable) xor one event pair around a contiguous run of ops (fdt_model_profile_segment):
  // the second form leaves the launch sequence as it is in production -- lazy grouped reduce passes, no event packet
  // between two kernels -- and measures what the per-op sums cannot: the ops back to back
  const bool seg = m->profile && m->seg_first >= 0;
  const bool prof = m->profile && !seg;
  if (prof) FDT_TRY(ensure_profile_events(m));
  const ConvArgs* pending[64];
  int npend = 0;
  auto flush = [&]() -> int {
    if (npend) FDT_TRY(launch_reduce_group(pending, npend, st));
    npend = 0;
    return FDT_OK;
  };
  exp_skip_reduce = m->passes > 0 && op_skipped("@reduce");   // (experiment builds only; once per pass)
  for (size_t i = first_op; i < m->ops.size(); ++i) {
    const Op& op = m->ops[i];
    if (op.flush_before) FDT_TRY(flush());
    if (prof) FDT_HIP(hipEventRecord(m->ev[i], st));
    if (seg && (int)i == m->seg_first) FDT_HIP(hipEventRecord(m->seg_ev[0], st));
    if (seg && (int)i == m->seg_last + 1) {   // the segment's deferred reduce passes belong to it
      FDT_TRY(flush());
      FDT_HIP(hipEventRecord(m->seg_ev[1], st));
    }
    if (m->passes > 0 && op_skipped(op.name)) continue;   // first pass complete: later ones read its (stale) maps
    switch (op.type) {
      case OP_CONV:
        if (op.u8_stem && m->u8_src) {
          FDT_TRY(launch_u8_stem(m, op, st));
          break;
        }
        FDT_TRY(launch_conv(op.kind, op.tile, op.ca, st, m->device));
        if (op.lazy) {
          if (npend == 64) FDT_TRY(flush());
          pending[npend++] = &op.ca;
          if (prof) FDT_TRY(flush());   // per-op timing: every layer with its own reduce pass, as before
        }
        break;
      case OP_POOL: {
        const Tensor& in = m->tensors[op.in_t];
        const Tensor& out = m->tensors[op.out_t];
        FDT_TRY(launch_maxpool3(in.d, B, in.C, in.H, in.W, op.stride, op.crelu, out.d, out.H, out.W, st));
        break;
      }
      case OP_PAD: {
        const Tensor& in = m->tensors[op.in_t];
        const Tensor& out = m->tensors[op.out_t];
        FDT_TRY(launch_pad1(in.d, B * in.C, in.H, in.W, out.d, st));
        break;
      }
      case OP_DW: {
        const Tensor& in = m->tensors[op.in_t];
        const Tensor& out = m->tensors[op.out_t];
        FDT_TRY(launch_dwconv(in.d, op.w, op.bias, B, in.C, in.H, in.W, op.ksize, op.stride, op.pad, op.dil, op.act,
                              out.d, out.H, out.W, st));
        break;
      }
      case OP_EXPDW: {
        const Tensor& in = m->tensors[op.in_t];
        const Tensor& out = m->tensors[op.out_t];
        if (op.oup)
          FDT_TRY(launch_ir_block(in.d, B, in.C, in.H, in.W, op.w, op.bias, op.w2, op.bias2, op.hid, op.stride, op.w3, op.bias3, op.oup,
                                  op.residual, out.d, out.H, out.W, st, m->device));
        else
          FDT_TRY(launch_expand_dw(in.d, B, in.C, in.H, in.W, op.w, op.bias, op.w2, op.bias2, op.hid, op.stride, out.d, out.H,
                                   out.W, st, m->device));
        break;
      }
      case OP_DWPROJ: {
        const Tensor& in = m->tensors[op.in_t];
        const Tensor& out = m->tensors[op.out_t];
        FDT_TRY(launch_dw_project(in.d, B, in.C, in.H, in.W, op.w, op.bias, op.w2, op.bias2, op.oup,
                                  op.in2_t >= 0 ? m->tensors[op.in2_t].d : nullptr, out.d, st));
        break;
      }
      case OP_HEADFIN:
        if (op.last_head) FDT_TRY(launch_head_finalize_all(m->headfin, B, st));   // all levels at once (setup_heads)
        break;
      case OP_MBOXFIN:
        if (op.last_head) FDT_TRY(launch_head_finalize_all(m->headfin, B, st));   // all multibox levels at once (setup_heads)
        break;
      default:
        break;
    }
  }
  FDT_TRY(flush());
  if (prof) FDT_HIP(hipEventRecord(m->ev[m->ops.size()], st));
  if (seg && m->seg_last + 1 == (int)m->ops.size()) FDT_HIP(hipEventRecord(m->seg_ev[1], st));
  exp_skip_reduce = false;
  ++m->passes;
  return FDT_OK;
}

int forward_impl(fdt_model* m, const void* frames, bool frames_on_device, int format, int B, int H, int W,
                 bool run_detect, float* out_dev, int* counts_dev, hipStream_t user_stream, int src_h = 0,
                 int src_w = 0) {
  FDT_REQUIRE(m && frames, FDT_ERR_ARG, "fdt_model_forward: null argument");
  FDT_REQUIRE(m->finalized, FDT_ERR_STATE, "fdt_model_forward: call fdt_model_finalize first");
  FDT_REQUIRE(B >= 1 && H >= 1 && W >= 1, FDT_ERR_ARG, "fdt_model_forward: bad shape");
  FDT_REQUIRE(format == FDT_FRAME_U8_HWC_BGR || format == FDT_FRAME_F32_NCHW, FDT_ERR_ARG,
              "fdt_model_forward: unknown frame format %d", format);
  FDT_HIP(hipSetDevice(m->device));
  const bool fresh = !(m->pB == B && m->pH == H && m->pW == W && !m->ops.empty()) || m->priors_dirty;
  FDT_TRY(make_plan(m, B, H, W));
  hipStream_t st = user_stream ? user_stream : m->stream;
  if (fresh && st != m->stream) FDT_HIP(hipStreamSynchronize(m->stream));   // priors were built there
  const hipMemcpyKind kind = frames_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  float* x = m->tensors[0].d;
  if (m->profile && m->seg_first < 0) {
    FDT_TRY(ensure_profile_events(m));
    FDT_HIP(hipEventRecord(m->ev[m->ops.size() + 2], st));
  }
  // Fused ingest (conv_stem_u8.h): with uint8 frames and a 7x7 stem on the input tensor, the (float)u8 - mean (/ 255) of
  // iouTracke_cal.py:40-46 / My_test_facebox.py:14-15 happens in the stem conv's staging -- the f32 NCHW frame is never
  // written.  The stem then reads the caller's device frames in place, the H2D landing buffer, or the resized uint8 image.
  const bool fused = m->fuse_stem && format == FDT_FRAME_U8_HWC_BGR && !m->ops.empty() && m->ops[0].u8_stem &&
                     (m->fuse_stem >= 2 || m->ops[0].u8_stream || m->ops[0].u8_kind == CONV_7x7_S2_U8 ||
                      m->ops[0].u8_kind == CONV_7x7_S2_U8B || m->ops[0].u8_kind == CONV_7x7_S4_U8B);
  m->u8_src = nullptr;
  if (fused) {
    const bool fb = m->arch == FDT_ARCH_FACEBOX;
    m->u8_mean[0] = fb ? 0.f : 104.f; m->u8_mean[1] = fb ? 0.f : 117.f; m->u8_mean[2] = fb ? 0.f : 123.f;
    m->u8_scale = fb ? 255.0f : 1.0f;
  }
  if (format == FDT_FRAME_U8_HWC_BGR && src_h > 0 && (src_h != H || src_w != W)) {
    // device-side ingest: cv2.resize(frame, (W, H)) + mean subtraction in one kernel
    const unsigned char* src = (const unsigned char*)frames;
    const size_t bytes = (size_t)B * src_h * src_w * 3;
    if (!frames_on_device) {
      if (bytes > m->src_bytes) {
        if (m->d_src_u8) (void)hipFree(m->d_src_u8);
        m->d_src_u8 = nullptr;
        m->src_bytes = 0;
        FDT_HIP(hipMalloc((void**)&m->d_src_u8, bytes));
        m->src_bytes = bytes;
      }
      FDT_HIP(hipMemcpyAsync(m->d_src_u8, frames, bytes, kind, st));
      src = m->d_src_u8;
    }
    if (fused) {
      FDT_TRY(launch_resize_u8(src, B, src_h, src_w, H, W, m->d_frames_u8, st));   // the resized uint8 image; the stem converts
      m->u8_src = m->d_frames_u8;
    } else if (m->arch == FDT_ARCH_FACEBOX)
      FDT_TRY(launch_resize_preprocess(src, B, src_h, src_w, H, W, 0.f, 0.f, 0.f, 255.0f, x, st));
    else
      FDT_TRY(launch_resize_preprocess(src, B, src_h, src_w, H, W, 104.f, 117.f, 123.f, 1.0f, x, st));
  } else if (format == FDT_FRAME_U8_HWC_BGR) {
    const unsigned char* src = (const unsigned char*)frames;
    if (!frames_on_device) {
      FDT_HIP(hipMemcpyAsync(m->d_frames_u8, frames, (size_t)B * H * W * 3, kind, st));
      src = m->d_frames_u8;
    }
    if (fused)
      m->u8_src = src;
    else if (m->arch == FDT_ARCH_FACEBOX)
      FDT_TRY(launch_preprocess(src, B, H, W, 0.f, 0.f, 0.f, 255.0f, x, st));
    else
      FDT_TRY(launch_preprocess(src, B, H, W, 104.f, 117.f, 123.f, 1.0f, x, st));
  } else {
    FDT_HIP(hipMemcpyAsync(x, frames, (size_t)B * 3 * H * W * 4, kind, st));
  }
  m->last_fused = m->u8_src != nullptr;
  m->last_u8_src = m->u8_src;
  // everything after the ingest kernel: one hipGraphLaunch once the plan has run eagerly (which also sets the
  // per-function LDS attributes) -- the graph bakes in the plan, the output buffers and the thresholds
  float* out_p = out_dev ? out_dev : m->d_out;
  int* counts_p = counts_dev ? counts_dev : m->d_counts;
  // graph path: the raw-frame stem reads a per-call pointer, so it runs eagerly in front of the graph (where the ingest kernel
  // used to), and the graph covers the ops behind it
  const size_t first_op = (m->use_graph && !m->profile && m->u8_src) ? 1 : 0;
  if (first_op) FDT_TRY(launch_u8_stem(m, m->ops[0], st));
  auto body = [&]() -> int {
    FDT_TRY(run_ops(m, B, st, first_op));
    if (run_detect && m->arch == FDT_ARCH_FACEBOX) {
      FDT_TRY(launch_facebox_decode(m->dplan, m->d_ws, m->d_loc, m->d_conf, m->d_priors, m->conf_t, m->nms_t,
                                    m->d_fb_boxes, m->d_fb_probs, counts_p, st));
      if (m->profile && m->seg_first < 0) FDT_HIP(hipEventRecord(m->ev[m->ops.size() + 1], st));
    } else if (run_detect) {
      if (!(m->passes > 1 && !m->profile && op_skipped("@detect")))   // (experiment hook: Detect left out of the un-profiled passes)
        FDT_TRY(launch_detect(m->dplan, m->d_ws, m->d_loc, m->d_conf, m->d_priors, 2, m->top_k, m->conf_t,
                              m->nms_t, 0.1f, 0.2f, out_p, counts_p, st));
      if (m->profile && m->seg_first < 0) FDT_HIP(hipEventRecord(m->ev[m->ops.size() + 1], st));
    }
    return FDT_OK;
  };
  if (m->use_graph && !m->profile) {
    const GraphKey key{run_detect ? (const void*)out_p : nullptr, run_detect ? (const void*)counts_p : nullptr,
                       run_detect ? 1 : 0, m->conf_t, m->nms_t, (int)first_op};
    auto it = m->graphs.find(key);
    if (it != m->graphs.end()) {
      m->graph_used[key] = ++m->graph_clock;
      FDT_HIP(hipGraphLaunch(it->second, st));
      return FDT_OK;
    }
    if (m->plan_runs >= 1) {
      if (m->graphs.size() >= kMaxGraphs) {
        // a caller that rotates its output buffers: evict the least recently replayed graph instead of silently running
        // eager from the 17th buffer on.  The victim may still be executing (on any stream it was replayed on).
        auto victim = m->graphs.begin();
        for (auto g = m->graphs.begin(); g != m->graphs.end(); ++g)
          if (m->graph_used[g->first] < m->graph_used[victim->first]) victim = g;
        FDT_HIP(fdt::device_sync());
        (void)hipGraphExecDestroy(victim->second);
        m->graph_used.erase(victim->first);
        m->graphs.erase(victim);
      }
      // no device-wide wait of another thread may fall inside the capture (common.h: device_sync)
      fdt::capture_lock_shared();
      const hipError_t be = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
      if (be != hipSuccess) {
        fdt::capture_unlock_shared();
        FDT_HIP(be);
      }
      const int rc = body();
      hipGraph_t g = nullptr;
      const hipError_t ce = hipStreamEndCapture(st, &g);
      fdt::capture_unlock_shared();
      if (rc != FDT_OK) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
      }
      FDT_REQUIRE(ce == hipSuccess && g, FDT_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
      hipGraphExec_t ex = nullptr;
      const hipError_t ie = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      FDT_REQUIRE(ie == hipSuccess && ex, FDT_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie));
      m->graphs[key] = ex;
      m->graph_used[key] = ++m->graph_clock;
      FDT_HIP(hipGraphLaunch(ex, st));
      return FDT_OK;
    }
  }
  FDT_TRY(body());
  m->plan_runs++;
  return FDT_OK;
}

}  // namespace

// ================================================================================== C ABI
extern "C" fdt_model* fdt_model_create(int arch, int device) {
  if (arch < FDT_ARCH_RES50 || arch > FDT_ARCH_TRY2) {
    set_error("fdt_model_create: unknown arch %d", arch);
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) {
    set_error("fdt_model_create: hipSetDevice(%d) failed: %s", device, hipGetErrorString(hipGetLastError()));
    return nullptr;
  }
  std::unique_ptr<fdt_model> m(new fdt_model());
  m->arch = arch;
  m->device = device;
  if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) {
    set_error("fdt_model_create: stream creation failed");
    return nullptr;
  }
  if (const char* g = getenv("FDT_GRAPH")) m->use_graph = atoi(g) != 0;
  if (const char* g = getenv("FDT_FB_FUSE")) m->fb_fuse = atoi(g);
  if (const char* g = getenv("FDT_STREAM_IR")) m->stream_ir = atoi(g) != 0;
  if (const char* g = getenv("FDT_STEM_B3")) m->stem_b3 = atoi(g) != 0;
  if (const char* g = getenv("FDT_FUSE_INGEST")) m->fuse_stem = atoi(g);   // A/B: 0 = ingest kernel + planar stem conv, 2 = also FaceBoxes
  if (arch == FDT_ARCH_TRY3 || arch == FDT_ARCH_TRY4 || arch == FDT_ARCH_TRY5) {   // pyramid_mb2_try3.py:216
    m->conf_t = 0.2f;
    m->nms_t = 0.35f;
  }
  if (arch == FDT_ARCH_TRY1) {   // pyramid_mobile_try1.py:220: Detect(num_classes, 0, 750, 0.3, 0.3); try2 keeps (0.3, 0.5)
    m->conf_t = 0.3f;
    m->nms_t = 0.3f;
  }
  if (arch == FDT_ARCH_FACEBOX) {   // decode_np(conf_thres=0.35), nms_np(threshold=0.5)  encoderl.py:217,308
    m->conf_t = 0.35f;
    m->nms_t = 0.5f;
  }
  // dry build: collect the state-dict keys the forward graph reads
  m->dry = true;
  int rc = build_graph(m.get(), 1, arch == FDT_ARCH_FACEBOX ? 1024 : 256, arch == FDT_ARCH_FACEBOX ? 1024 : 256);
  m->dry = false;
  m->tensors.clear();
  m->ops.clear();
  m->levels.clear();
  if (rc != FDT_OK) return nullptr;
  return m.release();
}

extern "C" void fdt_model_destroy(fdt_model* m) {
  if (!m) return;
  fdt::ExclusiveDevice quiet;
  (void)hipSetDevice(m->device);
  (void)fdt::device_sync();
  delete m;
}

// A second handle on the same net and GPU that SHARES the weights (state dict, folded host copies, tiled device
// copies) with `src` but has its own stream, activations, plan and Detect workspace: what a caller keeps per frame in
// flight.  Kernel-plan hints and the detect / priorbox configuration are copied.  Read-only on the weights: set_tensor
// on either handle fails while both exist.
extern "C" fdt_model* fdt_model_clone(fdt_model* src) {
  if (!src || !src->finalized) {
    set_error("fdt_model_clone: source handle is null or not finalized");
    return nullptr;
  }
  if (hipSetDevice(src->device) != hipSuccess) {
    set_error("fdt_model_clone: hipSetDevice(%d) failed", src->device);
    return nullptr;
  }
  std::unique_ptr<fdt_model> m(new fdt_model());
  m->arch = src->arch;
  m->device = src->device;
  if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) {
    set_error("fdt_model_clone: stream creation failed");
    return nullptr;
  }
  m->W = src->W;
  m->expected = src->expected;
  m->finalized = true;
  m->top_k = src->top_k;
  m->nms_top_k = src->nms_top_k;
  m->conf_t = src->conf_t;
  m->nms_t = src->nms_t;
  m->pb_set = src->pb_set;
  m->pb_w = src->pb_w;
  m->pb_h = src->pb_h;
  m->pb_stride = src->pb_stride;
  m->pb_box = src->pb_box;
  m->hints = src->hints;
  m->hB = src->hB;
  m->hH = src->hH;
  m->hW = src->hW;
  m->use_graph = src->use_graph;
  m->fuse_stem = src->fuse_stem;
  m->fb_fuse = src->fb_fuse;
  m->stream_ir = src->stream_ir;
  m->stem_b3 = src->stem_b3;
  return m.release();
}

extern "C" int fdt_model_fuse_ingest(fdt_model* m, int on) {
  FDT_REQUIRE(m, FDT_ERR_ARG, "fdt_model_fuse_ingest: null handle");
  m->fuse_stem = on < 0 ? 0 : (on > 2 ? 2 : on);
  return FDT_OK;
}

extern "C" int fdt_model_get_detect(fdt_model* m, int* top_k, float* conf_thresh, float* nms_thresh, int* nms_top_k) {
  FDT_REQUIRE(m, FDT_ERR_ARG, "fdt_model_get_detect: null handle");
  if (top_k) *top_k = m->top_k;
  if (conf_thresh) *conf_thresh = m->conf_t;
  if (nms_thresh) *nms_thresh = m->nms_t;
  if (nms_top_k) *nms_top_k = m->nms_top_k;
  return FDT_OK;
}

extern "C" int fdt_model_set_tensor(fdt_model* m, const char* name, const float* data, int ndim,
                                    const long long* dims) {
  FDT_REQUIRE(m && name && ndim >= 0 && ndim <= 8, FDT_ERR_ARG, "fdt_model_set_tensor: bad argument");
  std::string k(name);
  if (ignored_key(m, k)) return FDT_OK;
  FDT_REQUIRE(m->expected.count(k), FDT_ERR_NAME, "Unexpected key(s) in state_dict: \"%s\"", name);
  FDT_REQUIRE(data, FDT_ERR_ARG, "fdt_model_set_tensor: null data for %s", name);
  HostT t;
  long long n = 1;
  for (int i = 0; i < ndim; ++i) {
    FDT_REQUIRE(dims[i] >= 0, FDT_ERR_ARG, "fdt_model_set_tensor: negative dim");
    t.dims.push_back(dims[i]);
    n *= dims[i];
  }
  FDT_REQUIRE(m->W.use_count() == 1, FDT_ERR_STATE,
              "fdt_model_set_tensor: the weights are shared with fdt_model_clone() handles; destroy those first");
  t.v.assign(data, data + n);
  m->W->sd[k] = std::move(t);
  m->finalized = false;
  return FDT_OK;
}

extern "C" int fdt_model_missing(fdt_model* m, int* n) {
  FDT_REQUIRE(m && n, FDT_ERR_ARG, "fdt_model_missing: bad argument");
  int c = 0;
  for (auto& k : m->expected)
    if (!m->W->sd.count(k)) ++c;
  *n = c;
  return FDT_OK;
}

extern "C" int fdt_model_missing_name(fdt_model* m, int i, char* buf, int buflen) {
  FDT_REQUIRE(m && buf && buflen > 0, FDT_ERR_ARG, "fdt_model_missing_name: bad argument");
  int c = 0;
  for (auto& k : m->expected)
    if (!m->W->sd.count(k)) {
      if (c == i) {
        snprintf(buf, buflen, "%s", k.c_str());
        return FDT_OK;
      }
      ++c;
    }
  set_error("fdt_model_missing_name: index %d out of range", i);
  return FDT_ERR_ARG;
}

extern "C" int fdt_model_finalize(fdt_model* m) {
  FDT_REQUIRE(m, FDT_ERR_ARG, "fdt_model_finalize: null handle");
  for (auto& k : m->expected)
    FDT_REQUIRE(m->W->sd.count(k), FDT_ERR_STATE, "Missing key(s) in state_dict: \"%s\"", k.c_str());
  FDT_HIP(hipSetDevice(m->device));
  // weights may have changed: drop cached device copies and the plan
  FDT_HIP(hipStreamSynchronize(m->stream));
  m->free_plan();
  if (m->W.use_count() == 1) {
    m->W->drop_device();
  } else {
    // clones still read the device copies: only legal when nothing was re-loaded since (set_tensor refuses that)
    FDT_REQUIRE(m->finalized, FDT_ERR_STATE, "fdt_model_finalize: weights are shared with clones");
  }
  m->finalized = true;
  return FDT_OK;
}

extern "C" int fdt_model_set_priorbox(fdt_model* m, int width, int height, int n_levels, const int* stride,
                                      const int* box) {
  FDT_REQUIRE(m && width > 0 && height > 0 && n_levels >= 1 && stride && box, FDT_ERR_ARG,
              "fdt_model_set_priorbox: bad argument");
  m->pb_set = true;
  m->pb_w = width;
  m->pb_h = height;
  m->pb_stride.assign(stride, stride + n_levels);
  m->pb_box.assign(box, box + n_levels);
  m->priors_dirty = true;   // net.firstTime = True: priors are regenerated by the next forward
  return FDT_OK;
}

extern "C" int fdt_model_set_detect(fdt_model* m, int top_k, float conf_thresh, float nms_thresh,
                                    int nms_top_k) {
  FDT_REQUIRE(m && top_k >= 1 && nms_top_k >= 1, FDT_ERR_ARG, "fdt_model_set_detect: bad argument");
  FDT_REQUIRE(nms_thresh > 0.0f, FDT_ERR_ARG, "nms_threshold must be non negative.");
  if (top_k != m->top_k || nms_top_k != m->nms_top_k) {
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    m->free_plan();
  }
  m->top_k = top_k;
  m->nms_top_k = nms_top_k;
  m->conf_t = conf_thresh;
  m->nms_t = nms_thresh;
  return FDT_OK;
}

extern "C" int fdt_model_detect_facebox(fdt_model* m, const void* frames, int format, int B, int H, int W,
                                        float conf_thresh, float nms_thresh, float* boxes, float* probs,
                                        int* counts) {
  FDT_REQUIRE(m && boxes && probs && counts, FDT_ERR_ARG, "fdt_model_detect_facebox: null argument");
  FDT_REQUIRE(m->arch == FDT_ARCH_FACEBOX, FDT_ERR_ARG, "fdt_model_detect_facebox: not a FaceBox model");
  FDT_REQUIRE(nms_thresh > 0.f, FDT_ERR_ARG, "nms threshold must be positive");
  m->conf_t = conf_thresh;
  m->nms_t = nms_thresh;
  FDT_TRY(forward_impl(m, frames, false, format, B, H, W, true, nullptr, nullptr, nullptr));
  FDT_HIP(hipMemcpyAsync(counts, m->d_counts, (size_t)B * 4, hipMemcpyDeviceToHost, m->stream));
  FDT_HIP(hipMemcpyAsync(boxes, m->d_fb_boxes, (size_t)B * m->P * 16, hipMemcpyDeviceToHost, m->stream));
  FDT_HIP(hipMemcpyAsync(probs, m->d_fb_probs, (size_t)B * m->P * 4, hipMemcpyDeviceToHost, m->stream));
  FDT_HIP(hipStreamSynchronize(m->stream));
  return FDT_OK;
}

extern "C" int fdt_model_detect_facebox_dev(fdt_model* m, const void* frames_dev, int format, int B, int H,
                                            int W, float conf_thresh, float nms_thresh, int* counts_dev,
                                            void* stream) {
  FDT_REQUIRE(m && m->arch == FDT_ARCH_FACEBOX, FDT_ERR_ARG, "fdt_model_detect_facebox_dev: not a FaceBox model");
  FDT_REQUIRE(nms_thresh > 0.f, FDT_ERR_ARG, "nms threshold must be positive");
  m->conf_t = conf_thresh;
  m->nms_t = nms_thresh;
  return forward_impl(m, frames_dev, true, format, B, H, W, true, nullptr, counts_dev, (hipStream_t)stream);
}

// detect(im) of FACEBOX/My_test_facebox.py:12-36 INCLUDING its first line, im = cv2.resize(im, (1024, 1024)) (:13):
// B raw u8 BGR frames of src_h x src_w (e.g. 2160 x 3840) are resized on the GPU (8-bit INTER_LINEAR restatement, parity
// with cv2 itself unpinned), divided by 255, run through FaceBox, softmax, decode_np + nms_np.
extern "C" int fdt_model_detect_facebox_resized(fdt_model* m, const void* frames, int frames_on_device, int B,
                                                int src_h, int src_w, float conf_thresh, float nms_thresh,
                                                float* boxes, float* probs, int* counts, void* stream) {
  FDT_REQUIRE(m && m->arch == FDT_ARCH_FACEBOX, FDT_ERR_ARG, "fdt_model_detect_facebox_resized: not a FaceBox model");
  FDT_REQUIRE(nms_thresh > 0.f, FDT_ERR_ARG, "nms threshold must be positive");
  FDT_REQUIRE(src_h >= 1 && src_w >= 1, FDT_ERR_ARG, "fdt_model_detect_facebox_resized: bad source size");
  m->conf_t = conf_thresh;
  m->nms_t = nms_thresh;
  if (frames_on_device)
    return forward_impl(m, frames, true, FDT_FRAME_U8_HWC_BGR, B, 1024, 1024, true, nullptr, counts,
                        (hipStream_t)stream, src_h, src_w);
  FDT_REQUIRE(boxes && probs && counts, FDT_ERR_ARG, "fdt_model_detect_facebox_resized: null output");
  FDT_TRY(forward_impl(m, frames, false, FDT_FRAME_U8_HWC_BGR, B, 1024, 1024, true, nullptr, nullptr, nullptr, src_h,
                       src_w));
  FDT_HIP(hipMemcpyAsync(counts, m->d_counts, (size_t)B * 4, hipMemcpyDeviceToHost, m->stream));
  FDT_HIP(hipMemcpyAsync(boxes, m->d_fb_boxes, (size_t)B * m->P * 16, hipMemcpyDeviceToHost, m->stream));
  FDT_HIP(hipMemcpyAsync(probs, m->d_fb_probs, (size_t)B * m->P * 4, hipMemcpyDeviceToHost, m->stream));
  FDT_HIP(hipStreamSynchronize(m->stream));
  return FDT_OK;
}

extern "C" int fdt_model_forward(fdt_model* m, const void* frames, int format, int B, int H, int W,
                                 float* out, int* counts) {
  FDT_REQUIRE(out, FDT_ERR_ARG, "fdt_model_forward: null output");
  FDT_REQUIRE(m && m->arch != FDT_ARCH_FACEBOX, FDT_ERR_ARG,
              "fdt_model_forward: FaceBox has no Detect layer; use fdt_model_detect_facebox");
  FDT_TRY(forward_impl(m, frames, false, format, B, H, W, true, nullptr, nullptr, nullptr));
  FDT_HIP(hipMemcpyAsync(out, m->d_out, (size_t)B * 2 * m->top_k * 5 * 4, hipMemcpyDeviceToHost, m->stream));
  if (counts) FDT_HIP(hipMemcpyAsync(counts, m->d_counts, (size_t)B * 2 * 4, hipMemcpyDeviceToHost, m->stream));
  FDT_HIP(hipStreamSynchronize(m->stream));
  return FDT_OK;
}

extern "C" int fdt_model_forward_resized(fdt_model* m, const void* frames, int frames_on_device, int B,
                                         int src_h, int src_w, int H, int W, float* out, int* counts,
                                         void* stream) {
  FDT_REQUIRE(m && out && src_h >= 1 && src_w >= 1, FDT_ERR_ARG, "fdt_model_forward_resized: bad argument");
  FDT_REQUIRE(m->arch != FDT_ARCH_FACEBOX, FDT_ERR_ARG, "fdt_model_forward_resized: use fdt_model_detect_facebox");
  if (frames_on_device)
    return forward_impl(m, frames, true, FDT_FRAME_U8_HWC_BGR, B, H, W, true, out, counts, (hipStream_t)stream,
                        src_h, src_w);
  FDT_TRY(forward_impl(m, frames, false, FDT_FRAME_U8_HWC_BGR, B, H, W, true, nullptr, nullptr, nullptr, src_h,
                       src_w));
  FDT_HIP(hipMemcpyAsync(out, m->d_out, (size_t)B * 2 * m->top_k * 5 * 4, hipMemcpyDeviceToHost, m->stream));
  if (counts) FDT_HIP(hipMemcpyAsync(counts, m->d_counts, (size_t)B * 2 * 4, hipMemcpyDeviceToHost, m->stream));
  FDT_HIP(hipStreamSynchronize(m->stream));
  return FDT_OK;
}

extern "C" int fdt_model_forward_dev(fdt_model* m, const void* frames_dev, int format, int B, int H, int W,
                                     float* out_dev, int* counts_dev, void* stream) {
  FDT_REQUIRE(out_dev, FDT_ERR_ARG, "fdt_model_forward_dev: null output");
  FDT_REQUIRE(m && m->arch != FDT_ARCH_FACEBOX, FDT_ERR_ARG,
              "fdt_model_forward_dev: FaceBox has no Detect layer; use fdt_model_detect_facebox_dev");
  return forward_impl(m, frames_dev, true, format, B, H, W, true, out_dev, counts_dev, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------- pipelined host ingest
// iouTracke_cal.py:119-124 hands the detector host arrays (cv2 frames).  forward_async copies the caller's pageable
// buffer into a pinned slot (the caller may reuse its buffer at once), enqueues the H2D, the forward and the D2H of the
// Detect record on the handle's stream, and returns a ticket.  Two tickets per handle may be in flight; the copy overlaps
// the forwards of the OTHER handles in flight.
extern "C" int fdt_model_forward_async(fdt_model* m, const void* frames, int format, int B, int H, int W, int src_h,
                                       int src_w, int* ticket) {
  FDT_REQUIRE(m && frames && ticket, FDT_ERR_ARG, "fdt_model_forward_async: null argument");
  FDT_REQUIRE(m->arch != FDT_ARCH_FACEBOX, FDT_ERR_ARG, "fdt_model_forward_async: PyramidBox nets only");
  FDT_REQUIRE(B >= 1 && H >= 1 && W >= 1, FDT_ERR_ARG, "fdt_model_forward_async: bad shape");
  FDT_REQUIRE(format == FDT_FRAME_U8_HWC_BGR || format == FDT_FRAME_F32_NCHW, FDT_ERR_ARG,
              "fdt_model_forward_async: unknown frame format %d", format);
  const bool resized = format == FDT_FRAME_U8_HWC_BGR && src_h > 0 && src_w > 0 && (src_h != H || src_w != W);
  FDT_HIP(hipSetDevice(m->device));
  AsyncSlot& s = m->slots[m->next_ticket % kAsyncSlots];
  FDT_REQUIRE(!s.busy, FDT_ERR_STATE, "fdt_model_forward_async: ticket %d is still in flight; fdt_model_wait it first",
              s.ticket);
  if (!s.copied) {
    for (hipEvent_t* e : {&s.copied, &s.fwd, &s.done, &s.consumed})
      FDT_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
  }
  const size_t in_bytes = format == FDT_FRAME_F32_NCHW ? (size_t)B * 3 * H * W * 4
                                                       : (size_t)B * (resized ? src_h : H) * (resized ? src_w : W) * 3;
  const size_t out_floats = (size_t)B * 2 * m->top_k * 5, n_counts = (size_t)B * 2;
  if (in_bytes > s.in_bytes || out_floats != s.out_floats || n_counts != s.n_counts) {
    FDT_HIP(fdt::device_sync());   // a (rare) re-size: nothing may still read the old buffers
    hipEvent_t ev[4] = {s.copied, s.fwd, s.done, s.consumed};
    s.copied = s.fwd = s.done = s.consumed = nullptr;
    s.release();
    s.copied = ev[0]; s.fwd = ev[1]; s.done = ev[2]; s.consumed = ev[3];
    m->drop_graphs();                  // captured forwards bake in the old record pointer
    FDT_HIP(hipHostMalloc(&s.h_in, in_bytes, hipHostMallocDefault));
    FDT_HIP(hipMalloc(&s.d_in, in_bytes));
    FDT_HIP(hipMalloc((void**)&s.d_out, out_floats * 4));
    FDT_HIP(hipHostMalloc((void**)&s.h_out, out_floats * 4, hipHostMallocDefault));
    FDT_HIP(hipMalloc((void**)&s.d_counts, n_counts * 4));
    FDT_HIP(hipHostMalloc((void**)&s.h_counts, n_counts * 4, hipHostMallocDefault));
    s.in_bytes = in_bytes;
    s.out_floats = out_floats;
    s.n_counts = n_counts;
  }
  if (s.released) {
    // the ticket this slot carried last was retired without a host wait (fdt_model_release): its forward may still be
    // queued, and its H2D sits behind the forward before it on the same stream -- the pinned buffer is free once that copy
    // has run (the device buffer is protected by stream order)
    FDT_HIP(hipEventSynchronize(s.copied));
    s.released = false;
  }
  memcpy(s.h_in, frames, in_bytes);
  // The copy goes on the COMPUTE stream, in front of its forward.  A separate copy stream can be mapped onto the hardware
  // queue of ANOTHER handle's compute stream (streams share GPU_MAX_HW_QUEUES = 4 queues): its packets then wait behind a
  // whole forward of that handle, and the forward that needs the copy with them -- measured as a GPU 10 % slower than the
  // device-resident pipeline whatever the host did (tools/experiments/host_path_breakdown.py).  With two or more handles
  // in flight the 3 MB copy overlaps the other handles' forwards anyway.
  FDT_HIP(hipMemcpyAsync(s.d_in, s.h_in, in_bytes, hipMemcpyHostToDevice, m->stream));
  FDT_HIP(hipEventRecord(s.copied, m->stream));
  if (s.consumed_pending) {            // the last consumer of this slot's device record (e.g. the tracker)
    FDT_HIP(hipStreamWaitEvent(m->stream, s.consumed, 0));
    s.consumed_pending = false;
  }
  FDT_TRY(forward_impl(m, s.d_in, true, format, B, H, W, true, s.d_out, s.d_counts, nullptr, resized ? src_h : 0,
                       resized ? src_w : 0));
  FDT_HIP(hipEventRecord(s.fwd, m->stream));
  FDT_HIP(hipMemcpyAsync(s.h_out, s.d_out, out_floats * 4, hipMemcpyDeviceToHost, m->stream));
  FDT_HIP(hipMemcpyAsync(s.h_counts, s.d_counts, n_counts * 4, hipMemcpyDeviceToHost, m->stream));
  FDT_HIP(hipEventRecord(s.done, m->stream));
  s.busy = true;
  s.ticket = m->next_ticket++;
  *ticket = s.ticket;
  return FDT_OK;
}

namespace {
AsyncSlot* find_ticket(fdt_model* m, int ticket) {
  if (!m || ticket < 0) return nullptr;
  AsyncSlot& s = m->slots[ticket % kAsyncSlots];
  return (s.busy && s.ticket == ticket) ? &s : nullptr;
}
}  // namespace

// Device-side hand-over: `consumer_stream` waits (on the GPU) for the forward of `ticket`; *record_dev is its Detect
// record [B,2,top_k,5] in HBM (valid until fdt_model_wait).  Lets fdt_tracker_step_dev consume it with no host round trip.
extern "C" int fdt_model_async_record(fdt_model* m, int ticket, float** record_dev, void* consumer_stream) {
  AsyncSlot* s = find_ticket(m, ticket);
  FDT_REQUIRE(s, FDT_ERR_ARG, "fdt_model_async_record: ticket %d is not in flight", ticket);
  FDT_REQUIRE(record_dev, FDT_ERR_ARG, "fdt_model_async_record: null output");
  const hipStream_t cs = consumer_stream ? (hipStream_t)consumer_stream : fdt::thread_stream();   // never the legacy stream
  FDT_REQUIRE(cs, FDT_ERR_HIP, "fdt_model_async_record: could not create the calling thread's private stream");
  FDT_HIP(hipStreamWaitEvent(cs, s->fwd, 0));
  *record_dev = s->d_out;
  return FDT_OK;
}

// Host-side completion: blocks until the record of `ticket` is on the host, copies it to out [B,2,top_k,5] / counts
// [B,2] (either may be NULL) and frees the slot.  consumer_stream != NULL: work already enqueued there (the tracker
// step reading the device record) is ordered before the slot's next forward.
extern "C" int fdt_model_wait(fdt_model* m, int ticket, float* out, int* counts, void* consumer_stream) {
  AsyncSlot* s = find_ticket(m, ticket);
  FDT_REQUIRE(s, FDT_ERR_ARG, "fdt_model_wait: ticket %d is not in flight", ticket);
  if (consumer_stream) {
    FDT_HIP(hipEventRecord(s->consumed, (hipStream_t)consumer_stream));
    s->consumed_pending = true;
  }
  FDT_HIP(hipEventSynchronize(s->done));
  if (out) memcpy(out, s->h_out, s->out_floats * 4);
  if (counts) memcpy(counts, s->h_counts, s->n_counts * 4);
  s->busy = false;
  return FDT_OK;
}

// Retirement WITHOUT a host wait, for callers that consume the record on the device only (fdt_model_async_record +
// fdt_tracker_step_dev): nothing is copied back to the caller and the host does not block on the forward, so it can keep
// every handle's queue full; the slot's next forward_async is ordered on the device behind this ticket's forward and
// behind whatever `consumer_stream` holds now.  A host that waited for every frame (fdt_model_wait) left the GPU 10 %
// slower than the device-resident pipeline (tools/experiments/host_path_breakdown.py).
extern "C" int fdt_model_release(fdt_model* m, int ticket, void* consumer_stream) {
  AsyncSlot* s = find_ticket(m, ticket);
  FDT_REQUIRE(s, FDT_ERR_ARG, "fdt_model_release: ticket %d is not in flight", ticket);
  if (consumer_stream) {
    FDT_HIP(hipEventRecord(s->consumed, (hipStream_t)consumer_stream));
    s->consumed_pending = true;
  }
  s->busy = false;
  s->released = true;
  return FDT_OK;
}

extern "C" int fdt_model_forward_raw(fdt_model* m, const void* frames, int format, int B, int H, int W,
                                     float* loc, float* conf) {
  FDT_REQUIRE(loc && conf, FDT_ERR_ARG, "fdt_model_forward_raw: null output");
  FDT_TRY(forward_impl(m, frames, false, format, B, H, W, false, nullptr, nullptr, nullptr));
  FDT_HIP(hipMemcpyAsync(loc, m->d_loc, (size_t)B * m->P * 16, hipMemcpyDeviceToHost, m->stream));
  // PyramidBox forwards end in nn.Softmax (pyramid.py:332); FaceBox.forward returns raw conf_preds
  FDT_HIP(hipMemcpyAsync(conf, m->arch == FDT_ARCH_FACEBOX ? m->d_logits : m->d_conf, (size_t)B * m->P * 8,
                         hipMemcpyDeviceToHost, m->stream));
  FDT_HIP(hipStreamSynchronize(m->stream));
  return FDT_OK;
}

extern "C" int fdt_model_num_priors(fdt_model* m, int* P) {
  FDT_REQUIRE(m && P, FDT_ERR_ARG, "fdt_model_num_priors: bad argument");
  FDT_REQUIRE(m->pB > 0, FDT_ERR_STATE, "fdt_model_num_priors: no forward has run yet");
  *P = m->P;
  return FDT_OK;
}

extern "C" int fdt_model_get_tensor(fdt_model* m, const char* name, float* out, long long max_elems,
                                    long long* dims4) {
  FDT_REQUIRE(m && name, FDT_ERR_ARG, "fdt_model_get_tensor: bad argument");
  FDT_REQUIRE(m->pB > 0, FDT_ERR_STATE, "fdt_model_get_tensor: no forward has run yet");
  std::string k(name);
  const float* src = nullptr;
  long long d[4] = {m->pB, 0, 0, 0};
  if (k == "loc") {
    src = m->d_loc;
    d[1] = m->P;
    d[2] = 4;
    d[3] = 1;
  } else if (k == "conf") {
    src = m->d_conf;
    d[1] = m->P;
    d[2] = 2;
    d[3] = 1;
  } else if (k == "conf_logits") {
    src = m->d_logits;
    d[1] = m->P;
    d[2] = 2;
    d[3] = 1;
  } else if (k == "fb_boxes" && m->d_fb_boxes) {
    src = m->d_fb_boxes;
    d[1] = m->P;
    d[2] = 4;
    d[3] = 1;
  } else if (k == "fb_probs" && m->d_fb_probs) {
    src = m->d_fb_probs;
    d[1] = m->P;
    d[2] = 1;
    d[3] = 1;
  } else if (k == "priors") {
    src = m->d_priors;
    d[0] = 1;
    d[1] = m->P;
    d[2] = 4;
    d[3] = 1;
  } else {
    if (k == "input" && m->last_fused && m->last_u8_src && !m->tensors.empty()) {
      // the fused ingest never wrote the f32 frame: form it now from the uint8 frames of the last forward (valid while the
      // caller's device frames are; the library's own landing / resize buffers always are)
      const bool fb = m->arch == FDT_ARCH_FACEBOX;
      FDT_HIP(hipSetDevice(m->device));
      FDT_HIP(fdt::device_sync());
      FDT_TRY(launch_preprocess(m->last_u8_src, m->pB, m->pH, m->pW, fb ? 0.f : 104.f, fb ? 0.f : 117.f, fb ? 0.f : 123.f,
                                fb ? 255.0f : 1.0f, m->tensors[0].d, m->stream));
      FDT_HIP(hipStreamSynchronize(m->stream));
    }
    for (auto& t : m->tensors)
      if (t.name == k) {
        src = t.d;
        d[1] = t.C;
        d[2] = t.H;
        d[3] = t.W;
      }
  }
  FDT_REQUIRE(src, FDT_ERR_NAME, "fdt_model_get_tensor: no tensor named '%s'", name);
  if (dims4)
    for (int i = 0; i < 4; ++i) dims4[i] = d[i];
  long long n = d[0] * d[1] * d[2] * d[3];
  if (!out) return FDT_OK;
  FDT_REQUIRE(max_elems >= n, FDT_ERR_ARG, "fdt_model_get_tensor: buffer too small (%lld < %lld)", max_elems, n);
  FDT_HIP(hipStreamSynchronize(m->stream));
  FDT_HIP(copy_sync(out, src, (size_t)n * 4, hipMemcpyDeviceToHost, m->stream));
  return FDT_OK;
}

// Measure every conv layer of the current plan with each instantiated (tile, split-K) candidate on the
// layer's real buffers (HIP events, min of `iters` runs) and keep the fastest.  The analytic model in
// Builder::choose is the starting point; this replaces guessing with measurement ("measure, don't guess").
extern "C" int fdt_model_autotune(fdt_model* m, int iters) {
  FDT_REQUIRE(m && iters >= 1, FDT_ERR_ARG, "fdt_model_autotune: bad argument");
  FDT_REQUIRE(m->pB > 0 && !m->ops.empty(), FDT_ERR_STATE, "fdt_model_autotune: run a forward first");
  FDT_HIP(hipSetDevice(m->device));
  hipStream_t st = m->stream;
  FDT_HIP(fdt::device_sync());
  hipEvent_t e0, e1;
  FDT_HIP(hipEventCreate(&e0));
  FDT_HIP(hipEventCreate(&e1));
  const long long kMaxWsFloats = 256ll * 1024 * 1024;   // 1 GiB of partial sums at most per candidate
  // Experiment hook: only direct-kernel variants whose LDS footprint fits BESIDE a resident Winograd workgroup (132 KB of
  // the CU's 160 KB), so that the small layers of one frame can share CUs with the big layers of another
  const long long small_lds = getenv("FDT_TUNE_MAX_LDS") ? atoll(getenv("FDT_TUNE_MAX_LDS")) : 0;
  int rc = FDT_OK;
  // FDT_TUNE_ONLY=<substring>: re-measure only the layers whose name contains it (the others keep their plan entry)
  const char* only = getenv("FDT_TUNE_ONLY");
  const char* only_base = getenv("FDT_TUNE_BASE");
  const float split_penalty = getenv("FDT_TUNE_SPLIT_PENALTY") ? (float)atof(getenv("FDT_TUNE_SPLIT_PENALTY")) : 0.0f;
  // The in-kernel split-K combine (conv.h) is a candidate only on request (FDT_TUNE_COMBINE=1): measured, it wins a tenth of
  // the split layers in isolation and nothing in the multi-stream step (docs/EXPERIMENTS.md R3-4)
  const bool no_combine = !(getenv("FDT_TUNE_COMBINE") && atoi(getenv("FDT_TUNE_COMBINE")) != 0);
  for (auto& op : m->ops) {
    if (op.type != OP_CONV) continue;
    if (only && *only && op.name.find(only) == std::string::npos) continue;
    if (only_base && *only_base) {     // FDT_TUNE_BASE=0,1: only the layers of these base classes (conv.h: enum ConvKind)
      bool hit = false;
      for (const char* q = only_base; *q;) {
        char* e = nullptr;
        const long v = strtol(q, &e, 10);
        if (e == q) break;
        hit = hit || v == (long)conv_base_kind(op.kind);
        q = *e ? e + 1 : e;
      }
      if (!hit) continue;
    }
    struct Cand { int kind, tile, split, combine; float ms; };
    std::vector<Cand> cands;
    long long ws_need = 0, cnt_need = 0;
    const ConvKind base = conv_base_kind(op.kind);
    for (int k = 0; k < CONV_KIND_COUNT; ++k) {
      if (conv_base_kind((ConvKind)k) != base) continue;    // e.g. direct and Winograd 3x3/s1
      const int nstages = ceil_div(op.ca.Cin, conv_geom((ConvKind)k).kc);
      for (int t = 0; t < CONV_TILE_COUNT; ++t) {
        if (!conv_supported((ConvKind)k, (ConvTile)t)) continue;
        if (small_lds > 0 && !conv_geom((ConvKind)k).wino && (long long)conv_lds_bytes((ConvKind)k, (ConvTile)t) > small_lds)
          continue;
        for (int split = 1; split <= 64; split *= 2) {
          if (split > 1 && (split > nstages / 2 || nstages < 8)) break;
          ConvArgs a = op.ca;
          a.ksplit = split;
          {
            ConvArgs probe = a;         // shape limits of the class (e.g. the persistent 1x1 kernel: no split-K, no fused upsample)
            probe.ws = split > 1 ? (float*)16 : nullptr;
            probe.sk_count = nullptr;
            if (!conv_shape_supported((ConvKind)k, (ConvTile)t, probe)) continue;
          }
          long long wsf = conv_ws_floats(a);
          if (wsf > kMaxWsFloats) break;
          ws_need = std::max(ws_need, wsf);
          cands.push_back({k, t, split, 0, 0.f});
          a.ws = (float*)16;
          if (!no_combine && !op.head && conv_combine_supported((ConvKind)k, (ConvTile)t, a)) {
            cnt_need = std::max(cnt_need, conv_sk_counters((ConvKind)k, (ConvTile)t, a));
            cands.push_back({k, t, split, 1, 0.f});
          }
        }
      }
    }
    if (cands.empty()) continue;   // nothing to choose from (every variant filtered out): the layer keeps its kernel
    float* tmp_ws = nullptr;
    if (ws_need && hipMalloc((void**)&tmp_ws, (size_t)ws_need * 4) != hipSuccess) {
      set_error("fdt_model_autotune: workspace allocation failed");
      rc = FDT_ERR_HIP;
      break;
    }
    unsigned* tmp_cnt = nullptr;
    if (cnt_need && (hipMalloc((void**)&tmp_cnt, (size_t)cnt_need * sizeof(unsigned)) != hipSuccess ||
                     hipMemsetAsync(tmp_cnt, 0, (size_t)cnt_need * sizeof(unsigned), st) != hipSuccess)) {
      set_error("fdt_model_autotune: counter allocation failed");
      rc = FDT_ERR_HIP;
      if (tmp_ws) (void)hipFree(tmp_ws);
      break;
    }
    Cand best{(int)op.kind, (int)op.tile, op.ca.ksplit, op.combine ? 1 : 0, 1e30f};
    for (auto& c : cands) {
      DevW dw;
      rc = Builder::device_weights(m, op.name, (ConvKind)c.kind, (ConvTile)c.tile, dw);
      if (rc != FDT_OK) break;
      ConvArgs a = op.ca;
      a.w = dw.w;
      a.bias = dw.bias;
      a.ksplit = c.split;
      a.ws = c.split > 1 ? tmp_ws : nullptr;
      a.sk_count = c.combine ? tmp_cnt : nullptr;
      a.defer_reduce = op.head && c.split > 1 ? 1 : 0;   // a head conv's reduce pass is the grouped finalize's business
      float best_ms = 1e30f;
      for (int it = 0; it < iters + 1 && rc == FDT_OK; ++it) {
        (void)hipEventRecord(e0, st);
        rc = launch_conv((ConvKind)c.kind, (ConvTile)c.tile, a, st, m->device);
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) { set_error("autotune: kernel failed"); rc = FDT_ERR_HIP; }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it > 0) best_ms = std::min(best_ms, ms);
      }
      if (rc != FDT_OK) break;
      // Experiment hook FDT_TUNE_SPLIT_PENALTY=p: rank candidates by ms * (1 + p * log2(split)) -- split-K buys isolated latency
      // with extra partial-sum traffic, which a pipeline that keeps several frames in flight may not want to pay
      c.ms = best_ms * (1.0f + split_penalty * std::log2((float)c.split));
      if (c.ms < best.ms) best = c;
    }
    // the workgroup map of the winner (conv.h: rows / XCD-spatial / XCD-channel)
    int best_map = CONV_MAP_ROWS;
    if (rc == FDT_OK) {
      DevW dwb;
      rc = Builder::device_weights(m, op.name, (ConvKind)best.kind, (ConvTile)best.tile, dwb);
      ConvArgs a = op.ca;
      a.w = dwb.w;
      a.bias = dwb.bias;
      a.ksplit = best.split;
      a.ws = best.split > 1 ? tmp_ws : nullptr;
      a.sk_count = best.combine ? tmp_cnt : nullptr;
      a.defer_reduce = op.head && best.split > 1 ? 1 : 0;
      float map_ms[CONV_MAP_COUNT] = {1e30f, 1e30f, 1e30f, 1e30f};
      for (int mm = 0; mm < CONV_MAP_COUNT && rc == FDT_OK; ++mm) {
        a.map_mode = mm;
        for (int it = 0; it < iters + 2 && rc == FDT_OK; ++it) {
          (void)hipEventRecord(e0, st);
          rc = launch_conv((ConvKind)best.kind, (ConvTile)best.tile, a, st, m->device);
          (void)hipEventRecord(e1, st);
          if (hipEventSynchronize(e1) != hipSuccess) { set_error("autotune: kernel failed"); rc = FDT_ERR_HIP; }
          float ms = 0;
          (void)hipEventElapsedTime(&ms, e0, e1);
          if (it > 0) map_ms[mm] = std::min(map_ms[mm], ms);
        }
      }
      // an XCD-aware map that is no slower (within timer noise) is preferred: it re-fetches less from HBM
      for (int mm = 1; mm < CONV_MAP_COUNT; ++mm)
        if (map_ms[mm] <= 1.005f * map_ms[CONV_MAP_ROWS] && map_ms[mm] < map_ms[best_map == CONV_MAP_ROWS ? mm : best_map] * 1.0001f)
          best_map = mm;
    }
    if (tmp_ws) (void)hipFree(tmp_ws);
    if (tmp_cnt) (void)hipFree(tmp_cnt);
    if (rc != FDT_OK) break;
    DevW dw;
    rc = Builder::device_weights(m, op.name, (ConvKind)best.kind, (ConvTile)best.tile, dw);
    if (rc != FDT_OK) break;
    op.kind = (ConvKind)best.kind;
    op.tile = (ConvTile)best.tile;
    op.ca.ksplit = best.split;
    op.ca.map_mode = best_map;
    op.ca.w = dw.w;
    op.ca.bias = dw.bias;
    op.needs_ws = best.split > 1;
    op.combine = best.combine != 0;
    m->hints[op.name] = {best.kind, best.tile, best.split, best_map, best.combine};
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  FDT_TRY(rc);
  m->hB = m->pB;
  m->hH = m->pH;
  m->hW = m->pW;
  // final shared workspace
  long long need = 0;
  for (auto& op : m->ops)
    if (op.type == OP_CONV && op.needs_ws) need = std::max(need, conv_ws_floats(op.ca));
  FDT_TRY(grow_conv_workspace(m, need));
  for (auto& op : m->ops)
    if (op.type == OP_CONV) op.ca.ws = op.needs_ws ? m->d_convws : nullptr;
  long long cneed = 0;
  for (auto& op : m->ops)
    if (op.type == OP_CONV && op.combine) cneed = std::max(cneed, conv_sk_counters(op.kind, op.tile, op.ca));
  if (cneed > m->sk_counters || (cneed && !m->d_skcnt)) {
    unsigned* p = nullptr;
    FDT_HIP(hipMalloc((void**)&p, (size_t)cneed * sizeof(unsigned)));
    FDT_HIP(hipMemsetAsync(p, 0, (size_t)cneed * sizeof(unsigned), m->stream));   // never the legacy stream (include/fdt.h)
    FDT_HIP(hipStreamSynchronize(m->stream));
    m->plan_allocs.push_back(p);
    m->d_skcnt = p;
    m->sk_counters = cneed;
  }
  for (auto& op : m->ops)
    if (op.type == OP_CONV) op.ca.sk_count = op.combine ? m->d_skcnt : nullptr;
  FDT_TRY(setup_heads(m));   // the head convs' own slab regions and the grouped finalize table follow the new splits
  FDT_TRY(plan_reduces(m, m->pB));
  m->drop_graphs();   // captured forwards bake in the old kernel choice
  return FDT_OK;
}

// Plan hints as text: one "layer kind tile split map [1]" line per conv layer (trailing 1: in-kernel split-K combine), first
// line "shape B H W".
extern "C" int fdt_model_export_plan(fdt_model* m, char* buf, int buflen, int* needed) {
  FDT_REQUIRE(m && needed, FDT_ERR_ARG, "fdt_model_export_plan: bad argument");
  FDT_REQUIRE(m->pB > 0, FDT_ERR_STATE, "fdt_model_export_plan: no plan yet");
  std::string out = "shape " + std::to_string(m->pB) + " " + std::to_string(m->pH) + " " + std::to_string(m->pW) + "\n";
  for (auto& op : m->ops)
    if (op.type == OP_CONV)
      out += op.name + " " + std::to_string((int)op.kind) + " " + std::to_string((int)op.tile) + " " +
             std::to_string(op.ca.ksplit) + " " + std::to_string(op.ca.map_mode) + (op.combine ? " 1" : "") + "\n";
  *needed = (int)out.size() + 1;
  if (buf && buflen >= *needed) memcpy(buf, out.c_str(), out.size() + 1);
  return FDT_OK;
}

extern "C" int fdt_model_import_plan(fdt_model* m, const char* text) {
  FDT_REQUIRE(m && text, FDT_ERR_ARG, "fdt_model_import_plan: bad argument");
  std::map<std::string, fdt_model::Hint> hints;
  int B = 0, H = 0, W = 0;
  const char* p = text;
  while (*p) {
    const char* e = strchr(p, '\n');
    std::string line = e ? std::string(p, e - p) : std::string(p);
    p = e ? e + 1 : p + line.size();
    if (line.empty()) continue;
    char name[256];
    int a = 0, b = 0, c = 0, d = 0, cb = 0, nf = 0;
    if (sscanf(line.c_str(), "shape %d %d %d", &a, &b, &c) == 3) {
      B = a; H = b; W = c;
    } else if ((nf = sscanf(line.c_str(), "%255s %d %d %d %d %d", name, &a, &b, &c, &d, &cb)) >= 4) {
      if (nf == 4) d = CONV_MAP_ROWS;      // plans written before the workgroup map became a choice
      if (nf <= 5) cb = 0;                 // ... and before the in-kernel split-K combine existed
      FDT_REQUIRE(a >= 0 && a < CONV_KIND_COUNT && b >= 0 && b < CONV_TILE_COUNT && c >= 1 && c <= 4096 &&
                      d >= CONV_MAP_ROWS && d < CONV_MAP_COUNT && (cb == 0 || cb == 1),
                  FDT_ERR_ARG, "fdt_model_import_plan: bad entry '%s'", line.c_str());
      hints[name] = {a, b, c, d, cb};
    } else {
      set_error("fdt_model_import_plan: cannot parse '%s'", line.c_str());
      return FDT_ERR_ARG;
    }
  }
  FDT_REQUIRE(B > 0 && H > 0 && W > 0, FDT_ERR_ARG, "fdt_model_import_plan: missing shape line");
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  m->free_plan();          // the next forward at this shape builds the plan from the hints
  m->hints = std::move(hints);
  m->hB = B;
  m->hH = H;
  m->hW = W;
  return FDT_OK;
}

// Replay the forward as a captured HIP graph (default on; env FDT_GRAPH=0 turns it off at create time).
extern "C" int fdt_model_enable_graph(fdt_model* m, int on) {
  FDT_REQUIRE(m, FDT_ERR_ARG, "fdt_model_enable_graph: null handle");
  m->use_graph = on != 0;
  if (!on) {
    if (m->stream) (void)fdt::device_sync();
    m->drop_graphs();
  }
  return FDT_OK;
}

extern "C" int fdt_model_profile_enable(fdt_model* m, int on) {
  FDT_REQUIRE(m, FDT_ERR_ARG, "fdt_model_profile_enable: null handle");
  m->profile = on != 0;
  m->seg_first = m->seg_last = -1;
  return FDT_OK;
}

// Segment timing: ONE event pair around the contiguous ops [first_op, last_op] (indices as in fdt_model_profile_read) of
// every following forward, instead of an event in front of every op; first_op < 0 switches profiling off.
extern "C" int fdt_model_profile_segment(fdt_model* m, int first_op, int last_op) {
  FDT_REQUIRE(m, FDT_ERR_ARG, "fdt_model_profile_segment: null handle");
  if (first_op < 0) {
    m->profile = false;
    m->seg_first = m->seg_last = -1;
    return FDT_OK;
  }
  FDT_REQUIRE(m->pB > 0 && !m->ops.empty(), FDT_ERR_STATE, "fdt_model_profile_segment: run a forward first");
  FDT_REQUIRE(first_op <= last_op && last_op < (int)m->ops.size(), FDT_ERR_ARG,
              "fdt_model_profile_segment: ops [%d, %d] outside the plan's %d ops", first_op, last_op, (int)m->ops.size());
  FDT_HIP(hipSetDevice(m->device));
  for (auto& e : m->seg_ev)
    if (!e) FDT_HIP(hipEventCreate(&e));
  m->profile = true;
  m->seg_first = first_op;
  m->seg_last = last_op;
  return FDT_OK;
}

extern "C" int fdt_model_profile_segment_ms(fdt_model* m, float* ms) {
  FDT_REQUIRE(m && ms, FDT_ERR_ARG, "fdt_model_profile_segment_ms: bad argument");
  FDT_REQUIRE(m->profile && m->seg_first >= 0 && m->seg_ev[1], FDT_ERR_STATE,
              "fdt_model_profile_segment_ms: no segment set (fdt_model_profile_segment) or no forward since");
  FDT_HIP(hipEventSynchronize(m->seg_ev[1]));
  FDT_HIP(hipEventElapsedTime(ms, m->seg_ev[0], m->seg_ev[1]));
  return FDT_OK;
}

extern "C" int fdt_model_profile_read(fdt_model* m, int max, char* names, float* ms, double* flops, int* n) {
  FDT_REQUIRE(m && n, FDT_ERR_ARG, "fdt_model_profile_read: bad argument");
  FDT_REQUIRE(m->profile && m->ev.size() == m->ops.size() + 3, FDT_ERR_STATE,
              "fdt_model_profile_read: profiling not enabled or no forward since enabling");
  FDT_HIP(hipStreamSynchronize(m->stream));
  const int nops = (int)m->ops.size();
  int cnt = nops + 2;   // + the Detect stage + the ingest kernel (u8 -> f32 NCHW, optional resize) in front
  *n = cnt;
  for (int i = 0; i < cnt && i < max; ++i) {
    float t = 0.f;
    hipError_t e = i <= nops ? hipEventElapsedTime(&t, m->ev[i], m->ev[i + 1])
                             : hipEventElapsedTime(&t, m->ev[nops + 2], m->ev[0]);
    if (e != hipSuccess) t = 0.f;
    if (ms) ms[i] = t;
    if (flops) flops[i] = i < nops ? m->ops[i].flops : 0.0;
    if (names) {
      std::string nm = i < nops ? m->ops[i].name : std::string(i == nops ? "detect" : "ingest");
      if (i < (int)m->ops.size() && m->ops[i].type == OP_CONV) {
        const Op& o_ = m->ops[i];
        const bool u8 = o_.u8_stem && m->last_fused;      // the raw-frame stem ran in its place
        if (u8 && o_.u8_stream)
          nm += ".u8_stream";                               // stream_ir.hip: not a conv class (no "#k" suffix: a vector-ALU kernel)
        else
          nm += "#k" + std::to_string((int)(u8 ? o_.u8_kind : o_.kind)) + "t" + std::to_string((int)(u8 ? o_.u8_tile : o_.tile)) + "s" +
                std::to_string(u8 ? 1 : o_.ca.ksplit);
      }
      snprintf(names + (size_t)i * 48, 48, "%s", nm.c_str());
    }
  }
  return FDT_OK;
}

// Algorithmic (un-fused lower bound) HBM bytes of ONE forward of the current plan: every op reads its input tensor(s)
// and writes its output once (f32), every conv reads its weights once.  SURVEY.md 8(d): the per-unit figure the HBM
// roofline of the bandwidth-bound nets (FaceBoxes, the depthwise half of try3) is computed from.  Per-op values follow
// the order of fdt_model_profile_read (ops..., then 0 for "detect" / "ingest").
extern "C" int fdt_model_traffic(fdt_model* m, double* act_bytes, double* weight_bytes, int max, double* per_op,
                                 int* n) {
  FDT_REQUIRE(m, FDT_ERR_ARG, "fdt_model_traffic: null handle");
  FDT_REQUIRE(m->pB > 0 && !m->ops.empty(), FDT_ERR_STATE, "fdt_model_traffic: no forward has run yet");
  double act = 0, wts = 0;
  int i = 0;
  for (auto& op : m->ops) {
    double b = 0, w = 0;
    auto tb = [&](int t) { return t >= 0 ? 4.0 * m->pB * m->tensors[t].C * m->tensors[t].H * m->tensors[t].W : 0.0; };
    if (op.type == OP_CONV) {
      const ConvGeom g = conv_geom(conv_base_kind(op.kind));
      b = 4.0 * op.ca.B * ((double)op.ca.Cin * op.ca.Hin * op.ca.Win + (double)op.ca.Cout * op.ca.Hout * op.ca.Wout);
      if (op.ca.res) b += 4.0 * op.ca.B * (double)op.ca.Cout * op.ca.Hout * op.ca.Wout;
      if (op.ca.up) b += 4.0 * op.ca.B * (double)op.ca.Cout * op.ca.up_h * op.ca.up_w;
      w = 4.0 * ((double)op.ca.Cout * op.ca.Cin * g.kh * g.kw + op.ca.Cout);
    } else if (op.type == OP_EXPDW) {
      b = tb(op.in_t) + tb(op.out_t);        // (the residual of a whole-block launch is the staged input itself)
      w = 4.0 * op.hid * (m->tensors[op.in_t].C + 1 + 9 + 1) + 4.0 * op.oup * (op.hid + 1);
    } else if (op.type == OP_DWPROJ) {
      b = tb(op.in_t) + tb(op.out_t) + tb(op.in2_t);
      w = 4.0 * m->tensors[op.in_t].C * (9 + 1 + op.oup) + 4.0 * op.oup;
    } else if (op.type == OP_HEADFIN || op.type == OP_MBOXFIN) {
      b = 2.0 * tb(op.in_t);
    } else {
      b = tb(op.in_t) + tb(op.out_t);
      if (op.type == OP_DW) w = 4.0 * m->tensors[op.in_t].C * (op.ksize * op.ksize + 1);
    }
    act += b;
    wts += w;
    if (per_op && i < max) per_op[i] = b + w;
    ++i;
  }
  if (per_op)
    for (int k = i; k < i + 2 && k < max; ++k) per_op[k] = 0.0;
  if (n) *n = i + 2;
  if (act_bytes) *act_bytes = act;
  if (weight_bytes) *weight_bytes = wts;
  return FDT_OK;
}

extern "C" int fdt_model_flops(fdt_model* m, double* flops) {
  FDT_REQUIRE(m && flops, FDT_ERR_ARG, "fdt_model_flops: bad argument");
  FDT_REQUIRE(m->pB > 0, FDT_ERR_STATE, "fdt_model_flops: no forward has run yet");
  *flops = m->flops_per_frame;
  return FDT_OK;
}
