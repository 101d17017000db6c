// fdt_pipeline: the multi-stream detect + track loop of one GPU behind the C ABI (round 4).
//
// What bench.py times and what iouTracke_cal.py:117-156 is per frame -- detect(i) on slot i % inflight's handle and
// stream, the one exchange of the path, the sequential association on one more stream, HIP events ordering
//     detect(step i, slot k) -> exchange + track(step i) -> detect(step i + inflight, slot k)
// -- used to live in Python (pipeline.py: torch streams / events / tensors), so a C caller could reach every kernel but not
// the arrangement the headline number comes from.  This object owns that arrangement: `inflight` model handles (the caller's
// + fdt_model_clone()s: one weight copy), their streams, the per-slot Detect records and candidate counts, the device
// tracker, the events; frames are device pointers, nothing synchronises with the host until fdt_pipeline_sync /
// fdt_pipeline_tracker.  Built ONLY from the library's own C entry points (include/fdt.h): everything here is reachable
// without it, it is the shortest correct way to string them together.
#include <vector>

#include "common.h"

struct fdt_pipeline {
  int device = 0;
  int H = 0, W = 0, B = 1, NF = 1, world = 1, rank = 0, top_k = 0, src_h = 0, src_w = 0;
  long long REC = 0;                       // floats of one frame's Detect record [2, top_k, 5]
  float score_thresh = 0.4f;
  std::vector<fdt_model*> nets;            // [0] is the caller's, the rest are clones owned here
  std::vector<hipStream_t> det;
  hipStream_t trk = nullptr;
  std::vector<hipEvent_t> det_done, trk_done;
  std::vector<float*> mine, gathered;      // per slot: this rank's records [B][REC]; all ranks' [world][B][REC] (== mine at world 1)
  std::vector<int*> counts;                // per slot: [B][2]
  std::vector<unsigned char*> stage;       // per slot: [B][h][w][3] u8, frames handed over one at a time (step_frame)
  std::vector<unsigned char*> pinned;      // per slot: pinned host landing buffer of step_host (lazily allocated)
  std::vector<hipEvent_t> h2d_done;        // per slot: the H2D copy out of `pinned` has completed
  fdt_tracker* tracker = nullptr;
  fdt_comm* comm = nullptr;                // borrowed
  long long pend_group = -1;               // step_frame: the partly filled batch
  int pend_slot = 0, pend_n = 0;
  hipEvent_t mark[2] = {nullptr, nullptr};
};

namespace {
using fdt::set_error;

int forward_slot(fdt_pipeline* p, int k, const void* frames_dev) {
  if (p->src_h > 0)
    return fdt_model_forward_resized(p->nets[k], frames_dev, 1, p->B, p->src_h, p->src_w, p->H, p->W, p->mine[k], p->counts[k],
                                     (void*)p->det[k]);
  return fdt_model_forward_dev(p->nets[k], frames_dev, FDT_FRAME_U8_HWC_BGR, p->B, p->H, p->W, p->mine[k], p->counts[k],
                               (void*)p->det[k]);
}

// exchange + association of slot k's batch on the tracker stream; n_valid < B: a partly filled batch (the tail of a video)
int track_slot(fdt_pipeline* p, int k, int n_valid, bool frame_major) {
  FDT_HIP(hipEventRecord(p->det_done[k], p->det[k]));
  FDT_HIP(hipStreamWaitEvent(p->trk, p->det_done[k], 0));
  if (p->world > 1)
    FDT_TRY(fdt_allgather_dets(p->comm, 0, p->mine[k], p->gathered[k], (long long)p->B * p->REC, (void*)p->trk));
  if (!frame_major) {
    // step(): the gathered records are consumed in memory order (rank-major: rank r's batch holds consecutive frames)
    FDT_TRY(fdt_tracker_step_dev_multi(p->tracker, p->gathered[k], p->world * p->B, p->REC, 2, p->top_k, p->W, p->H,
                                       p->score_thresh, (void*)p->trk));
  } else {
    // step_frame(): frame (j, r) = j * world + r sits at rank r's batch entry j
    for (int j = 0; j < n_valid; ++j)
      FDT_TRY(fdt_tracker_step_dev_multi(p->tracker, p->gathered[k] + (long long)j * p->REC, p->world, (long long)p->B * p->REC, 2,
                                         p->top_k, p->W, p->H, p->score_thresh, (void*)p->trk));
  }
  FDT_HIP(hipEventRecord(p->trk_done[k], p->trk));
  return FDT_OK;
}
}  // namespace

extern "C" fdt_pipeline* fdt_pipeline_create(fdt_model* m, int device, int height, int width, int inflight, int batch,
                                             const char* plan_text, fdt_comm* comm, int rank, int world, int src_h, int src_w,
                                             float score_thresh, double sigma_iou, double sigma_h, int t_min, int log_frames) {
  if (!m || height < 1 || width < 1 || inflight < 1 || batch < 1 || world < 1 || rank < 0 || rank >= world ||
      (world > 1 && !comm) || (src_h > 0) != (src_w > 0)) {
    set_error("fdt_pipeline_create: bad argument (a communicator is required for world > 1)");
    return nullptr;
  }
  int top_k = 0;
  if (fdt_model_get_detect(m, &top_k, nullptr, nullptr, nullptr) != FDT_OK || top_k < 1) return nullptr;
  if (hipSetDevice(device) != hipSuccess) {
    set_error("fdt_pipeline_create: hipSetDevice(%d) failed", device);
    return nullptr;
  }
  fdt_pipeline* p = new fdt_pipeline();
  p->device = device;
  p->H = height; p->W = width; p->B = batch; p->NF = inflight; p->world = world; p->rank = rank;
  p->src_h = src_h; p->src_w = src_w;
  p->top_k = top_k;
  p->REC = 2ll * top_k * 5;
  p->score_thresh = score_thresh;
  p->comm = comm;
  bool ok = true;
  if (plan_text && *plan_text) ok = fdt_model_import_plan(m, plan_text) == FDT_OK;
  p->nets.push_back(m);
  for (int k = 1; ok && k < inflight; ++k) {
    fdt_model* c = fdt_model_clone(m);       // copies the Detect / PriorBox settings and the plan hints
    ok = c != nullptr;
    if (c) p->nets.push_back(c);
  }
  const size_t rec_bytes = (size_t)batch * p->REC * 4;
  for (int k = 0; ok && k < inflight; ++k) {
    hipStream_t s = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float *mine = nullptr, *all = nullptr;
    int* cnt = nullptr;
    ok = hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&e0, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&e1, hipEventDisableTiming) == hipSuccess &&
         hipMalloc((void**)&mine, rec_bytes) == hipSuccess && hipMalloc((void**)&cnt, (size_t)batch * 2 * 4) == hipSuccess &&
         (world == 1 || hipMalloc((void**)&all, rec_bytes * world) == hipSuccess);
    if (s) p->det.push_back(s);
    if (e0) p->det_done.push_back(e0);
    if (e1) p->trk_done.push_back(e1);
    if (mine) p->mine.push_back(mine);
    if (cnt) p->counts.push_back(cnt);
    p->gathered.push_back(world == 1 ? mine : all);
    p->stage.push_back(nullptr);
    p->pinned.push_back(nullptr);
    p->h2d_done.push_back(nullptr);
    if (ok) ok = hipMemsetAsync(mine, 0, rec_bytes, s) == hipSuccess && hipMemsetAsync(cnt, 0, (size_t)batch * 2 * 4, s) == hipSuccess;
  }
  ok = ok && hipStreamCreateWithFlags(&p->trk, hipStreamNonBlocking) == hipSuccess &&
       hipEventCreate(&p->mark[0]) == hipSuccess && hipEventCreate(&p->mark[1]) == hipSuccess;
  if (ok) {
    const int lf = log_frames > world * batch ? log_frames : world * batch;
    p->tracker = fdt_tracker_create(sigma_iou, sigma_h, t_min, 2 * top_k, lf);
    ok = p->tracker != nullptr;
  }
  if (ok) ok = hipDeviceSynchronize() == hipSuccess;
  if (!ok) {
    if (!*fdt::get_error()) set_error("fdt_pipeline_create: HIP resource creation failed: %s", hipGetErrorString(hipGetLastError()));
    fdt_pipeline_destroy(p);
    return nullptr;
  }
  return p;
}

extern "C" void fdt_pipeline_destroy(fdt_pipeline* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  (void)hipDeviceSynchronize();
  if (p->tracker) fdt_tracker_destroy(p->tracker);
  for (size_t k = 1; k < p->nets.size(); ++k) fdt_model_destroy(p->nets[k]);
  for (size_t k = 0; k < p->mine.size(); ++k) {
    if (p->world > 1 && k < p->gathered.size() && p->gathered[k]) (void)hipFree(p->gathered[k]);
    (void)hipFree(p->mine[k]);
  }
  for (auto c : p->counts) (void)hipFree(c);
  for (auto s : p->stage)
    if (s) (void)hipFree(s);
  for (auto s : p->pinned)
    if (s) (void)hipHostFree(s);
  for (auto e : p->h2d_done)
    if (e) (void)hipEventDestroy(e);
  for (auto e : p->det_done) (void)hipEventDestroy(e);
  for (auto e : p->trk_done) (void)hipEventDestroy(e);
  for (auto e : p->mark)
    if (e) (void)hipEventDestroy(e);
  for (auto s : p->det) (void)hipStreamDestroy(s);
  if (p->trk) (void)hipStreamDestroy(p->trk);
  delete p;
}

// One-time initialisation: every slot runs its forward twice (plan + weight tiling, then the HIP-graph capture keyed by the
// slot's own record / counts buffers); the tracker is not fed.
extern "C" int fdt_pipeline_prime(fdt_pipeline* p, const void* frames_dev) {
  FDT_REQUIRE(p && frames_dev, FDT_ERR_ARG, "fdt_pipeline_prime: bad argument");
  FDT_HIP(hipSetDevice(p->device));
  for (int k = 0; k < p->NF; ++k) {
    FDT_TRY(forward_slot(p, k, frames_dev));
    FDT_TRY(forward_slot(p, k, frames_dev));
  }
  FDT_HIP(hipDeviceSynchronize());
  return FDT_OK;
}

extern "C" int fdt_pipeline_step(fdt_pipeline* p, long long i, const void* frames_dev) {
  FDT_REQUIRE(p && frames_dev && i >= 0, FDT_ERR_ARG, "fdt_pipeline_step: bad argument");
  const int k = (int)(i % p->NF);
  FDT_HIP(hipStreamWaitEvent(p->det[k], p->trk_done[k], 0));     // the slot's previous record was consumed
  FDT_TRY(forward_slot(p, k, frames_dev));
  return track_slot(p, k, p->B, false);
}

// Step i from HOST frames (what cv2's capture hands over, iouTracke_cal.py:119-124): `batch` pageable uint8 frames are copied into
// the slot's pinned landing buffer (the caller may reuse its memory at once), the H2D copy runs on the slot's stream in front
// of its forward (a copy stream of its own can land on another slot's hardware queue and wait behind a whole forward there:
// DESIGN.md section 5), then as fdt_pipeline_step.  n_valid < batch: a partly filled last batch (only the first n_valid
// frames are shown to the tracker).
extern "C" int fdt_pipeline_step_host(fdt_pipeline* p, long long i, const void* frames_host, int n_valid) {
  FDT_REQUIRE(p && frames_host && i >= 0 && n_valid >= 1 && n_valid <= p->B, FDT_ERR_ARG, "fdt_pipeline_step_host: bad argument");
  const int k = (int)(i % p->NF);
  const size_t fb = (size_t)(p->src_h > 0 ? p->src_h : p->H) * (p->src_h > 0 ? p->src_w : p->W) * 3 * p->B;
  if (!p->pinned[k]) {
    FDT_HIP(hipHostMalloc((void**)&p->pinned[k], fb, hipHostMallocDefault));
    FDT_HIP(hipEventCreateWithFlags(&p->h2d_done[k], hipEventDisableTiming));
    if (!p->stage[k]) FDT_HIP(hipMalloc((void**)&p->stage[k], fb));
  } else {
    FDT_HIP(hipEventSynchronize(p->h2d_done[k]));             // the previous copy out of the landing buffer (long done)
  }
  memcpy(p->pinned[k], frames_host, fb);
  FDT_HIP(hipStreamWaitEvent(p->det[k], p->trk_done[k], 0));  // the slot's previous record was consumed
  FDT_HIP(hipMemcpyAsync(p->stage[k], p->pinned[k], fb, hipMemcpyHostToDevice, p->det[k]));
  FDT_HIP(hipEventRecord(p->h2d_done[k], p->det[k]));
  FDT_TRY(forward_slot(p, k, p->stage[k]));
  return track_slot(p, k, n_valid, n_valid < p->B);
}

extern "C" int fdt_pipeline_flush(fdt_pipeline* p) {
  FDT_REQUIRE(p, FDT_ERR_ARG, "fdt_pipeline_flush: null handle");
  if (p->pend_group < 0 || p->pend_n == 0) return FDT_OK;
  const int k = p->pend_slot, n = p->pend_n;
  p->pend_group = -1;
  p->pend_n = 0;
  FDT_TRY(forward_slot(p, k, p->stage[k]));
  return track_slot(p, k, n, true);
}

// Frames handed over ONE AT A TIME, executed `batch` at a time (cross-frame grouped launches, see pipeline.py: step_frame).
extern "C" int fdt_pipeline_step_frame(fdt_pipeline* p, long long i, const void* frame_dev) {
  FDT_REQUIRE(p && frame_dev && i >= 0, FDT_ERR_ARG, "fdt_pipeline_step_frame: bad argument");
  const long long g = i / p->B;
  const int k = (int)(g % p->NF), j = (int)(i % p->B);
  const size_t fb = (size_t)(p->src_h > 0 ? p->src_h : p->H) * (p->src_h > 0 ? p->src_w : p->W) * 3;
  if (!p->stage[k]) FDT_HIP(hipMalloc((void**)&p->stage[k], fb * p->B));
  if (j == 0) FDT_HIP(hipStreamWaitEvent(p->det[k], p->trk_done[k], 0));
  FDT_HIP(hipMemcpyAsync(p->stage[k] + fb * j, frame_dev, fb, hipMemcpyDeviceToDevice, p->det[k]));
  p->pend_group = g;
  p->pend_slot = k;
  p->pend_n = j + 1;
  if (j == p->B - 1) return fdt_pipeline_flush(p);
  return FDT_OK;
}

extern "C" int fdt_pipeline_sync(fdt_pipeline* p) {
  FDT_REQUIRE(p, FDT_ERR_ARG, "fdt_pipeline_sync: null handle");
  FDT_HIP(hipSetDevice(p->device));
  FDT_HIP(hipDeviceSynchronize());
  return FDT_OK;
}

extern "C" fdt_tracker* fdt_pipeline_tracker(fdt_pipeline* p) { return p ? p->tracker : nullptr; }

extern "C" int fdt_pipeline_slot(fdt_pipeline* p, int slot, fdt_model** model, void** det_stream, float** record_dev,
                                 float** gathered_dev, int** counts_dev) {
  FDT_REQUIRE(p && slot >= 0 && slot < p->NF, FDT_ERR_ARG, "fdt_pipeline_slot: bad slot");
  if (model) *model = p->nets[slot];
  if (det_stream) *det_stream = (void*)p->det[slot];
  if (record_dev) *record_dev = p->mine[slot];
  if (gathered_dev) *gathered_dev = p->gathered[slot];
  if (counts_dev) *counts_dev = p->counts[slot];
  return FDT_OK;
}

// which = 0 / 1: a timing event on the TRACKER stream (everything of the steps enqueued so far precedes it there)
extern "C" int fdt_pipeline_mark(fdt_pipeline* p, int which) {
  FDT_REQUIRE(p && (which == 0 || which == 1), FDT_ERR_ARG, "fdt_pipeline_mark: bad argument");
  FDT_HIP(hipEventRecord(p->mark[which], p->trk));
  return FDT_OK;
}

extern "C" int fdt_pipeline_elapsed_ms(fdt_pipeline* p, float* ms) {
  FDT_REQUIRE(p && ms, FDT_ERR_ARG, "fdt_pipeline_elapsed_ms: bad argument");
  FDT_HIP(hipEventSynchronize(p->mark[1]));
  FDT_HIP(hipEventElapsedTime(ms, p->mark[0], p->mark[1]));
  return FDT_OK;
}

// ---- plain device buffers for callers without a GPU runtime of their own ------------------------------------------------
extern "C" int fdt_dev_malloc(void** ptr, long long bytes) {
  FDT_REQUIRE(ptr && bytes > 0, FDT_ERR_ARG, "fdt_dev_malloc: bad argument");
  FDT_HIP(hipMalloc(ptr, (size_t)bytes));
  return FDT_OK;
}
extern "C" int fdt_dev_free(void* ptr) {
  if (ptr) FDT_HIP(hipFree(ptr));
  return FDT_OK;
}
extern "C" int fdt_dev_upload(void* dst_dev, const void* src_host, long long bytes) {
  FDT_REQUIRE(dst_dev && src_host && bytes > 0, FDT_ERR_ARG, "fdt_dev_upload: bad argument");
  const hipStream_t st = fdt::thread_stream();
  FDT_REQUIRE(st, FDT_ERR_HIP, "fdt_dev_upload: could not create the calling thread's private stream");
  FDT_HIP(fdt::copy_sync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, st));
  return FDT_OK;
}
extern "C" int fdt_dev_download(void* dst_host, const void* src_dev, long long bytes) {
  FDT_REQUIRE(dst_host && src_dev && bytes > 0, FDT_ERR_ARG, "fdt_dev_download: bad argument");
  const hipStream_t st = fdt::thread_stream();
  FDT_REQUIRE(st, FDT_ERR_HIP, "fdt_dev_download: could not create the calling thread's private stream");
  FDT_HIP(fdt::copy_sync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, st));
  return FDT_OK;
}
