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
  // per-group completion stamps (fdt_pipeline_stamps_enable): a timing event on the tracker stream after the association of
  // every track_slot() call, for latency measurements; off by default (no event in the product loop)
  std::vector<hipEvent_t> stamps;
  long long n_stamped = 0;
  size_t frame_bytes() const { return (size_t)(src_h > 0 ? src_h : H) * (src_h > 0 ? src_w : W) * 3; }
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
  if (!p->stamps.empty() && p->n_stamped < (long long)p->stamps.size()) FDT_HIP(hipEventRecord(p->stamps[p->n_stamped++], p->trk));
  return FDT_OK;
}

// the pinned landing buffers + copy events of step_host, all slots at once on the first host step (nothing is left half
// allocated: on a failure the slots already made stay valid and the call can be repeated)
int ensure_host_landing(fdt_pipeline* p) {
  const size_t fb = p->frame_bytes() * p->B;
  for (int k = 0; k < p->NF; ++k) {
    if (p->pinned[k]) continue;
    unsigned char* pin = nullptr;
    hipEvent_t ev = nullptr;
    if (hipHostMalloc((void**)&pin, fb, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      if (pin) (void)hipHostFree(pin);
      set_error("fdt_pipeline_step_host: pinned landing buffer of slot %d (%zu bytes): %s", k, fb, hipGetErrorString(hipGetLastError()));
      return FDT_ERR_HIP;
    }
    p->h2d_done[k] = ev;
    p->pinned[k] = pin;              // set last: a slot is either complete or absent
  }
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
    // the staging batch of step_frame / step_host is part of the pipeline, not of its first timed step
    unsigned char* stg = nullptr;
    if (ok) ok = hipMalloc((void**)&stg, p->frame_bytes() * batch) == hipSuccess;
    p->stage.push_back(stg);
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
  if (ok) ok = fdt::device_sync() == hipSuccess;
  if (!ok) {
    if (!*fdt::get_error()) set_error("fdt_pipeline_create: HIP resource creation failed: %s", hipGetErrorString(hipGetLastError()));
    fdt_pipeline_destroy(p);
    return nullptr;
  }
  return p;
}

extern "C" void fdt_pipeline_destroy(fdt_pipeline* p) {
  if (!p) return;
  fdt::ExclusiveDevice quiet;
  (void)hipSetDevice(p->device);
  (void)fdt::device_sync();
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
  for (auto e : p->stamps)
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
  FDT_HIP(fdt::device_sync());
  return FDT_OK;
}

extern "C" int fdt_pipeline_step(fdt_pipeline* p, long long i, const void* frames_dev) {
  FDT_REQUIRE(p && frames_dev && i >= 0, FDT_ERR_ARG, "fdt_pipeline_step: bad argument");
  FDT_HIP(hipSetDevice(p->device));
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
  // ONE frame order per video: a step's frames are consumed rank-major (rank r's batch holds consecutive frames, as in
  // fdt_pipeline_step).  A partly filled batch keeps that order at world 1 (the first n_valid frames of the one rank); at
  // world > 1 the ranks would have to know each other's n_valid, which only the caller does: refused.
  FDT_REQUIRE(p->world == 1 || n_valid == p->B, FDT_ERR_ARG,
              "fdt_pipeline_step_host: a partly filled batch (%d of %d) is only defined at world 1", n_valid, p->B);
  FDT_HIP(hipSetDevice(p->device));
  const int k = (int)(i % p->NF);
  const size_t fb = p->frame_bytes() * p->B;
  if (!p->pinned[k]) FDT_TRY(ensure_host_landing(p));
  else FDT_HIP(hipEventSynchronize(p->h2d_done[k]));          // the previous copy out of the landing buffer (long done)
  memcpy(p->pinned[k], frames_host, p->frame_bytes() * n_valid);
  FDT_HIP(hipStreamWaitEvent(p->det[k], p->trk_done[k], 0));  // the slot's previous record was consumed
  FDT_HIP(hipMemcpyAsync(p->stage[k], p->pinned[k], fb, hipMemcpyHostToDevice, p->det[k]));
  FDT_HIP(hipEventRecord(p->h2d_done[k], p->det[k]));
  FDT_TRY(forward_slot(p, k, p->stage[k]));
  if (n_valid == p->B) return track_slot(p, k, p->B, false);
  // world 1, partly filled: the first n_valid records of the one rank, in order (rank-major == frame-major here)
  FDT_HIP(hipEventRecord(p->det_done[k], p->det[k]));
  FDT_HIP(hipStreamWaitEvent(p->trk, p->det_done[k], 0));
  FDT_TRY(fdt_tracker_step_dev_multi(p->tracker, p->gathered[k], n_valid, p->REC, 2, p->top_k, p->W, p->H, p->score_thresh,
                                     (void*)p->trk));
  FDT_HIP(hipEventRecord(p->trk_done[k], p->trk));
  return FDT_OK;
}

extern "C" int fdt_pipeline_flush(fdt_pipeline* p) {
  FDT_REQUIRE(p, FDT_ERR_ARG, "fdt_pipeline_flush: null handle");
  if (p->pend_group < 0 || p->pend_n == 0) return FDT_OK;
  FDT_HIP(hipSetDevice(p->device));
  const int k = p->pend_slot, n = p->pend_n;
  p->pend_group = -1;                        // the group is closed: its remaining frame indices cannot be handed over any more
  p->pend_n = 0;
  FDT_TRY(forward_slot(p, k, p->stage[k]));
  return track_slot(p, k, n, true);
}

// Frames handed over ONE AT A TIME, executed `batch` at a time (cross-frame grouped launches, see pipeline.py: step_frame).
// The frames of a group arrive in order, starting with its first (i % batch == 0): that one waits for the slot's previous
// association, and a group that fdt_pipeline_flush has run partly filled is closed -- continuing it would overwrite records the
// tracker stream may still be reading and show it the flushed frames again (FDT_ERR_STATE).  World > 1: every rank hands over
// (and flushes) the same number of frames; frame i of rank r is frame i * world + r of the video.
extern "C" int fdt_pipeline_step_frame(fdt_pipeline* p, long long i, const void* frame_dev) {
  FDT_REQUIRE(p && frame_dev && i >= 0, FDT_ERR_ARG, "fdt_pipeline_step_frame: bad argument");
  const long long g = i / p->B;
  const int k = (int)(g % p->NF), j = (int)(i % p->B);
  FDT_REQUIRE(j == 0 ? p->pend_n == 0 : (p->pend_group == g && p->pend_n == j), FDT_ERR_STATE,
              "fdt_pipeline_step_frame: frame %lld is entry %d of group %lld, but %s", i, j, g,
              j == 0 ? "the previous group is still open (hand its frames over, or fdt_pipeline_flush)"
                     : "that group is not the open one at that entry (frames of a group arrive in order; a flushed group is closed)");
  FDT_HIP(hipSetDevice(p->device));
  const size_t fb = p->frame_bytes();
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
  FDT_HIP(fdt::device_sync());
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
  FDT_HIP(hipSetDevice(p->device));
  FDT_HIP(hipEventRecord(p->mark[which], p->trk));
  return FDT_OK;
}

extern "C" int fdt_pipeline_elapsed_ms(fdt_pipeline* p, float* ms) {
  FDT_REQUIRE(p && ms, FDT_ERR_ARG, "fdt_pipeline_elapsed_ms: bad argument");
  FDT_HIP(hipSetDevice(p->device));
  FDT_HIP(hipEventSynchronize(p->mark[1]));
  FDT_HIP(hipEventElapsedTime(ms, p->mark[0], p->mark[1]));
  return FDT_OK;
}

// Completion stamps for latency measurements: from now on the next `n` exchange + association groups (one per fdt_pipeline_step
// / _step_host call, one per launched group of _step_frame) each record a timing event on the tracker stream behind the
// association.  n = 0 switches them off.  fdt_pipeline_stamps_read: milliseconds from mark(0) to each stamp recorded so far
// (waits for them); *count = how many.
extern "C" int fdt_pipeline_stamps_enable(fdt_pipeline* p, int n) {
  FDT_REQUIRE(p && n >= 0 && n <= (1 << 20), FDT_ERR_ARG, "fdt_pipeline_stamps_enable: bad argument");
  FDT_HIP(hipSetDevice(p->device));
  FDT_HIP(fdt::device_sync());
  for (auto e : p->stamps) (void)hipEventDestroy(e);
  p->stamps.clear();
  p->n_stamped = 0;
  for (int i = 0; i < n; ++i) {
    hipEvent_t e = nullptr;
    FDT_HIP(hipEventCreate(&e));
    p->stamps.push_back(e);
  }
  return FDT_OK;
}
extern "C" int fdt_pipeline_stamps_read(fdt_pipeline* p, float* ms_since_mark0, int max, int* count) {
  FDT_REQUIRE(p && ms_since_mark0 && count && max >= 0, FDT_ERR_ARG, "fdt_pipeline_stamps_read: bad argument");
  FDT_HIP(hipSetDevice(p->device));
  const int n = (int)(p->n_stamped < max ? p->n_stamped : max);
  for (int i = 0; i < n; ++i) {
    FDT_HIP(hipEventSynchronize(p->stamps[i]));
    FDT_HIP(hipEventElapsedTime(&ms_since_mark0[i], p->mark[0], p->stamps[i]));
  }
  *count = n;
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
