// Device-resident IoU tracker: the inline tracker of reference iouTracke_cal.py:113-156 (per frame) and
// :174-177 (finalise) as one single-workgroup kernel per frame plus an event log that the host replays.
//
// Why one workgroup: the association is strictly sequential over the active tracks (greedy, order
// dependent, iouTracke_cal.py:129-148); see track_step_kernel for how the per-track arg-max is hoisted
// out of the sequential part.
// Nothing syncs with the host per frame: detections are read straight from the Detect output on the
// same stream, the active set lives in HBM, and each frame appends {dets, det->track id, finished ids}
// to a log that is copied back only at finish()/flush.
//
// Bit-exactness: IoU is f64 in the operand order of utils/calc_performance.py:4-31,54-74 (this file is
// compiled with -ffp-contract=off); arg-max follows numpy (first maximum, NaN wins); `>` tests are
// strict like the reference (:134, :146, :174).
#include <vector>

#include "common.h"

namespace fdt {
namespace {

struct TrkState {       // one per tracker, device memory
  int n_active;
  int next_id;
  int frame_num;
  int overflow;         // log capacity exceeded (host flushes before this can happen)
  long long log_cursor; // bytes used in the log
};

struct ActiveSet {      // structure of arrays, capacity M
  double* box;          // [M][4]
  double* max_score;    // [M]
  int* len;             // [M]
  int* id;              // [M]
};

__device__ __forceinline__ double npmin(double x, double y) {
  return (x != x) ? x : ((y != y) ? y : (x < y ? x : y));
}
__device__ __forceinline__ double npmax(double x, double y) {
  return (x != x) ? x : ((y != y) ? y : (x > y ? x : y));
}
// calculate_iou(dets[:, :4], [track_box])[j, 0]
__device__ __forceinline__ double iou64(const double* a, const double* b) {
  double dx = npmin(a[2], b[2]) - npmax(a[0], b[0]);
  double dy = npmin(a[3], b[3]) - npmax(a[1], b[1]);
  dx = npmax(dx, 0.0);
  dy = npmax(dy, 0.0);
  double inter = dx * dy;
  double area_a = (a[2] - a[0]) * (a[3] - a[1]);
  double area_b = (b[2] - b[0]) * (b[3] - b[1]);
  double uni = area_a + area_b - inter;
  return inter / uni;
}

// numpy argmax order: NaN beats everything, then larger value, then lower index.
__device__ __forceinline__ bool better(double av, int ai, double bv, int bi) {
  if (bi < 0) return ai >= 0;
  if (ai < 0) return false;
  bool an = av != av, bn = bv != bv;
  if (an || bn) return (an && bn) ? (ai < bi) : an;
  if (av != bv) return av > bv;
  return ai < bi;
}

constexpr int TRK_THREADS = 1024;             // one workgroup of 16 waves per frame
constexpr int TRK_LDS_PER_SLOT = 72;          // bytes of LDS per detection/track slot (see the carve-up below)

// G consecutive frames in one launch <<<1, TRK_THREADS, M * TRK_LDS_PER_SLOT>>> (G = 1 for the single-frame entry
// points; G = world size after the all-gather of a frame-parallel step: one launch instead of G).  `dets_in`
// (f64 [n_in,5], G == 1 only) xor `det_out` (f32, frame g at det_out + g * det_stride, each [C,top_k,5]) is given.
// The tracker state (counters, log cursor) lives in LDS across the G frames and the two active sets swap roles per
// frame; a frame's writes are ordered before the next frame's reads by the workgroup barrier (one workgroup == one
// CU, whose L1 all its waves share), so the result is bit-identical to G single-frame launches.
//
// The greedy loop of iouTracke_cal.py:129-148 is sequential over the tracks, but its expensive part is not:
//   phase 1 (16 waves, one track per wave at a time): arg-max of the track's IoU row over ALL detections;
//   phase 2 (wave 0, in track order): if that detection is still free it is also the arg-max over the
//     remaining list (deleting other rows cannot change the first maximum), so the track takes it with a
//     few LDS reads; only when an earlier track has taken it is the row re-evaluated over the free ones.
// Detections, the per-track results and the det->track map live in LDS for the whole kernel.
__global__ __launch_bounds__(TRK_THREADS) void track_step_kernel(
    TrkState* __restrict__ st, ActiveSet set_a, ActiveSet set_b, int M, double sigma_iou, double sigma_h,
    int t_min, const double* __restrict__ dets_in, int n_in, const float* __restrict__ det_out_base,
    long long det_stride, int G, int num_classes, int top_k, float fw, float fh, float score_thr,
    char* __restrict__ log, long long log_cap) {
  extern __shared__ double smem_d[];
  double* dbox = smem_d;                     // [M][4] this frame's boxes
  double* dscore = dbox + (size_t)M * 4;     // [M]
  double* best_v = dscore + M;               // [M] per active track: best IoU over all detections ...
  double* tmaxs = best_v + M;                // [M] ... and the track's max_score
  int* best_i = (int*)(tmaxs + M);           // [M] ... its arg-max
  int* tlens = best_i + M;                   // [M]
  int* tids = tlens + M;                     // [M]
  volatile int* det_tid = tids + M;          // [M] track id that took det j, -1 while free
  __shared__ int s_first_fail;
  __shared__ int s_n_active, s_next_id, s_frame_num;
  __shared__ long long s_cursor;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = TRK_THREADS / 64;
  if (tid == 0) {
    s_n_active = st->n_active;
    s_next_id = st->next_id;
    s_frame_num = st->frame_num;
    s_cursor = st->log_cursor;
  }
  __syncthreads();
  for (int g = 0; g < G; ++g) {
  const float* det_out = det_out_base ? det_out_base + (long long)g * det_stride : nullptr;
  const ActiveSet cur = (g & 1) ? set_b : set_a;
  const ActiveSet nxt = (g & 1) ? set_a : set_b;
  const int frame = s_frame_num + 1;         // iouTracke_cal.py:118 (1-based)

  // ---- record header + this frame's detections -------------------------------------------------
  const long long cursor = s_cursor;
  const long long rec_max = 16 + (long long)M * (40 + 4 + 4) + 8;
  if (cursor + rec_max > log_cap) {          // uniform over the block
    if (tid == 0) st->overflow = 1;
    break;
  }
  int* hdr = (int*)(log + cursor);
  double* dets = (double*)(log + cursor + 16);   // [n][5], filled below
  int n = 0;
  if (det_out) {
    // host unpack of iouTracke_cal.py:53-84 on device: per class plane take rows while
    // score >= thr (f32 compare), box = row[1:5] * (w,h,w,h) in f32, then widen.
    for (int c = 0; c < num_classes; ++c) {
      const float* plane = det_out + (long long)c * top_k * 5;
      if (tid == 0) s_first_fail = top_k;
      __syncthreads();
      for (int j = tid; j < top_k; j += TRK_THREADS)
        if (!(plane[j * 5] >= score_thr)) atomicMin(&s_first_fail, j);
      __syncthreads();
      int cnt = s_first_fail;
      if (n + cnt > M) cnt = M - n;
      for (int j = tid; j < cnt; j += TRK_THREADS) {
        const float* r = plane + j * 5;
        const double b0 = (double)(r[1] * fw), b1 = (double)(r[2] * fh);
        const double b2 = (double)(r[3] * fw), b3 = (double)(r[4] * fh), sc = (double)r[0];
        double* d = dets + (long long)(n + j) * 5;
        d[0] = b0; d[1] = b1; d[2] = b2; d[3] = b3; d[4] = sc;
        double* l = dbox + (size_t)(n + j) * 4;
        l[0] = b0; l[1] = b1; l[2] = b2; l[3] = b3;
        dscore[n + j] = sc;
      }
      n += cnt;
      __syncthreads();
    }
    if (n == 0) {   // :73-74 dummy row np.array([[0,0,0,0,0.4]]) (f64)
      if (tid == 0) {
        dets[0] = 0; dets[1] = 0; dets[2] = 0; dets[3] = 0; dets[4] = 0.4;
        dbox[0] = 0; dbox[1] = 0; dbox[2] = 0; dbox[3] = 0; dscore[0] = 0.4;
      }
      n = 1;
    }
  } else {
    n = n_in < M ? n_in : M;
    for (int j = tid; j < n; j += TRK_THREADS) {
      const double* r = dets_in + (long long)j * 5;
      double* d = dets + (long long)j * 5;
      double* l = dbox + (size_t)j * 4;
      for (int k = 0; k < 4; ++k) { d[k] = r[k]; l[k] = r[k]; }
      d[4] = r[4];
      dscore[j] = r[4];
    }
  }
  int* tid_log = (int*)(log + cursor + 16 + (long long)n * 40);
  int* fin_log = tid_log + n;
  for (int j = tid; j < n; j += TRK_THREADS) det_tid[j] = -1;
  __syncthreads();

  // ---- phase 1: every track's arg-max over all detections, tracks spread over the waves --------
  const int T = s_n_active;
  for (int t = wave; t < T; t += NW) {
    const double tb[4] = {cur.box[t * 4 + 0], cur.box[t * 4 + 1], cur.box[t * 4 + 2], cur.box[t * 4 + 3]};
    double bv = 0.0;
    int bi = -1;
    for (int j = lane; j < n; j += 64) {
      double v = iou64(dbox + (size_t)j * 4, tb);
      if (better(v, j, bv, bi)) { bv = v; bi = j; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      double ov = __shfl_xor(bv, o, 64);
      int oi = __shfl_xor(bi, o, 64);
      if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) {
      best_v[t] = bv;
      best_i[t] = bi;
      tmaxs[t] = cur.max_score[t];
      tlens[t] = cur.len[t];
      tids[t] = cur.id[t];
    }
  }
  __syncthreads();

  if (wave == 0) {
    // ---- phase 2: greedy association in track order (:129-148), one wave -----------------------
    int n_alive = n;
    int n_upd = 0, n_fin = 0;
    for (int t = 0; t < T; ++t) {
      if (n_alive == 0) break;               // :130 has no else: remaining tracks vanish
      double bv = best_v[t];
      int bi = best_i[t];
      const double tmax = tmaxs[t];
      const int tlen = tlens[t];
      const int tidv = tids[t];
      if (det_tid[bi] != -1) {               // taken by an earlier track: arg-max over the free rows
        const double tb[4] = {cur.box[t * 4 + 0], cur.box[t * 4 + 1], cur.box[t * 4 + 2], cur.box[t * 4 + 3]};
        bv = 0.0;
        bi = -1;
        for (int j = lane; j < n; j += 64) {
          if (det_tid[j] != -1) continue;
          double v = iou64(dbox + (size_t)j * 4, tb);
          if (better(v, j, bv, bi)) { bv = v; bi = j; }
        }
        for (int o = 32; o > 0; o >>= 1) {
          double ov = __shfl_xor(bv, o, 64);
          int oi = __shfl_xor(bi, o, 64);
          if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
      }
      if (bv > sigma_iou) {                  // :134 strict; NaN -> unmatched
        if (lane == 0) {
          const double* d = dbox + (size_t)bi * 4;
          const double sc = dscore[bi];
          nxt.box[n_upd * 4 + 0] = d[0];
          nxt.box[n_upd * 4 + 1] = d[1];
          nxt.box[n_upd * 4 + 2] = d[2];
          nxt.box[n_upd * 4 + 3] = d[3];
          nxt.max_score[n_upd] = (sc > tmax) ? sc : tmax;   // max(track, det)  :141
          nxt.len[n_upd] = tlen + 1;
          nxt.id[n_upd] = tidv;
          det_tid[bi] = tidv;
        }
        ++n_upd;
        --n_alive;
        __builtin_amdgcn_wave_barrier();     // LDS ops of one wave complete in order; keep the compiler honest
      } else if (tmax > sigma_h && tlen > t_min) {   // :146
        if (lane == 0) fin_log[n_fin] = tidv;
        ++n_fin;
      }
    }

    // ---- remaining detections start new tracks, in detection order (:150-155) -------------------
    const int next_id = s_next_id;
    int n_new = 0;
    for (int base = 0; base < n; base += 64) {
      int j = base + lane;
      bool fresh = (j < n) && det_tid[j] == -1;
      unsigned long long bal = __ballot(fresh);
      if (fresh) {
        int r = n_new + __popcll(bal & ((1ull << lane) - 1ull));
        int slot = n_upd + r;
        const double* d = dbox + (size_t)j * 4;
        nxt.box[slot * 4 + 0] = d[0];
        nxt.box[slot * 4 + 1] = d[1];
        nxt.box[slot * 4 + 2] = d[2];
        nxt.box[slot * 4 + 3] = d[3];
        nxt.max_score[slot] = dscore[j];
        nxt.len[slot] = 1;
        nxt.id[slot] = next_id + r;
        det_tid[j] = next_id + r;
      }
      n_new += __popcll(bal);
    }
    if (lane == 0) {
      hdr[0] = n;
      hdr[1] = n_fin;
      hdr[2] = frame;
      hdr[3] = 0;
      long long rec = 16 + (long long)n * 40 + (long long)(n + n_fin) * 4;
      rec = (rec + 7) & ~7ll;
      s_cursor = cursor + rec;
      s_n_active = n_upd + n_new;
      s_next_id = next_id + n_new;
      s_frame_num = frame;
    }
  }
  __syncthreads();
  for (int j = tid; j < n; j += TRK_THREADS) tid_log[j] = det_tid[j];
  __syncthreads();   // det_tid / dbox are rewritten by the next frame; the new active set is visible to every wave
  }
  if (tid == 0) {
    st->log_cursor = s_cursor;
    st->n_active = s_n_active;
    st->next_id = s_next_id;
    st->frame_num = s_frame_num;
  }
}

}  // namespace
}  // namespace fdt

// ================================================================================== host object
struct fdt_tracker {
  double sigma_iou, sigma_h;
  int t_min, M, log_frames;
  long long log_cap = 0;
  fdt::TrkState* d_state = nullptr;
  char* d_log = nullptr;
  double* d_dets_in = nullptr;
  fdt::ActiveSet set[2];
  void* d_sets = nullptr;
  int cur = 0;
  int frames_in_log = 0;
  int frames_total = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t last_stream = nullptr;
  hipEvent_t step_done = nullptr;   // recorded after every step: the state chain is ordered across caller streams

  struct Track {
    std::vector<double> boxes;   // 4 per box
    double max_score = 0;
    int start_frame = 0;
    bool touched = false;
  };
  std::vector<Track> live;       // indexed by id - id_base (sparse via map below)
  std::vector<int> live_ids;
  std::vector<Track> finished;
  bool finalized = false;
  std::vector<char> h_log;
};

namespace {
using fdt::set_error;

int find_live(fdt_tracker* t, int id) {
  for (size_t i = 0; i < t->live_ids.size(); ++i)
    if (t->live_ids[i] == id) return (int)i;
  return -1;
}

// Copy the log back and replay it into host-side tracks; resets the device cursor.
int flush_log(fdt_tracker* t) {
  hipStream_t st = t->last_stream ? t->last_stream : t->own_stream;
  FDT_HIP(hipStreamSynchronize(st));
  fdt::TrkState hs;
  FDT_HIP(hipMemcpy(&hs, t->d_state, sizeof(hs), hipMemcpyDeviceToHost));
  FDT_REQUIRE(!hs.overflow, FDT_ERR_STATE, "fdt_tracker: event log overflow");
  t->h_log.resize((size_t)hs.log_cursor);
  if (hs.log_cursor) FDT_HIP(hipMemcpy(t->h_log.data(), t->d_log, hs.log_cursor, hipMemcpyDeviceToHost));
  long long off = 0;
  while (off < hs.log_cursor) {
    const int* hdr = (const int*)(t->h_log.data() + off);
    int n = hdr[0], n_fin = hdr[1], frame = hdr[2];
    const double* dets = (const double*)(t->h_log.data() + off + 16);
    const int* tid = (const int*)(t->h_log.data() + off + 16 + (long long)n * 40);
    const int* fin = tid + n;
    for (auto& tr : t->live) tr.touched = false;
    // finished tracks keep the state they had before this frame
    std::vector<int> fin_pos;
    for (int k = 0; k < n_fin; ++k) {
      int p = find_live(t, fin[k]);
      FDT_REQUIRE(p >= 0, FDT_ERR_STATE, "fdt_tracker: log replay lost track %d", fin[k]);
      t->finished.push_back(t->live[p]);
    }
    std::vector<fdt_tracker::Track> nl;
    std::vector<int> nl_ids;
    // order of the next active list does not matter on the host (device keeps it); keep det order
    for (int j = 0; j < n; ++j) {
      const double* d = dets + (long long)j * 5;
      int p = find_live(t, tid[j]);
      if (p >= 0) {
        fdt_tracker::Track tr = std::move(t->live[p]);
        tr.boxes.insert(tr.boxes.end(), d, d + 4);
        tr.max_score = (d[4] > tr.max_score) ? d[4] : tr.max_score;
        nl.push_back(std::move(tr));
      } else {
        fdt_tracker::Track tr;
        tr.boxes.assign(d, d + 4);
        tr.max_score = d[4];
        tr.start_frame = frame;
        nl.push_back(std::move(tr));
      }
      nl_ids.push_back(tid[j]);
    }
    t->live.swap(nl);           // tracks neither matched nor finished are dropped (:130, :146)
    t->live_ids.swap(nl_ids);
    long long rec = 16 + (long long)n * 40 + (long long)(n + n_fin) * 4;
    off += (rec + 7) & ~7ll;
  }
  long long zero = 0;
  FDT_HIP(hipMemcpy((char*)t->d_state + offsetof(fdt::TrkState, log_cursor), &zero, 8,
                    hipMemcpyHostToDevice));
  t->frames_in_log = 0;
  return FDT_OK;
}

int step_common(fdt_tracker* t, const double* dets_dev, int n, const float* det_out, long long det_stride, int G,
                int nc, int top_k, int w, int h, float thr, hipStream_t st) {
  FDT_REQUIRE(!t->finalized, FDT_ERR_STATE, "fdt_tracker: already finished; call reset");
  FDT_REQUIRE(G >= 1 && G <= t->log_frames, FDT_ERR_ARG, "fdt_tracker: %d frames per launch > log_frames %d", G,
              t->log_frames);
  if (t->frames_in_log + G > t->log_frames) FDT_TRY(flush_log(t));
  // the tracker is one sequential state machine: a step on another stream than the previous one waits for it
  if (t->last_stream && t->last_stream != st) FDT_HIP(hipStreamWaitEvent(st, t->step_done, 0));
  hipLaunchKernelGGL(fdt::track_step_kernel, dim3(1), dim3(fdt::TRK_THREADS), (size_t)t->M * fdt::TRK_LDS_PER_SLOT,
                     st, t->d_state,
                     t->set[t->cur], t->set[t->cur ^ 1], t->M, t->sigma_iou, t->sigma_h, t->t_min,
                     dets_dev, n, det_out, det_stride, G, nc, top_k, (float)w, (float)h, thr, t->d_log, t->log_cap);
  FDT_LAUNCH_CHECK();
  FDT_HIP(hipEventRecord(t->step_done, st));
  t->cur ^= (G & 1);
  t->frames_in_log += G;
  t->frames_total += G;
  t->last_stream = st;
  return FDT_OK;
}
}  // namespace

extern "C" fdt_tracker* fdt_tracker_create(double sigma_iou, double sigma_h, int t_min, int max_dets,
                                           int log_frames) {
  if (max_dets < 1 || log_frames < 1) {
    set_error("fdt_tracker_create: max_dets and log_frames must be >= 1");
    return nullptr;
  }
  if ((long long)max_dets * fdt::TRK_LDS_PER_SLOT > 160 * 1024 - 64) {
    set_error("fdt_tracker_create: max_dets %d does not fit the LDS-resident frame state (limit %d)", max_dets,
              (160 * 1024 - 64) / fdt::TRK_LDS_PER_SLOT);
    return nullptr;
  }
  fdt_tracker* t = new fdt_tracker();
  t->sigma_iou = sigma_iou;
  t->sigma_h = sigma_h;
  t->t_min = t_min;
  t->M = max_dets;
  t->log_frames = log_frames;
  long long rec_max = 16 + (long long)max_dets * 48 + 8;
  t->log_cap = rec_max * (log_frames + 1);
  size_t per_set = (size_t)max_dets * (32 + 8 + 4 + 4);
  bool ok = hipStreamCreateWithFlags(&t->own_stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&t->step_done, hipEventDisableTiming) == hipSuccess &&
            hipMalloc((void**)&t->d_state, sizeof(fdt::TrkState)) == hipSuccess &&
            hipMalloc((void**)&t->d_log, t->log_cap) == hipSuccess &&
            hipMalloc((void**)&t->d_dets_in, (size_t)max_dets * 40) == hipSuccess &&
            hipMalloc(&t->d_sets, per_set * 2) == hipSuccess &&
            hipMemset(t->d_state, 0, sizeof(fdt::TrkState)) == hipSuccess &&
            hipFuncSetAttribute((const void*)fdt::track_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                max_dets * fdt::TRK_LDS_PER_SLOT) == hipSuccess;
  if (!ok) {
    set_error("fdt_tracker_create: HIP allocation failed: %s", hipGetErrorString(hipGetLastError()));
    fdt_tracker_destroy(t);
    return nullptr;
  }
  for (int s = 0; s < 2; ++s) {
    char* b = (char*)t->d_sets + per_set * s;
    t->set[s].box = (double*)b;
    t->set[s].max_score = (double*)(b + (size_t)max_dets * 32);
    t->set[s].len = (int*)(b + (size_t)max_dets * 40);
    t->set[s].id = (int*)(b + (size_t)max_dets * 44);
  }
  return t;
}

extern "C" void fdt_tracker_destroy(fdt_tracker* t) {
  if (!t) return;
  if (t->d_state) (void)hipFree(t->d_state);
  if (t->d_log) (void)hipFree(t->d_log);
  if (t->d_dets_in) (void)hipFree(t->d_dets_in);
  if (t->d_sets) (void)hipFree(t->d_sets);
  if (t->step_done) (void)hipEventDestroy(t->step_done);
  if (t->own_stream) (void)hipStreamDestroy(t->own_stream);
  delete t;
}

extern "C" int fdt_tracker_reset(fdt_tracker* t) {
  FDT_REQUIRE(t, FDT_ERR_ARG, "fdt_tracker_reset: null handle");
  hipStream_t st = t->last_stream ? t->last_stream : t->own_stream;
  FDT_HIP(hipStreamSynchronize(st));
  FDT_HIP(hipMemset(t->d_state, 0, sizeof(fdt::TrkState)));
  t->live.clear();
  t->live_ids.clear();
  t->finished.clear();
  t->finalized = false;
  t->frames_in_log = t->frames_total = 0;
  t->cur = 0;
  return FDT_OK;
}

extern "C" int fdt_tracker_step(fdt_tracker* t, const double* dets, int n) {
  FDT_REQUIRE(t && n >= 0 && (n == 0 || dets), FDT_ERR_ARG, "fdt_tracker_step: bad argument");
  FDT_REQUIRE(n <= t->M, FDT_ERR_ARG, "fdt_tracker_step: %d detections > max_dets %d", n, t->M);
  hipStream_t st = t->own_stream;
  // pageable-host memcpy on a stream is synchronous w.r.t. the host buffer: safe to return
  if (n) FDT_HIP(hipMemcpyAsync(t->d_dets_in, dets, (size_t)n * 40, hipMemcpyHostToDevice, st));
  return step_common(t, t->d_dets_in, n, nullptr, 0, 1, 0, 0, 0, 0, 0.f, st);
}

extern "C" int fdt_tracker_step_dev(fdt_tracker* t, const float* det_out, int num_classes, int top_k,
                                    int width, int height, float score_thresh, void* stream) {
  FDT_REQUIRE(t && det_out && num_classes >= 1 && top_k >= 1, FDT_ERR_ARG,
              "fdt_tracker_step_dev: bad argument");
  FDT_REQUIRE((long long)num_classes * top_k <= t->M, FDT_ERR_ARG,
              "fdt_tracker_step_dev: num_classes*top_k %d > max_dets %d", num_classes * top_k, t->M);
  hipStream_t st = stream ? (hipStream_t)stream : t->own_stream;
  return step_common(t, nullptr, 0, det_out, 0, 1, num_classes, top_k, width, height, score_thresh, st);
}

// The G frames of one frame-parallel step (rank order == frame order after the all-gather) in ONE
// launch: frame g's Detect record is at det_out + g * stride_floats.  Bit-identical to G fdt_tracker_step_dev calls.
extern "C" int fdt_tracker_step_dev_multi(fdt_tracker* t, const float* det_out, int n_frames, long long stride_floats,
                                          int num_classes, int top_k, int width, int height, float score_thresh,
                                          void* stream) {
  FDT_REQUIRE(t && det_out && num_classes >= 1 && top_k >= 1 && n_frames >= 1, FDT_ERR_ARG,
              "fdt_tracker_step_dev_multi: bad argument");
  FDT_REQUIRE(stride_floats >= (long long)num_classes * top_k * 5, FDT_ERR_ARG,
              "fdt_tracker_step_dev_multi: stride %lld smaller than one record", stride_floats);
  FDT_REQUIRE((long long)num_classes * top_k <= t->M, FDT_ERR_ARG,
              "fdt_tracker_step_dev_multi: num_classes*top_k %d > max_dets %d", num_classes * top_k, t->M);
  hipStream_t st = stream ? (hipStream_t)stream : t->own_stream;
  return step_common(t, nullptr, 0, det_out, stride_floats, n_frames, num_classes, top_k, width, height, score_thresh,
                     st);
}

extern "C" int fdt_tracker_finish(fdt_tracker* t) {
  FDT_REQUIRE(t, FDT_ERR_ARG, "fdt_tracker_finish: null handle");
  if (t->finalized) return FDT_OK;
  FDT_TRY(flush_log(t));
  // iouTracke_cal.py:174-175: surviving active tracks, in active-list order (device order)
  fdt::TrkState hs;
  FDT_HIP(hipMemcpy(&hs, t->d_state, sizeof(hs), hipMemcpyDeviceToHost));
  std::vector<int> ids(hs.n_active);
  if (hs.n_active)
    FDT_HIP(hipMemcpy(ids.data(), t->set[t->cur].id, (size_t)hs.n_active * 4, hipMemcpyDeviceToHost));
  for (int id : ids) {
    int p = find_live(t, id);
    FDT_REQUIRE(p >= 0, FDT_ERR_STATE, "fdt_tracker_finish: active track %d missing on host", id);
    const auto& tr = t->live[p];
    if (tr.max_score > t->sigma_h && (int)(tr.boxes.size() / 4) >= t->t_min) t->finished.push_back(tr);
  }
  t->finalized = true;
  return FDT_OK;
}

extern "C" int fdt_tracker_num_tracks(fdt_tracker* t, int* n) {
  FDT_REQUIRE(t && n, FDT_ERR_ARG, "fdt_tracker_num_tracks: bad argument");
  FDT_REQUIRE(t->finalized, FDT_ERR_STATE, "fdt_tracker: call fdt_tracker_finish first");
  *n = (int)t->finished.size();
  return FDT_OK;
}

extern "C" int fdt_tracker_track_info(fdt_tracker* t, int idx, int* n_boxes, double* max_score,
                                      int* start_frame) {
  FDT_REQUIRE(t && t->finalized, FDT_ERR_STATE, "fdt_tracker: call fdt_tracker_finish first");
  FDT_REQUIRE(idx >= 0 && idx < (int)t->finished.size(), FDT_ERR_ARG, "fdt_tracker: bad track index");
  const auto& tr = t->finished[idx];
  if (n_boxes) *n_boxes = (int)(tr.boxes.size() / 4);
  if (max_score) *max_score = tr.max_score;
  if (start_frame) *start_frame = tr.start_frame;
  return FDT_OK;
}

extern "C" int fdt_tracker_track_boxes(fdt_tracker* t, int idx, double* boxes) {
  FDT_REQUIRE(t && t->finalized && boxes, FDT_ERR_STATE, "fdt_tracker: call fdt_tracker_finish first");
  FDT_REQUIRE(idx >= 0 && idx < (int)t->finished.size(), FDT_ERR_ARG, "fdt_tracker: bad track index");
  const auto& tr = t->finished[idx];
  memcpy(boxes, tr.boxes.data(), tr.boxes.size() * 8);
  return FDT_OK;
}
