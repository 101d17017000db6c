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
#include <atomic>
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
  // which association form each frame ran (fdt_tracker_stats): [0] candidate form, [1] exact form because a pair's IoU was
  // NaN, [2] exact form because a track had more than TRK_CAND candidates, [3] exact form because sigma_iou < 0 (or NaN)
  long long form_frames[4];
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

// f32 image of a box for the overlap pre-test, or the everything-box when the pre-test would not be exact for it.
__device__ __forceinline__ float4 prefilter_box(double b0, double b1, double b2, double b3) {
  const float f0 = (float)b0, f1 = (float)b1, f2 = (float)b2, f3 = (float)b3;
  const float s = (f0 + f1) + (f2 + f3);
  const bool ok = (double)f0 == b0 && (double)f1 == b1 && (double)f2 == b2 && (double)f3 == b3 && (s - s) == 0.0f &&
                  f2 > f0 && f3 > f1;
  const float inf = __builtin_huge_valf();
  return ok ? make_float4(f0, f1, f2, f3) : make_float4(-inf, -inf, inf, inf);
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

#ifdef FDT_TRK_TIMING   // per-phase device clocks for tools/tracker_bench.py (not part of the product build)
__device__ long long g_trk_time[8];
#define TT(i) if (tid == 0) { long long c_ = wall_clock64(); g_trk_time[i] += c_ - tlast; tlast = c_; }
#else
#define TT(i)
#endif
constexpr int TRK_THREADS = 1024;             // one workgroup of 16 waves per frame
constexpr int TRK_CAND = 6;                   // candidate detections (IoU > sigma_iou) kept per track; more -> exact fallback
constexpr int TRK_LDS_PER_SLOT = 88 + 2 * TRK_CAND;   // bytes of LDS per detection/track slot (see the carve-up below)

// G consecutive frames in one launch <<<1, TRK_THREADS, M * TRK_LDS_PER_SLOT>>> (G = 1 for the single-frame entry
// points; G = world size after the all-gather of a frame-parallel step: one launch instead of G).  `dets_in`
// (f64 [n_in,5], G == 1 only) xor `det_out` (f32, frame g at det_out + g * det_stride, each [C,top_k,5]) is given.
// The tracker state (counters, log cursor) lives in LDS across the G frames and the two active sets swap roles per
// frame; a frame's writes are ordered before the next frame's reads by the workgroup barrier (one workgroup == one
// CU, whose L1 all its waves share), so the result is bit-identical to G single-frame launches.
//
// The greedy loop of iouTracke_cal.py:129-148 is sequential over the tracks, but its expensive part is not.
// CANDIDATE FORM (round 4, the default; sigma_iou >= 0): a track can only ever take a detection whose IoU with it
// exceeds sigma_iou (:134), whatever has been deleted before its turn, so
//   phase 1 (16 waves, one lane per track): the exact f64 IoU of every pair that passes the f32 overlap pre-test;
//     the detections above sigma_iou -- a handful per track even in crowded frames -- are kept as the track's
//     candidate list, sorted by numpy's arg-max order (larger IoU first, lower index on ties);
//   phase 2 (wave 0, in track order): the track takes its first candidate that is still free -- exactly the arg-max
//     over the remaining detections when that exceeds sigma_iou, and "unmatched" otherwise.  A conflict costs one
//     parallel LDS probe of <= 6 candidates instead of a re-evaluation of the whole IoU row.
// EXACT FORM (rounds 1-3; the fallback): per track the arg-max over ALL detections, re-evaluated over the free ones on
// a conflict.  A frame falls back to it when any pair's IoU is NaN (numpy's arg-max then returns the NaN and the track
// stays unmatched -- kept literally), a track has more than TRK_CAND candidates, or sigma_iou < 0 (zero-IoU
// detections would match).  Both forms make the decisions of the reference loop, bit for bit.
// Detections, the per-track results and the det->track map live in LDS for the whole kernel.
__global__ __launch_bounds__(TRK_THREADS) void track_step_kernel(
    TrkState* __restrict__ st, ActiveSet set_a, ActiveSet set_b, int M, double sigma_iou, double sigma_h,
    int t_min, const double* __restrict__ dets_in, int n_in, const float* __restrict__ det_out_base,
    long long det_stride, int G, int num_classes, int top_k, float fw, float fh, float score_thr,
    char* __restrict__ log, long long log_cap) {
  extern __shared__ __attribute__((aligned(16))) double smem_d[];
  // [M] f32 copy of this frame's boxes for the overlap pre-test of phase 1: the box itself when its coordinates are
  // finite, exactly representable in f32 and x2 > x1, y2 > y1 (every box the Detect layer emits), else
  // (-inf,-inf,+inf,+inf) = "may overlap anything", which sends every pair with it down the exact f64 path
  float4* fbox = (float4*)smem_d;
  double* dbox = smem_d + (size_t)M * 2;     // [M][4] this frame's boxes
  double* dscore = dbox + (size_t)M * 4;     // [M]
  double* best_v = dscore + M;               // [M] per active track: best IoU over all detections ...
  double* tmaxs = best_v + M;                // [M] ... and the track's max_score
  int* best_i = (int*)(tmaxs + M);           // [M] ... its arg-max
  int* tlens = best_i + M;                   // [M]
  int* tids = tlens + M;                     // [M]
  volatile int* det_tid = tids + M;          // [M] track id that took det j, -1 while free
  unsigned short* cand = (unsigned short*)(tids + 2 * (size_t)M);   // [M][TRK_CAND] per track: candidate detections, best first, 0xFFFF = none
  __shared__ int s_first_fail[2], s_fallback;
  __shared__ double s_pv[TRK_THREADS];       // phase 1: per-segment partial arg-max (value, index) per track
  __shared__ int s_pi[TRK_THREADS];
  __shared__ int s_n_active, s_next_id, s_frame_num, s_nupd, s_nfin;
  __shared__ long long s_cursor;
  __shared__ int s_form[4];                  // frames of this launch per association form (TrkState::form_frames)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = TRK_THREADS / 64;
  if (tid == 0) {
    s_n_active = st->n_active;
    s_next_id = st->next_id;
    s_frame_num = st->frame_num;
    s_cursor = st->log_cursor;
    s_form[0] = s_form[1] = s_form[2] = s_form[3] = 0;
  }
  __syncthreads();
#ifdef FDT_TRK_TIMING
  long long tlast = wall_clock64();
#endif
  for (int g = 0; g < G; ++g) {
  const float* det_out = det_out_base ? det_out_base + (long long)g * det_stride : nullptr;
  const ActiveSet cur = (g & 1) ? set_b : set_a;
  const ActiveSet nxt = (g & 1) ? set_a : set_b;
  const int frame = s_frame_num + 1;         // iouTracke_cal.py:118 (1-based)
  // The track boxes of this wave's first phase-1 work item (chunk = wave % n_chunks, whatever the segment count turns out to
  // be) are requested now, so that their HBM round trip runs under the unpack of the detections.
  const int T = s_n_active;
  double pf_tb[4] = {0, 0, 0, 0};
  {
    const int nch = (T + 63) >> 6;
    const int t = nch > 0 ? (wave % nch) * 64 + lane : T;
    if (t < T) {
      pf_tb[0] = cur.box[t * 4 + 0]; pf_tb[1] = cur.box[t * 4 + 1]; pf_tb[2] = cur.box[t * 4 + 2]; pf_tb[3] = cur.box[t * 4 + 3];
    }
  }

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
    // score >= thr (f32 compare), box = row[1:5] * (w,h,w,h) in f32, then widen.  All planes in one pass: a thread
    // loads its row of every plane once (the loads of all planes are in flight together), the first failing row per
    // plane is an LDS min, and the rows are written from the registers.
    constexpr int UNPACK_C = 2;                // planes per pass (the detectors here have num_classes = 2)
    for (int c0 = 0; c0 < num_classes; c0 += UNPACK_C) {
      float row[UNPACK_C][5];
      if (tid < UNPACK_C) s_first_fail[tid] = top_k;
#pragma unroll
      for (int cc = 0; cc < UNPACK_C; ++cc) {
        const bool live = c0 + cc < num_classes && tid < top_k;
        const float* r = det_out + ((long long)(c0 + cc) * top_k + (live ? tid : 0)) * 5;
#pragma unroll
        for (int k = 0; k < 5; ++k) row[cc][k] = live ? r[k] : 0.0f;
      }
      __syncthreads();
#pragma unroll
      for (int cc = 0; cc < UNPACK_C; ++cc) {
        if (c0 + cc < num_classes) {
          if (tid < top_k && !(row[cc][0] >= score_thr)) atomicMin(&s_first_fail[cc], tid);
          for (int j = tid + TRK_THREADS; j < top_k; j += TRK_THREADS)      // top_k > 1024: the rest from memory
            if (!(det_out[((long long)(c0 + cc) * top_k + j) * 5] >= score_thr)) atomicMin(&s_first_fail[cc], j);
        }
      }
      __syncthreads();
#pragma unroll
      for (int cc = 0; cc < UNPACK_C; ++cc) {
        if (c0 + cc >= num_classes) break;
        const float* plane = det_out + (long long)(c0 + cc) * top_k * 5;
        int cnt = s_first_fail[cc];
        if (n + cnt > M) cnt = M - n;
        for (int j = tid; j < cnt; j += TRK_THREADS) {
          float rr[5];
#pragma unroll
          for (int k = 0; k < 5; ++k) rr[k] = j == tid ? row[cc][k] : plane[j * 5 + k];
          const double b0 = (double)(rr[1] * fw), b1 = (double)(rr[2] * fh);
          const double b2 = (double)(rr[3] * fw), b3 = (double)(rr[4] * fh), sc = (double)rr[0];
          double* d = dets + (long long)(n + j) * 5;
          d[0] = b0; d[1] = b1; d[2] = b2; d[3] = b3; d[4] = sc;
          double* l = dbox + (size_t)(n + j) * 4;
          l[0] = b0; l[1] = b1; l[2] = b2; l[3] = b3;
          fbox[n + j] = prefilter_box(b0, b1, b2, b3);
          dscore[n + j] = sc;
        }
        n += cnt;
      }
      __syncthreads();
    }
    if (n == 0) {   // :73-74 dummy row np.array([[0,0,0,0,0.4]]) (f64)
      if (tid == 0) {
        dets[0] = 0; dets[1] = 0; dets[2] = 0; dets[3] = 0; dets[4] = 0.4;
        dbox[0] = 0; dbox[1] = 0; dbox[2] = 0; dbox[3] = 0; dscore[0] = 0.4;
        fbox[0] = prefilter_box(0, 0, 0, 0);
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
      fbox[j] = prefilter_box(r[0], r[1], r[2], r[3]);
      d[4] = r[4];
      dscore[j] = r[4];
    }
  }
  TT(0)
  int* tid_log = (int*)(log + cursor + 16 + (long long)n * 40);
  int* fin_log = tid_log + n;
  for (int j = tid; j < n; j += TRK_THREADS) det_tid[j] = -1;
  __syncthreads();

  // ---- phase 1: every track's arg-max over all detections -----------------------------------------
  // One LANE per track, 64 tracks per wave, the detections streamed past them as LDS broadcasts: no cross-lane
  // reduction, and the running arg-max of a lane sees its detections in index order.  When there are fewer than 16
  // chunks of 64 tracks the detection range is cut into S segments so that all 16 waves work; the per-segment results
  // are merged below (`better` is a total order, so the merge order does not matter).
  // Per (track, detection) pair only the f32 overlap pre-test runs (exact: both boxes are f32-exact with positive
  // extent, or one of them is the everything-box).  An empty intersection makes numpy's inter +0 and its IoU
  // 0 / (area_a + area_b) = +0 with both areas positive and finite, so all such detections tie at 0 and only the first
  // of them can be the arg-max; the (few) overlapping ones are remembered and evaluated exactly in f64 afterwards.
  // ---- phase 1, candidate form ---------------------------------------------------------------------
  bool fast = sigma_iou >= 0.0;              // uniform (false for a NaN threshold too)
  if (!fast && tid == 0) s_form[3] += 1;
  if (fast) {
    int* cnt = best_i;                       // candidates found so far per track (best_i is free until phase 2 stages results)
    for (int t = tid; t < T; t += TRK_THREADS) cnt[t] = 0;
    if (tid == 0) s_fallback = 0;
    __syncthreads();
    const int n_chunks = (T + 63) >> 6;
    // Work items = (chunk of 64 tracks) x (segment of the detections): S segments so that the 16 waves get equal shares
    // (a work item costs its detections + ~24 detection-visits of set-up)
    int S = 1;
    {
      long long best_cost = -1;
      for (int s_ = 1; s_ <= 16 && n_chunks > 0; ++s_) {
        if (s_ > 1 && (n + s_ - 1) / s_ < 16) break;
        const long long cost = (long long)((n_chunks * s_ + NW - 1) / NW) * ((n + s_ - 1) / s_ + 24);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; S = s_; }
      }
    }
    const double half_sigma = 0.5 * sigma_iou;
    const float inf = __builtin_huge_valf();
    int bad = 0;                               // 1: a NaN IoU, 2: more than TRK_CAND candidates for one track
    for (int w = wave; w < n_chunks * S; w += NW) {
      const int chunk = w % n_chunks, seg = w / n_chunks;
      const int t = chunk * 64 + lane;
      const bool in = t < T;
      double tb[4] = {0, 0, 0, 0};
      float4 tf = make_float4(inf, inf, -inf, -inf);          // overlaps nothing: lanes past the last track
      if (in) {
        if (w == wave) {                        // requested at the top of the frame
          tb[0] = pf_tb[0]; tb[1] = pf_tb[1]; tb[2] = pf_tb[2]; tb[3] = pf_tb[3];
        } else {
          tb[0] = cur.box[t * 4 + 0]; tb[1] = cur.box[t * 4 + 1]; tb[2] = cur.box[t * 4 + 2]; tb[3] = cur.box[t * 4 + 3];
        }
        tf = prefilter_box(tb[0], tb[1], tb[2], tb[3]);
        if (seg == 0) {
          tmaxs[t] = cur.max_score[t];
          tlens[t] = cur.len[t];
          tids[t] = cur.id[t];
        }
      }
      const int j_lo = (int)((long long)n * seg / S), j_hi = (int)((long long)n * (seg + 1) / S);
      // four compares per pair (exact: see prefilter_box); a pair that fails them has IoU +0 (never a candidate, never NaN)
      auto overlaps = [&](const float4 f) -> bool {
        return (int)(f.z > tf.x) & (int)(tf.z > f.x) & (int)(f.w > tf.y) & (int)(tf.w > f.y);
      };
      // iou64(det j, track) with its division skipped when the quotient is certainly below sigma_iou: inter < (sigma/2) * uni
      // implies ordered operands, uni > 0 and a quotient below sigma (the rounding of the product and of the division
      // is 2^-53 relative against a factor of two), so neither a candidate nor a NaN is lost
      auto exact = [&](int j) {
        const double* a = dbox + (size_t)j * 4;
        double dx = npmin(a[2], tb[2]) - npmax(a[0], tb[0]);
        double dy = npmin(a[3], tb[3]) - npmax(a[1], tb[1]);
        dx = npmax(dx, 0.0);
        dy = npmax(dy, 0.0);
        const double inter = dx * dy;
        const double area_a = (a[2] - a[0]) * (a[3] - a[1]);
        const double area_b = (tb[2] - tb[0]) * (tb[3] - tb[1]);
        const double uni = area_a + area_b - inter;
        if (inter < half_sigma * uni) return;
        const double v = inter / uni;
        if (v != v) {
          bad |= 1;
        } else if (v > sigma_iou) {
          const int k = atomicAdd(&cnt[t], 1);
          if (k < TRK_CAND) cand[(size_t)t * TRK_CAND + k] = (unsigned short)j;
          else bad |= 2;
        }
      };
      int j = j_lo;
      constexpr int SB = 4;                     // broadcast reads in flight per round trip
      for (; j + SB <= j_hi; j += SB) {
        float4 f[SB];
#pragma unroll
        for (int k = 0; k < SB; ++k) f[k] = fbox[j + k];
        bool m[SB], any = false;
#pragma unroll
        for (int k = 0; k < SB; ++k) { m[k] = overlaps(f[k]); any |= m[k]; }
        if (__ballot(any) == 0ull) continue;
#pragma unroll
        for (int k = 0; k < SB; ++k)
          if (__ballot(m[k]) != 0ull) { if (m[k]) exact(j + k); }
      }
      for (; j < j_hi; ++j) {
        const bool m = overlaps(fbox[j]);
        if (__ballot(m) != 0ull) { if (m) exact(j); }
      }
    }
    if (bad) atomicOr(&s_fallback, bad);
    __syncthreads();
    fast = s_fallback == 0;
    if (tid == 0) s_form[fast ? 0 : ((s_fallback & 1) ? 1 : 2)] += 1;   // a frame with both causes counts as NaN
    TT(5)
#ifdef FDT_TRK_TIMING
    if (tid == 0 && !fast) g_trk_time[6] += 1;   // frames that fell back to the exact form
#endif
    if (fast) {
      // sort each track's candidates into numpy's arg-max order (values recomputed: same function, same bits); a wave
      // spends only the rounds its fullest list needs (most tracks have one candidate or none)
      for (int t = tid; t < ((T + 63) & ~63); t += TRK_THREADS) {
        const bool in = t < T;
        const int k = in ? cnt[t] : 0;
        int kmax = k;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kmax = max(kmax, __shfl_xor(kmax, o, 64));
        int jj[TRK_CAND];
#pragma unroll
        for (int i = 0; i < TRK_CAND; ++i) jj[i] = i < k ? (int)cand[(size_t)t * TRK_CAND + i] : 0xFFFF;
        if (kmax >= 2) {
          const double tb[4] = {in ? cur.box[t * 4 + 0] : 0.0, in ? cur.box[t * 4 + 1] : 0.0, in ? cur.box[t * 4 + 2] : 0.0,
                                in ? cur.box[t * 4 + 3] : 0.0};
          double v[TRK_CAND];
#pragma unroll
          for (int i = 0; i < TRK_CAND; ++i) {
            v[i] = -1.0;
            if (i < kmax && i < k) v[i] = iou64(dbox + (size_t)jj[i] * 4, tb);
          }
#pragma unroll
          for (int i = 0; i < TRK_CAND - 1; ++i)
#pragma unroll
            for (int q = TRK_CAND - 1; q > i; --q) {
              if (q >= kmax) continue;                    // uniform: nothing beyond the fullest list
              const bool sw = better(v[q], jj[q], v[q - 1], jj[q - 1]);
              const double tv = v[q]; const int tj = jj[q];
              v[q] = sw ? v[q - 1] : v[q]; jj[q] = sw ? jj[q - 1] : jj[q];
              v[q - 1] = sw ? tv : v[q - 1]; jj[q - 1] = sw ? tj : jj[q - 1];
            }
        }
        if (in) {
#pragma unroll
          for (int i = 0; i < TRK_CAND; ++i) cand[(size_t)t * TRK_CAND + i] = (unsigned short)jj[i];
        }
      }
    }
  }
  // ---- phase 1, exact form (fallback) --------------------------------------------------------------
  if (!fast) {
    constexpr int KC = 4;                      // overlapping detections remembered per (track, segment)
    const int n_chunks = (T + 63) >> 6;
    const int S = (n_chunks > 0 && n_chunks < NW) ? NW / n_chunks : 1;
    const float inf = __builtin_huge_valf();
    for (int w = wave; w < n_chunks * S; w += NW) {
      const int chunk = w % n_chunks, seg = w / n_chunks;
      const int t = chunk * 64 + lane;
      const bool in = t < T;
      double tb[4] = {0, 0, 0, 0};
      float4 tf = make_float4(inf, inf, -inf, -inf);          // overlaps nothing: lanes past the last track
      if (in) {
        tb[0] = cur.box[t * 4 + 0]; tb[1] = cur.box[t * 4 + 1]; tb[2] = cur.box[t * 4 + 2]; tb[3] = cur.box[t * 4 + 3];
        tf = prefilter_box(tb[0], tb[1], tb[2], tb[3]);
        if (seg == 0) {
          tmaxs[t] = cur.max_score[t];
          tlens[t] = cur.len[t];
          tids[t] = cur.id[t];
        }
      }
      const int j_lo = (int)((long long)n * seg / S), j_hi = (int)((long long)n * (seg + 1) / S);
      // The pair test is four compares (min(a2,b2) > max(a0,b0) <=> a2 > b0 && b2 > a0 when a2 > a0 and b2 > b0, which
      // the pre-test boxes guarantee); a detection that no track of the wave touches -- nearly all of them -- costs
      // nothing else.  Overlapping detections go through a 4-deep shift register per lane (all of them are kept when
      // there are at most KC; their order is irrelevant to the arg-max).
      auto overlaps = [&](const float4 f) -> bool {
        return (int)(f.z > tf.x) & (int)(tf.z > f.x) & (int)(f.w > tf.y) & (int)(tf.w > f.y);
      };
      // first detection of the segment that the lane's track does NOT touch (usually the very first one)
      int zero_j = -1;
      for (int j = j_lo; j < j_hi; ++j) {
        if (zero_j < 0 && !overlaps(fbox[j])) zero_j = j;
        if (__ballot(zero_j < 0) == 0ull) break;
      }
      int cnt = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
      auto visit = [&](int j, const float4 f) {
        const bool maybe = overlaps(f);
        if (__ballot(maybe) != 0ull) {
          c3 = maybe ? c2 : c3;
          c2 = maybe ? c1 : c2;
          c1 = maybe ? c0 : c1;
          c0 = maybe ? j : c0;
          cnt += maybe ? 1 : 0;
        }
      };
      int j = j_lo;
      for (; j + 4 <= j_hi; j += 4) {           // four broadcast reads in flight per round trip
        const float4 f0 = fbox[j], f1 = fbox[j + 1], f2 = fbox[j + 2], f3 = fbox[j + 3];
        visit(j, f0); visit(j + 1, f1); visit(j + 2, f2); visit(j + 3, f3);
      }
      for (; j < j_hi; ++j) visit(j, fbox[j]);
      double bv = 0.0;
      int bi = zero_j;                                        // better(0, zero_j, 0, -1); -1 when the segment had none
      if (cnt > KC) {                                         // crowded (or an everything-box): the plain exact scan
        bv = 0.0;
        bi = -1;
        for (int j = j_lo; j < j_hi; ++j) {
          const double v = iou64(dbox + (size_t)j * 4, tb);
          if (better(v, j, bv, bi)) { bv = v; bi = j; }
        }
      } else {
#pragma unroll
        for (int k = 0; k < KC; ++k) {
          const int cj = k == 0 ? c0 : (k == 1 ? c1 : (k == 2 ? c2 : c3));
          if (k < cnt) {
            const double v = iou64(dbox + (size_t)cj * 4, tb);
            if (better(v, cj, bv, bi)) { bv = v; bi = cj; }
          }
        }
      }
      if (in) {
        if (S == 1) {
          best_v[t] = bv;
          best_i[t] = bi;
        } else {
          s_pv[seg * (n_chunks * 64) + t] = bv;
          s_pi[seg * (n_chunks * 64) + t] = bi;
        }
      }
    }
    if (S > 1) {
      __syncthreads();
      for (int t = tid; t < T; t += TRK_THREADS) {
        double bv = s_pv[t];
        int bi = s_pi[t];
        for (int sg = 1; sg < S; ++sg) {
          const double ov = s_pv[sg * (n_chunks * 64) + t];
          const int oi = s_pi[sg * (n_chunks * 64) + t];
          if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        best_v[t] = bv;
        best_i[t] = bi;
      }
    }
  }
  __syncthreads();
  TT(1)

  if (wave == 0) {
    // ---- phase 2: greedy association in track order (:129-148), one wave -----------------------
    // Surviving tracks and finished ids are STAGED IN LDS, in the prefix [0, t] of the per-track arrays that track t and
    // its predecessors have already consumed (n_upd + n_fin <= t before track t writes), and stored to HBM by the whole
    // workgroup after the loop -- no global store (and the vmcnt(0) drain the volatile LDS accesses force) per track.
    int* fin_ids = (int*)best_v;
    int* det_claim = const_cast<int*>(det_tid);
    int n_alive = n;
    int n_upd = 0, n_fin = 0;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // 64 tracks at a time: every lane loads its track's phase-1 result with ONE LDS read per field.  If no track of the
    // chunk wants a detection that is already gone or that an earlier lane also wants, and the detections cannot run
    // out inside the chunk, the sequential loop of :129-148 has no cross-track dependence here and the chunk commits in
    // parallel (slots by prefix sums == what the loop would have assigned).  Otherwise the chunk runs the loop as
    // written, one track after the other, with the per-track values broadcast from the lanes that hold them.
    // ---- candidate form: the track takes its first free candidate ----
    // Only tracks that can interact are serialised.  A track is SIMPLE when its first candidate is free and appears in no
    // other candidate list of the chunk (dcnt, a per-detection reference count over the chunk's lists): whatever the others
    // do it takes that detection, and nobody else can want it.  The remaining wanting tracks run the reference loop in
    // track order among themselves (lanes 0..5 probe one candidate each); then the whole chunk commits with prefix sums --
    // the slots the loop would have assigned.  The detections cannot run out inside a chunk with n_want < n_alive;
    // otherwise the chunk runs the loop as written (the `break` of :130 needs the exact position).
    int* dcnt = (int*)s_pv;                    // [M] ints of scratch (s_pv belongs to the exact form's phase 1), zero between chunks
    static_assert(sizeof(double) * TRK_THREADS >= sizeof(int) * 1600, "s_pv holds one int per detection slot (M <= 1512)");
    if (fast) {
      for (int j = lane; j < n; j += 64) dcnt[j] = 0;
      __builtin_amdgcn_wave_barrier();
    }
    for (int t0 = 0; fast && t0 < T && n_alive > 0; t0 += 64) {
      const int t = t0 + lane;
      const bool in = t < T;
      const unsigned* cw = (const unsigned*)(cand + (size_t)(in ? t : 0) * TRK_CAND);   // 12 bytes per track, 4-byte aligned
      const unsigned c01 = in ? cw[0] : 0xFFFFFFFFu, c23 = in ? cw[1] : 0xFFFFFFFFu, c45 = in ? cw[2] : 0xFFFFFFFFu;
      static_assert(TRK_CAND == 6, "three dwords of candidates per track");
      const double tmax = in ? tmaxs[t] : 0.0;
      const int tlen = in ? tlens[t] : 0;
      const int tidv = in ? tids[t] : 0;
      const int bi = (int)(c01 & 0xFFFFu);
      const bool want = in && bi != 0xFFFF;                 // :134: some detection is above sigma_iou
      const unsigned long long want_m = __ballot(want);
      const int n_want = __popcll(want_m);
      if (want_m == 0ull) {                                 // nobody matches: only :146 is left
        const bool fin = in && tmax > sigma_h && tlen > t_min;
        const unsigned long long fin_m = __ballot(fin);
        __builtin_amdgcn_wave_barrier();
        if (fin) fin_ids[n_fin + __popcll(fin_m & lt_mask)] = tidv;
        n_fin += __popcll(fin_m);
        __builtin_amdgcn_wave_barrier();
        continue;
      }
      if (n_want < n_alive) {
        const int ck[TRK_CAND] = {(int)(c01 & 0xFFFFu), (int)(c01 >> 16), (int)(c23 & 0xFFFFu), (int)(c23 >> 16),
                                  (int)(c45 & 0xFFFFu), (int)(c45 >> 16)};
#pragma unroll
        for (int k = 0; k < TRK_CAND; ++k)
          if (ck[k] != 0xFFFF) atomicAdd(&dcnt[ck[k]], 1);
        __builtin_amdgcn_wave_barrier();
        const bool simple = want && det_tid[bi] == -1 && dcnt[bi] == 1;
        unsigned long long seq_m = __ballot(want && !simple);
        int mine = -1;                                      // the detection a serialised track ends up with
        while (seq_m != 0ull) {
          const int l = __ffsll((long long)seq_m) - 1;
          seq_m &= seq_m - 1ull;
          const unsigned a0 = __builtin_amdgcn_readlane(c01, l), a1 = __builtin_amdgcn_readlane(c23, l),
                         a2 = __builtin_amdgcn_readlane(c45, l);
          const int stid = __builtin_amdgcn_readlane(tidv, l);
          const unsigned pair = lane < 2 ? a0 : (lane < 4 ? a1 : a2);
          const int my = lane < TRK_CAND ? (int)((pair >> ((lane & 1) * 16)) & 0xFFFFu) : 0xFFFF;
          const bool freec = my != 0xFFFF && det_tid[my] == -1;
          const unsigned long long fm = __ballot(freec);
          if (fm != 0ull) {
            const int k = __ffsll((long long)fm) - 1;
            const unsigned pk = k < 2 ? a0 : (k < 4 ? a1 : a2);
            const int sbi = (int)((pk >> ((k & 1) * 16)) & 0xFFFFu);
            if (lane == 0) det_tid[sbi] = stid;
            if (lane == l) mine = sbi;
            __builtin_amdgcn_wave_barrier();   // LDS ops of one wave complete in order; keep the compiler honest
          }
        }
        const bool matched = simple || mine >= 0;
        const int sel = simple ? bi : mine;
        const unsigned long long mat_m = __ballot(matched);
        const bool fin = in && !matched && tmax > sigma_h && tlen > t_min;   // :146
        const unsigned long long fin_m = __ballot(fin);
        __builtin_amdgcn_wave_barrier();                    // every lane holds its inputs in registers from here on
        if (matched) {
          const int slot = n_upd + __popcll(mat_m & lt_mask);
          best_i[slot] = sel;                               // the track's new last box = detection sel
          tmaxs[slot] = tmax;                               // max(track, det) (:141) is taken when the survivors are stored
          tlens[slot] = tlen + 1;
          tids[slot] = tidv;
          if (simple) det_tid[sel] = tidv;
        }
        if (fin) fin_ids[n_fin + __popcll(fin_m & lt_mask)] = tidv;
#pragma unroll
        for (int k = 0; k < TRK_CAND; ++k)
          if (ck[k] != 0xFFFF) dcnt[ck[k]] = 0;             // scratch back to zero for the next chunk
        n_upd += __popcll(mat_m);
        n_alive -= __popcll(mat_m);
        n_fin += __popcll(fin_m);
        __builtin_amdgcn_wave_barrier();
        continue;
      }
      // the detections may run out inside this chunk: the loop as written, one track after the other
      const int cnt = (T - t0) < 64 ? (T - t0) : 64;
      const unsigned tmax_lo = (unsigned)__double_as_longlong(tmax), tmax_hi = (unsigned)(__double_as_longlong(tmax) >> 32);
      for (int l = 0; l < cnt; ++l) {
        if (n_alive == 0) break;                            // :130 has no else: remaining tracks vanish
        const unsigned a0 = __builtin_amdgcn_readlane(c01, l), a1 = __builtin_amdgcn_readlane(c23, l),
                       a2 = __builtin_amdgcn_readlane(c45, l);
        const double stmax = __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane(tmax_hi, l) << 32) |
                                                              (unsigned long long)(unsigned)__builtin_amdgcn_readlane(tmax_lo, l)));
        const int stlen = __builtin_amdgcn_readlane(tlen, l);
        const int stid = __builtin_amdgcn_readlane(tidv, l);
        // lanes 0..5 probe one candidate each; the first free one in list order is the arg-max over the remaining rows
        const unsigned pair = lane < 2 ? a0 : (lane < 4 ? a1 : a2);
        const int my = lane < TRK_CAND ? (int)((pair >> ((lane & 1) * 16)) & 0xFFFFu) : 0xFFFF;
        const bool freec = my != 0xFFFF && det_tid[my] == -1;
        const unsigned long long fm = __ballot(freec);
        if (fm != 0ull) {
          const int k = __ffsll((long long)fm) - 1;
          const unsigned pk = k < 2 ? a0 : (k < 4 ? a1 : a2);
          const int sbi = (int)((pk >> ((k & 1) * 16)) & 0xFFFFu);
          if (lane == 0) {
            best_i[n_upd] = sbi;
            tmaxs[n_upd] = stmax;
            tlens[n_upd] = stlen + 1;
            tids[n_upd] = stid;
            det_tid[sbi] = stid;
          }
          ++n_upd;
          --n_alive;
          __builtin_amdgcn_wave_barrier();
        } else if (stmax > sigma_h && stlen > t_min) {
          if (lane == 0) fin_ids[n_fin] = stid;
          ++n_fin;
        }
      }
    }
    // ---- exact form (fallback) ----
    for (int t0 = 0; !fast && t0 < T && n_alive > 0; t0 += 64) {
      const int t = t0 + lane;
      const bool in = t < T;
      double bv = in ? best_v[t] : 0.0;
      int bi = in ? best_i[t] : -1;
      const double tmax = in ? tmaxs[t] : 0.0;
      const int tlen = in ? tlens[t] : 0;
      const int tidv = in ? tids[t] : 0;
      const bool want = in && (bv > sigma_iou);             // :134 strict; NaN -> unmatched
      bool claimed = false, clash = false;
      if (want) {
        const int old = atomicCAS(&det_claim[bi], -1, -2 - lane);
        claimed = old == -1;
        clash = !claimed;                                   // taken by an earlier chunk, or wanted twice in this one
      }
      const unsigned long long want_m = __ballot(want);
      const int n_want = __popcll(want_m);
      if (__ballot(clash) == 0ull && n_want < n_alive) {
        const bool fin = in && !want && tmax > sigma_h && tlen > t_min;   // :146
        const unsigned long long fin_m = __ballot(fin);
        __builtin_amdgcn_wave_barrier();                    // every lane holds its inputs in registers from here on
        if (want) {
          const int slot = n_upd + __popcll(want_m & lt_mask);
          const double sc = dscore[bi];
          best_i[slot] = bi;                                // the track's new last box = detection bi
          tmaxs[slot] = (sc > tmax) ? sc : tmax;            // max(track, det)  :141
          tlens[slot] = tlen + 1;
          tids[slot] = tidv;
          det_claim[bi] = tidv;
        }
        if (fin) fin_ids[n_fin + __popcll(fin_m & lt_mask)] = tidv;
        n_upd += n_want;
        n_alive -= n_want;
        n_fin += __popcll(fin_m);
        __builtin_amdgcn_wave_barrier();
        continue;
      }
      if (claimed) det_claim[bi] = -1;                      // undo the probes of this chunk
      __builtin_amdgcn_wave_barrier();
      const int cnt = (T - t0) < 64 ? (T - t0) : 64;
      for (int l = 0; l < cnt; ++l) {
        if (n_alive == 0) break;                            // :130 has no else: remaining tracks vanish
        const int tt = t0 + l;
        double sbv = __shfl(bv, l, 64);
        int sbi = __shfl(bi, l, 64);
        const double stmax = __shfl(tmax, l, 64);
        const int stlen = __shfl(tlen, l, 64);
        const int stid = __shfl(tidv, l, 64);
        // A track whose best IoU over ALL detections is not above sigma_iou stays unmatched whatever has been deleted
        // (removing rows cannot raise the maximum; a NaN row belongs to a zero-area detection, which no track can
        // take), so only a real conflict -- the wanted detection went to an earlier track -- re-evaluates the row.
        if (sbv > sigma_iou && det_tid[sbi] != -1) {
          const double tb[4] = {cur.box[tt * 4 + 0], cur.box[tt * 4 + 1], cur.box[tt * 4 + 2], cur.box[tt * 4 + 3]};
          sbv = 0.0;
          sbi = -1;
          for (int j = lane; j < n; j += 64) {
            if (det_tid[j] != -1) continue;
            double v = iou64(dbox + (size_t)j * 4, tb);
            if (better(v, j, sbv, sbi)) { sbv = v; sbi = j; }
          }
          for (int o = 32; o > 0; o >>= 1) {
            double ov = __shfl_xor(sbv, o, 64);
            int oi = __shfl_xor(sbi, o, 64);
            if (better(ov, oi, sbv, sbi)) { sbv = ov; sbi = oi; }
          }
        }
        if (sbv > sigma_iou) {
          if (lane == 0) {
            const double sc = dscore[sbi];
            best_i[n_upd] = sbi;
            tmaxs[n_upd] = (sc > stmax) ? sc : stmax;
            tlens[n_upd] = stlen + 1;
            tids[n_upd] = stid;
            det_tid[sbi] = stid;
          }
          ++n_upd;
          --n_alive;
          __builtin_amdgcn_wave_barrier();     // LDS ops of one wave complete in order; keep the compiler honest
        } else if (stmax > sigma_h && stlen > t_min) {
          if (lane == 0) fin_ids[n_fin] = stid;
          ++n_fin;
        }
      }
    }

    TT(2)
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
    TT(3)
    if (lane == 0) {
      s_nupd = n_upd;
      s_nfin = n_fin;
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
  for (int u = tid; u < s_nupd; u += TRK_THREADS) {        // staged survivors -> the next active set
    const double* d = dbox + (size_t)best_i[u] * 4;
    nxt.box[u * 4 + 0] = d[0];
    nxt.box[u * 4 + 1] = d[1];
    nxt.box[u * 4 + 2] = d[2];
    nxt.box[u * 4 + 3] = d[3];
    const double sc = dscore[best_i[u]];                  // :141 max(track['max_score'], det score) (idempotent for the exact form)
    nxt.max_score[u] = (sc > tmaxs[u]) ? sc : tmaxs[u];
    nxt.len[u] = tlens[u];
    nxt.id[u] = tids[u];
  }
  for (int k = tid; k < s_nfin; k += TRK_THREADS) fin_log[k] = ((const int*)best_v)[k];
  TT(4)
  __syncthreads();   // det_tid / dbox are rewritten by the next frame; the new active set is visible to every wave
  }
  if (tid == 0) {
    st->log_cursor = s_cursor;
    st->n_active = s_n_active;
    st->next_id = s_next_id;
    st->frame_num = s_frame_num;
    for (int k = 0; k < 4; ++k) st->form_frames[k] += s_form[k];
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

// Host-blocking copy / fill on the tracker's own stream, never on the legacy stream: ROCm refuses legacy-stream work while a
// HIP graph is being captured on another host thread (a detector handle capturing its first forward), and the periodic log
// flush of a running pipeline can fall exactly there.
hipError_t copy_sync(fdt_tracker* t, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, t->own_stream);
  return e != hipSuccess ? e : hipStreamSynchronize(t->own_stream);
}

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
  FDT_HIP(copy_sync(t, &hs, t->d_state, sizeof(hs), hipMemcpyDeviceToHost));
  FDT_REQUIRE(!hs.overflow, FDT_ERR_STATE, "fdt_tracker: event log overflow");
  t->h_log.resize((size_t)hs.log_cursor);
  if (hs.log_cursor) FDT_HIP(copy_sync(t, t->h_log.data(), t->d_log, hs.log_cursor, hipMemcpyDeviceToHost));
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
  FDT_HIP(copy_sync(t, (char*)t->d_state + offsetof(fdt::TrkState, log_cursor), &zero, 8, hipMemcpyHostToDevice));
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

namespace fdt {
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-(function, device) setting: it is raised ONCE per device to the
// constant ceiling, never to a tracker's own size (a second, smaller tracker must not lower it under a live larger one).
static hipError_t set_track_kernel_lds(int bytes) {
  static std::atomic<unsigned char> done[16];
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 16) return hipErrorInvalidDevice;
  if (done[dev].load(std::memory_order_acquire)) return hipSuccess;
  e = hipFuncSetAttribute((const void*)track_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done[dev].store(1, std::memory_order_release);
  return e;
}
}  // namespace fdt

extern "C" fdt_tracker* fdt_tracker_create(double sigma_iou, double sigma_h, int t_min, int max_dets,
                                           int log_frames) {
  if (max_dets < 1 || log_frames < 1) {
    set_error("fdt_tracker_create: max_dets and log_frames must be >= 1");
    return nullptr;
  }
  // 160 KB of LDS per CU minus the kernel's static arrays (the 12 KB of per-segment partial arg-max results + counters)
  constexpr long long kDynLds = 160 * 1024 - (12 * 1024 + 256);
  static_assert(1500ll * fdt::TRK_LDS_PER_SLOT <= kDynLds, "2 x top_k = 1500 slots (what the reference's Detect can emit) must fit");
  if ((long long)max_dets * fdt::TRK_LDS_PER_SLOT > kDynLds) {
    set_error("fdt_tracker_create: max_dets %d does not fit the LDS-resident frame state (limit %d)", max_dets,
              (int)(kDynLds / fdt::TRK_LDS_PER_SLOT));
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
            hipMemsetAsync(t->d_state, 0, sizeof(fdt::TrkState), t->own_stream) == hipSuccess &&
            hipStreamSynchronize(t->own_stream) == hipSuccess &&
            fdt::set_track_kernel_lds((int)kDynLds) == hipSuccess;
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
  FDT_HIP(hipMemsetAsync(t->d_state, 0, sizeof(fdt::TrkState), t->own_stream));
  FDT_HIP(hipStreamSynchronize(t->own_stream));
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
  FDT_HIP(copy_sync(t, &hs, t->d_state, sizeof(hs), hipMemcpyDeviceToHost));
  std::vector<int> ids(hs.n_active);
  if (hs.n_active)
    FDT_HIP(copy_sync(t, ids.data(), t->set[t->cur].id, (size_t)hs.n_active * 4, hipMemcpyDeviceToHost));
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

// Which association form the frames so far ran (since create / reset); synchronises with the stream of the last step.
extern "C" int fdt_tracker_stats(fdt_tracker* t, long long* frames, long long* form_frames) {
  FDT_REQUIRE(t, FDT_ERR_ARG, "fdt_tracker_stats: null handle");
  hipStream_t st = t->last_stream ? t->last_stream : t->own_stream;
  FDT_HIP(hipStreamSynchronize(st));
  fdt::TrkState hs;
  FDT_HIP(copy_sync(t, &hs, t->d_state, sizeof(hs), hipMemcpyDeviceToHost));
  if (frames) *frames = hs.frame_num;
  if (form_frames)
    for (int k = 0; k < 4; ++k) form_frames[k] = hs.form_frames[k];
  return FDT_OK;
}

#ifdef FDT_TRK_TIMING
extern "C" int fdt_debug_trk_times(long long* out) {
  FDT_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(fdt::g_trk_time), 64));
  long long z[8] = {0};
  FDT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(fdt::g_trk_time), z, 64));
  return FDT_OK;
}
#endif
