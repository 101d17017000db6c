// The one exchange step of the frame-parallel path (SURVEY.md 8(e), 8(b) "Multi-GPU"): an all-gather of the fixed-size
// per-frame Detect records over RCCL / xGMI, behind the C ABI.  The reference has no multi-GPU path (its only trace is
// the commented DataParallel line MyTrain_repo.py:71); the record layout is Detect's output [num_classes, top_k, 5]
// (layers/functions/detection.py:48,82), which the sequential association (iouTracke_cal.py:117-156) consumes in frame
// order == rank order.
//
// Two ways to build a communicator:
//   fdt_comm_init_rank : one process per GPU (what bench.py uses); rank 0 makes the 128-byte id with
//                        fdt_comm_unique_id and hands it to the other ranks by any host channel.
//   fdt_comm_init_all  : one process driving n GPUs (ncclCommInitAll); fdt_allgather_dets is then called once per
//                        local device inside fdt_comm_group_begin / _end.
// The collective is enqueued on the caller's stream; nothing here synchronises with the host.
//
// LOOP-BACK form (fdt_comm_unique_id_local): the `world` ranks are host THREADS of one process sharing one GPU -- RCCL refuses
// two ranks on one device, and the build and test boxes have one.  fdt_comm_init_rank recognises the id and joins the ranks in
// process memory; fdt_allgather_dets then is `world` device-to-device copies per rank on the rank's own stream, ordered by
// events, behind a host rendezvous of the ranks (a collective: every rank must call it, like the RCCL one).  Same entry points,
// same buffers, same stream semantics for the caller -- it exists so that everything ABOVE the collective (fdt_pipeline_* at
// world > 1: gathered-record indexing, frame order, the tracker stream) runs in the one-GPU test suite.
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "common.h"

namespace {
struct LocalGroup {                // the shared half of a loop-back communicator
  int world = 0, device = 0;
  std::mutex mu;
  std::condition_variable cv;
  int joined = 0;
  int arrived = 0;
  long long generation = 0;
  bool broken = false;             // a rank timed out or failed inside a collective: every later call fails
  std::vector<const float*> src;   // per rank: the local buffer of the collective in progress
  std::vector<long long> count;
  std::vector<hipEvent_t> ready, copied;
  ~LocalGroup() {
    for (auto e : ready)
      if (e) (void)hipEventDestroy(e);
    for (auto e : copied)
      if (e) (void)hipEventDestroy(e);
  }
  // host rendezvous of the ranks; false = some rank did not arrive within `seconds` (the group is broken from then on)
  bool barrier(double seconds) {
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return false;
    const long long gen = generation;
    if (++arrived == world) {
      arrived = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    const bool ok = cv.wait_for(lk, std::chrono::duration<double>(seconds), [&] { return generation != gen || broken; });
    if (!ok || broken) {
      broken = true;
      cv.notify_all();
      return false;
    }
    return true;
  }
};
constexpr char kLocalMagic[8] = {'F', 'D', 'T', 'L', 'O', 'O', 'P', '1'};
std::mutex g_local_mu;
std::map<unsigned long long, std::shared_ptr<LocalGroup>> g_local_groups;   // id token -> group, until every rank has joined
unsigned long long g_local_next = 1;
constexpr double kLocalTimeoutS = 120.0;
}  // namespace

struct fdt_comm {
  std::vector<ncclComm_t> comms;   // one per local device
  std::vector<int> devices;
  int world = 0;
  int rank0 = 0;                   // global rank of comms[0] (init_rank) / 0 (init_all)
  std::shared_ptr<LocalGroup> local;   // loop-back form: no RCCL communicator
};

#define FDT_NCCL(call)                                                                         \
  do {                                                                                         \
    ncclResult_t r_ = (call);                                                                  \
    if (r_ != ncclSuccess) {                                                                   \
      fdt::set_error("%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
      return FDT_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

static_assert(sizeof(ncclUniqueId) == FDT_COMM_ID_BYTES, "fdt.h: FDT_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");

extern "C" int fdt_comm_unique_id(char* id_out) {
  FDT_REQUIRE(id_out, FDT_ERR_ARG, "fdt_comm_unique_id: null buffer");
  ncclUniqueId id;
  FDT_NCCL(ncclGetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return FDT_OK;
}

extern "C" int fdt_comm_unique_id_local(char* id_out) {
  FDT_REQUIRE(id_out, FDT_ERR_ARG, "fdt_comm_unique_id_local: null buffer");
  memset(id_out, 0, FDT_COMM_ID_BYTES);
  memcpy(id_out, kLocalMagic, sizeof(kLocalMagic));
  std::lock_guard<std::mutex> lk(g_local_mu);
  const unsigned long long token = g_local_next++;
  memcpy(id_out + sizeof(kLocalMagic), &token, sizeof(token));
  g_local_groups[token] = std::make_shared<LocalGroup>();
  return FDT_OK;
}

// the loop-back branch of fdt_comm_init_rank: join the group named by the id; returns when all `world` ranks have joined
static fdt_comm* init_rank_local(int world, int rank, const char* id, int device) {
  unsigned long long token = 0;
  memcpy(&token, id + sizeof(kLocalMagic), sizeof(token));
  std::shared_ptr<LocalGroup> g;
  {
    std::lock_guard<std::mutex> lk(g_local_mu);
    auto it = g_local_groups.find(token);
    if (it != g_local_groups.end()) g = it->second;
  }
  if (!g) {
    fdt::set_error("fdt_comm_init_rank: unknown (or already complete) loop-back id");
    return nullptr;
  }
  std::unique_lock<std::mutex> lk(g->mu);
  if (g->world == 0) {             // the first rank to arrive shapes the group
    g->world = world;
    g->device = device;
    g->src.assign(world, nullptr);
    g->count.assign(world, 0);
    g->ready.assign(world, nullptr);
    g->copied.assign(world, nullptr);
    for (int r = 0; r < world; ++r)
      if (hipEventCreateWithFlags(&g->ready[r], hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&g->copied[r], hipEventDisableTiming) != hipSuccess) {
        g->broken = true;
        fdt::set_error("fdt_comm_init_rank: hipEventCreate failed (loop-back)");
        return nullptr;
      }
  }
  if (g->world != world || g->device != device || g->broken) {
    fdt::set_error("fdt_comm_init_rank: loop-back ranks disagree (world %d vs %d, device %d vs %d)", world, g->world, device,
                   g->device);
    return nullptr;
  }
  ++g->joined;
  g->cv.notify_all();
  if (!g->cv.wait_for(lk, std::chrono::duration<double>(kLocalTimeoutS), [&] { return g->joined == g->world || g->broken; }) ||
      g->broken) {
    g->broken = true;
    g->cv.notify_all();
    fdt::set_error("fdt_comm_init_rank: only %d of %d loop-back ranks joined within %.0f s", g->joined, world, kLocalTimeoutS);
    return nullptr;
  }
  lk.unlock();
  {
    std::lock_guard<std::mutex> lk2(g_local_mu);
    g_local_groups.erase(token);     // complete: the ranks' handles keep it alive
  }
  fdt_comm* fc = new fdt_comm();
  fc->devices.push_back(device);
  fc->world = world;
  fc->rank0 = rank;
  fc->local = g;
  return fc;
}

// all-gather of the loop-back form, called by every rank (from its own host thread) with its own stream
static int allgather_local(fdt_comm* c, const float* local_dev, float* all_dev, long long n, hipStream_t st) {
  LocalGroup* g = c->local.get();
  const int r = c->rank0;
  FDT_HIP(hipEventRecord(g->ready[r], st));      // everything that produces local_dev precedes this on the rank's stream
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->src[r] = local_dev;
    g->count[r] = n;
  }
  FDT_REQUIRE(g->barrier(kLocalTimeoutS), FDT_ERR_STATE,
              "fdt_allgather_dets (loop-back): not every rank entered the collective within %.0f s", kLocalTimeoutS);
  for (int q = 0; q < g->world; ++q) {
    FDT_REQUIRE(g->count[q] == n, FDT_ERR_ARG, "fdt_allgather_dets (loop-back): rank %d sends %lld floats, rank %d %lld", q,
                g->count[q], r, n);
    if (q != r) FDT_HIP(hipStreamWaitEvent(st, g->ready[q], 0));
    if (all_dev + q * n != g->src[q])
      FDT_HIP(hipMemcpyAsync(all_dev + q * n, g->src[q], (size_t)n * 4, hipMemcpyDeviceToDevice, st));
  }
  FDT_HIP(hipEventRecord(g->copied[r], st));
  // like the RCCL kernel, the collective is complete on a rank's stream only when every rank has read that rank's buffer
  // (the caller may overwrite local_dev behind it)
  FDT_REQUIRE(g->barrier(kLocalTimeoutS), FDT_ERR_STATE,
              "fdt_allgather_dets (loop-back): not every rank finished the collective within %.0f s", kLocalTimeoutS);
  for (int q = 0; q < g->world; ++q)
    if (q != r) FDT_HIP(hipStreamWaitEvent(st, g->copied[q], 0));
  return FDT_OK;
}

extern "C" fdt_comm* fdt_comm_init_rank(int world, int rank, const char* id, int device) {
  if (world < 1 || rank < 0 || rank >= world || !id) {
    fdt::set_error("fdt_comm_init_rank: bad argument (world %d, rank %d)", world, rank);
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) {
    fdt::set_error("fdt_comm_init_rank: hipSetDevice(%d) failed", device);
    return nullptr;
  }
  if (memcmp(id, kLocalMagic, sizeof(kLocalMagic)) == 0) return init_rank_local(world, rank, id, device);
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t c = nullptr;
  ncclResult_t r = ncclCommInitRank(&c, world, uid, rank);
  if (r != ncclSuccess) {
    fdt::set_error("ncclCommInitRank failed: %s", ncclGetErrorString(r));
    return nullptr;
  }
  fdt_comm* fc = new fdt_comm();
  fc->comms.push_back(c);
  fc->devices.push_back(device);
  fc->world = world;
  fc->rank0 = rank;
  return fc;
}

extern "C" fdt_comm* fdt_comm_init_all(int n_dev, const int* dev_ids) {
  if (n_dev < 1 || !dev_ids) {
    fdt::set_error("fdt_comm_init_all: bad argument");
    return nullptr;
  }
  fdt_comm* fc = new fdt_comm();
  fc->comms.resize(n_dev);
  fc->devices.assign(dev_ids, dev_ids + n_dev);
  fc->world = n_dev;
  ncclResult_t r = ncclCommInitAll(fc->comms.data(), n_dev, dev_ids);
  if (r != ncclSuccess) {
    fdt::set_error("ncclCommInitAll failed: %s", ncclGetErrorString(r));
    delete fc;
    return nullptr;
  }
  return fc;
}

extern "C" int fdt_comm_world(fdt_comm* c, int* world, int* n_local) {
  FDT_REQUIRE(c, FDT_ERR_ARG, "fdt_comm_world: null handle");
  if (world) *world = c->world;
  if (n_local) *n_local = c->local ? 1 : (int)c->comms.size();
  return FDT_OK;
}

extern "C" int fdt_comm_group_begin(void) {
  FDT_NCCL(ncclGroupStart());
  return FDT_OK;
}
extern "C" int fdt_comm_group_end(void) {
  FDT_NCCL(ncclGroupEnd());
  return FDT_OK;
}

extern "C" int fdt_allgather_dets(fdt_comm* c, int local_index, const float* local_dev, float* all_dev,
                                  long long floats_per_rank, void* stream) {
  FDT_REQUIRE(c && local_dev && all_dev && floats_per_rank >= 1, FDT_ERR_ARG, "fdt_allgather_dets: bad argument");
  if (c->local) {
    FDT_REQUIRE(local_index == 0, FDT_ERR_ARG, "fdt_allgather_dets: a loop-back rank has one local device");
    const hipStream_t lst = stream ? (hipStream_t)stream : fdt::thread_stream();
    FDT_REQUIRE(lst, FDT_ERR_HIP, "fdt_allgather_dets: could not create the calling thread's private stream");
    return allgather_local(c, local_dev, all_dev, floats_per_rank, lst);
  }
  FDT_REQUIRE(local_index >= 0 && local_index < (int)c->comms.size(), FDT_ERR_ARG,
              "fdt_allgather_dets: local index %d out of range (%d local devices)", local_index, (int)c->comms.size());
  if (c->comms.size() > 1) FDT_HIP(hipSetDevice(c->devices[local_index]));
  // NULL = the calling thread's private stream (fdt_thread_stream), never the legacy stream: legacy-stream work fails while
  // another host thread captures a HIP graph (fdt.h, threading contract)
  const hipStream_t st = stream ? (hipStream_t)stream : fdt::thread_stream();
  FDT_REQUIRE(st, FDT_ERR_HIP, "fdt_allgather_dets: could not create the calling thread's private stream");
  FDT_NCCL(ncclAllGather(local_dev, all_dev, (size_t)floats_per_rank, ncclFloat32, c->comms[local_index], st));
  return FDT_OK;
}

extern "C" void fdt_comm_destroy(fdt_comm* c) {
  if (!c) return;
  for (size_t i = 0; i < c->comms.size(); ++i)
    if (c->comms[i]) (void)ncclCommDestroy(c->comms[i]);
  delete c;
}
