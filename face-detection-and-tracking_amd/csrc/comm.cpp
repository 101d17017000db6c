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
#include <rccl/rccl.h>

#include <vector>

#include "common.h"

struct fdt_comm {
  std::vector<ncclComm_t> comms;   // one per local device
  std::vector<int> devices;
  int world = 0;
  int rank0 = 0;                   // global rank of comms[0] (init_rank) / 0 (init_all)
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

extern "C" fdt_comm* fdt_comm_init_rank(int world, int rank, const char* id, int device) {
  if (world < 1 || rank < 0 || rank >= world || !id) {
    fdt::set_error("fdt_comm_init_rank: bad argument (world %d, rank %d)", world, rank);
    return nullptr;
  }
  if (hipSetDevice(device) != hipSuccess) {
    fdt::set_error("fdt_comm_init_rank: hipSetDevice(%d) failed", device);
    return nullptr;
  }
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
  if (n_local) *n_local = (int)c->comms.size();
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
