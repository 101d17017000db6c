// Library-level entry points: error string, device queries.
#include <mutex>
#include <shared_mutex>

#include "common.h"

namespace fdt {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }
}  // namespace fdt

extern "C" const char* fdt_last_error(void) { return fdt::get_error(); }
extern "C" int fdt_version(void) { return 100; }

extern "C" int fdt_device_count(int* n) {
  FDT_REQUIRE(n, FDT_ERR_ARG, "fdt_device_count: null pointer");
  *n = 0;
  FDT_HIP(hipGetDeviceCount(n));
  return FDT_OK;
}

extern "C" int fdt_device_name(int dev, char* buf, int buflen) {
  FDT_REQUIRE(buf && buflen > 0, FDT_ERR_ARG, "fdt_device_name: bad buffer");
  hipDeviceProp_t p;
  FDT_HIP(hipGetDeviceProperties(&p, dev));
  snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
  return FDT_OK;
}

extern "C" int fdt_set_device(int dev) {
  FDT_HIP(hipSetDevice(dev));
  return FDT_OK;
}

namespace fdt {
hipStream_t thread_stream() {
  thread_local hipStream_t s[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!s[dev] && hipStreamCreateWithFlags(&s[dev], hipStreamNonBlocking) != hipSuccess) s[dev] = nullptr;
  return s[dev];
}
}  // namespace fdt

namespace fdt {
static std::shared_mutex g_capture_mu;
static thread_local int t_exclusive = 0, t_capturing = 0;
hipError_t device_sync() {
  if (t_capturing) return hipErrorStreamCaptureUnsupported;      // would wait for itself
  if (t_exclusive) return hipDeviceSynchronize();
  std::unique_lock<std::shared_mutex> lk(g_capture_mu);
  return hipDeviceSynchronize();
}
void capture_lock_shared() {
  g_capture_mu.lock_shared();
  ++t_capturing;
}
void capture_unlock_shared() {
  --t_capturing;
  g_capture_mu.unlock_shared();
}
void exclusive_begin() {
  if (t_exclusive++ == 0) g_capture_mu.lock();
}
void exclusive_end() {
  if (--t_exclusive == 0) g_capture_mu.unlock();
}
}  // namespace fdt

extern "C" int fdt_thread_stream(void** stream) {
  FDT_REQUIRE(stream, FDT_ERR_ARG, "fdt_thread_stream: null output");
  const hipStream_t st = fdt::thread_stream();
  FDT_REQUIRE(st, FDT_ERR_HIP, "fdt_thread_stream: could not create the calling thread's private stream");
  *stream = (void*)st;
  return FDT_OK;
}

extern "C" int fdt_device_synchronize(void) {
  FDT_HIP(fdt::device_sync());
  return FDT_OK;
}

extern "C" int fdt_device_mem_info(long long* free_bytes, long long* total_bytes) {
  size_t f = 0, t = 0;
  FDT_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = (long long)f;
  if (total_bytes) *total_bytes = (long long)t;
  return FDT_OK;
}

// A stream confined to a share of the compute units: partition `part` of `parts` (1, 2 or 4) equal slices of EVERY XCD's
// CUs.  On this chip bit i of hipExtStreamCreateWithCUMask's mask is CU i / 8 of XCD i % 8 and every XCD must keep at
// least one CU (tools/microbench/cu_mask_probe.hip), so a partition is a contiguous run of 256 / parts mask bits.  Kernels
// of streams on different partitions run side by side by construction instead of by the dispatcher's leave.
extern "C" int fdt_stream_create_partition(int part, int parts, void** stream) {
  FDT_REQUIRE(stream && (parts == 1 || parts == 2 || parts == 4) && part >= 0 && part < parts, FDT_ERR_ARG,
              "fdt_stream_create_partition: partition %d of %d", part, parts);
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const int per = 256 / parts;
  for (int i = part * per; i < (part + 1) * per; ++i) mask[i / 32] |= 1u << (i % 32);
  hipStream_t st = nullptr;
  FDT_HIP(hipExtStreamCreateWithCUMask(&st, 8, mask));
  *stream = (void*)st;
  return FDT_OK;
}

extern "C" int fdt_stream_destroy(void* stream) {
  FDT_REQUIRE(stream, FDT_ERR_ARG, "fdt_stream_destroy: null stream");
  FDT_HIP(hipStreamDestroy((hipStream_t)stream));
  return FDT_OK;
}
