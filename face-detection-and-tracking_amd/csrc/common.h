// Shared helpers for libfdt_hip.so (host side): error reporting + HIP call checking.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/fdt.h"

namespace fdt {

void set_error(const char* fmt, ...);
const char* get_error();

struct HipError {
  hipError_t code;
};

}  // namespace fdt

// Returns FDT_ERR_HIP from the enclosing function when a HIP call fails.
#define FDT_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      fdt::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,  \
                     __LINE__);                                                        \
      return FDT_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

#define FDT_REQUIRE(cond, code, ...)  \
  do {                                \
    if (!(cond)) {                    \
      fdt::set_error(__VA_ARGS__);    \
      return (code);                  \
    }                                 \
  } while (0)

#define FDT_TRY(expr)          \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != FDT_OK) return rc_; \
  } while (0)

// Launch check: catches bad launch configs immediately (kernel faults surface at the next sync).
#define FDT_LAUNCH_CHECK() FDT_HIP(hipGetLastError())

namespace fdt {

// RAII device buffer used by the host-pointer entry points.
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t n) {
    if (p) {
      (void)hipFree(p);
      p = nullptr;
    }
    bytes = n;
    if (n == 0) return FDT_OK;
    FDT_HIP(hipMalloc(&p, n));
    return FDT_OK;
  }
  template <class T>
  T* as() {
    return reinterpret_cast<T*>(p);
  }
};

// Host-blocking copy on an explicit stream.  Nothing in this library may use the legacy (null) stream: while a HIP graph is
// being captured on ANOTHER host thread (a detector handle recording its forward), ROCm fails legacy-stream work with
// "operation would make the legacy stream depend on a capturing blocking stream", and distinct handles / the host-pointer
// entry points must stay usable from distinct host threads (include/fdt.h).
static inline hipError_t copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
  return e != hipSuccess ? e : hipStreamSynchronize(st);
}
// The calling host thread's stream for the host-pointer convenience entry points (fdt_detect, fdt_nms, fdt_conv2d, ...) and
// for a NULL `stream` argument of the handle-less "_dev" entry points: created on first use, one per thread and device,
// non-blocking, kept for the life of the thread (api.cpp; exported as fdt_thread_stream).  nullptr = creation failed: callers
// turn that into FDT_ERR_HIP, never into legacy-stream work.
hipStream_t thread_stream();

// hipDeviceSynchronize() fails with "operation not permitted when stream is capturing" while ANY host thread has a stream
// capture open -- also a thread-local one of another thread (ROCm 7.2) --, and distinct handles are driven from distinct
// threads (one pipeline per rank thread in the loop-back tests, the reference-side threading contract of include/fdt.h).
// So the library's device-wide waits and its captures exclude each other: a capture holds the lock shared (captures of
// different threads may overlap), device_sync() takes it exclusively -- it waits for the few hundred microseconds an open
// capture lasts (captures only enqueue) and holds new ones back while the device drains.
hipError_t device_sync();
void capture_lock_shared();
void capture_unlock_shared();
// a whole teardown (device-wide wait + hipFree / hipGraphExecDestroy, which synchronise implicitly) with captures held back;
// re-entrant on a thread (fdt_pipeline_destroy destroys its clones)
void exclusive_begin();
void exclusive_end();
struct ExclusiveDevice {
  ExclusiveDevice() { exclusive_begin(); }
  ~ExclusiveDevice() { exclusive_end(); }
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }

}  // namespace fdt
