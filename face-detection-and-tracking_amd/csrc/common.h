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

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }

}  // namespace fdt
