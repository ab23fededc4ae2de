// Shared host-side helpers for libdiffsci_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <initializer_list>

#include "../../include/diffsci_hip.h"

namespace ds {

void set_error(const char* fmt, ...);
const char* get_error();

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s", what, hipGetErrorString(e));
  return DS_ERR_HIP;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device).  The attribute belongs to the
// device's copy of the code object, so a process that drives several GPUs must set it on each; the guard is
// a bit mask per kernel instantiation, updated atomically (two host threads racing on the first launch both
// set the attribute -- idempotent -- and neither launches before its own call has returned).
template <auto Kernel>
inline int ensure_dynamic_lds(int bytes, const char* what) {
  static std::atomic<uint64_t> done{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return DS_OK;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return hip_fail(e, what);
  done.fetch_or(bit, std::memory_order_release);
  return DS_OK;
}

}  // namespace ds

#define DS_REQUIRE(cond, code, ...)  \
  do {                               \
    if (!(cond)) {                   \
      ds::set_error(__VA_ARGS__);    \
      return (code);                 \
    }                                \
  } while (0)

#define DS_CHECK_LAUNCH(what)                          \
  do {                                                 \
    hipError_t e__ = hipGetLastError();                \
    if (e__ != hipSuccess) return ds::hip_fail(e__, what); \
  } while (0)
