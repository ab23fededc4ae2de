// Shared host-side helpers for libdiffsci_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/diffsci_hip.h"

namespace ds {

void set_error(const char* fmt, ...);
const char* get_error();

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s", what, hipGetErrorString(e));
  return DS_ERR_HIP;
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
