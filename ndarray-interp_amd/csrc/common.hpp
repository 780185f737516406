// csrc/common.hpp -- shared host-side plumbing of libndinterp_hip.so (error text, HIP checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/ndinterp.h"

namespace ndi {

inline std::string& tls_error() {
  static thread_local std::string s;
  return s;
}

inline ndi_status fail(ndi_status st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  tls_error() = buf;
  return st;
}

struct HipFailure {
  hipError_t err;
  const char* what;
  int line;
};

#define NDI_HIP(expr)                                                     \
  do {                                                                    \
    hipError_t e__ = (expr);                                              \
    if (e__ != hipSuccess) throw ::ndi::HipFailure{e__, #expr, __LINE__}; \
  } while (0)

inline ndi_status from_hip(const HipFailure& f) {
  return fail(NDI_HIP_ERROR, "HIP error %d (%s) at %s [csrc line %d]", (int)f.err,
              hipGetErrorString(f.err), f.what, f.line);
}

template <class T>
struct DType;
template <>
struct DType<float> {
  static constexpr int id = NDI_F32;
};
template <>
struct DType<double> {
  static constexpr int id = NDI_F64;
};

}  // namespace ndi
