// Host-side helpers shared by the two translation units of libreal3daug_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>

#include "../../include/real3daug_hip.h"

namespace r3d {

std::string &last_error_ref();

inline int fail(int code, const char *what) {
  last_error_ref() = what;
  return code;
}

inline int fail_hip(hipError_t e, const char *where) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s: %s", where, hipGetErrorString(e));
  last_error_ref() = buf;
  return R3D_E_HIP;
}

#define R3D_HIP(call)                                        \
  do {                                                       \
    hipError_t e_ = (call);                                  \
    if (e_ != hipSuccess) return r3d::fail_hip(e_, #call);   \
  } while (0)

#define R3D_LAUNCHED(name)                                   \
  do {                                                       \
    hipError_t e_ = hipGetLastError();                       \
    if (e_ != hipSuccess) return r3d::fail_hip(e_, name);    \
  } while (0)

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Carves a caller-provided workspace into aligned pieces.
struct Carver {
  uintptr_t base;
  size_t off = 0;
  explicit Carver(void *p) : base(reinterpret_cast<uintptr_t>(p)) {}
  template <class T>
  T *take(size_t count) {
    T *p = reinterpret_cast<T *>(base + off);
    off = align_up(off + count * sizeof(T));
    return p;
  }
};

inline int blocks_for(int64_t n, int per_block, int cap = 1 << 20) {
  int64_t b = (n + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

}  // namespace r3d
