// Shared host-side context for libsfm_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include "sfm_amd.h"

struct sfm_ctx {
  int device;
  hipStream_t stream;
  char err[512];
  double* pinned;          // SFM_SC_COUNT doubles of pinned host memory for scalar read-back
};

static inline int sfm_fail(sfm_ctx* h, int code, const char* what, const char* detail) {
  if (h) snprintf(h->err, sizeof(h->err), "%s: %s", what, detail ? detail : "");
  return code;
}

#define SFM_HIP(h, call)                                                        \
  do {                                                                          \
    hipError_t e_ = (call);                                                     \
    if (e_ != hipSuccess) return sfm_fail((h), SFM_ERR_HIP, #call, hipGetErrorString(e_)); \
  } while (0)

#define SFM_LAUNCH_CHECK(h, name)                                               \
  do {                                                                          \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ != hipSuccess) return sfm_fail((h), SFM_ERR_HIP, name, hipGetErrorString(e_)); \
  } while (0)

static inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }
static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }
