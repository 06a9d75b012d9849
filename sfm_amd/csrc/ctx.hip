// Handle management for libsfm_amd.so.
#include "ba_internal.h"

extern "C" const char* sfm_version(void) { return "sfm_amd 0.1 (gfx950)"; }

extern "C" int sfm_create(int device, sfm_handle* out) {
  if (!out) return SFM_ERR_ARG;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || device < 0 || device >= ndev) {
    fprintf(stderr, "sfm_create: hipGetDeviceCount -> %s, %d device(s), asked for %d\n", hipGetErrorString(e), ndev, device);
    return SFM_ERR_HIP;
  }
  if ((e = hipSetDevice(device)) != hipSuccess) {
    fprintf(stderr, "sfm_create: hipSetDevice(%d) -> %s\n", device, hipGetErrorString(e));
    return SFM_ERR_HIP;
  }
  sfm_ctx* h = new sfm_ctx();
  h->device = device;
  h->stream = nullptr;
  h->err[0] = 0;
  h->pinned = nullptr;
  h->profiling = 0;
  h->scratch = nullptr;
  h->scratch_bytes = 0;
  h->comm = nullptr; h->comm_owned = 0; h->comm_ranks = 0; h->comm_rank = 0;
  h->cgs_persist_off = 0;
  h->cgs_xcd_off = 0;
  memset(h->prof, 0, sizeof(h->prof));
  if ((e = hipHostMalloc((void**)&h->pinned, SFM_PINNED_DOUBLES * sizeof(double), hipHostMallocDefault)) != hipSuccess) {
    fprintf(stderr, "sfm_create: hipHostMalloc -> %s\n", hipGetErrorString(e));
    delete h;
    return SFM_ERR_HIP;
  }
  h->cg_event = nullptr;
  if ((e = hipEventCreateWithFlags(&h->cg_event, hipEventDisableTiming)) != hipSuccess) {
    fprintf(stderr, "sfm_create: hipEventCreate -> %s\n", hipGetErrorString(e));
    (void)hipHostFree(h->pinned);
    delete h;
    return SFM_ERR_HIP;
  }
  *out = h;
  return SFM_OK;
}

// Device scratch owned by the handle: grown (never shrunk) on demand.  Growing synchronises the stream first, so
// work already enqueued that still uses the old buffer has finished before it is freed.
void* sfm_scratch(sfm_ctx* h, size_t bytes) {
  if (bytes <= h->scratch_bytes) return h->scratch;
  (void)hipStreamSynchronize(h->stream);
  if (h->scratch) (void)hipFree(h->scratch);
  h->scratch = nullptr; h->scratch_bytes = 0;
  const size_t want = (bytes + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
  if (hipMalloc(&h->scratch, want) != hipSuccess) { h->scratch = nullptr; return nullptr; }
  h->scratch_bytes = want;
  return h->scratch;
}

void sfm_prof_fold(sfm_ctx* h, int slot) {
  sfm_prof_slot& s = h->prof[slot];
  for (int i = 0; i < s.pending; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(s.stop[i]) == hipSuccess && hipEventElapsedTime(&ms, s.start[i], s.stop[i]) == hipSuccess) {
      s.total_ms += ms;
      s.count++;
    }
  }
  s.pending = 0;
}

extern "C" int sfm_set_profiling(sfm_handle h, int enabled) {
  if (!h) return SFM_ERR_ARG;
  if (enabled && !h->prof[0].start[0]) {
    for (int k = 0; k < SFM_PROF_COUNT; ++k)
      for (int i = 0; i < SFM_PROF_RING; ++i) {
        SFM_HIP(h, hipEventCreate(&h->prof[k].start[i]));
        SFM_HIP(h, hipEventCreate(&h->prof[k].stop[i]));
      }
  }
  h->profiling = enabled ? 1 : 0;
  return SFM_OK;
}

extern "C" int sfm_profile_read(sfm_handle h, int slot, double* total_ms_host, int64_t* count_host) {
  if (!h || slot < 0 || slot >= SFM_PROF_COUNT || !total_ms_host || !count_host) return SFM_ERR_ARG;
  sfm_prof_fold(h, slot);
  *total_ms_host = h->prof[slot].total_ms;
  *count_host = h->prof[slot].count;
  h->prof[slot].total_ms = 0.0;
  h->prof[slot].count = 0;
  return SFM_OK;
}

extern "C" void sfm_destroy(sfm_handle h) {
  if (!h) return;
  (void)sfm_comm_destroy(h);
  if (h->prof[0].start[0])
    for (int k = 0; k < SFM_PROF_COUNT; ++k)
      for (int i = 0; i < SFM_PROF_RING; ++i) {
        (void)hipEventDestroy(h->prof[k].start[i]);
        (void)hipEventDestroy(h->prof[k].stop[i]);
      }
  if (h->cg_event) (void)hipEventDestroy(h->cg_event);
  if (h->pinned) (void)hipHostFree(h->pinned);
  if (h->scratch) (void)hipFree(h->scratch);
  delete h;
}

extern "C" int sfm_cgs_persist_enable(sfm_handle h, int enabled) {
  if (!h) return SFM_ERR_ARG;
  h->cgs_persist_off = enabled ? 0 : 1;
  if (enabled) h->cgs_xcd_off = 0;
  return SFM_OK;
}

extern "C" int sfm_cgs_persist_enabled(sfm_handle h) { return h && !h->cgs_persist_off ? 1 : 0; }

extern "C" const char* sfm_last_error(sfm_handle h) { return h ? h->err : "null handle"; }

extern "C" int sfm_set_stream(sfm_handle h, void* hip_stream) {
  if (!h) return SFM_ERR_ARG;
  h->stream = (hipStream_t)hip_stream;
  return SFM_OK;
}

extern "C" int sfm_synchronize(sfm_handle h) {
  if (!h) return SFM_ERR_ARG;
  SFM_HIP(h, hipStreamSynchronize(h->stream));
  return SFM_OK;
}

extern "C" int sfm_copy_to_host(sfm_handle h, void* dst_host, const void* src_device, int64_t bytes) {
  if (!h) return SFM_ERR_ARG;
  if (!dst_host || !src_device || bytes < 0) return sfm_fail(h, SFM_ERR_ARG, "sfm_copy_to_host", "bad argument");
  SFM_HIP(h, hipMemcpyAsync(dst_host, src_device, (size_t)bytes, hipMemcpyDeviceToHost, h->stream));
  SFM_HIP(h, hipStreamSynchronize(h->stream));
  return SFM_OK;
}
