// Handle management for libsfm_amd.so.
#include "common.h"

extern "C" const char* sfm_version(void) { return "sfm_amd 0.1 (gfx950)"; }

extern "C" int sfm_create(int device, sfm_handle* out) {
  if (!out) return SFM_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SFM_ERR_HIP;
  if (hipSetDevice(device) != hipSuccess) return SFM_ERR_HIP;
  sfm_ctx* h = new sfm_ctx();
  h->device = device;
  h->stream = nullptr;
  h->err[0] = 0;
  h->pinned = nullptr;
  if (hipHostMalloc((void**)&h->pinned, SFM_SC_COUNT * sizeof(double), hipHostMallocDefault) != hipSuccess) {
    delete h;
    return SFM_ERR_HIP;
  }
  *out = h;
  return SFM_OK;
}

extern "C" void sfm_destroy(sfm_handle h) {
  if (!h) return;
  if (h->pinned) (void)hipHostFree(h->pinned);
  delete h;
}

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
