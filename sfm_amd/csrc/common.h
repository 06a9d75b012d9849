// Shared host-side context for libsfm_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include "sfm_amd.h"

#define SFM_PROF_RING 128
#define SFM_PINNED_DOUBLES 64
#define SFM_PIN_CG1 32
#define SFM_PIN_CGB 40       /* tile-streaming CG: its status words as the deciding launch writes them, word 7 = the system's ticket */
struct sfm_prof_slot {
  hipEvent_t start[SFM_PROF_RING], stop[SFM_PROF_RING];
  int pending;             // recorded, not yet folded into total_ms
  double total_ms;
  long long count;
};

struct sfm_ctx {
  int device;
  hipStream_t stream;
  char err[512];
  double* pinned;          // SFM_PINNED_DOUBLES of pinned host memory: [0, SFM_SC_COUNT) scalar read-back, then the status words
                           // of the first camera-CG system of a damped solve (SFM_PIN_CG1, 8 doubles; the second system's live in
                           // the problem's own pinned block, sfm_ba_prob::host_sc)
  hipEvent_t cg_event;     // recorded behind the status copy of the first system (sfm_ba_schur_solve waits for it, not for the stream)
  int profiling;
  sfm_prof_slot prof[SFM_PROF_COUNT];
  void* scratch;           // growable device scratch for calls that take no workspace (sfm_scratch)
  size_t scratch_bytes;
  void* comm;              // ncclComm_t of this handle (comm_rccl.hip), or null
  int comm_owned, comm_ranks, comm_rank;
  int cgs_persist_off;     // set once a persistent CG launch had to be abandoned: per-launch kernel from then on
  int cgs_xcd_off;         // set once a one-XCD launch of the persistent CG had to be abandoned: device-wide form from then on
  double cgb_seq;          // tickets of the tile-streaming CG's systems (cgs_solve_big spins on word SFM_PIN_CGB + 7)
};

// HIP-event bracket around one kernel (or one short kernel sequence) on the handle's stream.
// No-ops unless sfm_set_profiling(h, 1).  A full ring is folded (which synchronises) before reuse.
void sfm_prof_fold(sfm_ctx* h, int slot);
static inline void sfm_prof_begin(sfm_ctx* h, int slot) {
  if (!h->profiling) return;
  sfm_prof_slot& s = h->prof[slot];
  if (s.pending == SFM_PROF_RING) sfm_prof_fold(h, slot);
  (void)hipEventRecord(s.start[s.pending], h->stream);
}
static inline void sfm_prof_end(sfm_ctx* h, int slot) {
  if (!h->profiling) return;
  sfm_prof_slot& s = h->prof[slot];
  (void)hipEventRecord(s.stop[s.pending], h->stream);
  s.pending++;
}

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
