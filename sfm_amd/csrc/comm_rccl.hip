// Collectives inside the library: RCCL (ncclAllReduce over xGMI) enqueued on the handle's HIP stream between the BA stages,
// so a multi-rank solve has no host round trip per exchange and a C host needs no Python / torch.distributed for it
// (SURVEY.md section 5 comm row, section 8e: one exchange step per damped solve - all-reduce of the packed [S | r] - plus a
// few short vectors).  RCCL is dlopen'ed on first use (librccl.so.1): the library itself loads on a box without it, and in
// a process that already holds an RCCL (PyTorch bundles one under the same soname) that instance is the one used.
// One communicator per handle = per (process, device); every rank calls the same sequence of stages, so the collectives
// line up by construction.  Reductions are in place, float64, SUM or MAX.
#include <dlfcn.h>
#include <mutex>
#include "ba_internal.h"

namespace {

typedef int nccl_result_t;                         // ncclSuccess == 0
typedef struct ncclComm* nccl_comm_t;
struct nccl_unique_id { char internal[128]; };     // NCCL_UNIQUE_ID_BYTES (rccl.h:40-43)
enum { NCCL_FLOAT64 = 8, NCCL_SUM = 0, NCCL_MAX = 2 };   // ncclDataType_t / ncclRedOp_t values (rccl.h)

struct Rccl {
  void* so = nullptr;
  nccl_result_t (*GetUniqueId)(nccl_unique_id*) = nullptr;
  nccl_result_t (*CommInitRank)(nccl_comm_t*, int, nccl_unique_id, int) = nullptr;
  nccl_result_t (*CommDestroy)(nccl_comm_t) = nullptr;
  nccl_result_t (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(nccl_result_t) = nullptr;
  nccl_result_t (*CommCount)(const nccl_comm_t, int*) = nullptr;       // optional: what RCCL itself says the communicator is
  nccl_result_t (*CommUserRank)(const nccl_comm_t, int*) = nullptr;
  char why[256] = {0};
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_load_once();
// Two threads that create communicators on different handles at the same time both come through here: the dlopen and the
// symbol table are set up exactly once, everybody else waits for that and reads the verdict.
bool rccl_load() {
  std::call_once(g_rccl_once, rccl_load_once);
  return g_rccl.ok;
}

void rccl_load_once() {
  Rccl& r = g_rccl;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* nm : names)
    if ((r.so = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!r.so) { snprintf(r.why, sizeof(r.why), "librccl.so.1 not found (%s)", dlerror()); return; }
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.so, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.so, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
  r.AllReduce = (decltype(r.AllReduce))dlsym(r.so, "ncclAllReduce");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
  r.CommCount = (decltype(r.CommCount))dlsym(r.so, "ncclCommCount");
  r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.so, "ncclCommUserRank");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) {
    snprintf(r.why, sizeof(r.why), "librccl.so.1 lacks an expected symbol");
    dlclose(r.so); r.so = nullptr;
    return;
  }
  r.ok = true;
}

int nccl_fail(sfm_ctx* h, const char* what, nccl_result_t e) {
  return sfm_fail(h, SFM_ERR_HIP, what, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "RCCL error");
}

}  // namespace

extern "C" int sfm_comm_unique_id(sfm_handle h, void* id_host) {
  if (!h) return SFM_ERR_ARG;
  if (!id_host) return sfm_fail(h, SFM_ERR_ARG, "sfm_comm_unique_id", "null id");
  if (!rccl_load()) return sfm_fail(h, SFM_ERR_HIP, "sfm_comm_unique_id", g_rccl.why);
  nccl_unique_id id;
  nccl_result_t e = g_rccl.GetUniqueId(&id);
  if (e) return nccl_fail(h, "ncclGetUniqueId", e);
  memcpy(id_host, id.internal, sizeof(id.internal));
  return SFM_OK;
}

extern "C" int sfm_comm_init_rank(sfm_handle h, const void* id_host, int32_t n_ranks, int32_t rank) {
  if (!h) return SFM_ERR_ARG;
  if (!id_host || n_ranks < 1 || rank < 0 || rank >= n_ranks) return sfm_fail(h, SFM_ERR_ARG, "sfm_comm_init_rank", "bad id / rank");
  if (h->comm) return sfm_fail(h, SFM_ERR_ARG, "sfm_comm_init_rank", "the handle already has a communicator (sfm_comm_destroy first)");
  if (!rccl_load()) return sfm_fail(h, SFM_ERR_HIP, "sfm_comm_init_rank", g_rccl.why);
  SFM_HIP(h, hipSetDevice(h->device));
  nccl_unique_id id;
  memcpy(id.internal, id_host, sizeof(id.internal));
  nccl_comm_t c = nullptr;
  nccl_result_t e = g_rccl.CommInitRank(&c, n_ranks, id, rank);
  if (e) return nccl_fail(h, "ncclCommInitRank", e);
  h->comm = c; h->comm_owned = 1; h->comm_ranks = n_ranks; h->comm_rank = rank;
  return SFM_OK;
}

extern "C" int sfm_comm_adopt(sfm_handle h, void* nccl_comm, int32_t n_ranks, int32_t rank) {
  if (!h) return SFM_ERR_ARG;
  if (!nccl_comm || n_ranks < 1 || rank < 0 || rank >= n_ranks) return sfm_fail(h, SFM_ERR_ARG, "sfm_comm_adopt", "bad communicator / rank");
  if (h->comm) return sfm_fail(h, SFM_ERR_ARG, "sfm_comm_adopt", "the handle already has a communicator (sfm_comm_destroy first)");
  if (!rccl_load()) return sfm_fail(h, SFM_ERR_HIP, "sfm_comm_adopt", g_rccl.why);
  h->comm = nccl_comm; h->comm_owned = 0; h->comm_ranks = n_ranks; h->comm_rank = rank;
  return SFM_OK;
}

extern "C" int sfm_comm_destroy(sfm_handle h) {
  if (!h) return SFM_ERR_ARG;
  if (h->comm && h->comm_owned && g_rccl.CommDestroy) {
    (void)hipStreamSynchronize(h->stream);
    g_rccl.CommDestroy((nccl_comm_t)h->comm);
  }
  h->comm = nullptr; h->comm_owned = 0; h->comm_ranks = 0; h->comm_rank = 0;
  return SFM_OK;
}

extern "C" int sfm_comm_info(sfm_handle h, int32_t* n_ranks, int32_t* rank) {
  if (!h || !n_ranks || !rank) return SFM_ERR_ARG;
  *n_ranks = h->comm ? h->comm_ranks : 0; *rank = h->comm ? h->comm_rank : 0;
  // RCCL's own word where it offers one: the ranks the communicator actually spans, not what the caller said it would
  if (h->comm && g_rccl.CommCount && g_rccl.CommUserRank) {
    int c = 0, u = 0;
    if (!g_rccl.CommCount((nccl_comm_t)h->comm, &c) && !g_rccl.CommUserRank((nccl_comm_t)h->comm, &u)) { *n_ranks = c; *rank = u; }
  }
  return SFM_OK;
}

extern "C" int sfm_comm_allreduce(sfm_handle h, double* data, int64_t count, int op) {
  if (!h) return SFM_ERR_ARG;
  if (!h->comm) return sfm_fail(h, SFM_ERR_ARG, "sfm_comm_allreduce", "no communicator (sfm_comm_init_rank / sfm_comm_adopt)");
  if (!data || count < 0 || (op != 0 && op != 1)) return sfm_fail(h, SFM_ERR_ARG, "sfm_comm_allreduce", "bad buffer / count / op");
  if (count == 0) return SFM_OK;
  nccl_result_t e = g_rccl.AllReduce(data, data, (size_t)count, NCCL_FLOAT64, op == 1 ? NCCL_MAX : NCCL_SUM, (nccl_comm_t)h->comm, h->stream);
  if (e) return nccl_fail(h, "ncclAllReduce", e);
  return SFM_OK;
}

// An sfm_reduce_fn: pass it as `reduce` with the handle as `reduce_user` (sfm_ba_trf_begin, sfm_ba_run_trf, sfm_ba_solve_pcg).
extern "C" int sfm_comm_reduce_hook(void* handle_as_user, void* data, int64_t count, int op) {
  return sfm_comm_allreduce((sfm_handle)handle_as_user, (double*)data, count, op);
}
