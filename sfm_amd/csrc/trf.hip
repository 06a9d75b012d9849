// Host state machine of the robust trust-region solve behind `bundle_adjust`: what
// scipy.optimize.least_squares(method='trf', loss='huber', ...) does at
// /root/reference/utils/sfm_reconstruction.py:506-514, as trf_no_bounds (scipy _lsq/trf.py:401-560) with the
// More' root finder for the Levenberg-Marquardt parameter (scipy _lsq/common.py:57-168), update_tr_radius and
// check_termination (common.py:222-248, 705-717).  Every data-parallel stage runs on the device (ba.hip); the
// host only sees a handful of scalars per stage.  sfm_amd/trf.py is the same state machine in the host language
// of the drop-in (it drives any backend, including the CPU stand-in of the multi-rank tests); the two are held
// together by tests/test_ba_gpu.py::test_c_loop_equals_python_loop.
//
// One deliberate difference from SciPy, shared with the oracle (oracle/ba_oracle.py): alpha is floored at
// ALPHA_FLOOR_REL * max diag(H) because a Cholesky route needs H + alpha I numerically positive definite.
#include <cmath>
#include <vector>
#include "ba_internal.h"
#include "trf_loop.h"      // the state machine itself, backend-agnostic (also built and checked on the CPU)

struct sfm_trf_state_s {
  sfm_ctx* h;
  sfm_ba_problem p;
  double* x;         // the caller's buffer: always the current iterate
  double* x_new;     // trial point (owned)
  sfm_trf_options opt;
  sfm_reduce_fn reduce;
  void* reduce_user;
  sfm_ba_layout lay;
  trf_core::State st;
  int n_solves;
  long long cg_iters;
};

namespace {

struct Backend {
  sfm_trf_state_s* s;
  double sc[SFM_SC_COUNT];

  int red(int64_t off_bytes, int64_t count, int op) {
    if (!s->reduce) return SFM_OK;
    int rc = s->reduce(s->reduce_user, (char*)s->p->workspace + off_bytes, count, op);
    return rc ? sfm_fail(s->h, SFM_ERR_HIP, "sfm_ba_trf", "the reduce hook failed") : SFM_OK;
  }
  int scalars() { return sfm_ba_read_scalars(s->h, s->p, sc); }

  int linearize(double* cost, double* g_norm, double* g_inf, double* hdiag) {
    int rc;
    if ((rc = sfm_ba_linearize(s->h, s->p, s->x))) return rc;
    if ((rc = red(s->lay.reduce_lin_off, s->lay.reduce_lin_count, 0))) return rc;
    if ((rc = red(s->lay.gmax_off, 2, 1))) return rc;
    if ((rc = sfm_ba_finish_linearize(s->h, s->p))) return rc;
    if ((rc = scalars())) return rc;
    *cost = sc[SFM_SC_COST]; *g_norm = std::sqrt(sc[SFM_SC_GNORM2]); *g_inf = sc[SFM_SC_GINF]; *hdiag = sc[SFM_SC_HDIAG];
    if (!std::isfinite(*cost)) return sfm_fail(s->h, SFM_ERR_NUMERIC, "sfm_ba_trf", "residuals are not finite");
    return SFM_OK;
  }
  int solve(double alpha, int want_q, double* p_norm, double* pq) {
    int rc;
    if (s->opt.solver == SFM_SOLVER_PCG) {
      const int n = s->p->n_cams * s->p->cam_dim;
      const double rtol = s->opt.pcg_rtol > 0 ? s->opt.pcg_rtol : 1e-13;
      const int mit = s->opt.pcg_max_iter > 0 ? s->opt.pcg_max_iter : 400;      // then the formed-S fallback (sfm_ba_solve_pcg)
      int32_t its = 0;
      if ((rc = sfm_ba_solve_pcg(s->h, s->p, alpha, want_q, rtol, mit, s->reduce, s->reduce_user, &its))) return rc;
      s->cg_iters += its;
      return after_solve(p_norm, pq);
    }
    if ((rc = sfm_ba_schur_build(s->h, s->p, alpha))) return rc;
    if (s->reduce) {
      if ((rc = sfm_ba_pack_system(s->h, s->p))) return rc;
      if ((rc = red(s->lay.reduce_Sp_off, s->lay.reduce_Sp_count, 0))) return rc;
      if ((rc = sfm_ba_unpack_system(s->h, s->p))) return rc;
    }
    if ((rc = sfm_ba_schur_solve(s->h, s->p, alpha, want_q))) return rc;
    if ((rc = red(s->lay.reduce_q_off, s->lay.reduce_q_count, 0))) return rc;
    if ((rc = sfm_ba_finish_solve(s->h, s->p, want_q))) return rc;
    return after_solve(p_norm, pq);
  }
  int after_solve(double* p_norm, double* pq) {
    int rc;
    if ((rc = scalars())) return rc;
    s->n_solves++;
    if (sc[SFM_SC_CHOL_FAIL] != 0.0)
      return sfm_fail(s->h, SFM_ERR_NUMERIC, "sfm_ba_trf",
                      sc[SFM_SC_CHOL_FAIL] == 1.0 ? "reduced camera system not positive definite"
                      : sc[SFM_SC_CHOL_FAIL] == 2.0 ? "triangular solve stalled" : "damped step is not finite");
    *p_norm = std::sqrt(sc[SFM_SC_PNORM2]); *pq = sc[SFM_SC_PQ];
    return SFM_OK;
  }
  int step(double scale, double* js2, double* gts, double* cost_new, double* step_norm, double* xnew_norm) {
    int rc;
    if ((rc = sfm_ba_step(s->h, s->p, s->x, scale, s->x_new))) return rc;
    if ((rc = red(s->lay.reduce_step_off, s->lay.reduce_step_count, 0))) return rc;
    if ((rc = sfm_ba_finish_step(s->h, s->p, s->x, scale, s->x_new))) return rc;
    if ((rc = scalars())) return rc;
    *js2 = sc[SFM_SC_JS2]; *gts = sc[SFM_SC_GTS]; *cost_new = sc[SFM_SC_COST_NEW];
    *step_norm = std::sqrt(sc[SFM_SC_SNORM2]); *xnew_norm = std::sqrt(sc[SFM_SC_XNEW_NORM2]);
    return SFM_OK;
  }
  int x_norm(double* out) {
    int rc;
    if ((rc = ba_xnorm_partial(s->h, s->p, s->x))) return rc;
    if ((rc = red(s->lay.reduce_step_off + 4 * 8, 1, 0))) return rc;
    if ((rc = ba_xnorm_finish(s->h, s->p, s->x))) return rc;
    if ((rc = scalars())) return rc;
    *out = std::sqrt(sc[SFM_SC_XNEW_NORM2]);
    return SFM_OK;
  }
  int accept() {     // x <- x_new
    const size_t bytes = ((size_t)s->p->n_cams * s->p->cam_dim + 3 * (size_t)s->p->n_pts) * sizeof(double);
    if (hipMemcpyAsync(s->x, s->x_new, bytes, hipMemcpyDeviceToDevice, s->h->stream) != hipSuccess)
      return sfm_fail(s->h, SFM_ERR_HIP, "sfm_ba_trf", "copy of the accepted iterate failed");
    return SFM_OK;
  }
};

}  // namespace

extern "C" int sfm_ba_trf_begin(sfm_handle h, sfm_ba_problem p, double* x, const sfm_trf_options* opt,
                                sfm_reduce_fn reduce, void* reduce_user, sfm_trf_state* out) {
  if (!h) return SFM_ERR_ARG;
  if (!p || !x || !opt || !out) return sfm_fail(h, SFM_ERR_ARG, "sfm_ba_trf_begin", "null argument");
  if (!p->workspace) return sfm_fail(h, SFM_ERR_WORKSPACE, "sfm_ba_trf_begin", "no workspace bound (sfm_ba_bind_workspace)");
  sfm_trf_state_s* s = new sfm_trf_state_s();
  s->h = h; s->p = p; s->x = x; s->x_new = nullptr; s->opt = *opt; s->reduce = reduce; s->reduce_user = reduce_user;
  s->cg_iters = 0;
  if (reduce) p->sharded = 1;          // one rank of several: no rank-local route switches in the replicated camera solve
  sfm_ba_get_layout(p, &s->lay);
  const size_t bytes = ((size_t)p->n_cams * p->cam_dim + 3 * (size_t)p->n_pts) * sizeof(double);
  if (hipMalloc((void**)&s->x_new, bytes) != hipSuccess) { delete s; return sfm_fail(h, SFM_ERR_HIP, "sfm_ba_trf_begin", "hipMalloc"); }
  Backend be{s};
  s->n_solves = 0;
  const int rc = trf_core::begin(be, s->st);
  if (rc) { (void)hipFree(s->x_new); delete s; return rc; }
  *out = s;
  return SFM_OK;
}

extern "C" int sfm_ba_trf_outer(sfm_trf_state s, int* more) {
  if (!s || !more) return SFM_ERR_ARG;
  const sfm_trf_options& o = s->opt;
  const trf_core::Options co = {o.ftol, o.xtol, o.gtol, o.max_nfev, o.max_outer, o.check_tolerances};
  Backend be{s};
  return trf_core::outer(be, s->st, co, more);
}

extern "C" int sfm_ba_trf_result(sfm_trf_state s, sfm_trf_result* out) {
  if (!s || !out) return SFM_ERR_ARG;
  out->cost = s->st.cost; out->optimality = s->st.g_inf;
  out->nfev = s->st.nfev; out->njev = s->st.njev; out->status = s->st.status < 0 ? 0 : s->st.status;
  out->n_solves = s->n_solves; out->n_outer = s->st.iteration; out->cg_iters = (int32_t)(s->cg_iters > 0x7fffffff ? 0x7fffffff : s->cg_iters);
  return SFM_OK;
}

extern "C" int sfm_ba_trf_trace(sfm_trf_state s, double* out_host, int32_t capacity_trials) {
  if (!s) return 0;
  const std::vector<double>& trace = s->st.trace;
  const int n = (int)(trace.size() / 4);
  if (out_host)
    for (int i = 0; i < n && i < capacity_trials; ++i)
      for (int q = 0; q < 4; ++q) out_host[4 * i + q] = trace[4 * (size_t)i + q];
  return n;
}

extern "C" void sfm_ba_trf_end(sfm_trf_state s) {
  if (!s) return;
  (void)hipStreamSynchronize(s->h->stream);
  if (s->x_new) (void)hipFree(s->x_new);
  delete s;
}

extern "C" int sfm_ba_run_trf(sfm_handle h, sfm_ba_problem p, double* x, const sfm_trf_options* opt,
                              sfm_reduce_fn reduce, void* reduce_user, sfm_trf_result* out) {
  if (!h) return SFM_ERR_ARG;
  if (!out) return sfm_fail(h, SFM_ERR_ARG, "sfm_ba_run_trf", "null result");
  sfm_trf_state st = nullptr;
  int rc = sfm_ba_trf_begin(h, p, x, opt, reduce, reduce_user, &st);
  if (rc) return rc;
  int more = 1;
  while (more && !(rc = sfm_ba_trf_outer(st, &more))) {}
  sfm_ba_trf_result(st, out);
  sfm_ba_trf_end(st);
  return rc;
}
