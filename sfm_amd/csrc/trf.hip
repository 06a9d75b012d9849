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

#define ALPHA_FLOOR_REL 1e-13

struct sfm_trf_state_s {
  sfm_ctx* h;
  sfm_ba_problem p;
  double* x;         // the caller's buffer: always the current iterate
  double* x_new;     // trial point (owned)
  sfm_trf_options opt;
  sfm_reduce_fn reduce;
  void* reduce_user;
  sfm_ba_layout lay;
  double cost, g_norm, g_inf, hdiag, x_norm, Delta, alpha;
  int nfev, njev, status /* -1: running */, iteration, n_solves;
  long long cg_iters;
  std::vector<double> trace;
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
      const int mit = s->opt.pcg_max_iter > 0 ? s->opt.pcg_max_iter : (4 * n < 20000 ? 4 * n : 20000);
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

// scipy common.py:222-248
void update_tr_radius(double& Delta, double actual, double predicted, double step_norm, bool bound_hit, double* ratio) {
  double r;
  if (predicted > 0) r = actual / predicted;
  else if (predicted == 0 && actual == 0) r = 1;
  else r = 0;
  if (r < 0.25) Delta = 0.25 * step_norm;
  else if (r > 0.75 && bound_hit) Delta *= 2.0;
  *ratio = r;
}

// scipy common.py:705-717; 0 = keep going
int check_termination(double dF, double F, double dx_norm, double x_norm, double ratio, double ftol, double xtol) {
  const bool f_ok = dF < ftol * F && ratio > 0.25;
  const bool x_ok = dx_norm < xtol * (xtol + x_norm);
  if (f_ok && x_ok) return 4;
  if (f_ok) return 2;
  if (x_ok) return 3;
  return 0;
}

// scipy common.py:57-168 with (H + alpha I) solves in place of the SVD; leaves p(alpha_final)
// in the workspace.  J has a 7-dof gauge null space: `full_rank` is never taken, alpha_lower starts at 0.
int solve_tr_more(Backend& be, double g_norm, double Delta, double& alpha, double alpha_floor, double* p_norm_out) {
  double alpha_upper = g_norm / Delta, alpha_lower = 0.0;
  if (alpha == 0) alpha = std::fmax(0.001 * alpha_upper, std::sqrt(alpha_lower * alpha_upper));
  int rc;
  for (int it = 0; it < 10; ++it) {
    if (alpha < alpha_lower || alpha > alpha_upper) alpha = std::fmax(0.001 * alpha_upper, std::sqrt(alpha_lower * alpha_upper));
    const bool on_floor = alpha <= alpha_floor;
    if (on_floor) alpha = alpha_floor;
    double p_norm, pq;
    if ((rc = be.solve(alpha, 1, &p_norm, &pq))) return rc;
    const double phi = p_norm - Delta;
    if (on_floor && phi < 0) { *p_norm_out = p_norm; return SFM_OK; }     // interior Gauss-Newton step: p(alpha_floor) is the answer
    const double phi_prime = -pq / p_norm;
    if (phi < 0) alpha_upper = alpha;
    const double ratio = phi / phi_prime;
    alpha_lower = std::fmax(alpha_lower, alpha - ratio);
    alpha -= (phi + Delta) * ratio / Delta;
    if (std::fabs(phi) < 0.01 * Delta) break;
  }
  alpha = std::fmax(std::fmax(alpha, alpha_floor), 1e-300);      // the Schur route needs alpha > 0 (SciPy's SVD form does not)
  double pq;
  return be.solve(alpha, 0, p_norm_out, &pq);
}

}  // namespace

extern "C" int sfm_ba_trf_begin(sfm_handle h, sfm_ba_problem p, double* x, const sfm_trf_options* opt,
                                sfm_reduce_fn reduce, void* reduce_user, sfm_trf_state* out) {
  if (!h) return SFM_ERR_ARG;
  if (!p || !x || !opt || !out) return sfm_fail(h, SFM_ERR_ARG, "sfm_ba_trf_begin", "null argument");
  if (!p->workspace) return sfm_fail(h, SFM_ERR_WORKSPACE, "sfm_ba_trf_begin", "no workspace bound (sfm_ba_bind_workspace)");
  sfm_trf_state_s* s = new sfm_trf_state_s();
  s->h = h; s->p = p; s->x = x; s->x_new = nullptr; s->opt = *opt; s->reduce = reduce; s->reduce_user = reduce_user;
  s->cg_iters = 0;
  sfm_ba_get_layout(p, &s->lay);
  const size_t bytes = ((size_t)p->n_cams * p->cam_dim + 3 * (size_t)p->n_pts) * sizeof(double);
  if (hipMalloc((void**)&s->x_new, bytes) != hipSuccess) { delete s; return sfm_fail(h, SFM_ERR_HIP, "sfm_ba_trf_begin", "hipMalloc"); }
  Backend be{s};
  int rc = be.linearize(&s->cost, &s->g_norm, &s->g_inf, &s->hdiag);
  if (!rc) rc = be.x_norm(&s->x_norm);
  if (rc) { (void)hipFree(s->x_new); delete s; return rc; }
  s->nfev = 1; s->njev = 1;
  s->Delta = s->x_norm > 0 ? s->x_norm : 1.0;
  s->alpha = 0.0;
  s->status = -1;
  s->iteration = 0;
  s->n_solves = 0;
  *out = s;
  return SFM_OK;
}

extern "C" int sfm_ba_trf_outer(sfm_trf_state s, int* more) {
  if (!s || !more) return SFM_ERR_ARG;
  *more = 0;
  const sfm_trf_options& o = s->opt;
  if (o.max_outer >= 0 && s->iteration >= o.max_outer) return SFM_OK;
  if (s->g_inf < o.gtol && o.check_tolerances && s->status < 0) s->status = 1;
  if (s->status >= 0 || s->nfev == o.max_nfev) return SFM_OK;
  Backend be{s};
  double actual = -1.0, cost_new = s->cost, xnew_norm = s->x_norm;
  int rc;
  while (actual <= 0 && s->nfev < o.max_nfev) {
    double p_norm;
    if ((rc = solve_tr_more(be, s->g_norm, s->Delta, s->alpha, ALPHA_FLOOR_REL * s->hdiag, &p_norm))) return rc;
    double js2, gts, step_norm;
    if ((rc = be.step(s->Delta / p_norm, &js2, &gts, &cost_new, &step_norm, &xnew_norm))) return rc;
    const double predicted = -(0.5 * js2 + gts);
    s->nfev++;
    if (!std::isfinite(cost_new)) { s->Delta = 0.25 * step_norm; continue; }
    actual = s->cost - cost_new;
    double Delta_new = s->Delta, ratio;
    update_tr_radius(Delta_new, actual, predicted, step_norm, step_norm > 0.95 * s->Delta, &ratio);
    s->trace.push_back(s->alpha); s->trace.push_back(s->Delta); s->trace.push_back(step_norm); s->trace.push_back(actual > 0 ? 1.0 : 0.0);
    if (o.check_tolerances) {
      const int t = check_termination(actual, s->cost, step_norm, s->x_norm, ratio, o.ftol, o.xtol);
      if (t) { s->status = t; break; }
    }
    s->alpha *= s->Delta / Delta_new;
    s->Delta = Delta_new;
  }
  if (actual > 0) {
    if ((rc = be.accept())) return rc;
    s->x_norm = xnew_norm;
    s->cost = cost_new;
    double c_unused;
    if ((rc = be.linearize(&c_unused, &s->g_norm, &s->g_inf, &s->hdiag))) return rc;
    s->njev++;
  }
  s->iteration++;
  *more = 1;
  return SFM_OK;
}

extern "C" int sfm_ba_trf_result(sfm_trf_state s, sfm_trf_result* out) {
  if (!s || !out) return SFM_ERR_ARG;
  out->cost = s->cost; out->optimality = s->g_inf;
  out->nfev = s->nfev; out->njev = s->njev; out->status = s->status < 0 ? 0 : s->status;
  out->n_solves = s->n_solves; out->n_outer = s->iteration; out->cg_iters = (int32_t)(s->cg_iters > 0x7fffffff ? 0x7fffffff : s->cg_iters);
  return SFM_OK;
}

extern "C" int sfm_ba_trf_trace(sfm_trf_state s, double* out_host, int32_t capacity_trials) {
  if (!s) return 0;
  const int n = (int)(s->trace.size() / 4);
  if (out_host)
    for (int i = 0; i < n && i < capacity_trials; ++i)
      for (int q = 0; q < 4; ++q) out_host[4 * i + q] = s->trace[4 * (size_t)i + q];
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
