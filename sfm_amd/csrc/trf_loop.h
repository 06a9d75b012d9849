// The trust-region state machine of sfm_ba_trf_* as plain C++ over any backend type BE with
//   int linearize(double* cost, double* g_norm, double* g_inf, double* hdiag);
//   int solve(double alpha, int want_q, double* p_norm, double* pq);
//   int step(double scale, double* js2, double* gts, double* cost_new, double* step_norm, double* xnew_norm);
//   int x_norm(double* out);
//   int accept();
// (0 = ok, anything else is passed up).  trf.hip instantiates it with the device backend; tests/native/trf_loop_check.cpp
// with a small dense problem, under the address / undefined-behaviour sanitizers, against sfm_amd/trf.py.
// What it restates: scipy _lsq/trf.py:401-560 (trf_no_bounds), common.py:57-168 (More'), 222-248, 705-717 - see trf.hip.
#pragma once
#include <cmath>
#include <vector>

namespace trf_core {

constexpr double ALPHA_FLOOR_REL = 1e-13;

struct Options { double ftol, xtol, gtol; int max_nfev, max_outer, check_tolerances; };

struct State {
  double cost = 0, g_norm = 0, g_inf = 0, hdiag = 0, x_norm = 0, Delta = 0, alpha = 0;
  int nfev = 0, njev = 0, status = -1 /* -1: running */, iteration = 0;
  std::vector<double> trace;      // (alpha, Delta, step_norm, accepted) per trial
};

// scipy common.py:222-248
inline void update_tr_radius(double& Delta, double actual, double predicted, double step_norm, bool bound_hit, double* ratio) {
  double r;
  if (predicted > 0) r = actual / predicted;
  else if (predicted == 0 && actual == 0) r = 1;
  else r = 0;
  if (r < 0.25) Delta = 0.25 * step_norm;
  else if (r > 0.75 && bound_hit) Delta *= 2.0;
  *ratio = r;
}

// scipy common.py:705-717; 0 = keep going
inline int check_termination(double dF, double F, double dx_norm, double x_norm, double ratio, double ftol, double xtol) {
  const bool f_ok = dF < ftol * F && ratio > 0.25;
  const bool x_ok = dx_norm < xtol * (xtol + x_norm);
  if (f_ok && x_ok) return 4;
  if (f_ok) return 2;
  if (x_ok) return 3;
  return 0;
}

// scipy common.py:57-168 with (H + alpha I) solves in place of the SVD; leaves p(alpha_final)
// in the backend.  J has a 7-dof gauge null space: `full_rank` is never taken, alpha_lower starts at 0.
template <class BE>
int solve_tr_more(BE& be, double g_norm, double Delta, double& alpha, double alpha_floor, double* p_norm_out) {
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
    if (on_floor && phi < 0) { *p_norm_out = p_norm; return 0; }     // interior Gauss-Newton step: p(alpha_floor) is the answer
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

template <class BE>
int begin(BE& be, State& s) {
  int rc = be.linearize(&s.cost, &s.g_norm, &s.g_inf, &s.hdiag);
  if (!rc) rc = be.x_norm(&s.x_norm);
  if (rc) return rc;
  s.nfev = 1; s.njev = 1;
  s.Delta = s.x_norm > 0 ? s.x_norm : 1.0;
  s.alpha = 0.0;
  s.status = -1;
  s.iteration = 0;
  return 0;
}

// one outer iteration: the trials of one linearisation up to the accepted step, then the next linearisation
template <class BE>
int outer(BE& be, State& s, const Options& o, int* more) {
  *more = 0;
  if (o.max_outer >= 0 && s.iteration >= o.max_outer) return 0;
  if (s.g_inf < o.gtol && o.check_tolerances) s.status = 1;       // overrides an ftol / xtol status of the same iteration, as scipy trf.py:451-453 does
  if (s.status >= 0 || s.nfev == o.max_nfev) return 0;
  double actual = -1.0, cost_new = s.cost, xnew_norm = s.x_norm;
  int rc;
  while (actual <= 0 && s.nfev < o.max_nfev) {
    double p_norm;
    if ((rc = solve_tr_more(be, s.g_norm, s.Delta, s.alpha, ALPHA_FLOOR_REL * s.hdiag, &p_norm))) return rc;
    double js2, gts, step_norm;
    if ((rc = be.step(s.Delta / p_norm, &js2, &gts, &cost_new, &step_norm, &xnew_norm))) return rc;
    const double predicted = -(0.5 * js2 + gts);
    s.nfev++;
    if (!std::isfinite(cost_new)) { s.Delta = 0.25 * step_norm; continue; }
    actual = s.cost - cost_new;
    double Delta_new = s.Delta, ratio;
    update_tr_radius(Delta_new, actual, predicted, step_norm, step_norm > 0.95 * s.Delta, &ratio);
    s.trace.push_back(s.alpha); s.trace.push_back(s.Delta); s.trace.push_back(step_norm); s.trace.push_back(actual > 0 ? 1.0 : 0.0);
    if (o.check_tolerances) {
      const int t = check_termination(actual, s.cost, step_norm, s.x_norm, ratio, o.ftol, o.xtol);
      if (t) { s.status = t; break; }
    }
    s.alpha *= s.Delta / Delta_new;
    s.Delta = Delta_new;
  }
  if (actual > 0) {
    if ((rc = be.accept())) return rc;
    s.x_norm = xnew_norm;
    s.cost = cost_new;
    double c_unused;
    if ((rc = be.linearize(&c_unused, &s.g_norm, &s.g_inf, &s.hdiag))) return rc;
    s.njev++;
  }
  s.iteration++;
  *more = 1;
  return 0;
}

}  // namespace trf_core
