"""Host state machine of the robust trust-region solve behind `bundle_adjust`.

The reference calls scipy.optimize.least_squares(method='trf', loss='huber', max_nfev=100,
ftol=1e-4, xtol=1e-4) (/root/reference/utils/sfm_reconstruction.py:506-514).  This module is
that solver's control flow - trf_no_bounds (scipy _lsq/trf.py:401-560), the More' root finder
for the Levenberg-Marquardt parameter (scipy _lsq/common.py:57-168), update_tr_radius and
check_termination (common.py:222-248, 705-717) - driving a *backend* that owns the parameters
and performs the three data-parallel stages on the GPU:

    backend.linearize()          -> cost, ||g||_2, ||g||_inf, max diag(H)   (Jacobian + Huber scaling)
    backend.solve(alpha, want_q) -> ||p||, p^T (H+alpha I)^-1 p       (damped Schur solve)
    backend.step(scale)          -> ||J~ s||^2, g^T s, cost(x+s), ||s||, ||x+s||   (s = scale*p)
    backend.accept()             -> x <- x + s

SciPy gets ||p(alpha)|| and its derivative from one SVD of the Jacobian; here they come from
solves of (H + alpha I), which is the same function of alpha.  J has a
7-dof gauge null space, so SciPy's `full_rank` shortcut is never taken and alpha_lower starts at 0.

One deliberate difference: when the Gauss-Newton step lies inside the trust region SciPy's iteration
drives alpha towards 0 (x 0.001 per pass) and its SVD step is then dominated by round-off in the gauge
directions (SURVEY.md section 0 fact 8 - SciPy does not reproduce itself there).  A Cholesky route needs
H + alpha I numerically positive definite, so alpha is floored at ALPHA_FLOOR_REL * max diag(H) and the
iteration stops once it sits on the floor with ||p|| < Delta.  The floor is never active in the damped
regime the reference's own runs exercise (all goldens), where the alpha sequence equals SciPy's.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

ALPHA_FLOOR_REL = 1e-13


@dataclass
class TRFResult:
    cost: float
    nfev: int
    njev: int
    status: int
    optimality: float
    n_solves: int = 0
    trace: list = field(default_factory=list)   # (alpha, Delta, step_norm, accepted) per trial

    @property
    def success(self):
        return self.status > 0


def update_tr_radius(Delta, actual_reduction, predicted_reduction, step_norm, bound_hit):
    if predicted_reduction > 0:
        ratio = actual_reduction / predicted_reduction
    elif predicted_reduction == actual_reduction == 0:
        ratio = 1
    else:
        ratio = 0
    if ratio < 0.25:
        Delta = 0.25 * step_norm
    elif ratio > 0.75 and bound_hit:
        Delta *= 2.0
    return Delta, ratio


def check_termination(dF, F, dx_norm, x_norm, ratio, ftol, xtol):
    ftol_satisfied = dF < ftol * F and ratio > 0.25
    xtol_satisfied = dx_norm < xtol * (xtol + x_norm)
    if ftol_satisfied and xtol_satisfied:
        return 4
    elif ftol_satisfied:
        return 2
    elif xtol_satisfied:
        return 3
    return None


def solve_tr_more(backend, g_norm, Delta, initial_alpha, alpha_floor=0.0, rtol=0.01, max_iter=10):
    """More' iteration on alpha; leaves p(alpha_final) in the backend.
    Returns (||p||, alpha, n_iter, n_solves)."""
    alpha_upper = g_norm / Delta
    alpha_lower = 0.0
    if initial_alpha is None or initial_alpha == 0:
        alpha = max(0.001 * alpha_upper, (alpha_lower * alpha_upper) ** 0.5)
    else:
        alpha = initial_alpha
    it = -1
    solves = 0
    for it in range(max_iter):
        if alpha < alpha_lower or alpha > alpha_upper:
            alpha = max(0.001 * alpha_upper, (alpha_lower * alpha_upper) ** 0.5)
        on_floor = alpha <= alpha_floor
        if on_floor:
            alpha = alpha_floor
        p_norm, pq = backend.solve(alpha, True)
        solves += 1
        phi = p_norm - Delta
        if on_floor and phi < 0:          # interior Gauss-Newton step: p(alpha_floor) is the answer
            return p_norm, alpha, it + 1, solves
        phi_prime = -pq / p_norm
        if phi < 0:
            alpha_upper = alpha
        ratio = phi / phi_prime
        alpha_lower = max(alpha_lower, alpha - ratio)
        alpha -= (phi + Delta) * ratio / Delta
        if abs(phi) < rtol * Delta:
            break
    alpha = max(alpha, alpha_floor, 1e-300)      # the Schur route needs alpha > 0 (SciPy's SVD form does not)
    p_norm, _ = backend.solve(alpha, False)
    return p_norm, alpha, it + 1, solves + 1


class TRFState:
    """trf_no_bounds as a steppable object: `outer()` runs one outer iteration (the trials of one
    linearisation up to the accepted step, then the next linearisation) and returns False once the
    loop would have ended.  bench.py times K calls of `outer()`."""

    def __init__(self, backend, ftol=1e-4, xtol=1e-4, gtol=1e-8, max_nfev=100, check_tolerances=True):
        self.be = backend
        self.ftol, self.xtol, self.gtol, self.max_nfev = ftol, xtol, gtol, max_nfev
        self.check = check_tolerances
        self.cost, self.g_norm, self.g_inf, self.hdiag = backend.linearize()
        self.nfev, self.njev = 1, 1
        self.x_norm = backend.x_norm()
        self.Delta = self.x_norm if self.x_norm > 0 else 1.0
        self.alpha = 0.0
        self.status = None
        self.iteration = 0
        self.n_solves = 0
        self.trace = []

    def outer(self):
        be = self.be
        if self.g_inf < self.gtol and self.check:
            self.status = 1
        if self.status is not None or self.nfev == self.max_nfev:
            return False
        actual_reduction = -1.0
        cost_new = self.cost
        xnew_norm = self.x_norm
        while actual_reduction <= 0 and self.nfev < self.max_nfev:
            p_norm, self.alpha, _, ns = solve_tr_more(be, self.g_norm, self.Delta, self.alpha,
                                                      ALPHA_FLOOR_REL * self.hdiag)
            self.n_solves += ns
            js2, gts, cost_new, step_norm, xnew_norm = be.step(self.Delta / p_norm)
            predicted_reduction = -(0.5 * js2 + gts)
            self.nfev += 1
            if not math.isfinite(cost_new):
                self.Delta = 0.25 * step_norm
                continue
            actual_reduction = self.cost - cost_new
            Delta_new, ratio = update_tr_radius(self.Delta, actual_reduction, predicted_reduction,
                                                step_norm, step_norm > 0.95 * self.Delta)
            self.trace.append((self.alpha, self.Delta, step_norm, actual_reduction > 0))
            if self.check:
                self.status = check_termination(actual_reduction, self.cost, step_norm, self.x_norm, ratio,
                                                self.ftol, self.xtol)
                if self.status is not None:
                    break
            self.alpha *= self.Delta / Delta_new
            self.Delta = Delta_new
        if actual_reduction > 0:
            be.accept()
            self.x_norm = xnew_norm
            self.cost = cost_new
            _, self.g_norm, self.g_inf, self.hdiag = be.linearize()
            self.njev += 1
        self.iteration += 1
        return True

    def result(self):
        return TRFResult(self.cost, self.nfev, self.njev, 0 if self.status is None else self.status,
                         self.g_inf, self.n_solves, self.trace)


def trf(backend, ftol=1e-4, xtol=1e-4, gtol=1e-8, max_nfev=100, max_outer=None,
        check_tolerances=True):
    """Trust-region-reflective loop without bounds.  `max_outer` / `check_tolerances=False`
    give the fixed-schedule runs used for throughput measurements."""
    st = TRFState(backend, ftol, xtol, gtol, max_nfev, check_tolerances)
    while (max_outer is None or st.iteration < max_outer) and st.outer():
        pass
    return st.result()
