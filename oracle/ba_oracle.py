"""CPU oracle for the bundle-adjustment hot path (TEST INFRASTRUCTURE - not product code).

NumPy fp64 restatement of what the reference's `StructureFromMotion.bundle_adjust`
(/root/reference/utils/sfm_reconstruction.py:401-549) computes, i.e. the closure
`objective` (:472-501) / `project_points` (:453-470) minimised by
`scipy.optimize.least_squares(method='trf', loss='huber', max_nfev=100, ftol=1e-4,
xtol=1e-4)` (:506-514).  The arithmetic of the solver lives in a third-party
dependency, SciPy (pinned `scipy>=1.7.0`, /root/reference/requirements.txt:6; this
image has 1.15.3); its published algorithm is restated here function by function:

  huber / scale_for_robust_loss_function  scipy/optimize/_lsq/least_squares.py:169-178,
                                          scipy/optimize/_lsq/common.py:720-731
  trf_no_bounds                           scipy/optimize/_lsq/trf.py:401-560
  solve_lsq_trust_region (More')          scipy/optimize/_lsq/common.py:57-168
  update_tr_radius / check_termination    scipy/optimize/_lsq/common.py:222-248,705-717

Two deliberate differences from the literal reference, both validated against the
reference itself by tests/golden/make_golden.py (parity pinned, see tests/golden/):
  * the Jacobian is analytic (the reference lets SciPy take 2-point finite
    differences of the closure);
  * the trust-region sub-problem is solved from the normal equations
    H = J~^T J~, g (Cholesky of H + alpha*I) instead of SciPy's dense SVD of J~.
    `full_rank` of scipy:common.py:120-127 is taken as False (7-dof gauge freedom).
    Where SciPy's iteration would drive alpha to ~0 (Gauss-Newton step inside the trust
    region; SciPy's own result is round-off dominated there)
    alpha is floored at ALPHA_FLOOR_REL * max diag(H); never active on the goldens.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

from dataclasses import dataclass, field
import numpy as np

EPS = np.finfo(np.float64).eps
ALPHA_FLOOR_REL = 1e-13


# --------------------------------------------------------------------------- problem
@dataclass
class BAProblem:
    """Arrays produced by the reference's packing step (sfm_reconstruction.py:409-451)."""
    n_cams: int
    n_pts: int
    d: int                      # 10 = [rvec,t,fx,fy,cx,cy] (reference), 6 = [rvec,t]
    cam_idx: np.ndarray         # [N] int64, point-major observation order (:430-435)
    pt_idx: np.ndarray          # [N] int64
    uv: np.ndarray              # [N,2] f64, the pixel the k-th observation is compared with
    K0: np.ndarray              # (fx0, fy0, cx0, cy0) of the pre-BA self.K
    width: float = 1024.0
    height: float = 768.0
    reg_weight: float = 0.1

    @property
    def n_obs(self):
        return int(self.cam_idx.shape[0])

    @property
    def n_var(self):
        return self.d * self.n_cams + 3 * self.n_pts

    @property
    def n_res(self):
        return 2 * self.n_obs + (4 * self.n_cams if self.d == 10 else 0)


def effective_uv(uv, cam_idx, order):
    """Pixel each observation is actually compared with.

    order="aligned": uv[k] belongs to observation k.
    order="reference": the reference stacks projections camera-by-camera
    (sfm_reconstruction.py:480-485) but subtracts `points2D` left in point-major order
    (:486), so the q-th observation in stable camera-sorted order is paired with uv[q].
    """
    uv = np.asarray(uv, dtype=np.float64).reshape(-1, 2)
    if order == "aligned":
        return uv.copy()
    if order != "reference":
        raise ValueError(f"unknown order {order!r}")
    perm = np.argsort(np.asarray(cam_idx), kind="stable")
    out = np.empty_like(uv)
    out[perm] = uv
    return out


# --------------------------------------------------------------------------- rotation
def _rod_coeffs(theta2):
    """a=sin(t)/t, b=(1-cos t)/t^2, a1=a'(t)/t, b1=b'(t)/t  (series for t^2 < 1e-4)."""
    theta2 = np.asarray(theta2, dtype=np.float64)
    small = theta2 < 1e-4
    t2 = np.where(small, 1.0, theta2)
    t = np.sqrt(t2)
    s, c = np.sin(t), np.cos(t)
    a = s / t
    b = (1.0 - c) / t2
    a1 = (t * c - s) / (t2 * t)
    b1 = (t * s - 2.0 * (1.0 - c)) / (t2 * t2)
    z = theta2
    a_s = 1.0 - z / 6.0 + z * z / 120.0
    b_s = 0.5 - z / 24.0 + z * z / 720.0
    a1_s = -1.0 / 3.0 + z / 30.0 - z * z / 840.0
    b1_s = -1.0 / 12.0 + z / 180.0 - z * z / 6720.0
    return (np.where(small, a_s, a), np.where(small, b_s, b),
            np.where(small, a1_s, a1), np.where(small, b1_s, b1))


def _skew(v):
    v = np.asarray(v, dtype=np.float64)
    out = np.zeros(v.shape[:-1] + (3, 3))
    out[..., 0, 1] = -v[..., 2]; out[..., 0, 2] = v[..., 1]
    out[..., 1, 0] = v[..., 2];  out[..., 1, 2] = -v[..., 0]
    out[..., 2, 0] = -v[..., 1]; out[..., 2, 1] = v[..., 0]
    return out


def rotation_and_derivs(rvec):
    """R = I + a[r]x + b[r]x^2 (what cv2.Rodrigues(rvec) returns, :465) and dR/dr_i.

    rvec [C,3] -> R [C,3,3], dR [C,3,3,3] with dR[c,i] = dR/dr_i.
    """
    r = np.asarray(rvec, dtype=np.float64).reshape(-1, 3)
    th2 = np.sum(r * r, axis=1)
    a, b, a1, b1 = _rod_coeffs(th2)
    S = _skew(r)
    S2 = S @ S
    R = np.eye(3)[None] + a[:, None, None] * S + b[:, None, None] * S2
    dR = np.empty((r.shape[0], 3, 3, 3))
    for i in range(3):
        e = np.zeros(3); e[i] = 1.0
        Ei = _skew(e)[None]
        dR[:, i] = (a[:, None, None] * Ei
                    + b[:, None, None] * (Ei @ S + S @ Ei)
                    + (a1 * r[:, i])[:, None, None] * S
                    + (b1 * r[:, i])[:, None, None] * S2)
    return R, dR


def rotation_to_rvec(R):
    """Inverse of Rodrigues for a proper rotation (cv2.Rodrigues(R) at :419)."""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = 0.5 * np.linalg.norm(v)
    c = 0.5 * (np.trace(R) - 1.0)
    theta = np.arctan2(s, c)
    if s > 1e-8:
        return v * (theta / (2.0 * s))
    if c > 0:
        return 0.5 * v
    # theta ~ pi: R ~ 2 k k^T - I
    M = 0.5 * (R + np.eye(3))
    k = np.sqrt(np.clip(np.diag(M), 0.0, None))
    i = int(np.argmax(k))
    sgn = np.sign(M[i]); sgn[i] = 1.0
    k = k * sgn
    k /= np.linalg.norm(k)
    return k * theta


# --------------------------------------------------------------------------- model
def split_x(x, prob):
    nc = prob.n_cams * prob.d
    return x[:nc].reshape(prob.n_cams, prob.d), x[nc:].reshape(prob.n_pts, 3)


def _intrinsics(cams, prob):
    if prob.d == 10:
        return cams[:, 6], cams[:, 7], cams[:, 8], cams[:, 9]
    C = cams.shape[0]
    fx0, fy0, cx0, cy0 = prob.K0
    return (np.full(C, fx0), np.full(C, fy0), np.full(C, cx0), np.full(C, cy0))


def residuals(x, prob):
    """objective(params) of sfm_reconstruction.py:472-501.

    Layout: [u_0 - uv_0x, v_0 - uv_0y, ...] over observations in point-major order, then
    (d=10 only) 4 regulariser rows per camera (:489-499).  The reference lists the
    reprojection rows camera-grouped; that is a permutation of the same scalars.
    """
    cams, pts = split_x(np.asarray(x, dtype=np.float64), prob)
    R, _ = rotation_and_derivs(cams[:, :3])
    fx, fy, cx, cy = _intrinsics(cams, prob)
    ci, pi = prob.cam_idx, prob.pt_idx
    Y = np.einsum("nij,nj->ni", R[ci], pts[pi]) + cams[ci, 3:6]
    u = fx[ci] * Y[:, 0] / Y[:, 2] + cx[ci]
    v = fy[ci] * Y[:, 1] / Y[:, 2] + cy[ci]
    f = np.empty(prob.n_res)
    f[0:2 * prob.n_obs:2] = u - prob.uv[:, 0]
    f[1:2 * prob.n_obs:2] = v - prob.uv[:, 1]
    if prob.d == 10:
        f[2 * prob.n_obs:] = regulariser(cams, prob).ravel()
    return f


def regulariser(cams, prob):
    fx0, _, cx0, cy0 = prob.K0
    fx, fy, cx, cy = cams[:, 6], cams[:, 7], cams[:, 8], cams[:, 9]
    w = prob.reg_weight
    return np.stack([(fx - fx0) / fx0 * w, (fy - fx) / fx * w,
                     (cx - cx0) / prob.width * w, (cy - cy0) / prob.height * w], axis=1)


def jacobian_blocks(x, prob):
    """Analytic d(residual)/d(camera block), d(residual)/d(point) per observation.

    Returns f [m], Jc [N,2,d], Jp [N,2,3], Jreg [C,4,4] (d=10; w.r.t. fx,fy,cx,cy) or None.
    """
    cams, pts = split_x(np.asarray(x, dtype=np.float64), prob)
    R, dR = rotation_and_derivs(cams[:, :3])
    fx, fy, cx, cy = _intrinsics(cams, prob)
    ci, pi = prob.cam_idx, prob.pt_idx
    X = pts[pi]
    Y = np.einsum("nij,nj->ni", R[ci], X) + cams[ci, 3:6]
    iz = 1.0 / Y[:, 2]
    xn, yn = Y[:, 0] * iz, Y[:, 1] * iz
    N = prob.n_obs
    Pi = np.zeros((N, 2, 3))
    Pi[:, 0, 0] = fx[ci] * iz; Pi[:, 0, 2] = -fx[ci] * xn * iz
    Pi[:, 1, 1] = fy[ci] * iz; Pi[:, 1, 2] = -fy[ci] * yn * iz
    Jc = np.zeros((N, 2, prob.d))
    dY = np.einsum("nkij,nj->nik", dR[ci], X)          # [N,3(out),3(k)] = d(RX)/dr_k
    Jc[:, :, 0:3] = Pi @ dY
    Jc[:, :, 3:6] = Pi
    if prob.d == 10:
        Jc[:, 0, 6] = xn; Jc[:, 1, 7] = yn
        Jc[:, 0, 8] = 1.0; Jc[:, 1, 9] = 1.0
    Jp = Pi @ R[ci]
    f = np.empty(prob.n_res)
    f[0:2 * N:2] = fx[ci] * xn + cx[ci] - prob.uv[:, 0]
    f[1:2 * N:2] = fy[ci] * yn + cy[ci] - prob.uv[:, 1]
    Jreg = None
    if prob.d == 10:
        f[2 * N:] = regulariser(cams, prob).ravel()
        w = prob.reg_weight
        fx0 = prob.K0[0]
        C = prob.n_cams
        Jreg = np.zeros((C, 4, 4))
        Jreg[:, 0, 0] = w / fx0
        Jreg[:, 1, 0] = -w * cams[:, 7] / cams[:, 6] ** 2
        Jreg[:, 1, 1] = w / cams[:, 6]
        Jreg[:, 2, 2] = w / prob.width
        Jreg[:, 3, 3] = w / prob.height
    return f, Jc, Jp, Jreg


def dense_jacobian(x, prob):
    f, Jc, Jp, Jreg = jacobian_blocks(x, prob)
    J = np.zeros((prob.n_res, prob.n_var))
    d, C = prob.d, prob.n_cams
    for k in range(prob.n_obs):
        c, p = prob.cam_idx[k], prob.pt_idx[k]
        J[2 * k:2 * k + 2, c * d:(c + 1) * d] = Jc[k]
        J[2 * k:2 * k + 2, C * d + 3 * p:C * d + 3 * p + 3] = Jp[k]
    if Jreg is not None:
        for c in range(C):
            r0 = 2 * prob.n_obs + 4 * c
            J[r0:r0 + 4, c * d + 6:c * d + 10] = Jreg[c]
    return f, J


# --------------------------------------------------------------------------- robust loss
def huber_rho(f):
    """scipy least_squares.py:169-178 with f_scale=1 -> rho0, rho1, rho2 per scalar."""
    z = f * f
    inl = z <= 1.0
    zs = np.where(inl, 1.0, z)
    rho0 = np.where(inl, z, 2.0 * zs ** 0.5 - 1.0)
    rho1 = np.where(inl, 1.0, zs ** -0.5)
    rho2 = np.where(inl, 0.0, -0.5 * zs ** -1.5)
    return rho0, rho1, rho2


def huber_cost(f):
    return 0.5 * float(np.sum(huber_rho(f)[0]))


def robust_row_scale(f):
    """scipy common.py:720-731: row scale sqrt(max(rho1+2 rho2 f^2, EPS)); also rho1."""
    _, rho1, rho2 = huber_rho(f)
    js = rho1 + 2.0 * rho2 * f * f
    js = np.where(js < EPS, EPS, js)
    return np.sqrt(js), rho1


@dataclass
class Linearization:
    """Block form of H = J~^T J~ and g = J^T (rho1 f)  (SURVEY.md Appendix D)."""
    cost: float
    f: np.ndarray
    Jc: np.ndarray      # [N,2,d] robust-scaled rows
    Jp: np.ndarray      # [N,2,3] robust-scaled rows
    B: np.ndarray       # [C,d,d]
    Cp: np.ndarray      # [P,3,3]
    g: np.ndarray       # [n]
    Jreg: np.ndarray | None = None   # [C,4,4] scaled


def linearize(x, prob):
    f, Jc, Jp, Jreg = jacobian_blocks(x, prob)
    scale, rho1 = robust_row_scale(f)
    N, C, P, d = prob.n_obs, prob.n_cams, prob.n_pts, prob.d
    gf = (rho1 * f)                                       # gradient weights per row
    s_obs = scale[:2 * N].reshape(N, 2)
    gf_obs = gf[:2 * N].reshape(N, 2)
    g_c = np.zeros((C, d)); g_p = np.zeros((P, 3))
    np.add.at(g_c, prob.cam_idx, np.einsum("nr,nrd->nd", gf_obs, Jc))
    np.add.at(g_p, prob.pt_idx, np.einsum("nr,nrd->nd", gf_obs, Jp))
    Jc_s = Jc * s_obs[:, :, None]
    Jp_s = Jp * s_obs[:, :, None]
    B = np.zeros((C, d, d)); Cp = np.zeros((P, 3, 3))
    np.add.at(B, prob.cam_idx, np.einsum("nri,nrj->nij", Jc_s, Jc_s))
    np.add.at(Cp, prob.pt_idx, np.einsum("nri,nrj->nij", Jp_s, Jp_s))
    Jreg_s = None
    if Jreg is not None:
        s_reg = scale[2 * N:].reshape(C, 4)
        gf_reg = gf[2 * N:].reshape(C, 4)
        g_c[:, 6:10] += np.einsum("cr,crj->cj", gf_reg, Jreg)
        Jreg_s = Jreg * s_reg[:, :, None]
        B[:, 6:10, 6:10] += np.einsum("cri,crj->cij", Jreg_s, Jreg_s)
    g = np.concatenate([g_c.ravel(), g_p.ravel()])
    return Linearization(huber_cost(f), f, Jc_s, Jp_s, B, Cp, g, Jreg_s)


def dense_H(lin, prob):
    C, P, d, N = prob.n_cams, prob.n_pts, prob.d, prob.n_obs
    n = prob.n_var
    H = np.zeros((n, n))
    for c in range(C):
        H[c * d:(c + 1) * d, c * d:(c + 1) * d] = lin.B[c]
    o = C * d
    for p in range(P):
        H[o + 3 * p:o + 3 * p + 3, o + 3 * p:o + 3 * p + 3] = lin.Cp[p]
    W = np.einsum("nri,nrj->nij", lin.Jc, lin.Jp)        # [N,d,3]
    for k in range(N):
        c, p = prob.cam_idx[k], prob.pt_idx[k]
        H[c * d:(c + 1) * d, o + 3 * p:o + 3 * p + 3] += W[k]
        H[o + 3 * p:o + 3 * p + 3, c * d:(c + 1) * d] += W[k].T
    return H


def apply_J(lin, prob, p):
    """||J~ p||^2 pieces: returns the vector J~ p (robust-scaled rows)."""
    pc, pp = split_x(p, prob)
    jp = (np.einsum("nrd,nd->nr", lin.Jc, pc[prob.cam_idx])
          + np.einsum("nrd,nd->nr", lin.Jp, pp[prob.pt_idx])).ravel()
    if lin.Jreg is not None:
        jr = np.einsum("crj,cj->cr", lin.Jreg, pc[:, 6:10]).ravel()
        jp = np.concatenate([jp, jr])
    return jp


# --------------------------------------------------------------------------- damped solve
def schur_solve(lin, prob, alpha, rhs=None):
    """Solve (H + alpha I) s = rhs by eliminating the points.

    rhs defaults to -g.  Returns s [n].  Dense reduced camera system, Cholesky.
    """
    C, P, d = prob.n_cams, prob.n_pts, prob.d
    b = -lin.g if rhs is None else np.asarray(rhs, dtype=np.float64)
    bc = b[:C * d].reshape(C, d)
    bp = b[C * d:].reshape(P, 3)
    Ca_inv = np.linalg.inv(lin.Cp + alpha * np.eye(3)[None])
    W = np.einsum("nri,nrj->nij", lin.Jc, lin.Jp)         # [N,d,3]
    Y = np.einsum("nij,njk->nik", W, Ca_inv[prob.pt_idx])  # W C^-1
    S = np.zeros((C * d, C * d))
    for c in range(C):
        S[c * d:(c + 1) * d, c * d:(c + 1) * d] = lin.B[c] + alpha * np.eye(d)
    r = bc.copy()
    np.add.at(r, prob.cam_idx, -np.einsum("nij,nj->ni", Y, bp[prob.pt_idx]))
    # S -= sum over points of Y_k W_k'^T for every pair (k,k') on the same track
    order = np.argsort(prob.pt_idx, kind="stable")
    bounds = np.searchsorted(prob.pt_idx[order], np.arange(P + 1))
    for j in range(P):
        ks = order[bounds[j]:bounds[j + 1]]
        for k in ks:
            ck = prob.cam_idx[k]
            for k2 in ks:
                c2 = prob.cam_idx[k2]
                S[ck * d:(ck + 1) * d, c2 * d:(c2 + 1) * d] -= Y[k] @ W[k2].T
    L = np.linalg.cholesky(S)
    sc = np.linalg.solve(L.T, np.linalg.solve(L, r.ravel())).reshape(C, d)
    t = bp.copy()
    np.add.at(t, prob.pt_idx, -np.einsum("nij,ni->nj", W, sc[prob.cam_idx]))
    sp = np.einsum("pij,pj->pi", Ca_inv, t)
    return np.concatenate([sc.ravel(), sp.ravel()])


def dense_solve(H, alpha, rhs):
    import scipy.linalg as sla
    cf = sla.cho_factor(H + alpha * np.eye(H.shape[0]), lower=True)
    return sla.cho_solve(cf, rhs)


# --------------------------------------------------------------------------- TRF
def solve_tr_more(solve, g, Delta, initial_alpha, alpha_floor=0.0, rtol=0.01, max_iter=10):
    """scipy common.py:57-168 with the SVD replaced by solves of (H+alpha I).

    `solve(alpha, rhs)` returns (H + alpha I)^-1 rhs.  full_rank is False (gauge freedom),
    so alpha_lower starts at 0 and the Gauss-Newton shortcut is never taken.
    Returns p (rescaled to ||p|| = Delta), alpha, n_iter.
    """
    alpha_upper = np.linalg.norm(g) / Delta
    alpha_lower = 0.0
    if initial_alpha is None or initial_alpha == 0:
        alpha = max(0.001 * alpha_upper, (alpha_lower * alpha_upper) ** 0.5)
    else:
        alpha = initial_alpha
    it = -1
    for it in range(max_iter):
        if alpha < alpha_lower or alpha > alpha_upper:
            alpha = max(0.001 * alpha_upper, (alpha_lower * alpha_upper) ** 0.5)
        on_floor = alpha <= alpha_floor
        if on_floor:
            alpha = alpha_floor
        p = solve(alpha, -g)
        p_norm = np.linalg.norm(p)
        phi = p_norm - Delta
        if on_floor and phi < 0:
            return p * (Delta / p_norm), alpha, it + 1
        q = solve(alpha, p)
        phi_prime = -float(np.dot(p, q)) / p_norm
        if phi < 0:
            alpha_upper = alpha
        ratio = phi / phi_prime
        alpha_lower = max(alpha_lower, alpha - ratio)
        alpha -= (phi + Delta) * ratio / Delta
        if abs(phi) < rtol * Delta:
            break
    alpha = max(alpha, alpha_floor, 1e-300)
    p = solve(alpha, -g)
    p *= Delta / np.linalg.norm(p)
    return p, alpha, it + 1


def update_tr_radius(Delta, actual_reduction, predicted_reduction, step_norm, bound_hit):
    """scipy common.py:222-248."""
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
    """scipy common.py:705-717."""
    ftol_satisfied = dF < ftol * F and ratio > 0.25
    xtol_satisfied = dx_norm < xtol * (xtol + x_norm)
    if ftol_satisfied and xtol_satisfied:
        return 4
    elif ftol_satisfied:
        return 2
    elif xtol_satisfied:
        return 3
    return None


@dataclass
class TRFResult:
    x: np.ndarray
    cost: float
    nfev: int
    njev: int
    status: int
    optimality: float
    trace: list = field(default_factory=list)   # (alpha, Delta, step_norm, accepted) per trial

    @property
    def success(self):
        return self.status > 0


def trf(prob, x0, ftol=1e-4, xtol=1e-4, gtol=1e-8, max_nfev=100, solver="dense",
        max_outer=None):
    """scipy trf.py:401-560 (trf_no_bounds, x_scale=1, loss='huber', tr_solver='exact').

    solver="dense": Cholesky of the dense H + alpha I;  "schur": point elimination.
    max_outer: optional cap on accepted+rejected outer iterations (fixed-schedule runs).
    """
    x = np.asarray(x0, dtype=np.float64).copy()
    lin = linearize(x, prob)
    cost = lin.cost
    nfev, njev = 1, 1
    g = lin.g
    Delta = np.linalg.norm(x0)
    if Delta == 0:
        Delta = 1.0
    alpha = 0.0
    status = None
    iteration = 0
    trace = []
    g_norm = 0.0

    def make_solver(lin):
        if solver == "dense":
            H = dense_H(lin, prob)
            return lambda a, rhs: dense_solve(H, a, rhs)
        return lambda a, rhs: schur_solve(lin, prob, a, rhs)

    solve = make_solver(lin)
    hdiag = lambda l: max(float(np.max(np.einsum("cii->ci", l.B))), float(np.max(np.einsum("pii->pi", l.Cp))))
    floor = ALPHA_FLOOR_REL * hdiag(lin)
    while True:
        g_norm = float(np.linalg.norm(g, ord=np.inf))
        if g_norm < gtol:
            status = 1
        if status is not None or nfev == max_nfev:
            break
        if max_outer is not None and iteration >= max_outer:
            break
        actual_reduction = -1.0
        x_new = x
        cost_new = cost
        while actual_reduction <= 0 and nfev < max_nfev:
            step, alpha, _ = solve_tr_more(solve, g, Delta, alpha, floor)
            jp = apply_J(lin, prob, step)
            predicted_reduction = -(0.5 * float(np.dot(jp, jp)) + float(np.dot(g, step)))
            x_new = x + step
            f_new = residuals(x_new, prob)
            nfev += 1
            step_norm = float(np.linalg.norm(step))
            if not np.all(np.isfinite(f_new)):
                Delta = 0.25 * step_norm
                continue
            cost_new = huber_cost(f_new)
            actual_reduction = cost - cost_new
            Delta_new, ratio = update_tr_radius(Delta, actual_reduction, predicted_reduction,
                                                step_norm, step_norm > 0.95 * Delta)
            trace.append((alpha, Delta, step_norm, actual_reduction > 0))
            status = check_termination(actual_reduction, cost, step_norm,
                                       float(np.linalg.norm(x)), ratio, ftol, xtol)
            if status is not None:
                break
            alpha *= Delta / Delta_new
            Delta = Delta_new
        if actual_reduction > 0:
            x = x_new
            cost = cost_new
            lin = linearize(x, prob)
            njev += 1
            g = lin.g
            solve = make_solver(lin)
            floor = ALPHA_FLOOR_REL * hdiag(lin)
        iteration += 1
    if status is None:
        status = 0
    return TRFResult(x, cost, nfev, njev, status, g_norm, trace)


# --------------------------------------------------------------------------- drop-in pieces
def pack_state(poses, points3D, point_tracks, K, d=10, order="reference",
               width=1024, height=768):
    """Mirror of the packing at sfm_reconstruction.py:409-451 -> (BAProblem, x0, ids)."""
    ids = list(poses.keys())
    id_to_idx = {img_id: i for i, img_id in enumerate(ids)}
    cams = np.zeros((len(ids), d))
    for i, img_id in enumerate(ids):
        R, t = poses[img_id]
        cams[i, :3] = rotation_to_rvec(R)
        cams[i, 3:6] = np.asarray(t, dtype=np.float64).reshape(3)
        if d == 10:
            cams[i, 6:] = (K[0, 0], K[1, 1], K[0, 2], K[1, 2])
    pts = np.array([np.asarray(p, dtype=np.float64).ravel() for p in points3D]).reshape(-1, 3)
    cam_idx, pt_idx, uv = [], [], []
    for i, track in enumerate(point_tracks):
        for img_id, pt2d in track.items():
            pt_idx.append(i); cam_idx.append(id_to_idx[img_id])
            uv.append(np.asarray(pt2d, dtype=np.float64).ravel())
    cam_idx = np.asarray(cam_idx, dtype=np.int64)
    pt_idx = np.asarray(pt_idx, dtype=np.int64)
    uv = np.asarray(uv, dtype=np.float64).reshape(-1, 2)
    prob = BAProblem(len(ids), pts.shape[0], d, cam_idx, pt_idx,
                     effective_uv(uv, cam_idx, order),
                     np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]], dtype=np.float64),
                     float(width), float(height))
    x0 = np.concatenate([cams.ravel(), pts.ravel()])
    return prob, x0, ids


def reconstruction_stats(poses, points3D, point_tracks, K):
    """compute_reconstruction_stats, sfm_reconstruction.py:582-631 (vectorised)."""
    errs, lens = [], []
    for X, track in zip(points3D, point_tracks):
        X = np.asarray(X, dtype=np.float64)
        for img_id, p2 in track.items():
            R, t = poses[img_id]
            P = K @ np.hstack([R, np.asarray(t, dtype=np.float64).reshape(3, 1)])
            pr = P @ np.append(X, 1.0)
            errs.append(np.linalg.norm(pr[:2] / pr[2] - np.asarray(p2, dtype=np.float64)))
        lens.append(len(track))
    if not errs:
        return dict(mean_reproj_error=0, max_reproj_error=0, mean_track_length=0,
                    max_track_length=0, num_points=len(points3D), num_cameras=len(poses))
    return dict(mean_reproj_error=float(np.mean(errs)), max_reproj_error=float(np.max(errs)),
                mean_track_length=float(np.mean(lens)), max_track_length=float(np.max(lens)),
                num_points=len(points3D), num_cameras=len(poses))
