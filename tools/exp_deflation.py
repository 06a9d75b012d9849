#!/usr/bin/env python3
"""Would deflating the k smallest eigen-directions of the block-scaled camera system cut its CG iterations?  S(alpha) from the
GPU workspace (after six outer iterations), scaled by its 10 x 10 diagonal blocks, exact eigenvectors from NumPy, deflated CG to
a relative residual of 1e-13.  python tools/exp_deflation.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import synth
from sfm_amd.ba import GpuBA

def cg_deflated(A, b, Z, rtol=1e-13, max_it=400):
    n = A.shape[0]
    if Z is not None:
        AZ = A @ Z; E = Z.T @ AZ; Einv = np.linalg.inv(E)
        P = lambda v: v - AZ @ (Einv @ (Z.T @ v))          # P A is singular on span(Z); solve P A x = P b
    else:
        P = lambda v: v
    x = np.zeros(n); r = P(b); p = r.copy(); rr = r @ r; rr0 = rr
    for it in range(1, max_it + 1):
        Ap = P(A @ p)
        a = rr / (p @ Ap)
        x += a * p; r -= a * Ap
        rr_new = r @ r
        if rr_new <= rtol * rtol * rr0:
            return it
        p = r + (rr_new / rr) * p; rr = rr_new
    return max_it

for vis in ("random", "nearest"):
    sc = synth.make_scene(200, 100000, obs_per_point=10, seed=1004, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002, visibility=vis)
    be = GpuBA(sc.cams0, sc.pts0, sc.cam_idx, sc.pt_idx, sc.uv, synth.K_REF)
    st = be.trf_begin(max_nfev=2 ** 31 - 1, check_tolerances=False)
    for _ in range(6):
        st.outer()
    _, gnorm, _, hd = be.linearize()
    n = be.C * be.d
    for mult in (1e-5, 1e-3):
        alpha = mult * hd
        be.solve(alpha, False)
        torch.cuda.synchronize()
        Sfull = be.view(be.lay.reduce_S_off, n * n + n).cpu().numpy().copy()
        S = Sfull[:n * n].reshape(n, n); rhs = Sfull[n * n:]
        S = np.tril(S) + np.tril(S, -1).T + alpha * np.eye(n)
        Einv = np.zeros_like(S)
        for i in range(0, n, 10):
            Einv[i:i + 10, i:i + 10] = np.linalg.inv(np.linalg.cholesky(S[i:i + 10, i:i + 10]))
        A = Einv @ S @ Einv.T; b = Einv @ rhs
        w, V = np.linalg.eigh(A)
        its = [cg_deflated(A, b, None)] + [cg_deflated(A, b, V[:, :k]) for k in (7, 20, 50)]
        print(vis, "alpha = %.0e hdiag: eigenvalues %.2e .. %.2e (8th %.2e, 21st %.2e); CG iterations plain / 7 / 20 / 50 deflated:" % (
            mult, w[0], w[-1], w[7], w[20]), its, flush=True)
    st.close()


# ------------------------------------------------------------------ the same with the ANALYTIC gauge directions
def rodrigues(r):
    th = np.linalg.norm(r)
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * (K @ K)

def right_jacobian_inv(r):
    th = np.linalg.norm(r)
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    if th < 1e-8:
        return np.eye(3) + 0.5 * K
    return np.eye(3) + 0.5 * K + (1 / th ** 2 - (1 + np.cos(th)) / (2 * th * np.sin(th))) * (K @ K)

def gauge_directions(cams, d):
    """[n, 7]: camera parts of the 7 similarity generators (3 translations, scale, 3 rotations) for blocks [rvec, t, ...]."""
    C = cams.shape[0]
    D = np.zeros((C * d, 7))
    for c in range(C):
        r, t = cams[c, :3], cams[c, 3:6]
        R = rodrigues(r)
        o = c * d
        D[o + 3:o + 6, 0:3] = -R                       # X' = X + tau  ->  t' = t - R tau
        D[o + 3:o + 6, 3] = t                          # X' = (1 + e) X  ->  t' = (1 + e) t
        D[o:o + 3, 4:7] = -right_jacobian_inv(r)       # X' = (I + [w]x) X  ->  R' = R exp(-[w]x)
    return D

print("--- analytic gauge directions", flush=True)
for vis in ("random", "nearest"):
    sc = synth.make_scene(200, 100000, obs_per_point=10, seed=1004, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002, visibility=vis)
    be = GpuBA(sc.cams0, sc.pts0, sc.cam_idx, sc.pt_idx, sc.uv, synth.K_REF)
    st = be.trf_begin(max_nfev=2 ** 31 - 1, check_tolerances=False)
    for _ in range(6):
        st.outer()
    _, gnorm, _, hd = be.linearize()
    n = be.C * be.d
    cams, _ = be.params()
    Dg = gauge_directions(np.asarray(cams), be.d)
    for mult in (1e-7, 1e-5, 1e-3, 1e-1):
        alpha = mult * hd
        be.solve(alpha, False)
        torch.cuda.synchronize()
        Sfull = be.view(be.lay.reduce_S_off, n * n + n).cpu().numpy().copy()
        S = Sfull[:n * n].reshape(n, n); rhs = Sfull[n * n:]
        S = np.tril(S) + np.tril(S, -1).T + alpha * np.eye(n)
        Einv = np.zeros_like(S); Ef = np.zeros_like(S)
        for i in range(0, n, 10):
            L = np.linalg.cholesky(S[i:i + 10, i:i + 10])
            Ef[i:i + 10, i:i + 10] = L; Einv[i:i + 10, i:i + 10] = np.linalg.inv(L)
        A = Einv @ S @ Einv.T; b = Einv @ rhs
        Z = Ef.T @ Dg
        Z, _ = np.linalg.qr(Z)
        res = np.linalg.norm(A @ Z, axis=0)
        print(vis, "alpha = %.0e hdiag: ||A z|| of the 7 analytic directions %.1e .. %.1e; CG iterations plain / analytic-deflated:" % (
            mult, res.min(), res.max()), [cg_deflated(A, b, None), cg_deflated(A, b, Z)], flush=True)
    st.close()
