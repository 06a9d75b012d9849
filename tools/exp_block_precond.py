#!/usr/bin/env python3
"""How many CG iterations (relative residual 1e-13) would the camera system need with LARGER diagonal blocks as the
preconditioner?  Takes S(alpha) from the GPU workspace and runs block-Jacobi PCG in NumPy with blocks of 1, 2, 5, 10, 20
consecutive cameras, on the random and the spatially coherent scene.  python tools/exp_block_precond.py"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import synth
from sfm_amd.ba import GpuBA

def pcg_iters(S, b, blk, rtol=1e-13, max_it=400):
    n = S.shape[0]
    Ls = []
    for i in range(0, n, blk):
        Ls.append(np.linalg.cholesky(S[i:i + blk, i:i + blk]))
    def prec(r):
        z = np.empty_like(r)
        for j, i in enumerate(range(0, n, blk)):
            L = Ls[j]
            z[i:i + blk] = np.linalg.solve(L.T, np.linalg.solve(L, r[i:i + blk]))
        return z
    x = np.zeros(n); r = b.copy(); z = prec(r); p = z.copy(); rz = r @ z
    # stop on the residual of the SCALED system (what the product iterates on): ||E^-1 r||
    def snorm(r):
        out = 0.0
        for j, i in enumerate(range(0, n, blk)):
            y = np.linalg.solve(Ls[j], r[i:i + blk]); out += y @ y
        return out
    r0 = snorm(r)
    for it in range(1, max_it + 1):
        Ap = S @ p
        a = rz / (p @ Ap)
        x += a * p; r -= a * Ap
        if snorm(r) <= rtol * rtol * r0:
            return it
        z = prec(r); rz_new = r @ z
        p = z + (rz_new / rz) * p; rz = rz_new
    return max_it

for vis in ("random", "nearest"):
    sc = synth.make_scene(200, 100000, obs_per_point=10, seed=1004, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002, visibility=vis)
    be = GpuBA(sc.cams0, sc.pts0, sc.cam_idx, sc.pt_idx, sc.uv, synth.K_REF)
    st = be.trf_begin(max_nfev=2 ** 31 - 1, check_tolerances=False)
    for _ in range(6):
        st.outer()                                  # a state as in the timed part of the bench schedule
    _, gnorm, _, hd = be.linearize()
    n = be.C * be.d
    for mult in (1e-5, 1e-3, 1e-1):
        alpha = mult * hd
        be.solve(alpha, False)
        torch.cuda.synchronize()
        Sfull = be.view(be.lay.reduce_S_off, n * n + n).cpu().numpy().copy()
        S = Sfull[:n * n].reshape(n, n); rhs = Sfull[n * n:]
        S = np.tril(S) + np.tril(S, -1).T + alpha * np.eye(n)
        its = [pcg_iters(S, rhs, 10 * k) for k in (1, 2, 5, 10, 20)]
        print(vis, "alpha = %.0e hdiag: CG iterations with blocks of 1 / 2 / 5 / 10 / 20 cameras:" % mult, its, "(product:", "see solver_stats)", flush=True)
    st.close()
