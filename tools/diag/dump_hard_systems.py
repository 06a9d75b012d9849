#!/usr/bin/env python3
"""Which damped camera systems of a run are hard for the block-Jacobi CG, and what do they look like?  Drives the Python
trust-region loop, records (outer iteration, alpha, alpha / max diag H, CG iterations of the step system and of the q system,
fallback) per damped solve, and saves the packed lower triangle of S (+ alpha, rhs) of the hardest ones for offline experiments
(np.savez under gpurun_out/).  usage: dump_hard_systems.py <visibility> <outer iterations> <how many to save> [cams pts]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sfm_amd import synth
from sfm_amd.ba import GpuBA
from sfm_amd.trf import trf

vis, n_outer, n_save = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
C_, P_ = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (200, 100000)
sc = synth.make_scene(C_, P_, obs_per_point=10, seed=1004, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002, visibility=vis)
be = GpuBA(sc.cams0, sc.pts0, sc.cam_idx, sc.pt_idx, sc.uv, synth.K_REF)
n = be.C * be.d
log, kept = [], []
state = {"lin": 0, "hd": 1.0}
orig_solve, orig_lin = be.solve, be.linearize

def lin():
    out = orig_lin()
    state["lin"] += 1; state["hd"] = out[3]
    return out

def solve(alpha, want_q):
    i0, f0 = be.solver_stats()
    out = orig_solve(alpha, want_q)
    i1, f1 = be.solver_stats()
    rec = (state["lin"], alpha, alpha / state["hd"], i1 - i0, f1 - f0, int(bool(want_q)))
    log.append(rec)
    if n_save and (len(kept) < n_save or (i1 - i0) > min(k[0] for k in kept)):
        torch.cuda.synchronize()
        Sfull = be.view(be.lay.reduce_S_off, n * n + n).cpu().numpy()
        S = Sfull[:n * n].reshape(n, n)
        tri = S[np.tril_indices(n)].copy()
        if len(kept) >= n_save:
            kept.remove(min(kept, key=lambda k: k[0]))
        kept.append((i1 - i0, rec, tri, Sfull[n * n:].copy()))
    return out

be.solve, be.linearize = solve, lin
res = trf(be, max_nfev=10 ** 9, max_outer=n_outer, check_tolerances=False)
print("visibility %s: %d outer iterations, %d damped solves, CG iterations %d, fallbacks %d" % ((vis, n_outer, len(log)) + be.solver_stats()))
print("lin  alpha        alpha/hdiag  CG-its(both systems)  fallback want_q")
for r in log:
    print("%3d  %.4e  %.3e   %4d   %d  %d" % r)
out = os.path.join(ROOT, "gpurun_out", "hard_systems_%s.npz" % vis)
np.savez_compressed(out, n=n, d=be.d, **{"tri%d" % i: k[2] for i, k in enumerate(kept)}, **{"rhs%d" % i: k[3] for i, k in enumerate(kept)},
                    **{"rec%d" % i: np.array(k[1]) for i, k in enumerate(kept)})
print("saved", out, [k[1] for k in kept])
