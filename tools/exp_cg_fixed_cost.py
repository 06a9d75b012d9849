#!/usr/bin/env python3
"""Camera-solve slots (HIP events) against CG iterations per system at cfg4, over a range of alpha: what a system costs
before its first iteration.  usage: exp_cg_fixed_cost.py [cams pts]; SFM_CGS_RTOL=1 gives zero iterations, SFM_CGS_XCD=0 the
device-wide exchange at sizes where the one-XCD form would run."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import synth
from sfm_amd.ba import GpuBA
C_, P_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 100000)
sc = synth.make_scene(C_, P_, obs_per_point=10, seed=1004, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002)
be = GpuBA(sc.cams0, sc.pts0, sc.cam_idx, sc.pt_idx, sc.uv, synth.K_REF)
_, gnorm, _, hd = be.linearize()
for mult in (1e-6, 1e-4, 1e-2, 1.0, 1e2, 1e4):
    alpha = mult * hd
    for _ in range(3):
        be.solve(alpha, True)
    its0 = be.solver_stats()[0]
    be.h.set_profiling(True); be.h.profile()
    reps = 10
    for _ in range(reps):
        be.solve(alpha, True)
    torch.cuda.synchronize()
    prof = be.h.profile(); be.h.set_profiling(False)
    its = (be.solver_stats()[0] - its0) / (2.0 * reps)
    print("alpha = %.0e hdiag: %.1f CG iterations per system; slots us: step system (einv + scale + CG) %.1f, q system %.1f, schur %.1f, backsub %.1f" % (
        mult, its, prof["chol"][0] / prof["chol"][1] * 1e3, prof["trsv"][0] / prof["trsv"][1] * 1e3,
        prof["schur"][0] / prof["schur"][1] * 1e3, prof["backsub"][0] / prof["backsub"][1] * 1e3), flush=True)
