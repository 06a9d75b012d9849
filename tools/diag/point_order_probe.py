#!/usr/bin/env python3
"""Does the ORDER OF THE POINTS matter to the Schur gather?  G is stored point-major; on the spatially coherent scene the points
come in random order (a point's neighbours in space - which share its cameras - are anywhere in G).  This probe renumbers the
points so that points with the same cameras are neighbours (lexicographic order of their camera lists, or of the lowest camera
alone) and times the kernels of a damped solve on the scene as given and renumbered.
usage: point_order_probe.py [cams pts [visibility]]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sfm_amd import synth
from sfm_amd.ba import GpuBA

C_, P_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 100000)
vis = sys.argv[3] if len(sys.argv) > 3 else "nearest"
L = 10
sc = synth.make_scene(C_, P_, obs_per_point=L, seed=1004, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002, visibility=vis)


def renumber(perm):
    """Scene arrays with point perm[i] as new point i (observations stay point-major)."""
    cam = sc.cam_idx.reshape(P_, L)[perm].ravel()
    uv = sc.uv.reshape(P_, L, 2)[perm].reshape(-1, 2)
    return sc.pts0[perm], cam, np.repeat(np.arange(P_, dtype=np.int64), L), uv


cams = sc.cam_idx.reshape(P_, L)
orders = {"as given": np.arange(P_),
          "by lowest camera": np.argsort(cams[:, 0], kind="stable"),
          "by camera list": np.lexsort(cams.T[::-1])}
for name, perm in orders.items():
    pts0, ci, pi, uv = renumber(perm)
    be = GpuBA(sc.cams0, pts0, ci, pi, uv, synth.K_REF)
    cost, gnorm, _, hd = be.linearize()
    alpha = 1e-4 * hd
    for _ in range(3):
        be.solve(alpha, True)
    be.h.set_profiling(True); be.h.profile()
    reps = 10
    for _ in range(reps):
        be.solve(alpha, True)
    torch.cuda.synchronize()
    prof = be.h.profile(); be.h.set_profiling(False)
    us = {k: v[0] / v[1] * 1e3 for k, v in prof.items() if v[1] > 0}
    print("%-18s cost %.6e  us per launch: build_G %.1f schur %.1f (gather %.1f) backsub %.1f chol %.1f trsv %.1f" % (
        name, cost, us["build_G"], us["schur"], us["schur_items"], us["backsub"], us["chol"], us["trsv"]), flush=True)
    del be
    torch.cuda.empty_cache()
