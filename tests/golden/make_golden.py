#!/usr/bin/env python3
"""Generate golden BA vectors by running the REFERENCE's own bundle_adjust.

Runs only in the build container (needs /root/reference; never on the GPU box).  The
reference module is imported unmodified with a NumPy stand-in for the single OpenCV call
it needs on this path (cv2.Rodrigues, sfm_reconstruction.py:419,465,544); `cv2` itself is
not installable here.  `scipy.optimize.least_squares` is wrapped only to record its
arguments and result.  For the "aligned" variant the closure cell `points2D` is permuted
into camera-grouped order before the solve.

Output: tests/golden/ba_<name>.npz holding inputs (K, R0, t0, pts0, cam_idx, pt_idx, uv)
and the reference's outputs (x0, x, nfev, njev, status, cost, f0_norm, f1_norm, K_after,
ret).  Only data is stored - no reference source.
"""
import importlib
import os
import sys
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

from oracle import ba_oracle as bo          # noqa: E402  (Rodrigues stand-in only)
from sfm_amd import synth                   # noqa: E402


def _cv2_stub():
    cv2 = types.ModuleType("cv2")

    def Rodrigues(a):
        a = np.asarray(a, dtype=np.float64)
        if a.size == 3:
            R, _ = bo.rotation_and_derivs(a.reshape(1, 3))
            return R[0], None
        return bo.rotation_to_rvec(a.reshape(3, 3)).reshape(3, 1), None

    cv2.Rodrigues = Rodrigues
    return cv2


def load_reference():
    sys.modules["cv2"] = _cv2_stub()
    sys.path.insert(0, "/root/reference")
    return importlib.import_module("utils.sfm_reconstruction")


MAX_NFEV_OVERRIDE = None      # --max-nfev N: the same reference run cut short (parity after a fixed iteration count)


def run_reference(m, scene, aligned):
    poses, pts, tracks, K = scene.state()
    s = object.__new__(m.StructureFromMotion)
    s.K = K.copy(); s.image_width = 1024; s.image_height = 768
    s.poses = poses; s.points3D = pts; s.point_tracks = tracks
    rec = {}
    real = m.optimize.least_squares

    def wrapper(fun, x0, **kw):
        cells = dict(zip(fun.__code__.co_freevars, fun.__closure__))
        if aligned:
            cam_idxs = cells["camera_idxs"].cell_contents
            p2d = cells["points2D"].cell_contents
            perm = np.argsort(cam_idxs, kind="stable")
            p2d[:] = p2d[perm].copy()
        if MAX_NFEV_OVERRIDE is not None:
            kw = dict(kw, max_nfev=MAX_NFEV_OVERRIDE)
        res = real(fun, x0, **kw)
        rec.update(x0=np.array(x0), x=res.x.copy(), nfev=res.nfev, njev=res.njev,
                   status=res.status, cost=res.cost, f0_norm=np.linalg.norm(fun(x0)),
                   f1_norm=np.linalg.norm(fun(res.x)), kwargs=repr(sorted(kw.items())))
        return res

    m.optimize.least_squares = wrapper
    try:
        t0 = time.time()
        ret = s.bundle_adjust()
        rec["seconds"] = time.time() - t0
    finally:
        m.optimize.least_squares = real
    rec["ret"] = -1 if ret is None else int(bool(ret))
    rec["K_after"] = np.array(s.K)
    rec["t_shape_after"] = np.array(np.asarray(next(iter(s.poses.values()))[1]).shape)
    return rec


CASES = [
    # name, n_cams, n_pts, obs_per_point, noise_px, pt_sigma, cam_sigma, seed
    ("c3p20_n0",    3,  20, None, 0.0, 0.02, 0.0, 11),
    ("c3p20_n05",   3,  20, None, 0.5, 0.02, 0.0, 12),
    ("c5p50_n05",   5,  50, None, 0.5, 0.02, 0.0, 13),
    ("c10p100_n05", 10, 100, None, 0.5, 0.02, 0.0, 14),
    ("c8p60_L4",    8,  60, 4,    0.5, 0.02, 0.01, 15),
    ("c6p40_cam",   6,  40, None, 0.3, 0.01, 0.005, 16),
    ("c15p200_L6",  15, 200, 6,   0.5, 0.02, 0.005, 17),
    ("c20p300_L5",  20, 300, 5,   0.7, 0.015, 0.008, 18),
    # BASELINE.json configs[0] at its stated size (10 cameras / 1,000 points / 10,000 observations, n = 3,100,
    # m = 20,040): ~25 s (literal) + ~55 s (aligned) of the reference's own CPU path on 8 cores
    ("cfg1_c10p1000", 10, 1000, None, 0.5, 0.02, 0.0, 1001),
]


def main():
    global MAX_NFEV_OVERRIDE
    argv = sys.argv[1:]
    suffix = ""
    variants = (False, True)
    if "--max-nfev" in argv:
        i = argv.index("--max-nfev")
        MAX_NFEV_OVERRIDE = int(argv[i + 1])
        suffix = f"_nfev{MAX_NFEV_OVERRIDE}"
        del argv[i:i + 2]
    if "--literal-only" in argv:
        argv.remove("--literal-only")
        variants = (False,)
    m = load_reference()
    only = set(argv)
    for name, C, P, L, noise, ps, cs, seed in CASES:
        if only and name not in only:
            continue
        scene = synth.make_scene(C, P, obs_per_point=L, seed=seed, noise_px=noise,
                                 pt_sigma=ps, cam_sigma=cs)
        poses, pts, tracks, K = scene.state()
        R0 = np.stack([poses[k][0] for k in poses])
        t0 = np.stack([np.asarray(poses[k][1]).reshape(3) for k in poses])
        for aligned in variants:
            rec = run_reference(m, scene, aligned)
            tag = "aligned" if aligned else "reference"
            out = os.path.join(HERE, f"ba_{name}_{tag}{suffix}.npz")
            np.savez_compressed(
                out, K=K, R0=R0, t0=t0, pts0=np.asarray(pts), cam_idx=scene.cam_idx,
                pt_idx=scene.pt_idx, uv=scene.uv, order=tag,
                x0=rec["x0"], x=rec["x"], nfev=rec["nfev"], njev=rec["njev"],
                status=rec["status"], cost=rec["cost"], f0_norm=rec["f0_norm"],
                f1_norm=rec["f1_norm"], K_after=rec["K_after"], ret=rec["ret"],
                t_shape_after=rec["t_shape_after"], solver_kwargs=rec["kwargs"])
            print(f"{name:14s} {tag:9s} nfev={rec['nfev']:3d} njev={rec['njev']:2d} "
                  f"status={rec['status']} cost={rec['cost']:.6g} "
                  f"|f| {rec['f0_norm']:.4g}->{rec['f1_norm']:.4g} ret={rec['ret']} "
                  f"{rec['seconds']:.1f}s", flush=True)


if __name__ == "__main__":
    main()
