#!/usr/bin/env python3
"""Goldens for the harness parts of the path (a5 packing, a10 statistics) - build container only.

* the arrays the reference's bundle_adjust packs (sfm_reconstruction.py:409-451), captured from the closure of
  its `objective` at the moment it calls scipy.optimize.least_squares (the solve itself is skipped here):
  camera_idxs, point2D_idxs, points2D, x0;
* compute_reconstruction_stats() (:582-631) run by the reference itself;
on the shipped bunny state (35 cameras, int image ids) and on two synthetic scenes (string ids, mixed track lengths).
Only data is stored."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                      # noqa: E402
from sfm_amd import synth                     # noqa: E402


class _Abort(Exception):
    pass


def capture(m, poses, pts, tracks, K):
    s = object.__new__(m.StructureFromMotion)
    s.K = K.copy(); s.image_width = 1024; s.image_height = 768
    s.poses = poses; s.points3D = pts; s.point_tracks = tracks
    stats = s.compute_reconstruction_stats()
    rec = {}
    real = m.optimize.least_squares

    def wrapper(fun, x0, **kw):
        cells = dict(zip(fun.__code__.co_freevars, fun.__closure__))
        rec.update(camera_idxs=np.array(cells["camera_idxs"].cell_contents),
                   point2D_idxs=np.array(cells["point2D_idxs"].cell_contents),
                   points2D=np.array(cells["points2D"].cell_contents), x0=np.array(x0))
        raise _Abort()

    m.optimize.least_squares = wrapper
    try:
        s.bundle_adjust()
    except _Abort:
        pass
    finally:
        m.optimize.least_squares = real
    return rec, stats


def main():
    m = mg.load_reference()
    out = {}
    b = dict(np.load(os.path.join(HERE, "bunny_state.npz")))
    ids = [int(i) for i in b["ids"]]
    poses = {k: (b["R"][i], b["t"][i].reshape(3, 1)) for i, k in enumerate(ids)}
    tracks = [dict() for _ in range(b["pts"].shape[0])]
    for k in range(len(b["cam_idx"])):
        tracks[int(b["pt_idx"][k])][ids[int(b["cam_idx"][k])]] = b["uv"][k].tolist()
    K = np.array([[1228, 0, 512], [0, 1228, 384], [0, 0, 1]], dtype=np.float64)
    cases = {"bunny": (poses, b["pts"].tolist(), tracks, K)}
    for name, (C, P, L, seed) in {"c7p60": (7, 60, None, 31), "c12p150_L4": (12, 150, 4, 32)}.items():
        sc = synth.make_scene(C, P, obs_per_point=L, seed=seed, noise_px=0.8, pt_sigma=0.02, cam_sigma=0.005)
        cases[name] = sc.state()
    out["cases"] = np.asarray(list(cases))
    for name, (poses, pts, tracks, K) in cases.items():
        rec, stats = capture(m, poses, pts, tracks, K)
        for k, v in rec.items():
            out[f"{name}_{k}"] = v
        for k, v in stats.items():
            out[f"{name}_stat_{k}"] = np.asarray(v)
        print(name, {k: v.shape for k, v in rec.items()}, stats)
    np.savez_compressed(os.path.join(HERE, "pack_stats.npz"), **out)


if __name__ == "__main__":
    main()
