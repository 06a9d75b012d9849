#!/usr/bin/env python3
"""Digests and layouts of the artefacts the reference ships (build container only): SHA-256 + size of
bunny_data/reconstruction/{poses.json,points3D.json,reconstruction.ply} and
bunny_data/exports/colmap/{cameras,images,points3D}.txt, and the array names / dtypes / shapes of one
pair's three files.  tests/test_interchange.py regenerates the files from tests/golden/bunny_state.npz
with sfm_amd.interchange and compares.  Only digests and metadata are stored."""
import hashlib
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/bunny_data"
FILES = ["reconstruction/poses.json", "reconstruction/points3D.json", "reconstruction/reconstruction.ply",
         "exports/colmap/cameras.txt", "exports/colmap/images.txt", "exports/colmap/points3D.txt"]


def main():
    out = {"files": {}, "pair_layout": {}}
    for rel in FILES:
        b = open(f"{SRC}/{rel}", "rb").read()
        out["files"][rel] = {"sha256": hashlib.sha256(b).hexdigest(), "bytes": len(b)}
    pair = "pair_10_11"
    for rel in (f"correspondences/{pair}_pts1.npy", f"correspondences/{pair}_pts2.npy"):
        a = np.load(f"{SRC}/{rel}", allow_pickle=False)
        out["pair_layout"][rel] = {"dtype": str(a.dtype), "ndim": a.ndim, "cols": int(a.shape[1])}
    for rel in (f"fundamental/{pair}_F.npz", f"matches/{pair}_matches.npz"):
        z = np.load(f"{SRC}/{rel}", allow_pickle=False)
        out["pair_layout"][rel] = {k: {"dtype": str(z[k].dtype), "ndim": z[k].ndim} for k in z.files}
    json.dump(out, open(os.path.join(HERE, "bunny_artifacts.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out["files"], indent=1))


if __name__ == "__main__":
    main()
