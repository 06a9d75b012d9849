#!/usr/bin/env python3
"""Golden BA vector on the reconstruction state the reference SHIPS (bunny_data/reconstruction/*.json,
via tests/golden/bunny_state.npz): runs the reference's own bundle_adjust (same harness as
make_golden.py) on 35 cameras / 2,555 points / 5,110 observations, literal (bug-compatible) pairing.
Build container only; takes several minutes (dense finite-difference Jacobian + SVD of 10,360 x 8,015)."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg      # noqa: E402


class _Scene:
    def __init__(self, b):
        self.b = b
        self.cam_idx, self.pt_idx, self.uv = b["cam_idx"], b["pt_idx"], b["uv"]

    def state(self):
        b = self.b
        ids = [str(i) for i in b["ids"]]
        poses = {k: (b["R"][i], b["t"][i].reshape(3, 1)) for i, k in enumerate(ids)}
        tracks = [dict() for _ in range(b["pts"].shape[0])]
        for k in range(len(b["cam_idx"])):
            tracks[int(b["pt_idx"][k])][ids[int(b["cam_idx"][k])]] = b["uv"][k].tolist()
        K = np.array([[1228, 0, 512], [0, 1228, 384], [0, 0, 1]], dtype=np.float64)
        return poses, b["pts"].tolist(), tracks, K


def main():
    b = dict(np.load(os.path.join(HERE, "bunny_state.npz")))
    m = mg.load_reference()
    sc = _Scene(b)
    for aligned in (False,):
        t0 = time.time()
        rec = mg.run_reference(m, sc, aligned)
        tag = "aligned" if aligned else "reference"
        poses, pts, tracks, K = sc.state()
        np.savez_compressed(os.path.join(HERE, f"ba_bunny_{tag}.npz"), K=K, R0=b["R"], t0=b["t"], pts0=b["pts"],
                            cam_idx=b["cam_idx"], pt_idx=b["pt_idx"], uv=b["uv"], order=tag,
                            x0=rec["x0"], x=rec["x"], nfev=rec["nfev"], njev=rec["njev"], status=rec["status"],
                            cost=rec["cost"], f0_norm=rec["f0_norm"], f1_norm=rec["f1_norm"], K_after=rec["K_after"],
                            ret=rec["ret"], t_shape_after=rec["t_shape_after"], solver_kwargs=rec["kwargs"])
        print(tag, "nfev", rec["nfev"], "njev", rec["njev"], "status", rec["status"], "cost", rec["cost"],
              "|f|", rec["f0_norm"], "->", rec["f1_norm"], f"{time.time() - t0:.0f}s", flush=True)


if __name__ == "__main__":
    main()
