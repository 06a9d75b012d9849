#!/usr/bin/env python3
"""Extract DATA-ONLY fixtures from the reference's shipped bunny_data/ (build container only).

* bunny_state.npz   - the 35-camera / 2,555-point / 5,110-observation reconstruction state
  the reference ships (bunny_data/reconstruction/{poses,points3D}.json): a realistic BA input
  (it is pre-BA state, not a BA output - SURVEY.md section 4).
* bunny_matches.npz - queryIdx / trainIdx / distance of the 148 shipped match files
  (bunny_data/matches/*.npz), concatenated with offsets: evidence for the matcher's output
  contract (query-sorted, one per query, distance == sqrtf(int)).
Loaded with json / numpy.load(allow_pickle=False) only.
"""
import glob
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/bunny_data"


def main():
    poses = json.load(open(f"{SRC}/reconstruction/poses.json"))
    pts = json.load(open(f"{SRC}/reconstruction/points3D.json"))
    ids = list(poses.keys())
    id_to_idx = {k: i for i, k in enumerate(ids)}
    R = np.stack([np.asarray(poses[k]["R"], dtype=np.float64) for k in ids])
    t = np.stack([np.asarray(poses[k]["t"], dtype=np.float64).reshape(3) for k in ids])
    P = np.asarray(pts["points3D"], dtype=np.float64)
    cam_idx, pt_idx, uv = [], [], []
    for j, tr in enumerate(pts["tracks"]):
        for k, p2 in tr.items():
            cam_idx.append(id_to_idx[k]); pt_idx.append(j); uv.append(p2)
    np.savez_compressed(os.path.join(HERE, "bunny_state.npz"), ids=np.asarray([int(i) for i in ids]),
                        R=R, t=t, pts=P, cam_idx=np.asarray(cam_idx, np.int64),
                        pt_idx=np.asarray(pt_idx, np.int64), uv=np.asarray(uv, np.float64))
    q, tr_, d, off, names = [], [], [], [0], []
    for fn in sorted(glob.glob(f"{SRC}/matches/*.npz")):
        m = np.load(fn, allow_pickle=False)
        q.append(m["queryIdx"]); tr_.append(m["trainIdx"]); d.append(m["distance"])
        off.append(off[-1] + len(m["queryIdx"])); names.append(os.path.basename(fn))
    np.savez_compressed(os.path.join(HERE, "bunny_matches.npz"), queryIdx=np.concatenate(q),
                        trainIdx=np.concatenate(tr_), distance=np.concatenate(d),
                        offsets=np.asarray(off, np.int64), names=np.asarray(names))
    print("cams", len(ids), "pts", P.shape[0], "obs", len(cam_idx), "match files", len(names),
          "matches", off[-1])


if __name__ == "__main__":
    main()
