#!/usr/bin/env python3
"""Golden vectors for the driver-side rows - build container only.

1. bunny_pairs.npz  - DATA the reference ships for its 148 verified pairs: F, all matched pts1/pts2
   (float32), the inlier mask geometric_verification produced for them
   (bunny_data/fundamental/*.npz: F, pts1, pts2, mask) and the metrics it logged
   (bunny_data/matching_results.csv).  A known-answer set for find_matches.py:157-201.
2. driver_bunny.npz - outputs of the reference's OWN methods run here on the shipped state
   (tests/golden/bunny_state.npz) and shipped correspondence files:
     * find_2d3d_matches(image_id) for a few images (pair order as find_image_pairs globbed it);
     * add_new_matches(pair, image_id) replayed on a truncated state, with triangulate_dlt (the
       restated cv2.triangulatePoints; OpenCV is not installable here) as the stand-in - pins the
       dedupe / gating / append logic, not the DLT arithmetic (that is pinned by the shipped points).
Only data is stored - no reference source."""
import csv
import glob
import os
import sys
from pathlib import Path

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                      # noqa: E402
from oracle import driver_oracle as do        # noqa: E402

SRC = "/root/reference/bunny_data"


def extract_pairs():
    rows = {r["pair_name"]: r for r in csv.DictReader(open(f"{SRC}/matching_results.csv"))}
    names, F, p1, p2, mk, off = [], [], [], [], [], [0]
    nm, ni, ir, re_, wd = [], [], [], [], []
    for fn in sorted(glob.glob(f"{SRC}/fundamental/*_F.npz")):
        name = os.path.basename(fn)[:-6]
        z = np.load(fn, allow_pickle=False)
        r = rows[name]
        names.append(name); F.append(z["F"]); p1.append(z["pts1"]); p2.append(z["pts2"]); mk.append(z["mask"])
        off.append(off[-1] + len(z["mask"]))
        nm.append(int(r["num_matches"])); ni.append(int(r["num_inliers"])); ir.append(float(r["inlier_ratio"]))
        re_.append(r["reprojection_error"]); wd.append(r["well_distributed"] == "True")
        c1 = np.load(f"{SRC}/correspondences/{name}_pts1.npy", allow_pickle=False)
        c2 = np.load(f"{SRC}/correspondences/{name}_pts2.npy", allow_pickle=False)
        assert np.array_equal(c1, z["pts1"][z["mask"]]) and np.array_equal(c2, z["pts2"][z["mask"]])
    np.savez_compressed(os.path.join(HERE, "bunny_pairs.npz"), names=np.asarray(names), F=np.stack(F),
                        pts1=np.concatenate(p1), pts2=np.concatenate(p2), mask=np.concatenate(mk),
                        offsets=np.asarray(off, np.int64), num_matches=np.asarray(nm), num_inliers=np.asarray(ni),
                        inlier_ratio=np.asarray(ir), reprojection_error=np.asarray(re_),
                        well_distributed=np.asarray(wd))
    print("pairs", len(names), "matches", off[-1])


def _stub_triangulate(m):
    def triangulatePoints(P0, P1, x0, x1):
        return do.triangulate_dlt(P0, P1, np.asarray(x0).T, np.asarray(x1).T).T
    m.cv2.triangulatePoints = triangulatePoints


def _state(m, b, n_tracks=None):
    s = object.__new__(m.StructureFromMotion)
    s.K = np.array([[1228, 0, 512], [0, 1228, 384], [0, 0, 1]], dtype=np.float64)
    s.image_width = 1024; s.image_height = 768
    s.data_dir = Path(SRC); s.matches_dir = s.data_dir / "matches"; s.fund_dir = s.data_dir / "fundamental"
    s.corr_dir = s.data_dir / "correspondences"
    ids = [int(i) for i in b["ids"]]
    s.poses = {k: (b["R"][i], b["t"][i].reshape(3, 1)) for i, k in enumerate(ids)}
    P = b["pts"].shape[0] if n_tracks is None else n_tracks
    tracks = [dict() for _ in range(b["pts"].shape[0])]
    for k in range(len(b["cam_idx"])):
        tracks[int(b["pt_idx"][k])][ids[int(b["cam_idx"][k])]] = b["uv"][k].tolist()
    s.point_tracks = tracks[:P]
    s.points3D = b["pts"][:P].tolist()
    s.constructed = [f"{i:04d}.ppm" for i in ids]
    return s, ids


def main():
    extract_pairs()
    b = dict(np.load(os.path.join(HERE, "bunny_state.npz")))
    m = mg.load_reference()
    _stub_triangulate(m)
    out = {}
    # ---- find_2d3d_matches on the full shipped state
    images = [3, 12, 20, 33]
    out["f_images"] = np.asarray(images)
    for img in images:
        s, ids = _state(m, b)
        s.constructed = [f"{i:04d}.ppm" for i in ids if i != img]
        pairs = s.find_image_pairs(img)
        p3, p2 = s.find_2d3d_matches(img)
        out[f"f{img}_pairs"] = np.asarray(pairs)
        out[f"f{img}_points3D"] = p3; out[f"f{img}_points2D"] = p2
        print("find_2d3d_matches", img, len(pairs), "pairs ->", p3.shape, p2.shape, p2.dtype)
    # ---- add_new_matches replay on a truncated state (first 600 tracks)
    n0 = 600
    s, ids = _state(m, b, n0)
    replay = ["pair_3_4", "pair_12_13", "pair_20_21", "pair_1_3", "pair_33_34"]
    replay = [p for p in replay if os.path.exists(f"{SRC}/correspondences/{p}_pts1.npy")]
    counts = []
    for pair in replay:
        before = len(s.points3D)
        ret = s.add_new_matches(pair, int(pair.split("_")[2]))
        counts.append(len(s.points3D) - before)
        print("add_new_matches", pair, "ret", ret, "added", counts[-1])
    out["a_n0"] = np.asarray(n0); out["a_pairs"] = np.asarray(replay); out["a_added"] = np.asarray(counts)
    out["a_points3D"] = np.asarray(s.points3D[n0:], dtype=np.float64)
    new_tracks = s.point_tracks[n0:]
    out["a_track_ids"] = np.asarray([list(t.keys()) for t in new_tracks], dtype=np.int64).reshape(-1, 2)
    out["a_track_uv"] = np.asarray([list(t.values()) for t in new_tracks], dtype=np.float64).reshape(-1, 2, 2)
    np.savez_compressed(os.path.join(HERE, "driver_bunny.npz"), **out)


if __name__ == "__main__":
    main()
