"""CPU tests pinning the driver-row oracle (oracle/driver_oracle.py) against what the reference ships and
against outputs of the reference's own methods (tests/golden/make_golden_driver.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import driver_oracle as do

K_BUNNY = np.array([[1228, 0, 512], [0, 1228, 384], [0, 0, 1]], dtype=np.float64)


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def bunny_tracks(b, n=None):
    ids = [int(i) for i in b["ids"]]
    tracks = [dict() for _ in range(b["pts"].shape[0])]
    for k in range(len(b["cam_idx"])):
        tracks[int(b["pt_idx"][k])][ids[int(b["cam_idx"][k])]] = b["uv"][k].tolist()
    poses = {k: (b["R"][i], b["t"][i].reshape(3, 1)) for i, k in enumerate(ids)}
    n = len(tracks) if n is None else n
    return poses, b["pts"][:n].tolist(), tracks[:n]


def pair_corr(bp):
    """pair name -> (pts1, pts2) correspondence files (= matched points under the shipped inlier mask,
    find_matches.py:314-316; make_golden_driver.py asserts the equality with the shipped .npy files)."""
    out = {}
    off = bp["offsets"]
    for i, name in enumerate(bp["names"]):
        sl = slice(off[i], off[i + 1])
        m = bp["mask"][sl]
        out[str(name)] = (bp["pts1"][sl][m], bp["pts2"][sl][m])
    return out


def test_geometric_verification_reproduces_shipped_masks_and_metrics():
    """148 shipped pairs: `mask` in fundamental/*.npz is geometric_verification's inlier_mask for the shipped
    F / pts1 / pts2 and matching_results.csv holds its metrics (find_matches.py:289-310)."""
    bp = load("bunny_pairs.npz")
    off = bp["offsets"]
    for i, name in enumerate(bp["names"]):
        sl = slice(off[i], off[i + 1])
        res = do.geometric_verification(bp["pts1"][sl], bp["pts2"][sl], bp["F"][i])
        assert np.array_equal(res["inlier_mask"], bp["mask"][sl]), name
        m = res["metrics"]
        assert m["total_matches"] == bp["num_matches"][i]
        assert int(m["inliers"]) == bp["num_inliers"][i]
        assert float(m["inlier_ratio"]) == bp["inlier_ratio"][i]
        assert str(m["reprojection_error"]) == str(bp["reprojection_error"][i]), name    # float32 repr, exact
        assert bool(m["well_distributed"]) == bool(bp["well_distributed"][i])
        assert do.verify_match_quality(res)          # every shipped pair passed the quality gate (:286)


def test_epilines_are_unit_normal_float32():
    rng = np.random.default_rng(3)
    F = rng.normal(size=(3, 3))
    p = (rng.random((50, 2)) * 1000).astype(np.float32)
    for which in (1, 2):
        l = do.epilines(p, which, F)
        assert l.dtype == np.float32
        assert np.allclose(np.hypot(l[:, 0], l[:, 1]), 1.0, atol=1e-6)
        M = F if which == 1 else F.T
        ref = (M @ np.c_[p.astype(np.float64), np.ones(50)].T).T
        ref /= np.hypot(ref[:, 0], ref[:, 1])[:, None]
        assert np.allclose(l, ref, rtol=1e-5, atol=1e-5)


def test_dlt_reproduces_shipped_triangulated_points():
    """Every point after the initial pair in bunny_data/reconstruction/points3D.json came out of
    triangulate_point (sfm_reconstruction.py:263-307) on its shipped two-view track: the restated DLT must
    reproduce it and it must pass the 4 px gate.  (The first 229 are the float32 initial-pair batch, :138.)"""
    b = load("bunny_state.npz")
    poses, pts, tracks = bunny_tracks(b)
    worst = 0.0
    for j in range(229, len(tracks)):
        tr = tracks[j]
        assert len(tr) == 2
        Ps = [do.projection(K_BUNNY, *poses[i]) for i in tr]
        X = do.triangulate_point(Ps, [tr[i] for i in tr])
        assert X is not None
        worst = max(worst, np.linalg.norm(X - pts[j]) / np.linalg.norm(pts[j]))
    assert worst < 1e-9, worst
    for j in range(0, 229, 7):
        tr = tracks[j]
        Ps = [do.projection(K_BUNNY, *poses[i]) for i in tr]
        X = do.triangulate_dlt(Ps[0], Ps[1], [tr[i] for i in tr][0], [tr[i] for i in tr][1])[0]
        X = X[:3] / X[3]
        assert np.linalg.norm(X - pts[j]) / np.linalg.norm(pts[j]) < 5e-6


def test_triangulate_gate_rejects_and_nan_passes():
    P0 = do.projection(K_BUNNY, np.eye(3), np.zeros(3))
    P1 = do.projection(K_BUNNY, np.eye(3), np.array([-1.0, 0, 0]))
    X = np.array([0.2, -0.1, 5.0, 1.0])
    x0 = (P0 @ X)[:2] / (P0 @ X)[2]; x1 = (P1 @ X)[:2] / (P1 @ X)[2]
    got = do.triangulate_point([P0, P1], [x0, x1])
    assert np.allclose(got, X[:3], rtol=1e-9)
    assert do.triangulate_point([P0, P1], [x0, x1 + [0, 30.0]]) is None        # epipolar-inconsistent: > 4 px


def test_associate_threshold_is_strict_and_order_is_row_major():
    t = np.array([[10.0, 10.0], [512.0, 0.0], [10.0, 12.0]])
    c = np.array([[10.0, 12.0], [10.0, 10.0], [512.0, np.nextafter(2.0, 0.0)], [10.0, 11.0]],
                 dtype=np.float64)
    rows, cols = do.associate(t, c)
    # row 0: c0 at distance exactly 2.0 is NOT < 2.0; c1 (0) and c3 (1) are.  row 1: c2 just below 2.
    assert rows.tolist() == [0, 0, 1, 2, 2] and cols.tolist() == [1, 3, 2, 0, 3]


def test_find_2d3d_matches_equals_reference_run():
    g = load("driver_bunny.npz")
    b = load("bunny_state.npz")
    corr = pair_corr(load("bunny_pairs.npz"))
    poses, pts, tracks = bunny_tracks(b)
    for img in g["f_images"]:
        pairs = [(str(n),) + corr[str(n)] for n in g[f"f{img}_pairs"]]
        p3, p2 = do.find_2d3d_matches(pts, tracks, pairs, int(img))
        assert np.array_equal(p3, g[f"f{img}_points3D"])
        assert p2.dtype == g[f"f{img}_points2D"].dtype and np.array_equal(p2, g[f"f{img}_points2D"])


def test_add_new_matches_equals_reference_replay():
    g = load("driver_bunny.npz")
    b = load("bunny_state.npz")
    corr = pair_corr(load("bunny_pairs.npz"))
    n0 = int(g["a_n0"])
    poses, pts, tracks = bunny_tracks(b, n0)
    added = [do.add_new_matches(pts, tracks, poses, K_BUNNY, str(p), *corr[str(p)]) for p in g["a_pairs"]]
    assert added == g["a_added"].tolist()
    assert np.array_equal(np.asarray(pts[n0:]), g["a_points3D"])
    assert np.array_equal(np.asarray([list(t.keys()) for t in tracks[n0:]]), g["a_track_ids"])
    assert np.array_equal(np.asarray([list(t.values()) for t in tracks[n0:]]), g["a_track_uv"])
