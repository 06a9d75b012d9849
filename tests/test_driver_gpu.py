"""GPU parity tests of the driver rows (sfm_amd.driver -> libsfm_amd.so) against the CPU oracle, the
reference-generated goldens and the outputs the reference ships.  Index / mask / float32 results are
bit-exact; triangulated points agree to 1e-9 relative (fp64 SVD by a different iteration order)."""
import logging
import os

import numpy as np
import pytest

from test_driver_oracle import K_BUNNY, bunny_tracks, load, pair_corr

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------ association
def pixels(rng, n, dup_from=None, frac=0.5, jitter=1.5):
    """float32 pixel coordinates; a fraction re-uses rows of `dup_from` exactly or within a few px."""
    p = (rng.random((n, 2)) * [1024, 768]).astype(np.float32)
    if dup_from is not None and len(dup_from):
        k = int(n * frac)
        src = dup_from[rng.integers(0, len(dup_from), k)]
        noise = (rng.normal(size=(k, 2)) * jitter).astype(np.float32)
        noise[: k // 2] = 0
        p[rng.permutation(n)[:k]] = src + noise
    return p


@pytest.mark.parametrize("T,M,seed", [(1, 1, 0), (7, 300, 1), (256, 256, 2), (257, 513, 3), (3000, 229, 4), (5000, 4000, 5)])
def test_associate_matches_oracle(gpu_ready, T, M, seed):
    from oracle import driver_oracle as do
    from sfm_amd import driver
    rng = np.random.default_rng(seed)
    c = pixels(rng, M)
    t = pixels(rng, T, c).astype(np.float64)
    rows, cols = driver.associate(t, c)
    r0, c0 = do.associate(t, c)
    assert np.array_equal(rows, r0) and np.array_equal(cols, c0)
    assert rows.dtype == np.int64


def test_associate_threshold_edge_cases(gpu_ready):
    from oracle import driver_oracle as do
    from sfm_amd import driver
    t = np.array([[10.0, 10.0], [512.0, 0.0], [10.0, 12.0], [np.nan, 1.0], [1.2, 1.6]])
    c = np.array([[10.0, 12.0], [10.0, 10.0], [512.0, np.nextafter(2.0, 0.0)], [10.0, 11.0], [0.0, 0.0],
                  [np.inf, 0.0]])
    rows, cols = driver.associate(t, c)
    with np.errstate(invalid="ignore"):
        r0, c0 = do.associate(t, c)
    assert rows.tolist() == r0.tolist() and cols.tolist() == c0.tolist()
    assert (0, 0) not in set(zip(rows.tolist(), cols.tolist()))        # distance exactly 2.0 is not < 2.0
    for thr in (0.0, 1e-300, 2.0000000000000004, 1e6):
        rows, cols = driver.associate(t[:3], c[:4], thr)
        r0, c0 = do.associate(t[:3], c[:4], thr)
        assert rows.tolist() == r0.tolist() and cols.tolist() == c0.tolist()


def test_associate_segments_ragged_and_empty(gpu_ready):
    """Several image pairs in one launch: empty segments, segments crossing 256-row blocks, and a dense
    segment whose hit count overflows the first output capacity (second pass)."""
    from oracle import driver_oracle as do
    from sfm_amd import driver
    rng = np.random.default_rng(11)
    sizes = [(0, 10), (300, 0), (5, 7), (700, 90), (0, 0), (255, 1), (1, 255), (40, 40)]
    tl, cl = [], []
    for T, M in sizes:
        c = pixels(rng, M)
        tl.append(pixels(rng, T, c).astype(np.float64)); cl.append(c)
    tl.append(np.tile([[100.0, 100.0]], (600, 1))); cl.append(np.tile(np.float32([[100.5, 100.5]]), (500, 1)))
    got = driver.associate_segments(tl, cl)
    assert len(got) == len(tl)
    for (rows, cols), t, c in zip(got, tl, cl):
        if len(t) and len(c):
            r0, c0 = do.associate(t, c)
        else:
            r0 = c0 = np.zeros(0, np.int64)
        assert np.array_equal(rows, r0) and np.array_equal(cols, c0)
    assert got[-1][0].size == 600 * 500


def test_associate_random_segment_layouts(gpu_ready):
    """40 random launches: 1-12 segments of 0-700 tracks x 0-400 correspondences each (empty ones included), so
    segment boundaries fall anywhere inside the 256-row blocks and the column splits."""
    from oracle import driver_oracle as do
    from sfm_amd import driver
    rng = np.random.default_rng(2024)
    for trial in range(40):
        n_seg = int(rng.integers(1, 13))
        tl, cl = [], []
        for _ in range(n_seg):
            T = int(rng.integers(0, 701)) if rng.random() > 0.15 else 0
            M = int(rng.integers(0, 401)) if rng.random() > 0.15 else 0
            c = pixels(rng, M)
            tl.append(pixels(rng, T, c).astype(np.float64)); cl.append(c)
        got = driver.associate_segments(tl, cl)
        for (rows, cols), t, c in zip(got, tl, cl):
            if len(t) and len(c):
                r0, c0 = do.associate(t, c)
            else:
                r0 = c0 = np.zeros(0, np.int64)
            assert np.array_equal(rows, r0) and np.array_equal(cols, c0), (trial, len(t), len(c))


def test_associate_large_properties(gpu_ready):
    """100,000 tracks x 20,000 correspondences (2e9 pair tests): transposing the problem gives the
    transposed pair set, and sampled rows equal the oracle."""
    from oracle import driver_oracle as do
    from sfm_amd import driver
    rng = np.random.default_rng(5)
    c = pixels(rng, 20000)
    t = pixels(rng, 100000, c, 0.4).astype(np.float64)
    rows, cols = driver.associate(t, c)
    assert rows.size > 30000
    assert np.all(np.diff(rows) >= 0) and np.all((np.diff(cols) > 0) | (np.diff(rows) > 0))     # np.where order
    r2, c2 = driver.associate(c.astype(np.float64), t)
    a = np.stack([rows, cols], 1); b = np.stack([c2, r2], 1)
    b = b[np.lexsort((b[:, 1], b[:, 0]))]
    assert np.array_equal(a, b)
    pick = np.sort(rng.choice(len(t), 300, replace=False))
    r0, c0 = do.associate(t[pick], c)
    sel = np.isin(rows, pick)
    assert np.array_equal(np.searchsorted(pick, rows[sel]), r0) and np.array_equal(cols[sel], c0)


def _Sfm(tmp_path, corr, n_tracks=None):
    """The reference-side state find_2d3d_matches / add_new_matches read, on files written from fixtures."""
    from sfm_amd.reconstruction import StructureFromMotion
    b = load("bunny_state.npz")
    s = StructureFromMotion(tmp_path)
    for d in (s.matches_dir, s.corr_dir):
        d.mkdir(parents=True, exist_ok=True)
    for name, (p1, p2) in corr.items():
        np.save(s.corr_dir / f"{name}_pts1.npy", p1); np.save(s.corr_dir / f"{name}_pts2.npy", p2)
        np.savez(s.matches_dir / f"{name}_matches.npz", inlier_mask=np.ones(len(p1), bool))
    s.poses, s.points3D, s.point_tracks = bunny_tracks(b, n_tracks)
    s.constructed = [f"{int(i):04d}.ppm" for i in b["ids"]]
    return s


def test_find_2d3d_matches_equals_reference_run(gpu_ready, tmp_path, caplog):
    """Drop-in method on the shipped state + correspondence files == the reference's own run (golden)."""
    g = load("driver_bunny.npz")
    corr = pair_corr(load("bunny_pairs.npz"))
    s = _Sfm(tmp_path, corr)
    for img in g["f_images"]:
        order = [str(n) for n in g[f"f{img}_pairs"]]
        s.constructed = [f"{int(i):04d}.ppm" for i in load("bunny_state.npz")["ids"] if int(i) != int(img)]
        assert sorted(s.find_image_pairs(int(img))) == sorted(order)
        s.find_image_pairs = lambda image_id, order=order: order          # directory order of the golden run
        with caplog.at_level(logging.INFO):
            caplog.clear()
            p3, p2 = s.find_2d3d_matches(int(img))
        del s.find_image_pairs
        assert np.array_equal(p3, g[f"f{img}_points3D"])
        assert p2.dtype == np.float32 and np.array_equal(p2, g[f"f{img}_points2D"])
        msgs = [r.getMessage() for r in caplog.records]
        assert msgs == [f"Found {len(order)} pairs for image {img}", f"Found {len(p3)} 2D-3D matches for image {img}"]


def test_find_2d3d_matches_edge_cases(gpu_ready, tmp_path, caplog):
    corr = pair_corr(load("bunny_pairs.npz"))
    s = _Sfm(tmp_path, {k: corr[k] for k in ("pair_3_4", "pair_1_3")})
    s.find_image_pairs = lambda image_id: ["pair_3_4", "pair_3_99", "pair_1_3"]     # one file is missing
    with caplog.at_level(logging.WARNING):
        p3, p2 = s.find_2d3d_matches(3)
    assert any("Failed to process pair pair_3_99" in r.getMessage() for r in caplog.records)
    assert len(p3) == len(p2) > 0
    s.point_tracks, s.points3D = [], []                                             # nothing reconstructed yet
    p3, p2 = s.find_2d3d_matches(3)
    assert p3.shape == (0,) and p2.shape == (0,)


# ---------------------------------------------------------------------------------------- triangulation
def test_triangulate_matches_oracle_and_shipped_points(gpu_ready):
    from oracle import driver_oracle as do
    from sfm_amd import driver
    b = load("bunny_state.npz")
    poses, pts, tracks = bunny_tracks(b)
    ids = list(poses)
    proj = np.stack([driver.projection_matrix(K_BUNNY, *poses[i]) for i in ids])
    pos = {k: i for i, k in enumerate(ids)}
    js = range(229, len(tracks))
    c0 = [pos[list(tracks[j])[0]] for j in js]; c1 = [pos[list(tracks[j])[1]] for j in js]
    x0 = [list(tracks[j].values())[0] for j in js]; x1 = [list(tracks[j].values())[1] for j in js]
    X, valid, err = driver.triangulate_two_view(proj, c0, c1, x0, x1)
    assert valid.all() and err.max() < 4.0
    ship = np.asarray(pts[229:])
    assert np.max(np.linalg.norm(X - ship, axis=1) / np.linalg.norm(ship, axis=1)) < 1e-9
    for k in range(0, len(c0), 97):
        ref = do.triangulate_point([proj[c0[k]], proj[c1[k]]], [x0[k], x1[k]])
        assert np.linalg.norm(X[k] - ref) / np.linalg.norm(ref) < 1e-10


def test_triangulate_gate_and_degenerate_inputs(gpu_ready):
    from oracle import driver_oracle as do
    from sfm_amd import driver
    rng = np.random.default_rng(2)
    P0 = do.projection(K_BUNNY, np.eye(3), np.zeros(3))
    P1 = do.projection(K_BUNNY, np.eye(3), np.array([-1.0, 0, 0]))
    n = 2000
    Xt = np.c_[rng.uniform(-1, 1, (n, 2)), rng.uniform(3, 8, n), np.ones(n)]
    x0 = (Xt @ P0.T); x0 = x0[:, :2] / x0[:, 2:]
    x1 = (Xt @ P1.T); x1 = x1[:, :2] / x1[:, 2:]
    x1[:, 1] += rng.uniform(-12, 12, n)                       # vertical disparity -> some fail the 4 px gate
    X, valid, err = driver.triangulate_two_view([P0, P1], np.zeros(n, int), np.ones(n, int), x0, x1)
    ref_valid = np.zeros(n, bool); ref_X = np.zeros((n, 3))
    for k in range(n):
        r = do.triangulate_point([P0, P1], [x0[k], x1[k]])
        ref_valid[k] = r is not None
        X4 = do.triangulate_dlt(P0, P1, x0[k], x1[k])[0]; ref_X[k] = X4[:3] / X4[3]
    near = np.abs(err - 4.0).min(axis=1) < 1e-9               # knife-edge cases may legitimately differ
    assert np.array_equal(valid[~near], ref_valid[~near]) and 0.2 < valid.mean() < 0.8
    assert np.max(np.linalg.norm(X - ref_X, axis=1) / np.linalg.norm(ref_X, axis=1)) < 1e-9
    # identical cameras: no unique null vector; whatever comes out must not crash and NaN passes the gate
    X, valid, err = driver.triangulate_two_view([P0, P0], [0], [1], [[500.0, 300.0]], [[500.0, 300.0]])
    assert X.shape == (1, 3)
    X, valid, err = driver.triangulate_two_view(np.zeros((2, 3, 4)), [0], [1], [[1.0, 2.0]], [[3.0, 4.0]])
    assert not np.isfinite(X).any() and valid[0]              # `nan > 4.0` is False in the reference too
    X, valid, err = driver.triangulate_two_view([P0, P1], [], [], np.zeros((0, 2)), np.zeros((0, 2)))
    assert X.shape == (0, 3) and valid.shape == (0,)
    with pytest.raises(ValueError):
        driver.triangulate_two_view([P0, P1], [0], [2], [[1.0, 2.0]], [[3.0, 4.0]])


def test_add_new_matches_equals_reference_replay(gpu_ready, tmp_path, caplog):
    g = load("driver_bunny.npz")
    corr = pair_corr(load("bunny_pairs.npz"))
    n0 = int(g["a_n0"])
    s = _Sfm(tmp_path, {str(p): corr[str(p)] for p in g["a_pairs"]}, n0)
    added = []
    with caplog.at_level(logging.INFO):
        for p in g["a_pairs"]:
            before = len(s.points3D)
            assert s.add_new_matches(str(p), int(str(p).split("_")[2])) is True
            added.append(len(s.points3D) - before)
    assert added == g["a_added"].tolist()
    msgs = [r.getMessage() for r in caplog.records]
    assert msgs == [f"Added {a} new tracks from pair {p}" for a, p in zip(added, g["a_pairs"])]
    new_tracks = s.point_tracks[n0:]
    assert np.array_equal(np.asarray([list(t.keys()) for t in new_tracks]), g["a_track_ids"])
    assert np.array_equal(np.asarray([list(t.values()) for t in new_tracks]), g["a_track_uv"])
    got, ref = np.asarray(s.points3D[n0:]), g["a_points3D"]
    assert np.max(np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)) < 1e-9
    assert isinstance(s.points3D[-1], np.ndarray) and isinstance(new_tracks[-1][int(g["a_track_ids"][-1][0])], list)
    # replaying a pair adds nothing (every correspondence now exists in a track) and still returns True
    n = len(s.points3D)
    with caplog.at_level(logging.WARNING):
        caplog.clear()
        assert s.add_new_matches(str(g["a_pairs"][0]), 4) is True
    assert len(s.points3D) == n and "No valid tracks found" in caplog.records[-1].getMessage()
    assert s.add_new_matches("pair_98_99", 99) is False          # missing files -> False, as the reference
    one = s.triangulate_point(new_tracks[0])
    assert np.linalg.norm(one - ref[0]) / np.linalg.norm(ref[0]) < 1e-9
    assert s.triangulate_point({3: [1.0, 2.0]}) is None


# ------------------------------------------------------------------------------- epipolar verification
def test_geometric_verification_148_shipped_pairs_one_launch(gpu_ready):
    from oracle import driver_oracle as do
    from sfm_amd import driver
    bp = load("bunny_pairs.npz")
    off = bp["offsets"]
    pairs = [(bp["pts1"][off[i]:off[i + 1]], bp["pts2"][off[i]:off[i + 1]], bp["F"][i]) for i in range(len(off) - 1)]
    res = driver.verify_pairs(pairs)
    for i, (r, (p1, p2, F)) in enumerate(zip(res, pairs)):
        assert r["symmetric_errors"].dtype == np.float32
        assert np.array_equal(r["symmetric_errors"], do.symmetric_epipolar_errors(p1, p2, F))       # bit-exact
        assert np.array_equal(r["inlier_mask"], bp["mask"][off[i]:off[i + 1]])                       # shipped mask
        m = r["metrics"]
        assert int(m["inliers"]) == bp["num_inliers"][i] and m["total_matches"] == bp["num_matches"][i]
        assert str(m["reprojection_error"]) == str(bp["reprojection_error"][i])
        assert float(m["inlier_ratio"]) == bp["inlier_ratio"][i]
        assert bool(m["well_distributed"]) == bool(bp["well_distributed"][i])


def test_geometric_verification_method_and_edge_cases(gpu_ready):
    from oracle import driver_oracle as do
    from sfm_amd.matcher import ImageMatcher
    im = ImageMatcher()
    rng = np.random.default_rng(9)
    F = rng.normal(size=(3, 3)) * [1e-6, 1e-6, 1e-3]
    p1 = (rng.random((5000, 2)) * 1000).astype(np.float32); p2 = (rng.random((5000, 2)) * 1000).astype(np.float32)
    got = im.geometric_verification(p1, p2, F)
    ref = do.geometric_verification(p1, p2, F)
    assert np.array_equal(got["symmetric_errors"], ref["symmetric_errors"])
    assert np.array_equal(got["inlier_mask"], ref["inlier_mask"])
    for k in ref["metrics"]:
        assert str(got["metrics"][k]) == str(ref["metrics"][k]), k
    assert im.verify_match_quality(got) == do.verify_match_quality(ref)
    # F = 0: nu == 0 branch of computeCorrespondEpilines -> 0/0 = NaN errors, nothing is an inlier
    with np.errstate(all="ignore"):
        got = im.geometric_verification(p1[:10], p2[:10], np.zeros((3, 3)))
    assert np.isnan(got["symmetric_errors"]).all() and not got["inlier_mask"].any()
    assert got["metrics"]["reprojection_error"] == float("inf") and got["metrics"]["well_distributed"] is False
    assert im.verify_match_quality(got) is False
