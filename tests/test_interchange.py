"""Interchange writers/readers (sfm_amd.interchange) against the artefacts the reference ships: the state in
tests/golden/bunny_state.npz is exactly what bunny_data/reconstruction/*.json hold, so re-writing it must give
byte-identical files (digests in tests/golden/bunny_artifacts.json)."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_driver_oracle import bunny_tracks, load
from sfm_amd import interchange as ix


def digest(path):
    b = open(path, "rb").read()
    return {"sha256": hashlib.sha256(b).hexdigest(), "bytes": len(b)}


@pytest.fixture(scope="module")
def art():
    return json.load(open(os.path.join(GOLDEN, "bunny_artifacts.json")))


@pytest.fixture()
def shipped_state():
    poses, pts, tracks = bunny_tracks(load("bunny_state.npz"))
    poses = {k: (R, t.reshape(3, 1)) for k, (R, t) in poses.items()}
    return poses, pts, tracks


def test_reconstruction_files_are_byte_identical_to_shipped(tmp_path, art, shipped_state):
    ix.save_reconstruction(*shipped_state, tmp_path / "reconstruction")
    for name in ("poses.json", "points3D.json", "reconstruction.ply"):
        assert digest(tmp_path / "reconstruction" / name) == art["files"][f"reconstruction/{name}"], name


def test_colmap_export_is_byte_identical_to_shipped(tmp_path, art, shipped_state):
    ix.save_reconstruction(*shipped_state, tmp_path / "reconstruction")
    ex = ix.SfMExporter(tmp_path / "reconstruction")
    ex.export_all(tmp_path / "exports")
    for name in ("cameras.txt", "images.txt", "points3D.txt"):
        assert digest(tmp_path / "exports" / "colmap" / name) == art["files"][f"exports/colmap/{name}"], name
    import sqlite3
    con = sqlite3.connect(tmp_path / "exports" / "colmap" / "database.db")
    row = con.execute("SELECT camera_id, model, width, height, params FROM cameras").fetchall()
    con.close()
    assert len(row) == 1 and row[0][:4] == (1, 1, 1024, 768)
    assert np.frombuffer(row[0][4], np.float64).tolist() == list(ix.COLMAP_CAMERA_PARAMS)


def test_load_reconstruction_round_trip(tmp_path, shipped_state):
    poses, pts, tracks = shipped_state
    ix.save_reconstruction(poses, [np.asarray(p) for p in pts], tracks, tmp_path / "r")     # ndarray points too
    p2, pts2, tr2 = ix.load_reconstruction(tmp_path / "r")
    assert list(p2) == list(poses) and pts2 == pts and tr2 == tracks
    for k in poses:
        assert np.array_equal(p2[k][0], poses[k][0]) and p2[k][1].shape == (3, 1)
        assert np.array_equal(p2[k][1], poses[k][1])
    with pytest.raises(ValueError):
        ix.SfMExporter(tmp_path / "missing")


def test_exporter_drops_single_view_points(tmp_path, shipped_state):
    poses, pts, tracks = shipped_state
    tracks = [dict(t) for t in tracks[:10]]
    tracks[3] = {25: tracks[3][25]}
    ix.save_reconstruction(poses, pts[:10], tracks, tmp_path / "r")
    ex = ix.SfMExporter(tmp_path / "r")
    assert len(ex.points3D) == 9 and len(ex.tracks) == 9


def test_pair_files_layout_and_round_trip(tmp_path, art):
    from sfm_amd.matcher import DMatch
    bp = load("bunny_pairs.npz")
    bm = load("bunny_matches.npz")
    i = list(bp["names"]).index("pair_10_11")
    sl = slice(bp["offsets"][i], bp["offsets"][i + 1])
    j = list(bm["names"]).index("pair_10_11_matches.npz")
    ms = slice(bm["offsets"][j], bm["offsets"][j + 1])
    matches = [DMatch(q, t, d) for q, t, d in zip(bm["queryIdx"][ms], bm["trainIdx"][ms], bm["distance"][ms])]
    ix.save_pair_data(tmp_path, "pair_10_11", bp["pts1"][sl], bp["pts2"][sl], bp["F"][i], bp["mask"][sl], matches)
    lay = art["pair_layout"]
    for rel in ("correspondences/pair_10_11_pts1.npy", "correspondences/pair_10_11_pts2.npy"):
        a = np.load(tmp_path / rel, allow_pickle=False)
        assert {"dtype": str(a.dtype), "ndim": a.ndim, "cols": a.shape[1]} == lay[rel]
    for rel in ("fundamental/pair_10_11_F.npz", "matches/pair_10_11_matches.npz"):
        z = np.load(tmp_path / rel, allow_pickle=False)
        assert {k: {"dtype": str(z[k].dtype), "ndim": z[k].ndim} for k in z.files} == lay[rel]
    got = ix.load_pair_data(tmp_path, "pair_10_11")
    assert np.array_equal(got["corr_pts1"], bp["pts1"][sl][bp["mask"][sl]])
    assert np.array_equal(got["F"], bp["F"][i]) and np.array_equal(got["inlier_mask"], bp["mask"][sl])
    assert np.array_equal(got["queryIdx"], bm["queryIdx"][ms]) and np.array_equal(got["distance"], bm["distance"][ms])


def test_quaternion_branches_are_rotations():
    from sfm_amd.rotation import rodrigues
    rng = np.random.default_rng(0)
    cases = [np.eye(3), np.diag([1.0, -1.0, -1.0]), np.diag([-1.0, 1.0, -1.0]), np.diag([-1.0, -1.0, 1.0])]
    cases += [rodrigues(rng.normal(size=3) * s) for s in (0.1, 1.0, 2.5, 3.1) for _ in range(5)]
    for R in cases:
        w, x, y, z = ix.rotation_to_quaternion(R)
        assert abs(w * w + x * x + y * y + z * z - 1) < 1e-12
        Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        assert np.allclose(Rq, R, atol=1e-12)
