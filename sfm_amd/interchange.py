"""On-disk interchange with the reference's artefacts (host-side, no GPU work).

Writers produce byte-for-byte what the reference's writers produce, so results can be diffed against
`bunny_data/`; readers accept those files.  Reference code these mirror:

* pair files            save_pair_data            /root/reference/utils/find_matches.py:312-327
* poses/points JSON+PLY save_reconstruction/_ply  /root/reference/utils/sfm_reconstruction.py:711-767
* COLMAP text + db      SfMExporter               /root/reference/utils/export.py:9-187

tests/test_interchange.py regenerates the five artefacts the reference ships under
bunny_data/{reconstruction,exports/colmap}/ from the shipped state and compares SHA-256 digests.
"""
from __future__ import annotations

import json
import logging
import sqlite3
from pathlib import Path

import numpy as np

COLMAP_CAMERA_LINE = "1 PINHOLE 1024 768 2393.95 2398.12 932.38 628.26"     # export.py:59 (literal)
COLMAP_CAMERA_PARAMS = (2393.95, 2398.12, 932.38, 628.26)                    # export.py:175


# ------------------------------------------------------------------------------------------- pair files
def save_pair_data(data_dir, pair_name, pts1, pts2, F, inlier_mask, matches):
    """Three files per verified pair: inlier correspondences (.npy), F + mask + all matched points (.npz),
    match indices/distances (.npz).  `matches`: objects with queryIdx / trainIdx / distance."""
    data_dir = Path(data_dir)
    corr, fund, mdir = data_dir / 'correspondences', data_dir / 'fundamental', data_dir / 'matches'
    for d in (corr, fund, mdir):
        d.mkdir(parents=True, exist_ok=True)
    inlier_mask = np.asarray(inlier_mask)
    np.save(corr / f'{pair_name}_pts1.npy', pts1[inlier_mask])
    np.save(corr / f'{pair_name}_pts2.npy', pts2[inlier_mask])
    np.savez(fund / f'{pair_name}_F.npz', F=F, mask=inlier_mask, pts1=pts1, pts2=pts2)
    if hasattr(matches, "queryIdx") and len(matches) > 0:
        # a DMatchList (sfm_amd.matcher): the arrays are there already; dtypes as np.array() of Python ints / floats gives them
        qi, ti, di = (np.asarray(matches.queryIdx, dtype=np.int64), np.asarray(matches.trainIdx, dtype=np.int64),
                      np.asarray(matches.distance, dtype=np.float64))
    else:
        qi, ti, di = (np.array([m.queryIdx for m in matches]), np.array([m.trainIdx for m in matches]),
                      np.array([m.distance for m in matches]))
    np.savez(mdir / f'{pair_name}_matches.npz', queryIdx=qi, trainIdx=ti, distance=di, inlier_mask=inlier_mask)


def load_pair_data(data_dir, pair_name):
    """Everything save_pair_data wrote for one pair (plain arrays; nothing is unpickled)."""
    data_dir = Path(data_dir)
    out = {'corr_pts1': np.load(data_dir / 'correspondences' / f'{pair_name}_pts1.npy', allow_pickle=False),
           'corr_pts2': np.load(data_dir / 'correspondences' / f'{pair_name}_pts2.npy', allow_pickle=False)}
    with np.load(data_dir / 'fundamental' / f'{pair_name}_F.npz', allow_pickle=False) as z:
        out.update(F=z['F'], mask=z['mask'], pts1=z['pts1'], pts2=z['pts2'])
    with np.load(data_dir / 'matches' / f'{pair_name}_matches.npz', allow_pickle=False) as z:
        out.update(queryIdx=z['queryIdx'], trainIdx=z['trainIdx'], distance=z['distance'],
                   inlier_mask=z['inlier_mask'])
    return out


# ---------------------------------------------------------------------------------------- reconstruction
def _plain(v):
    return v.tolist() if isinstance(v, np.ndarray) else v


def save_ply(points3D, filepath):
    pts = np.array(points3D)
    head = ["ply", "format ascii 1.0", f"element vertex {len(pts)}",
            "property float x", "property float y", "property float z", "end_header"]
    with open(filepath, 'w') as f:
        f.write("\n".join(head) + "\n")
        f.writelines(f"{p[0]} {p[1]} {p[2]}\n" for p in pts)


def save_reconstruction(poses, points3D, point_tracks, output_dir):
    """poses.json, points3D.json (points + tracks, image ids as strings) and reconstruction.ply."""
    output_dir = Path(output_dir)
    output_dir.mkdir(exist_ok=True)
    pose_json = {str(k): {'R': np.asarray(R).tolist(), 't': np.asarray(t).ravel().tolist()}
                 for k, (R, t) in poses.items()}
    with open(output_dir / 'poses.json', 'w') as f:
        json.dump(pose_json, f, indent=2)
    body = {'points3D': [_plain(p) for p in points3D],
            'tracks': [{str(k): _plain(v) for k, v in tr.items()} for tr in point_tracks]}
    with open(output_dir / 'points3D.json', 'w') as f:
        json.dump(body, f, indent=2)
    save_ply(points3D, output_dir / 'reconstruction.ply')
    logging.info(f"Saved reconstruction to {output_dir}")


def load_reconstruction(recon_dir, int_keys=True):
    """(poses, points3D, point_tracks) from poses.json / points3D.json.  int_keys=True restores the in-memory
    form the driver uses (image ids as ints, R as arrays, t as (3,1)); False keeps the JSON strings."""
    recon_dir = Path(recon_dir)
    with open(recon_dir / 'poses.json') as f:
        raw = json.load(f)
    with open(recon_dir / 'points3D.json') as f:
        body = json.load(f)
    key = int if int_keys else str
    poses = {key(k): (np.asarray(v['R'], dtype=np.float64), np.asarray(v['t'], dtype=np.float64).reshape(3, 1))
             for k, v in raw.items()}
    tracks = [{key(k): p for k, p in tr.items()} for tr in body['tracks']]
    return poses, body['points3D'], tracks


class ReconstructionIOMixin:
    """save_reconstruction / save_ply with the reference's method signatures."""

    def save_reconstruction(self, output_dir):
        save_reconstruction(self.poses, self.points3D, self.point_tracks, output_dir)

    def save_ply(self, filepath):
        save_ply(self.points3D, filepath)


# ------------------------------------------------------------------------------------------------ COLMAP
def rotation_to_quaternion(R):
    """(qw, qx, qy, qz) by the largest-pivot branch rule of export.py:125-151."""
    R = np.asarray(R)
    d = (R[0, 0], R[1, 1], R[2, 2])
    tr = np.trace(R)
    if tr > 0:
        S = np.sqrt(tr + 1.0) * 2
        return 0.25 * S, (R[2, 1] - R[1, 2]) / S, (R[0, 2] - R[2, 0]) / S, (R[1, 0] - R[0, 1]) / S
    if d[0] > d[1] and d[0] > d[2]:
        S = np.sqrt(1.0 + d[0] - d[1] - d[2]) * 2
        return (R[2, 1] - R[1, 2]) / S, 0.25 * S, (R[0, 1] + R[1, 0]) / S, (R[0, 2] + R[2, 0]) / S
    if d[1] > d[2]:
        S = np.sqrt(1.0 + d[1] - d[0] - d[2]) * 2
        return (R[0, 2] - R[2, 0]) / S, (R[0, 1] + R[1, 0]) / S, 0.25 * S, (R[1, 2] + R[2, 1]) / S
    S = np.sqrt(1.0 + d[2] - d[0] - d[1]) * 2
    return (R[1, 0] - R[0, 1]) / S, (R[0, 2] + R[2, 0]) / S, (R[1, 2] + R[2, 1]) / S, 0.25 * S


class SfMExporter:
    """COLMAP text export of a saved reconstruction (export.py:9-187).  Points with fewer than two
    observations are dropped on load, as the reference does."""

    def __init__(self, reconstruction_dir):
        self.recon_dir = Path(reconstruction_dir)
        try:
            self.poses, pts, tracks = load_reconstruction(self.recon_dir, int_keys=False)
        except FileNotFoundError as e:
            raise ValueError(f"Failed to load reconstruction data: {e}")
        self.poses = {k: {'R': R.tolist(), 't': t.ravel().tolist()} for k, (R, t) in self.poses.items()}
        keep = [i for i, tr in enumerate(tracks) if len(tr) >= 2]
        self.points3D = [pts[i] for i in keep]
        self.tracks = [tracks[i] for i in keep]
        logging.info(f"Loaded poses for {len(self.poses)} images, {len(self.points3D)} valid points")

    def export_colmap(self, output_dir):
        output_dir = Path(output_dir)
        output_dir.mkdir(exist_ok=True)
        with open(output_dir / 'cameras.txt', 'w') as f:
            f.write("# Camera list with one line of data per camera:\n"
                    "#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n" + COLMAP_CAMERA_LINE + "\n")
        per_image = {}                       # image id -> ["x y point_id", ...] in point order
        for j, tr in enumerate(self.tracks):
            for img, (x, y) in tr.items():
                per_image.setdefault(img, []).append(f"{x} {y} {j + 1}")
        refs = 0
        with open(output_dir / 'images.txt', 'w') as f:
            f.write("# Image list with two lines of data per image:\n"
                    "#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n"
                    "#   POINTS2D[] as (X, Y, POINT3D_ID)\n")
            for img, pose in self.poses.items():
                qw, qx, qy, qz = rotation_to_quaternion(np.array(pose['R']))
                t = np.array(pose['t']).reshape(3)
                obs = per_image.get(str(img), [])
                f.write(f"{img} {qw} {qx} {qy} {qz} {t[0]} {t[1]} {t[2]} 1 {int(img):08d}.jpg\n")
                f.write(" ".join(obs) + "\n")
                refs += len(obs)
        logging.info(f"Total point references in images.txt: {refs}")
        written = 0
        with open(output_dir / 'points3D.txt', 'w') as f:
            f.write("# 3D point list with one line of data per point:\n"
                    "#   POINT3D_ID, X, Y, Z, R, G, B, ERROR, TRACK[] as (IMAGE_ID, POINT2D_IDX)\n")
            for j, ((x, y, z), tr) in enumerate(zip(self.points3D, self.tracks)):
                if len(tr) < 2:
                    continue
                views = " ".join(f"{k} 0" for k in sorted(tr.keys()))       # string order, as the reference
                f.write(f"{j + 1} {x} {y} {z} 255 255 255 1.0 {views}\n")
                written += 1
        logging.info(f"Wrote {written} points to points3D.txt")

    def _create_colmap_database(self, db_path):
        """Empty COLMAP database holding the one camera row (export.py:153-187)."""
        db_path = Path(db_path)
        if db_path.exists():
            db_path.unlink()
        conn = sqlite3.connect(db_path)
        try:
            cur = conn.cursor()
            cur.execute("CREATE TABLE cameras (camera_id INTEGER PRIMARY KEY, model INTEGER, width INTEGER, "
                        "height INTEGER, params BLOB)")
            cur.execute("CREATE TABLE images (image_id INTEGER PRIMARY KEY, name TEXT, camera_id INTEGER, "
                        "prior_qw REAL, prior_qx REAL, prior_qy REAL, prior_qz REAL, prior_tx REAL, "
                        "prior_ty REAL, prior_tz REAL)")
            cur.execute("INSERT INTO cameras VALUES (?, ?, ?, ?, ?)",
                        (1, 1, 1024, 768, np.array(COLMAP_CAMERA_PARAMS, dtype=np.float64).tobytes()))
            conn.commit()
        except sqlite3.Error:
            conn.rollback()
            raise
        finally:
            conn.close()

    def export_all(self, output_dir):
        colmap_dir = Path(output_dir) / 'colmap'
        colmap_dir.mkdir(parents=True, exist_ok=True)
        self._create_colmap_database(colmap_dir / 'database.db')
        self.export_colmap(colmap_dir)
        logging.info(f"Exported all formats to {output_dir}")
