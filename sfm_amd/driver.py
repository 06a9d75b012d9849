"""GPU drop-ins for the driver-side steps either side of the hot path:

* `DriverMixin.find_2d3d_matches`   /root/reference/utils/sfm_reconstruction.py:157-230
* `DriverMixin.triangulate_point`   :263-307
* `DriverMixin.add_new_matches`     :341-399
* `VerificationMixin.geometric_verification / verify_match_quality`
                                    /root/reference/utils/find_matches.py:157-214

The reference loops over image pairs and tracks in Python; here every pair of a driver step is one
segment of a single launch (`sfm_assoc_radius`, `sfm_triangulate2`, `sfm_epipolar_errors` in
libsfm_amd.so).  State contract, return values and log lines are the reference's.  No CPU fallback:
without the library or a GPU these raise.
"""
from __future__ import annotations

import ctypes as C
import logging

import numpy as np

from . import _lib

MATCHING_THRESHOLD = 2.0          # sfm_reconstruction.py:14
TRIANGULATION_MAX_ERROR = 4.0     # sfm_reconstruction.py:299


def _p(t):
    return C.c_void_p(t.data_ptr() if t is not None and t.numel() else 0)


def _dev(a, dtype, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(dev)


def _ptr_array(lengths, dev):
    import torch
    ptr = np.zeros(len(lengths) + 1, dtype=np.int64)
    np.cumsum(lengths, out=ptr[1:])
    return ptr, torch.from_numpy(ptr).to(dev)


# ------------------------------------------------------------------------------------------ association
def associate_segments(track_pts, corr_pts, threshold=MATCHING_THRESHOLD, device=0):
    """For each segment s: all (row, col) with ||track_pts[s][row] - corr_pts[s][col]||_2 < threshold in
    float64, in np.where (row-major) order - the test of sfm_reconstruction.py:212-213 for every image pair
    of a driver step in one launch.  Returns [(rows, cols)] (int64, segment-local)."""
    import torch
    if len(track_pts) != len(corr_pts):
        raise ValueError("segment lists differ in length")
    h = _lib.get_handle(device)
    dev = torch.device("cuda", device)
    n_seg = len(track_pts)
    tl = [np.asarray(a, dtype=np.float64).reshape(-1, 2) for a in track_pts]
    cl = [np.asarray(a, dtype=np.float64).reshape(-1, 2) for a in corr_pts]
    t_ptr_h, t_ptr = _ptr_array([a.shape[0] for a in tl], dev)
    m_ptr_h, m_ptr = _ptr_array([a.shape[0] for a in cl], dev)
    T, M = int(t_ptr_h[-1]), int(m_ptr_h[-1])
    empty = (np.zeros(0, np.int64), np.zeros(0, np.int64))
    if n_seg == 0 or T == 0 or M == 0:
        return [empty for _ in range(n_seg)]
    tp = _dev(np.concatenate(tl), np.float64, dev)
    cp = _dev(np.concatenate(cl), np.float64, dev)
    need = C.c_int64()
    h.check(h.lib.sfm_assoc_workspace_bytes(T, C.byref(need)), "sfm_assoc_workspace_bytes")
    ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    capacity = max(1024, 2 * max(T, M))
    while True:
        out_row = torch.empty(capacity, dtype=torch.int32, device=dev)
        out_col = torch.empty(capacity, dtype=torch.int32, device=dev)
        h.call("sfm_assoc_radius", _p(tp), _p(t_ptr), _p(cp), _p(m_ptr), n_seg, T, C.c_double(threshold),
               _p(out_row), _p(out_col), capacity, _p(total), _p(ws), need.value)
        n = int(total.item())
        if n <= capacity:
            break
        capacity = n
    rows = out_row[:n].cpu().numpy().astype(np.int64)
    cols = out_col[:n].cpu().numpy().astype(np.int64)
    seg_end = np.searchsorted(rows, t_ptr_h[1:], side="left")      # rows ascend across segments
    out, b = [], 0
    for s in range(n_seg):
        e = int(seg_end[s])
        out.append((rows[b:e] - t_ptr_h[s], cols[b:e] - m_ptr_h[s]))
        b = e
    return out


def associate(track_pts, other_pts, threshold=MATCHING_THRESHOLD, device=0):
    """np.where(np.linalg.norm(track_pts[:, None] - other_pts, axis=2) < threshold) on the GPU."""
    return associate_segments([track_pts], [other_pts], threshold, device)[0]


# ---------------------------------------------------------------------------------------- triangulation
def triangulate_two_view(proj, cam0, cam1, x0, x1, max_error=TRIANGULATION_MAX_ERROR, device=0):
    """Batched triangulate_point for two-view tracks (sfm_reconstruction.py:287-307).
    proj [n_cams,3,4] = K[R|t]; cam0/cam1 [n] index it; x0/x1 [n,2] pixels.
    Returns X [n,3] float64, valid [n] bool, err [n,2] float64 (reprojection error in each view)."""
    import torch
    h = _lib.get_handle(device)
    dev = torch.device("cuda", device)
    proj = np.asarray(proj, dtype=np.float64).reshape(-1, 12)
    cam0 = np.asarray(cam0, dtype=np.int32).ravel(); cam1 = np.asarray(cam1, dtype=np.int32).ravel()
    x0 = np.asarray(x0, dtype=np.float64).reshape(-1, 2); x1 = np.asarray(x1, dtype=np.float64).reshape(-1, 2)
    n = cam0.shape[0]
    if not (cam1.shape[0] == n and x0.shape[0] == n and x1.shape[0] == n):
        raise ValueError("triangulate_two_view: per-track arrays differ in length")
    if n == 0:
        return np.zeros((0, 3)), np.zeros(0, bool), np.zeros((0, 2))
    if proj.shape[0] < 1 or min(cam0.min(), cam1.min()) < 0 or max(cam0.max(), cam1.max()) >= proj.shape[0]:
        raise ValueError("triangulate_two_view: camera index out of range")
    d_proj, d_c0, d_c1 = _dev(proj, np.float64, dev), _dev(cam0, np.int32, dev), _dev(cam1, np.int32, dev)
    d_x0, d_x1 = _dev(x0, np.float64, dev), _dev(x1, np.float64, dev)
    X = torch.empty((n, 3), dtype=torch.float64, device=dev)
    valid = torch.empty(n, dtype=torch.int32, device=dev)
    err = torch.empty((n, 2), dtype=torch.float64, device=dev)
    h.call("sfm_triangulate2", _p(d_proj), proj.shape[0], _p(d_c0), _p(d_c1), _p(d_x0), _p(d_x1), n,
           C.c_double(max_error), _p(X), _p(valid), _p(err))
    return X.cpu().numpy(), valid.cpu().numpy().astype(bool), err.cpu().numpy()


def projection_matrix(K, R, t):
    """P = K [R | t] exactly as sfm_reconstruction.py:281-283 forms it."""
    return K @ np.hstack([R, np.asarray(t).reshape(3, 1)])


# ------------------------------------------------------------------------------- epipolar verification
def symmetric_epipolar_errors(pts1, pts2, F, threshold=3.0, device=0):
    """Per-match symmetric epipolar distance (float32) and `err < threshold` mask for a LIST of pairs
    (pts1[s], pts2[s], F[s]) in one launch - find_matches.py:160-174.  Returns [(err, mask)]."""
    import torch
    h = _lib.get_handle(device)
    dev = torch.device("cuda", device)
    n_seg = len(F)
    if not (len(pts1) == n_seg and len(pts2) == n_seg):
        raise ValueError("segment lists differ in length")
    p1 = [np.asarray(a, dtype=np.float32).reshape(-1, 2) for a in pts1]
    p2 = [np.asarray(a, dtype=np.float32).reshape(-1, 2) for a in pts2]
    for a, b in zip(p1, p2):
        if a.shape[0] != b.shape[0]:
            raise ValueError("pts1 / pts2 differ in length")
    ptr_h, ptr = _ptr_array([a.shape[0] for a in p1], dev)
    n = int(ptr_h[-1])
    if n == 0:
        return [(np.zeros(0, np.float32), np.zeros(0, bool)) for _ in range(n_seg)]
    d_F = _dev(np.stack([np.asarray(f, dtype=np.float64).reshape(9) for f in F]), np.float64, dev)
    d_p1, d_p2 = _dev(np.concatenate(p1), np.float32, dev), _dev(np.concatenate(p2), np.float32, dev)
    err = torch.empty(n, dtype=torch.float32, device=dev)
    mask = torch.empty(n, dtype=torch.uint8, device=dev)
    h.call("sfm_epipolar_errors", _p(d_F), _p(ptr), n_seg, _p(d_p1), _p(d_p2), n, C.c_float(threshold),
           _p(err), _p(mask))
    e, m = err.cpu().numpy(), mask.cpu().numpy().astype(bool)
    return [(e[ptr_h[s]:ptr_h[s + 1]], m[ptr_h[s]:ptr_h[s + 1]]) for s in range(n_seg)]


def _verification_result(pts1, pts2, symmetric_errors, inlier_mask):
    """The metrics dictionary of find_matches.py:176-201, from the per-match values the GPU produced."""
    any_in = bool(inlier_mask.any())
    if any_in:
        reproj_error = np.mean(symmetric_errors[inlier_mask])
        spread1 = np.std(pts1[inlier_mask], axis=0)
        spread2 = np.std(pts2[inlier_mask], axis=0)
        well_distributed = np.all(spread1 > 20) and np.all(spread2 > 20)
    else:
        reproj_error = float('inf')
        well_distributed = False
    return {
        'metrics': {
            'total_matches': len(pts1),
            'inliers': np.sum(inlier_mask),
            'inlier_ratio': np.mean(inlier_mask),
            'reprojection_error': reproj_error,
            'symmetric_error': np.mean(symmetric_errors),
            'well_distributed': well_distributed,
        },
        'inlier_mask': inlier_mask,
        'symmetric_errors': symmetric_errors,
    }


def verify_pairs(pairs, threshold=3.0, device=0):
    """geometric_verification for a list of (pts1, pts2, F) in one launch."""
    p1 = [np.asarray(p[0], dtype=np.float32).reshape(-1, 2) for p in pairs]
    p2 = [np.asarray(p[1], dtype=np.float32).reshape(-1, 2) for p in pairs]
    res = symmetric_epipolar_errors(p1, p2, [p[2] for p in pairs], threshold, device)
    return [_verification_result(a, b, e, m) for a, b, (e, m) in zip(p1, p2, res)]


class VerificationMixin:
    """geometric_verification / verify_match_quality of the reference's ImageMatcher."""
    device = 0

    def geometric_verification(self, pts1, pts2, F, threshold=3.0):
        return verify_pairs([(pts1, pts2, F)], threshold, getattr(self, "device", 0))[0]

    def verify_match_quality(self, geometric_results, min_inliers=15, min_ratio=0.3, max_error=2.0):
        m = geometric_results['metrics']
        return all([m['inliers'] >= min_inliers, m['inlier_ratio'] >= min_ratio,
                    m['reprojection_error'] <= max_error, m['well_distributed']])


# ------------------------------------------------------------------------------------------- the mixin
class DriverMixin:
    """find_2d3d_matches / triangulate_point / add_new_matches on the GPU, state contract of the reference
    (`self.points3D`, `self.point_tracks`, `self.poses`, `self.K`, `self.corr_dir`, `find_image_pairs`)."""
    ba_device = 0

    def _tracks_by_image(self):
        """image id -> (track indices, their 2-D points) in track order: the `if other_img_id in track`
        walk of sfm_reconstruction.py:200-203, done once per call instead of once per pair."""
        idx, pts = {}, {}
        for j, track in enumerate(self.point_tracks):
            for img_id, p in track.items():
                idx.setdefault(img_id, []).append(j)
                pts.setdefault(img_id, []).append(p)
        return idx, pts

    def find_2d3d_matches(self, image_id):
        points3D_array = np.array(self.points3D)
        image_pairs = self.find_image_pairs(image_id)
        logging.info(f"Found {len(image_pairs)} pairs for image {image_id}")
        by_idx, by_pts = self._tracks_by_image()
        seg_tracks, seg_corr, seg_meta = [], [], []
        for pair in image_pairs:
            try:
                pts1 = np.load(self.corr_dir / f'{pair}_pts1.npy')
                pts2 = np.load(self.corr_dir / f'{pair}_pts2.npy')
                id1, id2 = map(int, pair.split('_')[1:])
                if id1 == image_id:
                    new_img_pts, other_img_pts, other_img_id = pts1, pts2, id2
                else:
                    new_img_pts, other_img_pts, other_img_id = pts2, pts1, id1
                if other_img_id not in by_idx:
                    continue
                valid_track_points = np.array(by_pts[other_img_id])
                if valid_track_points.ndim != 2 or valid_track_points.shape[1] != 2 or \
                        other_img_pts.ndim != 2 or other_img_pts.shape[1] != 2:
                    raise ValueError(f"expected [n,2] points, got {valid_track_points.shape} / {other_img_pts.shape}")
                seg_tracks.append(valid_track_points)
                seg_corr.append(other_img_pts)
                seg_meta.append((np.asarray(by_idx[other_img_id], dtype=np.int64), new_img_pts))
            except (FileNotFoundError, ValueError) as e:
                logging.warning(f"Failed to process pair {pair}: {e}")
                continue
        out3, out2 = [], []
        if seg_tracks:
            hits = associate_segments(seg_tracks, seg_corr, MATCHING_THRESHOLD, getattr(self, "ba_device", 0))
            for (rows, cols), (track_idx, new_img_pts) in zip(hits, seg_meta):
                if rows.size:
                    out3.append(points3D_array[track_idx[rows]])
                    out2.append(new_img_pts[cols])
        points3D = np.concatenate(out3) if out3 else np.array([])
        points2D = np.concatenate(out2) if out2 else np.array([])
        logging.info(f"Found {len(points3D)} 2D-3D matches for image {image_id}")
        return points3D, points2D

    def triangulate_point(self, image_points):
        if len(image_points) < 2:
            return None
        Ps, points = [], []
        for img_id, point in image_points.items():
            R, t = self.poses[img_id]
            Ps.append(projection_matrix(self.K, R, t))
            points.append(point)
        Ps = np.array(Ps); points = np.array(points)
        X, valid, _ = triangulate_two_view(Ps[:2], [0], [1], points[0].reshape(1, 2), points[1].reshape(1, 2),
                                           TRIANGULATION_MAX_ERROR, getattr(self, "ba_device", 0))
        if not valid[0]:
            return None
        point3D = X[0]
        for P, point2D in zip(Ps[2:], points[2:]):          # views beyond the first two: gate only (:300-305)
            projected = P @ np.append(point3D, 1)
            projected = projected[:2] / projected[2]
            if np.linalg.norm(projected - point2D) > TRIANGULATION_MAX_ERROR:
                return None
        return point3D

    def add_new_matches(self, pair, image_id):
        try:
            pts1 = np.asarray(np.load(self.corr_dir / f'{pair}_pts1.npy')).reshape(-1, 2)
            pts2 = np.asarray(np.load(self.corr_dir / f'{pair}_pts2.npy')).reshape(-1, 2)
            id1, id2 = map(int, pair.split('_')[1:])
            existing = set()
            for track in self.point_tracks:
                for img_id, point in track.items():
                    existing.add((img_id, tuple(np.asarray(point).ravel())))
            keep = [k for k, (a, b) in enumerate(zip(pts1, pts2))
                    if (id1, tuple(a.ravel())) not in existing and (id2, tuple(b.ravel())) not in existing]
            n_valid = 0
            if keep and id1 != id2:
                Ps = [projection_matrix(self.K, *self.poses[id1]), projection_matrix(self.K, *self.poses[id2])]
                a, b = pts1[keep], pts2[keep]
                zeros = np.zeros(len(keep), np.int32)
                X, valid, _ = triangulate_two_view(Ps, zeros, zeros + 1, a, b, TRIANGULATION_MAX_ERROR,
                                                   getattr(self, "ba_device", 0))
                sel = np.flatnonzero(valid)
                n_valid = int(sel.size)
                if n_valid:
                    self.points3D.extend(X[sel])
                    self.point_tracks.extend({id1: a[k].tolist(), id2: b[k].tolist()} for k in sel)
            if n_valid:
                logging.info(f"Added {n_valid} new tracks from pair {pair}")
            else:
                logging.warning(f"No valid tracks found for pair {pair}")
        except (FileNotFoundError, ValueError, KeyError, IndexError, TypeError) as e:   # data problems only:
            # the reference swallows every Exception here (:395-397); a missing library / GPU must stay loud
            logging.warning(f"Failed to add matches for pair {pair}: {e}")
            return False
        return True
