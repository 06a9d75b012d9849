"""CPU oracle for the driver-side rows that sit either side of the hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and nothing else.  The product
(sfm_amd.driver) never imports this module; it fails loudly when libsfm_amd.so or the GPU is missing.

What is restated, and what pins it:

* associate()            - /root/reference/utils/sfm_reconstruction.py:209-218 (track point <-> pair
                           correspondence association, < MATCHING_THRESHOLD px, np.where order).
  find_2d3d_matches()    - :157-230 around it.  Pinned by tests/golden/driver_bunny.npz, produced by the
                           reference's OWN find_2d3d_matches run on its shipped bunny_data/ files
                           (tests/golden/make_golden_driver.py).
* triangulate_dlt()      - cv2.triangulatePoints (opencv-python 4.11.0, requirements.txt:2; not installable
                           here): per point the 4x4 system rows x*P[2]-P[0], y*P[2]-P[1] of both views, null
                           vector = last right singular vector.  Pinned by the reference's shipped outputs:
                           every non-initial point of bunny_data/reconstruction/points3D.json is the output
                           of triangulate_point() (:263-307) on its shipped two-view track, and this
                           restatement reproduces all 2,326 of them to <= 6e-11 relative.
  triangulate_point()    - :263-307 (first two views, then the 4 px gate over all views).
  add_new_matches()      - :341-399 (dedupe against existing track points, triangulate, append).  The
                           golden for its control flow runs the reference's method with triangulate_dlt as
                           the cv2.triangulatePoints stand-in, so it pins ordering/dedupe/gating only.
* epilines()             - cv2.computeCorrespondEpilines: double accumulate, normalise a^2+b^2=1, store in
                           the points' precision (float32 here).
  geometric_verification() - /root/reference/utils/find_matches.py:157-201, float32 NumPy arithmetic as the
                           reference performs it.  Pinned by the 148 shipped pairs: `mask` in
                           bunny_data/fundamental/*.npz IS this function's inlier_mask for the shipped
                           F/pts1/pts2, and matching_results.csv holds its metrics.
"""
import numpy as np

MATCHING_THRESHOLD = 2.0          # sfm_reconstruction.py:14
TRIANGULATION_MAX_ERROR = 4.0     # sfm_reconstruction.py:299


# ------------------------------------------------------------------------------------------ association
def associate(track_pts, other_pts, threshold=MATCHING_THRESHOLD):
    """sfm_reconstruction.py:212-213 - all (track, correspondence) pairs closer than `threshold`, row-major."""
    track_pts = np.asarray(track_pts)
    distances = np.linalg.norm(track_pts[:, None] - other_pts, axis=2)
    rows, cols = np.where(distances < threshold)
    return rows, cols


def find_2d3d_matches(points3D, point_tracks, pairs, image_id, threshold=MATCHING_THRESHOLD):
    """sfm_reconstruction.py:167-230.  `pairs` = [(name, pts1, pts2)] in the order find_image_pairs gave."""
    out3, out2 = [], []
    points3D_array = np.array(points3D)
    for name, pts1, pts2 in pairs:
        id1, id2 = map(int, name.split('_')[1:])
        if id1 == image_id:
            new_img_pts, other_img_pts, other_img_id = pts1, pts2, id2
        else:
            new_img_pts, other_img_pts, other_img_id = pts2, pts1, id1
        valid_idx = [i for i, tr in enumerate(point_tracks) if other_img_id in tr]
        if not valid_idx:
            continue
        valid_pts = np.array([point_tracks[i][other_img_id] for i in valid_idx])
        rows, cols = associate(valid_pts, other_img_pts, threshold)
        for r, c in zip(rows, cols):
            out3.append(points3D_array[valid_idx[r]])
            out2.append(new_img_pts[c])
    return np.array(out3), np.array(out2)


# ---------------------------------------------------------------------------------------- triangulation
def triangulate_dlt(P0, P1, x0, x1):
    """cv2.triangulatePoints restated: x0, x1 are [n,2]; returns homogeneous [n,4] (sign/scale arbitrary)."""
    P0 = np.asarray(P0, np.float64); P1 = np.asarray(P1, np.float64)
    x0 = np.asarray(x0, np.float64).reshape(-1, 2); x1 = np.asarray(x1, np.float64).reshape(-1, 2)
    out = np.empty((x0.shape[0], 4))
    for i in range(x0.shape[0]):
        A = np.stack([x0[i, 0] * P0[2] - P0[0], x0[i, 1] * P0[2] - P0[1],
                      x1[i, 0] * P1[2] - P1[0], x1[i, 1] * P1[2] - P1[1]])
        out[i] = np.linalg.svd(A)[2][3]
    return out


def triangulate_point(Ps, points, max_error=TRIANGULATION_MAX_ERROR):
    """sfm_reconstruction.py:287-307: DLT on views 0,1; reject when any view reprojects > max_error px."""
    Ps = np.asarray(Ps, np.float64); points = np.asarray(points, np.float64)
    X4 = triangulate_dlt(Ps[0], Ps[1], points[0], points[1])[0]
    with np.errstate(all="ignore"):
        X = X4[:3] / X4[3]
        for P, uv in zip(Ps, points):
            proj = P @ np.append(X, 1)
            proj = proj[:2] / proj[2]
            if np.linalg.norm(proj - uv) > max_error:
                return None
    return X


def projection(K, R, t):
    """sfm_reconstruction.py:281-283."""
    return K @ np.hstack([R, np.asarray(t).reshape(3, 1)])


def add_new_matches(points3D, point_tracks, poses, K, pair, pts1, pts2):
    """sfm_reconstruction.py:353-393; mutates points3D / point_tracks like the reference, returns #added."""
    pts1 = np.asarray(pts1).reshape(-1, 2); pts2 = np.asarray(pts2).reshape(-1, 2)
    id1, id2 = map(int, pair.split('_')[1:])
    existing = set()
    for tr in point_tracks:
        for img_id, p in tr.items():
            existing.add((img_id, tuple(np.asarray(p).ravel())))
    new_tracks = []
    for p1, p2 in zip(pts1, pts2):
        if (id1, tuple(p1.ravel())) not in existing and (id2, tuple(p2.ravel())) not in existing:
            new_tracks.append({id1: p1.tolist(), id2: p2.tolist()})
    added = 0
    for tr in new_tracks:
        Ps = [projection(K, *poses[i]) for i in tr]
        X = triangulate_point(Ps, [tr[i] for i in tr])
        if X is not None:
            points3D.append(X); point_tracks.append(tr); added += 1
    return added


# ------------------------------------------------------------------------------- epipolar verification
def epilines(points, which_image, F):
    """cv2.computeCorrespondEpilines for float32 points: l = F x (image 1) or F^T x (image 2), double
    accumulate, scaled so a^2+b^2 = 1, stored float32."""
    f = np.asarray(F, np.float64)
    if which_image == 2:
        f = f.T
    p = np.asarray(points, np.float32).reshape(-1, 2).astype(np.float64)
    a = f[0, 0] * p[:, 0] + f[0, 1] * p[:, 1] + f[0, 2]
    b = f[1, 0] * p[:, 0] + f[1, 1] * p[:, 1] + f[1, 2]
    c = f[2, 0] * p[:, 0] + f[2, 1] * p[:, 1] + f[2, 2]
    nu = a * a + b * b
    with np.errstate(all="ignore"):
        nu = np.where(nu != 0, 1.0 / np.sqrt(nu), 1.0)
    return np.stack([a * nu, b * nu, c * nu], axis=1).astype(np.float32)


def symmetric_epipolar_errors(pts1, pts2, F):
    """find_matches.py:160-171 (float32 arithmetic as NumPy performs it on float32 operands)."""
    pts1 = np.asarray(pts1, np.float32).reshape(-1, 2); pts2 = np.asarray(pts2, np.float32).reshape(-1, 2)
    lines1 = epilines(pts2, 2, F)
    lines2 = epilines(pts1, 1, F)
    e1 = np.abs(np.sum(np.multiply(pts1, lines1[:, :2]), axis=1) + lines1[:, 2]) / \
        np.sqrt(np.sum(np.square(lines1[:, :2]), axis=1))
    e2 = np.abs(np.sum(np.multiply(pts2, lines2[:, :2]), axis=1) + lines2[:, 2]) / \
        np.sqrt(np.sum(np.square(lines2[:, :2]), axis=1))
    return (e1 + e2) / 2


def verification_metrics(pts1, pts2, symmetric_errors, threshold=3.0):
    """find_matches.py:173-201 given the per-match errors."""
    inlier_mask = symmetric_errors < threshold
    reproj_error = np.mean(symmetric_errors[inlier_mask]) if np.any(inlier_mask) else float('inf')
    if np.any(inlier_mask):
        s1 = np.std(pts1[inlier_mask], axis=0); s2 = np.std(pts2[inlier_mask], axis=0)
        well = bool(np.all(s1 > 20) and np.all(s2 > 20))
    else:
        well = False
    return {
        'metrics': {
            'total_matches': len(pts1),
            'inliers': np.sum(inlier_mask),
            'inlier_ratio': np.mean(inlier_mask),
            'reprojection_error': reproj_error,
            'symmetric_error': np.mean(symmetric_errors),
            'well_distributed': well,
        },
        'inlier_mask': inlier_mask,
        'symmetric_errors': symmetric_errors,
    }


def geometric_verification(pts1, pts2, F, threshold=3.0):
    pts1 = np.asarray(pts1, np.float32).reshape(-1, 2); pts2 = np.asarray(pts2, np.float32).reshape(-1, 2)
    return verification_metrics(pts1, pts2, symmetric_epipolar_errors(pts1, pts2, F), threshold)


def verify_match_quality(res, min_inliers=15, min_ratio=0.3, max_error=2.0):
    """find_matches.py:203-214."""
    m = res['metrics']
    return bool(m['inliers'] >= min_inliers and m['inlier_ratio'] >= min_ratio
                and m['reprojection_error'] <= max_error and m['well_distributed'])
