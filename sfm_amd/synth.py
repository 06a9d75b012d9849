"""Seeded synthetic scenes and descriptor sets.

Geometry mirrors the reference's capture set-up: K = (1228, 1228, 512, 384) on a 1024x768
image (/root/reference/utils/sfm_reconstruction.py:40-49), cameras on a hemisphere of
radius 6 looking at the origin, points in the box [-1,1]^3.  NumPy only.
"""
from __future__ import annotations

from dataclasses import dataclass
import numpy as np

K_REF = (1228.0, 1228.0, 512.0, 384.0)
WIDTH, HEIGHT = 1024.0, 768.0


from .rotation import rodrigues as _rodrigues, log_so3 as _log_so3


@dataclass
class Scene:
    """Ground truth + noisy observations + perturbed start, as flat arrays."""
    cams_true: np.ndarray     # [C,10] rvec,t,fx,fy,cx,cy
    pts_true: np.ndarray      # [P,3]
    cams0: np.ndarray         # [C,10] initial guess
    pts0: np.ndarray          # [P,3]
    cam_idx: np.ndarray       # [N] int64, point-major
    pt_idx: np.ndarray        # [N] int64
    uv: np.ndarray            # [N,2] f64 (float32-exact pixels, like the reference's tracks)
    K: np.ndarray             # 3x3

    @property
    def n_obs(self):
        return int(self.cam_idx.shape[0])

    def state(self):
        """(poses, points3D, point_tracks, K) in the reference's container types
        (sfm_reconstruction.py:57-59): dict id->(R, t(3,1)), list of 3-lists, list of dicts."""
        poses = {}
        for c in range(self.cams0.shape[0]):
            poses[f"{c:04d}.ppm"] = (_rodrigues(self.cams0[c, :3]), self.cams0[c, 3:6].reshape(3, 1).copy())
        ids = list(poses.keys())
        tracks = [dict() for _ in range(self.pts0.shape[0])]
        for k in range(self.n_obs):
            tracks[int(self.pt_idx[k])][ids[int(self.cam_idx[k])]] = [float(self.uv[k, 0]), float(self.uv[k, 1])]
        return poses, self.pts0.tolist(), tracks, self.K.copy()


def coherent_camera_order(centres):
    """Permutation that numbers cameras on the hemisphere so that neighbours in index are neighbours in space: latitude
    rings, serpentine in azimuth - the order a turntable / orbit capture adds its images in (the reference's shipped bunny
    set: 36 views around the object, tracks between neighbouring views only)."""
    n = centres.shape[0]
    z = centres[:, 2] / np.linalg.norm(centres, axis=1)
    n_rings = max(1, int(round(np.sqrt(n / 2.0))))
    ring = np.minimum((np.argsort(np.argsort(z)) * n_rings) // n, n_rings - 1)
    phi = np.arctan2(centres[:, 1], centres[:, 0])
    key = np.where(ring % 2 == 0, phi, -phi)
    return np.lexsort((key, ring))


def make_scene(n_cams, n_pts, obs_per_point=None, seed=0, noise_px=0.5,
               pt_sigma=0.02, cam_sigma=0.0, radius=6.0, visibility="random"):
    """Synthetic pinhole scene.  obs_per_point=None -> every camera sees every point.
    visibility="random": the L cameras of a track are drawn uniformly (the BASELINE scene).
    visibility="nearest": spatially coherent, as a real capture is - the points lie on the surface of the unit sphere
    and each is seen by the L cameras whose centres are closest in direction, and the cameras are numbered along the
    hemisphere (coherent_camera_order)."""
    if visibility not in ("random", "nearest"):
        raise ValueError(f"unknown visibility {visibility!r}")
    rng = np.random.default_rng(seed)
    fx, fy, cx, cy = K_REF
    # camera centres on the upper hemisphere (golden-angle spiral, deterministic)
    i = np.arange(n_cams) + 0.5
    zc = 0.15 + 0.8 * i / n_cams
    phi = i * np.pi * (3.0 - np.sqrt(5.0))
    rad = np.sqrt(1.0 - zc * zc)
    centres = radius * np.stack([rad * np.cos(phi), rad * np.sin(phi), zc], axis=1)
    if visibility == "nearest":
        centres = centres[coherent_camera_order(centres)]
    cams = np.zeros((n_cams, 10))
    for c in range(n_cams):
        zax = -centres[c] / np.linalg.norm(centres[c])
        up = np.array([0.0, 0.0, 1.0])
        xax = np.cross(up, zax); xax /= np.linalg.norm(xax)
        yax = np.cross(zax, xax)
        R = np.stack([xax, yax, zax])
        cams[c, :3] = _log_so3(R)
        cams[c, 3:6] = -R @ centres[c]
        cams[c, 6:] = (fx, fy, cx, cy)
    pts = rng.uniform(-1.0, 1.0, size=(n_pts, 3))
    L = n_cams if obs_per_point is None else int(obs_per_point)
    if visibility == "nearest":
        # surface points of the upper unit sphere (what an object in front of the cameras shows them)
        v = rng.normal(size=(n_pts, 3)); v[:, 2] = np.abs(v[:, 2]) * 0.8 + 0.15
        pts = v / np.linalg.norm(v, axis=1, keepdims=True)
    if L >= n_cams:
        cam_idx = np.tile(np.arange(n_cams, dtype=np.int64), n_pts)
    elif visibility == "nearest":
        cdir = centres / np.linalg.norm(centres, axis=1, keepdims=True)
        cam_idx = np.empty((n_pts, L), dtype=np.int64)
        for a in range(0, n_pts, 65536):                       # chunks keep the [points, cameras] score matrix small
            score = pts[a:a + 65536] @ cdir.T
            cam_idx[a:a + 65536] = np.sort(np.argpartition(-score, L, axis=1)[:, :L], axis=1)
        cam_idx = cam_idx.ravel()
    else:
        # L distinct cameras per point, ascending camera id inside a track
        keys = rng.random((n_pts, n_cams))
        sel = np.argpartition(keys, L, axis=1)[:, :L]
        cam_idx = np.sort(sel, axis=1).astype(np.int64).ravel()
        L = sel.shape[1]
    pt_idx = np.repeat(np.arange(n_pts, dtype=np.int64), L if L < n_cams else n_cams)
    Rs = np.stack([_rodrigues(cams[c, :3]) for c in range(n_cams)])
    Y = np.einsum("nij,nj->ni", Rs[cam_idx], pts[pt_idx]) + cams[cam_idx, 3:6]
    uv = np.stack([fx * Y[:, 0] / Y[:, 2] + cx, fy * Y[:, 1] / Y[:, 2] + cy], axis=1)
    uv += rng.normal(0.0, noise_px, size=uv.shape) if noise_px > 0 else 0.0
    uv = uv.astype(np.float32).astype(np.float64)
    cams0 = cams.copy()
    if cam_sigma > 0:
        cams0[:, :6] += rng.normal(0.0, cam_sigma, size=(n_cams, 6))
    pts0 = pts + (rng.normal(0.0, pt_sigma, size=pts.shape) if pt_sigma > 0 else 0.0)
    Kmat = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    return Scene(cams, pts, cams0, pts0, cam_idx, pt_idx, uv, Kmat)


def make_descriptors(n1, n2, seed=0, dim=128, kind="sift"):
    """Two descriptor sets.  kind="sift": integer-valued float32 in [0,255] (what
    cv2.SIFT emits and what the shipped bunny fixtures imply);
    the second set is a permuted copy with noise on 60 % of its rows + 40 % fresh rows.
    kind="uniform": non-integer float32 in [0,1)."""
    rng = np.random.default_rng(seed)

    def sift_like(n):
        v = rng.gamma(0.6, 1.0, size=(n, dim))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        return np.clip(np.rint(v * 512.0), 0, 255).astype(np.float32)

    if kind == "uniform":
        return rng.random((n1, dim), dtype=np.float32), rng.random((n2, dim), dtype=np.float32)
    d1 = sift_like(n1)
    n_copy = min(n1, int(0.6 * n2))
    src = rng.permutation(n1)[:n_copy]
    noisy = d1[src] + np.rint(rng.normal(0.0, 6.0, size=(n_copy, dim))).astype(np.float32)
    d2 = np.concatenate([np.clip(noisy, 0, 255), sift_like(n2 - n_copy)], axis=0)
    d2 = d2[rng.permutation(n2)].astype(np.float32)
    return d1, d2
