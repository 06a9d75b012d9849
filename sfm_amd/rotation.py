"""Rodrigues vector <-> rotation matrix on the host (what the reference gets from
cv2.Rodrigues at /root/reference/utils/sfm_reconstruction.py:419,465,544)."""
from __future__ import annotations

import numpy as np


def rodrigues(rvec):
    """3-vector -> 3x3 rotation, R = I + a [r]x + b [r]x^2."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    th2 = float(r @ r)
    S = np.array([[0.0, -r[2], r[1]], [r[2], 0.0, -r[0]], [-r[1], r[0], 0.0]])
    if th2 < 1e-4:
        a = 1.0 - th2 / 6.0 + th2 * th2 / 120.0
        b = 0.5 - th2 / 24.0 + th2 * th2 / 720.0
    else:
        th = np.sqrt(th2)
        a = np.sin(th) / th
        b = (1.0 - np.cos(th)) / th2
    return np.eye(3) + a * S + b * (S @ S)


def log_so3(R):
    """3x3 rotation -> Rodrigues 3-vector (angle in [0, pi])."""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = 0.5 * np.linalg.norm(v)
    c = 0.5 * (np.trace(R) - 1.0)
    theta = np.arctan2(s, c)
    if s > 1e-8:
        return v * (theta / (2.0 * s))
    if c > 0:
        return 0.5 * v
    M = 0.5 * (R + np.eye(3))            # theta ~ pi: R ~ 2 k k^T - I
    k = np.sqrt(np.clip(np.diag(M), 0.0, None))
    i = int(np.argmax(k))
    sg = np.sign(M[i]); sg[i] = 1.0
    k = k * sg
    return k / np.linalg.norm(k) * theta
