"""CPU oracle for descriptor matching (TEST INFRASTRUCTURE - not product code).

Restates `ImageMatcher.match_features` (/root/reference/utils/find_matches.py:141-155):
brute-force kNN (k=2) of every desc1 row in desc2, then Lowe's ratio test
`m.distance < 0.75 * n.distance` evaluated in double on float32 distances (:150-153).

The distance arithmetic lives in a third-party dependency that is absent from
/root/reference and from this image: opencv-python 4.11.0 (requirements.txt:2; call
sites find_matches.py:144,147: cv2.BFMatcher(...).knnMatch(k=2)).  Its documented
behaviour is restated: float32 distances; NORM_L2 = sqrtf of a float32 sum of squared
differences; NORM_HAMMING = popcount; top-2 ascending, lowest train index first on ties
(ties are decided on the float32 distance).  The reference holds no input->output vector
for this path (descriptors are not stored anywhere): PARITY UNPINNED with respect to
OpenCV's tie/rounding details.  What IS pinned (tests/test_matcher_oracle.py): the output
contract visible in the reference's shipped results bunny_data/matches/*.npz - one match
per query, queryIdx strictly increasing, distance == sqrtf(integer d^2)
(tests/golden/bunny_matches.npz, extracted by tests/golden/extract_bunny.py).

For integer-valued descriptors (SIFT: uint8-quantised floats) every partial sum of the
squared distance is an integer < 2^24, so the float32 sum is exact in any order and the
result is independent of OpenCV's SIMD summation order; for general floats this oracle
fixes the order as sequential over k with separate multiply and add roundings.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import numpy as np

_POP8 = np.array([bin(i).count("1") for i in range(256)], dtype=np.int32)


def _is_small_int(a):
    a = np.asarray(a)
    if a.dtype == np.uint8:
        return True
    return bool(np.all(a == np.rint(a)) and a.min(initial=0) >= 0 and a.max(initial=0) <= 255)


def sq_l2(desc1, desc2):
    """float32 [Nq,Nt] squared L2 distances."""
    a = np.asarray(desc1); b = np.asarray(desc2)
    if _is_small_int(a) and _is_small_int(b):
        ai = a.astype(np.int64); bi = b.astype(np.int64)
        d2 = (ai * ai).sum(1)[:, None] + (bi * bi).sum(1)[None, :] - 2 * (ai @ bi.T)
        return d2.astype(np.float32)          # exact: d2 <= 128*255^2 < 2^24
    a = a.astype(np.float32); b = b.astype(np.float32)
    acc = np.zeros((a.shape[0], b.shape[0]), dtype=np.float32)
    for k in range(a.shape[1]):
        diff = a[:, k, None] - b[None, :, k]
        acc = acc + diff * diff               # two roundings, k ascending
    return acc


def hamming(desc1, desc2):
    a = np.asarray(desc1, dtype=np.uint8); b = np.asarray(desc2, dtype=np.uint8)
    out = np.zeros((a.shape[0], b.shape[0]), dtype=np.int32)
    for k in range(a.shape[1]):
        out += _POP8[a[:, k, None] ^ b[None, :, k]]
    return out.astype(np.float32)


def knn2(desc1, desc2, metric="l2", chunk=2048):
    """cv2.BFMatcher(norm).knnMatch(desc1, desc2, k=2) -> idx1, idx2 [Nq] int32, d1, d2 [Nq] f32."""
    desc1 = np.asarray(desc1); desc2 = np.asarray(desc2)
    nq, nt = desc1.shape[0], desc2.shape[0]
    if nt < 2:
        raise ValueError("knn2 needs at least 2 train descriptors")
    idx1 = np.empty(nq, np.int32); idx2 = np.empty(nq, np.int32)
    d1 = np.empty(nq, np.float32); d2 = np.empty(nq, np.float32)
    for s in range(0, nq, chunk):
        e = min(nq, s + chunk)
        if metric == "l2":
            dist = np.sqrt(sq_l2(desc1[s:e], desc2))      # correctly rounded sqrtf
        elif metric == "hamming":
            dist = hamming(desc1[s:e], desc2)
        else:
            raise ValueError(metric)
        rows = np.arange(e - s)
        i1 = np.argmin(dist, axis=1)                      # first occurrence = lowest index
        v1 = dist[rows, i1].copy()
        dist[rows, i1] = np.inf
        i2 = np.argmin(dist, axis=1)
        idx1[s:e] = i1; idx2[s:e] = i2
        d1[s:e] = v1; d2[s:e] = dist[rows, i2]
    return idx1, idx2, d1, d2


def ratio_filter(idx1, d1, d2, ratio=0.75):
    """find_matches.py:150-153 - comparison in double, strict '<'."""
    keep = d1.astype(np.float64) < ratio * d2.astype(np.float64)
    q = np.nonzero(keep)[0].astype(np.int32)
    return q, idx1[q].astype(np.int32), d1[q].astype(np.float32)


def match_features(desc1, desc2, ratio=0.75, metric="l2"):
    """Returns (queryIdx, trainIdx, distance) arrays in query order.
    No query or no train rows: knnMatch returns [] and so does the loop (:147-155).  Exactly one train row:
    each knn row has one entry and the unpacking `for m, n in matches` (:151) raises ValueError."""
    if np.asarray(desc2).shape[0] == 0 or np.asarray(desc1).shape[0] == 0:
        z = np.zeros(0, np.int32)
        return z, z.copy(), np.zeros(0, np.float32)
    if np.asarray(desc2).shape[0] == 1:
        raise ValueError("not enough values to unpack (expected 2, got 1)")
    idx1, _, d1, d2 = knn2(desc1, desc2, metric)
    return ratio_filter(idx1, d1, d2, ratio)
