"""Drop-in for the reference's `ImageMatcher.match_features`
(/root/reference/utils/find_matches.py:141-155): brute-force kNN (k=2) + Lowe ratio test on
the GPU through libsfm_amd.so; `geometric_verification` (:157-214) comes from sfm_amd.driver.
No CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .driver import VerificationMixin


class DMatch:
    """The attributes of cv2.DMatch the reference reads (find_matches.py:230-232,278-279,323-327)."""
    __slots__ = ("queryIdx", "trainIdx", "distance", "imgIdx")

    def __init__(self, queryIdx, trainIdx, distance, imgIdx=0):
        self.queryIdx = int(queryIdx)
        self.trainIdx = int(trainIdx)
        self.distance = float(distance)
        self.imgIdx = int(imgIdx)

    def __repr__(self):
        return f"DMatch(queryIdx={self.queryIdx}, trainIdx={self.trainIdx}, distance={self.distance})"


def _to_device(a, device):
    import torch
    if isinstance(a, torch.Tensor):
        return a.to(device).contiguous()
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a).to(device)


def _resolve_metric(q, t, metric):
    import torch
    if metric == "auto":
        if q.dtype == torch.uint8:
            metric = "hamming" if q.shape[1] <= 64 and q.shape[1] != 128 else "l2"
        else:
            metric = "l2"
    if metric not in ("l2", "hamming"):
        raise ValueError(f"unknown metric {metric!r}")
    return metric


def knn2(desc1, desc2, metric="auto", device=0):
    """cv2.BFMatcher(norm).knnMatch(desc1, desc2, k=2) -> (idx1, idx2, d1, d2) CUDA tensors.

    metric: "l2" (float32 or uint8 descriptors, SIFT), "hamming" (uint8 bit strings, ORB),
    "auto" = hamming for uint8 of <= 64 bytes, l2 otherwise.  float32 descriptors whose values
    are all integers in [0,255] (what SIFT emits) take the exact integer path on the matrix cores.
    """
    import torch
    h = _lib.get_handle(device)
    dev = torch.device("cuda", device)
    q, t = _to_device(desc1, dev), _to_device(desc2, dev)
    if q.dim() != 2 or t.dim() != 2 or q.shape[1] != t.shape[1]:
        raise ValueError("descriptors must be [n, dim] with equal dim")
    if q.dtype != t.dtype:
        raise ValueError("descriptor dtypes differ")
    nq, dim = q.shape
    nt = t.shape[0]
    if nt < 2 or nq < 1:
        raise ValueError("knn2 needs at least 1 query and 2 train descriptors")
    metric = _resolve_metric(q, t, metric)
    if metric == "hamming":
        if q.dtype != torch.uint8:
            raise ValueError("hamming needs uint8 descriptors")
        code = _lib.METRIC_HAMMING
    elif q.dtype == torch.uint8:
        if dim in (32, 64, 128):
            code = _lib.METRIC_L2_U8
        else:
            q, t, code = q.float(), t.float(), _lib.METRIC_L2_F32
    else:
        q, t = q.float(), t.float()
        code = _lib.METRIC_L2_F32
        if dim in (32, 64, 128):
            flag = torch.ones(2, dtype=torch.int32, device=dev)
            q8 = torch.empty(q.shape, dtype=torch.uint8, device=dev)
            t8 = torch.empty(t.shape, dtype=torch.uint8, device=dev)
            h.call("sfm_match_f32_to_u8", C.c_void_p(q.data_ptr()), q.numel(), C.c_void_p(q8.data_ptr()),
                   C.c_void_p(flag[0:1].data_ptr()))
            h.call("sfm_match_f32_to_u8", C.c_void_p(t.data_ptr()), t.numel(), C.c_void_p(t8.data_ptr()),
                   C.c_void_p(flag[1:2].data_ptr()))
            if bool((flag == 1).all().item()):
                q, t, code = q8, t8, _lib.METRIC_L2_U8
    need = C.c_int64()
    h.check(h.lib.sfm_match_workspace_bytes(code, nq, nt, dim, C.byref(need)), "sfm_match_workspace_bytes")
    ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    idx1 = torch.empty(nq, dtype=torch.int32, device=dev)
    idx2 = torch.empty(nq, dtype=torch.int32, device=dev)
    d1 = torch.empty(nq, dtype=torch.float32, device=dev)
    d2 = torch.empty(nq, dtype=torch.float32, device=dev)
    h.call("sfm_match_knn2", code, C.c_void_p(q.data_ptr()), nq, C.c_void_p(t.data_ptr()), nt, dim,
           C.c_void_p(idx1.data_ptr()), C.c_void_p(idx2.data_ptr()), C.c_void_p(d1.data_ptr()),
           C.c_void_p(d2.data_ptr()), C.c_void_p(ws.data_ptr()), need.value)
    return idx1, idx2, d1, d2


def ratio_filter(idx1, d1, d2, ratio=0.75, device=0):
    """`m.distance < ratio * n.distance` (find_matches.py:152) + compaction in query order."""
    import torch
    h = _lib.get_handle(device)
    dev = idx1.device
    nq = idx1.shape[0]
    qi = torch.empty(nq, dtype=torch.int32, device=dev)
    ti = torch.empty(nq, dtype=torch.int32, device=dev)
    dd = torch.empty(nq, dtype=torch.float32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    ws_bytes = ((nq + 255) // 256) * 8 + 64
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    h.call("sfm_match_ratio", nq, C.c_void_p(idx1.data_ptr()), C.c_void_p(d1.data_ptr()),
           C.c_void_p(d2.data_ptr()), C.c_double(ratio), C.c_void_p(qi.data_ptr()), C.c_void_p(ti.data_ptr()),
           C.c_void_p(dd.data_ptr()), C.c_void_p(cnt.data_ptr()), C.c_void_p(ws.data_ptr()), ws_bytes)
    m = int(cnt.item())
    return qi[:m], ti[:m], dd[:m]


def match_arrays(desc1, desc2, ratio=0.75, metric="auto", device=0):
    """(queryIdx, trainIdx, distance) NumPy arrays in query order.

    Degenerate inputs follow the reference: with no query or no train descriptors knnMatch hands back
    an empty list and the ratio loop yields [] (find_matches.py:147-155); with exactly ONE train
    descriptor every knn row holds a single DMatch and `for m, n in matches` (find_matches.py:151)
    raises ValueError, which the per-pair try/except at :344-350 turns into a skipped pair."""
    n1 = int(desc1.shape[0]) if desc1 is not None else 0
    n2 = int(desc2.shape[0]) if desc2 is not None else 0
    if n1 == 0 or n2 == 0:
        return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32)
    if n2 == 1:
        raise ValueError("not enough values to unpack (expected 2, got 1)")
    idx1, _, d1, d2 = knn2(desc1, desc2, metric, device)
    q, t, d = ratio_filter(idx1, d1, d2, ratio, device)
    return q.cpu().numpy(), t.cpu().numpy(), d.cpu().numpy()


class ImageMatcher(VerificationMixin):
    """`match_features` with the reference's call shape (find_matches.py:141).  The in-tree
    reference matches ORB bit strings with NORM_HAMMING and ratio 0.75; its shipped results come
    from SIFT / L2.  metric="auto" follows the descriptor type."""

    def __init__(self, data_dir=None, ratio=0.75, metric="auto", device=0):
        self.data_dir = data_dir
        self.ratio = float(ratio)
        self.metric = metric
        self.device = device

    def match_features(self, desc1, desc2):
        q, t, d = match_arrays(desc1, desc2, self.ratio, self.metric, self.device)
        return [DMatch(a, b, c) for a, b, c in zip(q.tolist(), t.tolist(), d.tolist())]
