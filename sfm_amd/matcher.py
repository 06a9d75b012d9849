"""Drop-in for the reference's `ImageMatcher.match_features`
(/root/reference/utils/find_matches.py:141-155): brute-force kNN (k=2) + Lowe ratio test on
the GPU through libsfm_amd.so; `geometric_verification` (:157-214) comes from sfm_amd.driver.
No CPU fallback."""
from __future__ import annotations

import ctypes as C
from collections.abc import Sequence

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

    def __eq__(self, other):
        return (isinstance(other, DMatch) and (self.queryIdx, self.trainIdx, self.distance, self.imgIdx) ==
                (other.queryIdx, other.trainIdx, other.distance, other.imgIdx))

    __hash__ = None


def _dmatch(q, t, d):
    """A DMatch from values that already are Python int / float (ndarray.tolist()): no conversions, no __init__ call."""
    m = DMatch.__new__(DMatch)
    m.queryIdx, m.trainIdx, m.distance, m.imgIdx = q, t, d, 0
    return m


class DMatchList(Sequence):
    """What `match_features` returns: the matches of one image pair as a read-only sequence over the three result arrays.

    Everything the reference does with the list works - `len(matches)` (find_matches.py:274,305), iteration in query order with
    `.queryIdx / .trainIdx / .distance` on every element (:230-232, 278-279, 323-327), indexing, slicing, truthiness,
    `list(matches)` - but no object exists until one is asked for: building 30,000 DMatch objects up front cost two orders of
    magnitude more than the kernels that found the matches (50k x 50k: 0.4 ms on the GPU).  Array consumers read
    `.queryIdx`, `.trainIdx`, `.distance` (int32 / int32 / float32 NumPy arrays, query order) and skip the objects altogether:
    `pts1 = kp_xy[matches.queryIdx]` is the vector form of find_matches.py:278."""

    __slots__ = ("queryIdx", "trainIdx", "distance")

    def __init__(self, queryIdx, trainIdx, distance):
        self.queryIdx = np.asarray(queryIdx, dtype=np.int32)
        self.trainIdx = np.asarray(trainIdx, dtype=np.int32)
        self.distance = np.asarray(distance, dtype=np.float32)
        if not (self.queryIdx.ndim == 1 and self.queryIdx.shape == self.trainIdx.shape == self.distance.shape):
            raise ValueError("queryIdx, trainIdx and distance must be 1-D arrays of one length")

    def __len__(self):
        return int(self.queryIdx.shape[0])

    def __getitem__(self, i):
        if isinstance(i, slice):
            return DMatchList(self.queryIdx[i], self.trainIdx[i], self.distance[i])
        i = int(i)
        n = len(self)
        if i < -n or i >= n:
            raise IndexError("DMatchList index out of range")
        return _dmatch(int(self.queryIdx[i]), int(self.trainIdx[i]), float(self.distance[i]))

    def __iter__(self):
        # one tolist() per array, then objects one at a time: nothing of the pair is held beyond the element in hand
        return map(_dmatch, self.queryIdx.tolist(), self.trainIdx.tolist(), self.distance.tolist())

    def __eq__(self, other):
        if isinstance(other, DMatchList):
            return (np.array_equal(self.queryIdx, other.queryIdx) and np.array_equal(self.trainIdx, other.trainIdx) and
                    np.array_equal(self.distance, other.distance))
        if isinstance(other, (list, tuple)):
            return len(other) == len(self) and all(a == b for a, b in zip(self, other))
        return NotImplemented

    __hash__ = None

    def __repr__(self):
        return f"DMatchList({len(self)} matches)"


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


_STAGE = {}


def _pinned_stage(n_bytes):
    """A page-locked uint8 buffer of at least n_bytes, kept across calls (grow-only, doubled on growth)."""
    import torch
    buf = _STAGE.get("buf")
    if buf is None or buf.numel() < n_bytes:
        size = max(n_bytes, 2 * (buf.numel() if buf is not None else 0), 1 << 20)
        buf = torch.empty(size, dtype=torch.uint8, pin_memory=True)
        _STAGE["buf"] = buf
    return buf


def _upload_sets(arrs, dev):
    """Descriptor sets of several images -> one device array (rows back to back) and its row offsets."""
    import torch
    dim = arrs[0].shape[1]
    if any(a.ndim != 2 or a.shape[1] != dim or a.dtype != arrs[0].dtype for a in arrs):
        raise ValueError("descriptor sets must be [n, dim] arrays of one dtype and dim")
    ptr = np.zeros(len(arrs) + 1, dtype=np.int64)
    np.cumsum([a.shape[0] for a in arrs], out=ptr[1:])
    # rows back to back straight into a page-locked staging buffer (kept and grown across calls), one asynchronous copy from
    # there: the concatenation into pageable memory plus its staged upload were 0.3 ms of a 0.8 ms call at 18 x 2,000 x 128
    n_bytes = int(ptr[-1]) * dim * arrs[0].dtype.itemsize
    stage = _pinned_stage(max(n_bytes, 1))
    host = stage[:n_bytes].numpy().view(arrs[0].dtype).reshape(int(ptr[-1]), dim)
    if int(ptr[-1]) > 0:
        np.concatenate(arrs, axis=0, out=host)
    allrows = torch.from_numpy(host).to(dev, non_blocking=True)
    torch.cuda.current_stream(dev).synchronize()          # the staging buffer is reused by the next call
    return allrows, ptr, dim


def _run_batch(h, dev, rows, ptr, code, dim, seg_pairs, ratio):
    """One sfm_match_knn2_batched + sfm_match_ratio_batched over the image pairs `seg_pairs` (indices into ptr)."""
    import torch
    n_seg = len(seg_pairs)
    q_beg = np.array([ptr[i] for i, _ in seg_pairs], dtype=np.int64)
    q_end = np.array([ptr[i + 1] for i, _ in seg_pairs], dtype=np.int64)
    t_beg = np.array([ptr[j] for _, j in seg_pairs], dtype=np.int64)
    t_end = np.array([ptr[j + 1] for _, j in seg_pairs], dtype=np.int64)
    hp = lambda a: C.c_void_p(a.ctypes.data)
    n_out, need = C.c_int64(), C.c_int64()
    n_rows = int(rows.shape[0])
    h.check(h.lib.sfm_match_batched_workspace_bytes(code, n_seg, hp(q_beg), hp(q_end), hp(t_beg), hp(t_end), n_rows, n_rows,
                                                    C.byref(n_out), C.byref(need)), "sfm_match_batched_workspace_bytes")
    n = n_out.value
    ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    idx1 = torch.empty(n, dtype=torch.int32, device=dev)
    idx2 = torch.empty(n, dtype=torch.int32, device=dev)
    d1 = torch.empty(n, dtype=torch.float32, device=dev)
    d2 = torch.empty(n, dtype=torch.float32, device=dev)
    out_ptr = torch.empty(n_seg + 1, dtype=torch.int64, device=dev)
    dp = lambda t: C.c_void_p(t.data_ptr())
    h.call("sfm_match_knn2_batched", code, dp(rows), n_rows, dp(rows), n_rows, dim, n_seg, hp(q_beg), hp(q_end), hp(t_beg),
           hp(t_end), dp(idx1), dp(idx2), dp(d1), dp(d2), dp(out_ptr), dp(ws), need.value)
    qi = torch.empty(n, dtype=torch.int32, device=dev)
    ti = torch.empty(n, dtype=torch.int32, device=dev)
    dd = torch.empty(n, dtype=torch.float32, device=dev)
    seg_ptr = torch.empty(n_seg + 1, dtype=torch.int64, device=dev)
    h.call("sfm_match_ratio_batched", n, n_seg, dp(out_ptr), dp(idx1), dp(d1), dp(d2), C.c_double(ratio), dp(qi), dp(ti),
           dp(dd), dp(seg_ptr), dp(ws), need.value)
    sp = seg_ptr.cpu().numpy()
    m = int(sp[-1])
    qh, th, dh = qi[:m].cpu().numpy(), ti[:m].cpu().numpy(), dd[:m].cpu().numpy()
    # disjoint views of the three result arrays (444 copies were 0.25 ms of a 1.2 ms call)
    return [(qh[int(sp[k]):int(sp[k + 1])], th[int(sp[k]):int(sp[k + 1])], dh[int(sp[k]):int(sp[k + 1])]) for k in range(n_seg)]


def match_pairs(descs, pairs, ratio=0.75, metric="auto", device=0):
    """match_features for MANY image pairs in one launch (sfm_match_knn2_batched + sfm_match_ratio_batched).

    descs: list of per-image descriptor arrays (None = an image without keypoints, as cv2 returns it); pairs: list of
    (i, j) = match image i's descriptors (queries) against image j's (train), exactly what the reference does once per
    pair in its serial loop (find_matches.py:329-350, call at :272).  Returns one entry per pair, each what
    match_arrays(descs[i], descs[j]) gives for THAT pair: a (queryIdx, trainIdx, distance) triple of NumPy arrays, bit
    for bit - empty arrays for a pair with an empty side, and the ValueError of the reference's unpacking (:151) AS THE
    ENTRY (not raised) for a pair whose train image has exactly one descriptor: the reference's per-pair try / except
    (:344-350) skips that pair only.  Only images some live pair refers to are uploaded.  The exact uint8 path is decided
    per image: float32 sets whose values are all integers in [0, 255] pair up on the matrix cores, pairs touching any
    other float set run the float32 kernel (one launch per group; on integer-valued rows both kernels give the same bits:
    sums below 2^24 are exact in float32)."""
    import torch
    empty = (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32))
    pairs = [(int(i), int(j)) for i, j in pairs]
    sizes = [0 if d is None else int(np.asarray(d).shape[0]) for d in descs]
    out = [empty] * len(pairs)
    live = []
    for s, (i, j) in enumerate(pairs):
        if sizes[i] > 0 and sizes[j] == 1:
            out[s] = ValueError("not enough values to unpack (expected 2, got 1)")
        elif sizes[i] > 0 and sizes[j] >= 2:
            live.append(s)
    if not live:
        return out
    h = _lib.get_handle(device)
    dev = torch.device("cuda", device)
    used = sorted({i for s in live for i in pairs[s]})
    slot = {img: k for k, img in enumerate(used)}
    rows, ptr, dim = _upload_sets([np.ascontiguousarray(descs[i]) for i in used], dev)
    metric = _resolve_metric(rows, rows, metric)
    seg = [(slot[pairs[s][0]], slot[pairs[s][1]]) for s in live]
    groups = []                                            # (rows, metric code, positions in `live`)
    if metric == "hamming":
        if rows.dtype != torch.uint8:
            raise ValueError("hamming needs uint8 descriptors")
        groups.append((rows, _lib.METRIC_HAMMING, list(range(len(live)))))
    elif rows.dtype == torch.uint8:
        groups.append((rows, _lib.METRIC_L2_U8, list(range(len(live)))) if dim in (32, 64, 128)
                      else (rows.float(), _lib.METRIC_L2_F32, list(range(len(live)))))
    else:
        rows = rows.float()
        if dim in (32, 64, 128):
            # per image: every value an integer in [0, 255]?  (what SIFT emits)
            row_ok = ((rows == rows.round()) & (rows >= 0) & (rows <= 255)).all(dim=1)
            csum = torch.zeros(rows.shape[0] + 1, dtype=torch.int64, device=dev)
            torch.cumsum(row_ok.to(torch.int64), 0, out=csum[1:])
            tp = torch.from_numpy(ptr).to(dev)
            img_ok = ((csum[tp[1:]] - csum[tp[:-1]]) == (tp[1:] - tp[:-1])).cpu().numpy()
            exact = [k for k, (a, b) in enumerate(seg) if img_ok[a] and img_ok[b]]
            other = [k for k, (a, b) in enumerate(seg) if not (img_ok[a] and img_ok[b])]
            if exact:
                groups.append((rows.to(torch.uint8), _lib.METRIC_L2_U8, exact))     # rows of other images are never read by this group
            if other:
                groups.append((rows, _lib.METRIC_L2_F32, other))
        else:
            groups.append((rows, _lib.METRIC_L2_F32, list(range(len(live)))))
    for grows, code, pos in groups:
        for k, res in zip(pos, _run_batch(h, dev, grows, ptr, code, dim, [seg[k] for k in pos], ratio)):
            out[live[k]] = res
    return out


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
        """Returns a DMatchList: the reference's list of cv2.DMatch as a lazy sequence (see DMatchList)."""
        q, t, d = match_arrays(desc1, desc2, self.ratio, self.metric, self.device)
        return DMatchList(q, t, d)

    def match_features_batched(self, descs, pairs):
        """match_features(descs[i], descs[j]) for every (i, j) of `pairs` in one launch: the list the reference's
        pair loop (find_matches.py:329-350) would have collected call by call.  A pair the reference's per-pair
        try / except (:344-350) would have logged and skipped (one train descriptor) is logged and yields None."""
        import logging
        out = []
        for (i, j), res in zip(pairs, match_pairs(descs, pairs, self.ratio, self.metric, self.device)):
            if isinstance(res, Exception):
                logging.error(f"Error processing pair ({i}, {j}): {res}")
                out.append(None)
                continue
            out.append(DMatchList(*res))
        return out
