"""Host-side index structures of a bundle-adjustment problem (NumPy, built once per problem).

The reference packs observations point-major (/root/reference/utils/sfm_reconstruction.py:430-435);
everything here is derived from that order: the per-point track ranges, the per-camera
observation lists and the camera-pair lists that drive the Schur-complement kernel, plus the
point sharding used when the problem is split over several GPUs.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
import numpy as np


@dataclass
class BAStructure:
    n_cams: int
    n_pts: int
    cam_idx: np.ndarray     # [N] int32
    pt_idx: np.ndarray      # [N] int32, non-decreasing
    pt_ptr: np.ndarray      # [P+1] int32
    cam_ptr: np.ndarray     # [C+1] int32
    cam_obs: np.ndarray     # [N] int32
    blk_ptr: np.ndarray     # [C(C+1)/2 + 1] int32
    pair_k: np.ndarray      # [n_pairs] int32
    pair_k2: np.ndarray     # [n_pairs] int32
    item_ptr: np.ndarray    # [C(C+1)/2 + 1] int32: work items (<= ITEM_PAIRS pairs of ONE block) per block
    item_beg: np.ndarray    # [n_items] int32 ranges into pair_k / pair_k2
    item_end: np.ndarray    # [n_items] int32
    xcd_ptr: np.ndarray     # [9] int32: items of block rows c = x (mod 8) are xcd_items[xcd_ptr[x]:xcd_ptr[x+1]]
    xcd_items: np.ndarray   # [n_items] int32: item ids grouped by c mod 8 (an XCD keeps one camera's G in its L2)
    cch_ptr: np.ndarray     # [C+1] int32: chunks (<= CHUNK_OBS entries of cam_obs of ONE camera) per camera
    cch_beg: np.ndarray     # [n_cchunks] int32 ranges into cam_obs
    cch_end: np.ndarray     # [n_cchunks] int32

    @property
    def n_obs(self):
        return int(self.cam_idx.shape[0])

    @property
    def n_pairs(self):
        return int(self.pair_k.shape[0])

    @property
    def n_items(self):
        return int(self.item_beg.shape[0])

    @property
    def n_cchunks(self):
        return int(self.cch_beg.shape[0])


ITEM_PAIRS = 256     # pairs per Schur work item (one wavefront)
CHUNK_OBS = 256      # observations per camera-major reduction chunk (one workgroup)


def _split_ranges(ptr, size):
    """Split every CSR range [ptr[i], ptr[i+1]) into pieces of at most `size`.
    Returns (piece_ptr [len(ptr)], beg [n_pieces], end [n_pieces])."""
    ptr = np.asarray(ptr, dtype=np.int64)
    cnt = np.diff(ptr)
    npiece = (cnt + size - 1) // size
    piece_ptr = np.zeros(len(ptr), dtype=np.int64)
    np.cumsum(npiece, out=piece_ptr[1:])
    total = int(piece_ptr[-1])
    owner = np.repeat(np.arange(len(cnt), dtype=np.int64), npiece)
    within = np.arange(total, dtype=np.int64) - np.repeat(piece_ptr[:-1], npiece)
    beg = ptr[owner] + within * size
    end = np.minimum(beg + size, ptr[owner + 1])
    return piece_ptr, beg, end


def block_index(c, c2, n_cams):
    """Linear index of the upper-triangular camera pair (c <= c2)."""
    c = np.asarray(c, dtype=np.int64)
    c2 = np.asarray(c2, dtype=np.int64)
    return c * n_cams - c * (c - 1) // 2 + (c2 - c)


def build_structure(cam_idx, pt_idx, n_cams, n_pts):
    cam_idx = np.ascontiguousarray(cam_idx, dtype=np.int64)
    pt_idx = np.ascontiguousarray(pt_idx, dtype=np.int64)
    N = cam_idx.shape[0]
    if N == 0:
        raise ValueError("no observations")
    if np.any(np.diff(pt_idx) < 0):
        raise ValueError("observations must be point-major (pt_idx non-decreasing)")
    if cam_idx.min() < 0 or cam_idx.max() >= n_cams or pt_idx.min() < 0 or pt_idx.max() >= n_pts:
        raise ValueError("index out of range")
    pt_ptr = np.zeros(n_pts + 1, dtype=np.int64)
    np.cumsum(np.bincount(pt_idx, minlength=n_pts), out=pt_ptr[1:])
    cam_ptr = np.zeros(n_cams + 1, dtype=np.int64)
    np.cumsum(np.bincount(cam_idx, minlength=n_cams), out=cam_ptr[1:])
    cam_obs = np.argsort(cam_idx, kind="stable")
    # every ordered pair (k, k2) of observations on one track with cam(k) < cam(k2), and all
    # ordered pairs (k == k2 included) with cam(k) == cam(k2)
    L = (pt_ptr[1:] - pt_ptr[:-1])[pt_idx]                 # track length seen from each observation
    total = int(L.sum())
    k = np.repeat(np.arange(N, dtype=np.int64), L)
    first = np.cumsum(L) - L
    k2 = np.repeat(pt_ptr[pt_idx], L) + (np.arange(total, dtype=np.int64) - np.repeat(first, L))
    ck, ck2 = cam_idx[k], cam_idx[k2]
    keep = ck <= ck2
    k, k2, ck, ck2 = k[keep], k2[keep], ck[keep], ck2[keep]
    blk = block_index(ck, ck2, n_cams)
    order = np.argsort(blk, kind="stable")
    nblk = n_cams * (n_cams + 1) // 2
    blk_ptr = np.zeros(nblk + 1, dtype=np.int64)
    np.cumsum(np.bincount(blk, minlength=nblk), out=blk_ptr[1:])
    if blk_ptr[-1] >= 2 ** 31 or N >= 2 ** 31:
        raise ValueError("problem too large for int32 indices")
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    item_ptr, item_beg, item_end = _split_ranges(blk_ptr, ITEM_PAIRS)
    cch_ptr, cch_beg, cch_end = _split_ranges(cam_ptr, CHUNK_OBS)
    # block row of every item -> 8 groups (stable: rows, then blocks, stay in order inside a group): rows dealt back and forth, or with
    # SFM_XCD_GROUP=contig contiguous row ranges balanced by item count (the same rule as sfm_amd/csrc/problem.hip)
    n_blk_row = np.arange(n_cams, 0, -1, dtype=np.int64)                 # row c holds blocks (c, c..C-1)
    blk_row = np.repeat(np.arange(n_cams, dtype=np.int64), n_blk_row)
    item_row = np.repeat(blk_row, np.diff(item_ptr))
    # ... unless SFM_XCD_GROUP says which, a locality statistic decides (problem.hip, k_band_stat): the mean distance from the
    # diagonal of the non-empty off-diagonal blocks against the C / 3 of evenly spread ones; below 0.8 x: contiguous ranges
    grp_env = os.environ.get("SFM_XCD_GROUP")
    contig = grp_env is not None and grp_env[:1] == "c"
    if grp_env is None and n_cams >= 16:
        blk_col = np.concatenate([np.arange(c, n_cams, dtype=np.int64) for c in range(n_cams)])
        work = (np.diff(item_ptr) > 0) & (blk_col > blk_row)
        if work.any():
            contig = bool(int((blk_col - blk_row)[work].sum()) / int(work.sum()) < 0.8 * (n_cams / 3.0))
    if contig:
        row_cnt = np.bincount(item_row, minlength=n_cams).astype(np.int64)
        total = int(row_cnt.sum())
        mid = np.cumsum(row_cnt) - row_cnt + row_cnt // 2
        row_grp = np.minimum((mid * 8) // max(total, 1), 7) if total > 0 else np.zeros(n_cams, dtype=np.int64)
    else:
        r = np.arange(n_cams, dtype=np.int64)
        row_grp = np.where(r & 8, 7 - (r & 7), r & 7)        # dealt back and forth: balances the shrinking rows over the groups
    item_grp = row_grp[item_row]
    xcd_items = np.argsort(item_grp, kind="stable")
    xcd_ptr = np.zeros(9, dtype=np.int64)
    np.cumsum(np.bincount(item_grp, minlength=8), out=xcd_ptr[1:])
    return BAStructure(int(n_cams), int(n_pts), i32(cam_idx), i32(pt_idx), i32(pt_ptr), i32(cam_ptr),
                       i32(cam_obs), i32(blk_ptr), i32(k[order]), i32(k2[order]),
                       i32(item_ptr), i32(item_beg), i32(item_end), i32(xcd_ptr), i32(xcd_items),
                       i32(cch_ptr), i32(cch_beg), i32(cch_end))


def partition_points(pt_ptr, world_size):
    """Contiguous point ranges [lo, hi) per rank, balanced by observation count.

    Points keep all their observations on one rank, so per-point blocks
    never cross ranks and only the reduced camera system is exchanged.
    """
    pt_ptr = np.asarray(pt_ptr, dtype=np.int64)
    P = pt_ptr.shape[0] - 1
    N = int(pt_ptr[-1])
    bounds = [0]
    for r in range(1, world_size):
        target = N * r / world_size
        j = int(np.searchsorted(pt_ptr, target, side="left"))
        j = min(max(j, bounds[-1]), P)
        bounds.append(j)
    bounds.append(P)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def shard_arrays(cam_idx, pt_idx, uv, pts, lo, hi):
    """Observations / points of the point range [lo, hi) with point ids rebased to 0."""
    pt_idx = np.asarray(pt_idx)
    a = int(np.searchsorted(pt_idx, lo, side="left"))
    b = int(np.searchsorted(pt_idx, hi, side="left"))
    return (np.asarray(cam_idx)[a:b].copy(), (pt_idx[a:b] - lo).copy(), np.asarray(uv)[a:b].copy(),
            np.asarray(pts)[lo:hi].copy())
