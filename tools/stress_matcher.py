#!/usr/bin/env python3
"""Randomised parity sweep of the uint8 kNN kernels against the C oracle (every row): sizes across the kernel switch
(12,288 and 28,672 queries), the filter switch (2,048 train rows), ragged tiles / windows / splits, SIFT-like, uniform, duplicate-heavy
and far-apart (float32 re-ranking) data, single pairs and batched segments.  Prints one line per case; exits 1 on a mismatch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ba_c
from sfm_amd import synth, matcher

def make(kind, nq, nt, rng):
    if kind == "sift":
        d1, d2 = synth.make_descriptors(nq, nt, seed=int(rng.integers(1 << 30)))
        return d1.astype(np.uint8), d2.astype(np.uint8)
    if kind == "uniform":
        return rng.integers(0, 256, size=(nq, 128), dtype=np.uint8), rng.integers(0, 256, size=(nt, 128), dtype=np.uint8)
    if kind == "far":
        q = np.zeros((nq, 128), np.uint8); t = np.full((nt, 128), 255, np.uint8)
        q[:, 96:] = 128 + rng.integers(0, 2, size=(nq, 32)); t[:, 96:] = 128 + rng.integers(0, 3, size=(nt, 32))
        return q, t
    base = rng.integers(0, 256, size=(50, 128), dtype=np.uint8)                     # "dups"
    t = base[rng.integers(0, 50, size=nt)]; q = base[rng.integers(0, 50, size=nq)].copy()
    q[::2] = np.clip(q[::2].astype(np.int32) + rng.integers(-2, 3, size=q[::2].shape), 0, 255).astype(np.uint8)
    return q, t

def same(got, ref):
    return all(np.array_equal(g, r) for g, r in zip(got, ref))

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
bad = 0
for c in range(n_cases):
    kind = ["sift", "uniform", "far", "dups"][c % 4]
    big = c % 3 != 2
    nq = int(rng.integers(12288, 30000)) if big else int(rng.integers(1, 12288))
    nt = int(rng.choice([rng.integers(2, 300), rng.integers(300, 2048), rng.integers(2048, 9000), rng.integers(9000, 26000)]))
    if kind == "far":
        nt = min(nt, 6000)                                                          # every query is re-ranked over all train rows
    q, t = make(kind, nq, nt, rng)
    t0 = time.perf_counter()
    i1, i2, a, b = matcher.knn2(torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda(), "l2")
    got = (i1.cpu().numpy(), i2.cpu().numpy(), a.cpu().numpy(), b.cpu().numpy())
    ok = same(got, ba_c.knn2_u8(q, t))
    bad += not ok
    print(f"case {c:2d} {kind:8s} nq={nq:6d} nt={nt:6d} {'ok' if ok else 'MISMATCH'}  {time.perf_counter() - t0:.2f}s", flush=True)
# Hamming (ORB) on 128 / 256 / 512 bits against the NumPy oracle: the first two run as uint8 L2 over unpacked bits
from oracle import matcher_oracle as mo
for c in range(6):
    nbytes = [16, 32, 64][c % 3]
    nq = int(rng.integers(1, 14000)); nt = int(rng.integers(2, 9000))
    b1 = rng.integers(0, 256, size=(nq, nbytes), dtype=np.uint8); b2 = rng.integers(0, 256, size=(nt, nbytes), dtype=np.uint8)
    k = min(nq, nt) // 2
    b2[:k] = b1[:k] ^ (rng.integers(0, 256, size=(k, nbytes), dtype=np.uint8) & rng.integers(0, 256, size=(k, nbytes), dtype=np.uint8))
    i1, i2, a, b = matcher.knn2(torch.from_numpy(b1).cuda(), torch.from_numpy(b2).cuda(), "hamming")
    ok = same((i1.cpu().numpy(), i2.cpu().numpy(), a.cpu().numpy(), b.cpu().numpy()), mo.knn2(b1, b2, "hamming"))
    bad += not ok
    print(f"hamming {c} bytes={nbytes} nq={nq} nt={nt} {'ok' if ok else 'MISMATCH'}", flush=True)
# batched segments, some with more than 2,048 train rows (filter on inside a batch)
for c in range(4):
    sizes = [int(rng.integers(50, 5000)) for _ in range(6)]
    imgs = [make(["sift", "dups", "uniform", "sift"][c], s, 2, rng)[0] for s in sizes]
    pairs = [(i, j) for i in range(6) for j in range(6) if i != j][:20]
    got = matcher.match_pairs(imgs, pairs)
    ok = True
    for (i, j), g in zip(pairs, got):
        r = matcher.match_arrays(imgs[i], imgs[j])
        ok = ok and all(np.array_equal(x, y) for x, y in zip(g, r))
        i1, i2, a, b = ba_c.knn2_u8(imgs[i], imgs[j])
        keep = a.astype(np.float64) < 0.75 * b.astype(np.float64)
        ok = ok and np.array_equal(g[0], np.nonzero(keep)[0]) and np.array_equal(g[1], i1[keep])
    bad += not ok
    print(f"batch {c} sizes={sizes} {'ok' if ok else 'MISMATCH'}", flush=True)
sys.exit(1 if bad else 0)
