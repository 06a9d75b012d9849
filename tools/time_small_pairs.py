#!/usr/bin/env python3
"""Per-call cost of matcher.match_arrays on image-sized descriptor sets (host arrays in, match arrays out)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import synth, matcher as mt
rng = np.random.default_rng(7)
for per_image in (500, 2000):
    base, _ = synth.make_descriptors(per_image + 200, 2, seed=77)
    imgs = [np.clip(base[rng.permutation(base.shape[0])[:per_image]] + np.rint(rng.normal(0, 5.0, size=(per_image, 128))), 0, 255).astype(np.uint8) for _ in range(6)]
    pairs = [(i, j) for i in range(6) for j in range(i + 1, 6)]
    [mt.match_arrays(imgs[i], imgs[j]) for i, j in pairs[:4]]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        [mt.match_arrays(imgs[i], imgs[j]) for i, j in pairs]
    torch.cuda.synchronize()
    print(per_image, "match_arrays us per call", (time.perf_counter() - t0) / (4 * len(pairs)) * 1e6, flush=True)
    q = torch.from_numpy(imgs[0]).cuda(); t = torch.from_numpy(imgs[1]).cuda()
    for _ in range(3): mt.knn2(q, t, "l2")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): mt.knn2(q, t, "l2")
    torch.cuda.synchronize()
    print(per_image, "knn2 us per call", (time.perf_counter() - t0) / 50 * 1e6, flush=True)
