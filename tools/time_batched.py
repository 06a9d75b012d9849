#!/usr/bin/env python3
"""Where the time of one batched matching call goes (148 image pairs, 500 / 2000 descriptors per image)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import synth, matcher as mt
rng = np.random.default_rng(7)
for per_image in (500, 2000):
    base, _ = synth.make_descriptors(per_image + 200, 2, seed=77)
    imgs = [np.clip(base[rng.permutation(base.shape[0])[:per_image]] + np.rint(rng.normal(0, 5.0, size=(per_image, 128))), 0, 255).astype(np.uint8) for _ in range(18)]
    pairs = [(i, j) for i in range(18) for j in range(i + 1, 18)][:148]
    for _ in range(3): mt.match_pairs(imgs, pairs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): mt.match_pairs(imgs, pairs)
    torch.cuda.synchronize()
    print(per_image, "match_pairs ms per call", (time.perf_counter() - t0) / 10 * 1e3, flush=True)
