#!/usr/bin/env python3
"""Time the matcher's kNN kernel alone (HIP events through the handle's profiling slots).
usage: python tools/time_matcher.py [n=50000] [dim=128]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import synth, matcher, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
d1, d2 = synth.make_descriptors(n, n, seed=1002, dim=dim)
q = torch.from_numpy(d1.astype(np.uint8)).cuda(); t = torch.from_numpy(d2.astype(np.uint8)).cuda()
h = _lib.get_handle(0)
for _ in range(2):
    matcher.knn2(q, t, "l2")
h.set_profiling(True); h.profile()
for _ in range(10):
    matcher.knn2(q, t, "l2")
ms, cnt = h.profile()["knn"]
us = ms / cnt * 1e3
print(f"k_knn2 {n}x{n}x{dim}: {us:.1f} us per launch, {n * n / (us * 1e-6):.3e} pairs/s, "
      f"{2.0 * dim * n * n / (us * 1e-6) / 1e12:.0f} TOP/s = {2.0 * dim * n * n / (us * 1e-6) / 5e15:.3f} of the i8 MFMA peak")
