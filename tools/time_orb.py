#!/usr/bin/env python3
"""Time kNN(2) on ORB-like 256-bit descriptors (Hamming): python tools/time_orb.py [n] [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import matcher
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(1006)
oq = torch.from_numpy(rng.integers(0, 256, size=(n, 32), dtype=np.uint8)).cuda()
ot = torch.from_numpy(rng.integers(0, 256, size=(n, 32), dtype=np.uint8)).cuda()
for _ in range(5):
    r = matcher.knn2(oq, ot, "hamming")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    r = matcher.knn2(oq, ot, "hamming")
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
knobs = {k: v for k, v in os.environ.items() if k.startswith("SFM_MATCH_")}
print(f"n={n} 256-bit Hamming {knobs} {dt * 1e6:.1f} us per call  {n * n / dt:.3e} pairs/s  checksum {int(r[0].sum())} {int(r[1].sum())} {float(r[2].sum()):.1f} {float(r[3].sum()):.1f}", flush=True)
