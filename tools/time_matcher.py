#!/usr/bin/env python3
"""Time the kNN(2) launch (50k x 50k x 128 uint8 by default) under the tuning knobs given in the environment."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import synth, matcher
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nq = int(sys.argv[3]) if len(sys.argv) > 3 else n          # optional: fewer queries than train rows
d1, d2 = synth.make_descriptors(n, n, seed=1002)
d1 = d1[:nq]
q = torch.from_numpy(d1.astype(np.uint8)).cuda(); t = torch.from_numpy(d2.astype(np.uint8)).cuda()
for _ in range(3):
    matcher.knn2(q, t, "l2")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    matcher.knn2(q, t, "l2")
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
knobs = {k: v for k, v in os.environ.items() if k.startswith("SFM_MATCH_")}
print(f"nq={nq} nt={n} {knobs} {dt * 1e6:.1f} us per call  {nq * n / dt:.3e} pairs/s", flush=True)
