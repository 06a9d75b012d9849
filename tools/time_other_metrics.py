import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
from sfm_amd import matcher
rng = np.random.default_rng(0)
for n in (2000, 10000, 30000):
    q = torch.from_numpy(rng.integers(0, 256, size=(n, 32), dtype=np.uint8)).cuda()
    t = torch.from_numpy(rng.integers(0, 256, size=(n, 32), dtype=np.uint8)).cuda()
    for _ in range(3): matcher.knn2(q, t, "hamming")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): matcher.knn2(q, t, "hamming")
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"hamming n={n}: {dt*1e6:.1f} us per call, {n*n/dt:.3e} pairs/s", flush=True)
    qf = torch.from_numpy(rng.random((n, 128), dtype=np.float32)).cuda(); tf = torch.from_numpy(rng.random((n, 128), dtype=np.float32)).cuda()
    for _ in range(2): matcher.knn2(qf, tf, "l2")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): matcher.knn2(qf, tf, "l2")
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"l2 f32 n={n}: {dt*1e6:.1f} us per call, {n*n/dt:.3e} pairs/s", flush=True)
