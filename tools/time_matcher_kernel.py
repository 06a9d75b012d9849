#!/usr/bin/env python3
"""HIP-event time of the distance kernel alone (profiling slot "knn") for the 50k x 50k x 128 uint8 case, under the tuning
knobs given in the environment.  python tools/time_matcher_kernel.py [n] [reps] [nq]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import synth, matcher, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nq = int(sys.argv[3]) if len(sys.argv) > 3 else n
d1, d2 = synth.make_descriptors(n, n, seed=1002)
q = torch.from_numpy(d1[:nq].astype(np.uint8)).cuda(); t = torch.from_numpy(d2.astype(np.uint8)).cuda()
for _ in range(3):
    matcher.knn2(q, t, "l2")
torch.cuda.synchronize()
h = _lib.get_handle(0)
h.set_profiling(True); h.profile()
for _ in range(reps):
    matcher.knn2(q, t, "l2")
torch.cuda.synchronize()
ms, launches = h.profile()["knn"]
h.set_profiling(False)
knobs = {k: v for k, v in os.environ.items() if k.startswith("SFM_MATCH_")}
us = ms / launches * 1e3
print(f"nq={nq} nt={n} {knobs} distance kernel {us:.1f} us (HIP events, {launches} launches) = {2.0 * 128 * nq * n / (us * 1e-6) / 5e15:.3f} of 5 POP/s", flush=True)
