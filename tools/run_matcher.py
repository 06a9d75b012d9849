#!/usr/bin/env python3
"""Run only the matcher kernels (50k x 50k x 128 uint8) a few times - a small target for rocprofv3 --pmc."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import synth, matcher
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
d1, d2 = synth.make_descriptors(n, n, seed=1002)
q = torch.from_numpy(d1.astype(np.uint8)).cuda(); t = torch.from_numpy(d2.astype(np.uint8)).cuda()
for _ in range(4):
    matcher.knn2(q, t, "l2")
torch.cuda.synchronize()
