#!/usr/bin/env python3
"""Where a persistent camera-CG launch spends its time (workgroup 0, thread 0; library built with -DSFM_CGS_STAMPS=1 and
selected with SFM_AMD_LIB): start -> rhs scaled -> ||rhs||^2 -> matrix rows arrived -> first iteration -> last -> end."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import synth
from sfm_amd.ba import GpuBA
sc = synth.make_scene(200, 100000, obs_per_point=10, seed=1004, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002)
be = GpuBA(sc.cams0, sc.pts0, sc.cam_idx, sc.pt_idx, sc.uv, synth.K_REF)
st = be.trf_begin(max_nfev=2 ** 31 - 1, check_tolerances=False)
for _ in range(12):
    st.outer()
torch.cuda.synchronize()
lib = be.h.lib
buf = np.zeros(8 * 4096, dtype=np.uint64)
lib.sfm_debug_cgs_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.sfm_debug_cgs_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(4096, 8)
s = s[s[:, 0] > 0].astype(np.int64)
d = np.diff(s[:, :7], axis=1) / 100.0          # us
its = s[:, 7]
names = ["scale rhs", "||rhs||^2", "rows arrived", "to first iteration", "iterations", "epilogue"]
print("launches:", s.shape[0], "iterations mean %.1f" % its.mean())
for i, nme in enumerate(names):
    print("%-20s mean %6.2f us  p50 %6.2f  p95 %6.2f" % (nme, d[:, i].mean(), np.percentile(d[:, i], 50), np.percentile(d[:, i], 95)))
print("per iteration: %.2f us" % (d[:, 4].sum() / max(its.sum(), 1)))
st.close()
