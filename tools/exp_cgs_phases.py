#!/usr/bin/env python3
"""Where an iteration of the persistent camera CG spends its time (workgroup 0, thread 0; library built with -DSFM_CGS_STAMPS=1
by tools/exp_cgs_phases.sh and selected with SFM_AMD_LIB).  Stamps: 0 launch start, 1 matrix rows in registers, 2 product + wave
sums done (about to publish), 3 gather complete, 4 recurrences done."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import synth
from sfm_amd.ba import GpuBA
C_, P_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 100000)
sc = synth.make_scene(C_, P_, obs_per_point=10, seed=1004, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002)
be = GpuBA(sc.cams0, sc.pts0, sc.cam_idx, sc.pt_idx, sc.uv, synth.K_REF)
st = be.trf_begin(max_nfev=2 ** 31 - 1, check_tolerances=False)
for _ in range(10):
    st.outer()
torch.cuda.synchronize()
lib = be.h.lib
buf = np.zeros(1 << 16, dtype=np.uint64)
used = ctypes.c_uint(0)
lib.sfm_debug_cgs_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_uint)]
assert lib.sfm_debug_cgs_stamps(buf.ctypes.data, buf.size, ctypes.byref(used)) == 0
n = min(int(used.value), buf.size) // 2
rec = buf[:2 * n].reshape(n, 2).astype(np.int64)
tags, t = rec[:, 0], rec[:, 1] / 100.0                  # us
launch = np.cumsum(tags == 0) - 1
d = {"start -> rows": [], "product + wave sums (4 -> 2, or 1 -> 2)": [], "publish -> gather complete (2 -> 3)": [], "recurrences (3 -> 4)": []}
its, span = [], []
for L in range(launch.max() + 1):
    sel = np.flatnonzero(launch == L)
    tg, tt = tags[sel], t[sel]
    its.append(int((tg == 4).sum()))
    span.append(tt[-1] - tt[0])
    for a in range(1, len(sel)):
        pa, pb = tg[a - 1], tg[a]
        dt = tt[a] - tt[a - 1]
        if (pa, pb) == (0, 1): d["start -> rows"].append(dt)
        elif pb == 2 and pa in (1, 4): d["product + wave sums (4 -> 2, or 1 -> 2)"].append(dt)
        elif (pa, pb) == (2, 3): d["publish -> gather complete (2 -> 3)"].append(dt)
        elif (pa, pb) == (3, 4): d["recurrences (3 -> 4)"].append(dt)
print("launches %d, iterations per launch mean %.1f, stamped span per launch mean %.1f us" % (len(its), np.mean(its), np.mean(span)))
for k, v in d.items():
    v = np.array(v)
    print("%-44s n %6d  mean %6.2f us  p50 %6.2f  p95 %6.2f" % (k, len(v), v.mean(), np.percentile(v, 50), np.percentile(v, 95)))
st.close()
