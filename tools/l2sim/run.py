#!/usr/bin/env python3
"""Work lists of k_schur_items for a synthetic scene -> tools/l2sim/l2sim (an offline LRU model of an XCD's L2).
usage: run.py <visibility> [cams pts] [order]   order: shipped | track (pairs of a block sorted by track = by k)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sfm_amd import synth
from sfm_amd.structure import build_structure
vis = sys.argv[1]
C_, P_ = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (200, 100000)
sc = synth.make_scene(C_, P_, obs_per_point=10, seed=1004, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002, visibility=vis)
st = build_structure(sc.cam_idx, sc.pt_idx, C_, P_)
here = os.path.dirname(os.path.abspath(__file__))
exe = os.path.join("/tmp", "l2sim")
subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(here, "l2sim.c")], check=True)
out = "/tmp/l2sim_%s.bin" % vis
with open(out, "wb") as f:
    np.array([st.n_pairs, st.n_items, 2, 0], dtype=np.int64).tofile(f)
    for a in (st.pair_k, st.pair_k2, st.item_beg, st.item_end, st.xcd_ptr, st.xcd_items):
        np.ascontiguousarray(a, dtype=np.int32).tofile(f)
print(vis, "pairs", st.n_pairs, "items", st.n_items, "items per XCD group", np.diff(st.xcd_ptr).tolist(), flush=True)
for W, mb in ((640, 4.0), (160, 4.0), (640, 8.0), (640, 16.0)):
    subprocess.run([exe, out, str(W), str(mb)], check=True)
