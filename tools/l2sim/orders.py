#!/usr/bin/env python3
"""Alternative item orders for k_schur_items, evaluated in the offline L2 model (tools/l2sim/l2sim.c).
usage: orders.py <visibility> [cams pts]"""
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
exe = "/tmp/l2sim"
subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(here, "l2sim.c")], check=True)
n_blk_row = np.arange(C_, 0, -1)
blk_row = np.repeat(np.arange(C_), n_blk_row)
blk_col = np.concatenate([np.arange(c, C_) for c in range(C_)])
it_row = np.repeat(blk_row, np.diff(st.item_ptr)); it_col = np.repeat(blk_col, np.diff(st.item_ptr))
n_items = st.n_items
row_cnt = np.bincount(it_row, minlength=C_)
mid = np.cumsum(row_cnt) - row_cnt + row_cnt // 2
contig = np.minimum(mid * 8 // n_items, 7)
r = np.arange(C_)
mod8 = np.where(r & 8, 7 - (r & 7), r & 7)

def run(name, grp_of_row, key):
    grp = grp_of_row[it_row]
    order = np.lexsort(tuple(reversed([grp] + key)))           # primary: group, then the keys in order
    xp = np.zeros(9, dtype=np.int32); np.cumsum(np.bincount(grp, minlength=8), out=xp[1:])
    out = "/tmp/l2sim_o.bin"
    with open(out, "wb") as f:
        np.array([st.n_pairs, n_items, 2, 0], dtype=np.int64).tofile(f)
        for a in (st.pair_k, st.pair_k2, st.item_beg, st.item_end, xp, order):
            np.ascontiguousarray(a, dtype=np.int32).tofile(f)
    print("%-46s" % name, end=" ", flush=True)
    subprocess.run([exe, out, "640", "4"], check=True)

ids = np.arange(n_items)
run("shipped: rows mod 8, row-major", mod8, [it_row, it_col, ids])
run("contiguous rows, row-major", contig, [it_row, it_col, ids])
run("contiguous rows, column-major", contig, [it_col, it_row, ids])
for T in (4, 8, 16):
    run("contiguous rows, column tiles of %d, row-major" % T, contig, [it_col // T, it_row, it_col, ids])
    run("rows mod 8, column tiles of %d, row-major" % T, mod8, [it_col // T, it_row, it_col, ids])
for T in (2, 4, 8):
    run("contiguous rows, row tiles of %d, column-major" % T, contig, [it_row // T, it_col, it_row, ids])
