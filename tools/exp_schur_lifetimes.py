#!/usr/bin/env python3
"""Wave lifetimes of k_schur_items at cfg4 (needs a library built with -DSFM_SCHUR_STAMPS=1: tools/exp_schur_lifetimes.sh).
Prints how full the chip is over the span of ONE launch (resident waves over time) and how wave time depends on item size."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import synth, _lib
from sfm_amd.ba import GpuBA
C_, P_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 100000)
vis = sys.argv[3] if len(sys.argv) > 3 else "random"
sc = synth.make_scene(C_, P_, obs_per_point=10, seed=1004, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002, visibility=vis)
be = GpuBA(sc.cams0, sc.pts0, sc.cam_idx, sc.pt_idx, sc.uv, synth.K_REF)
_, gnorm, _, hd = be.linearize()
alpha = 1e-4 * hd
for _ in range(3):
    be.solve(alpha, True)
torch.cuda.synchronize()
lib = be.h.lib
NW = 1 << 16
buf = np.zeros(NW * 4, dtype=np.uint64)
lib.sfm_debug_schur_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.sfm_debug_schur_stamps(buf.ctypes.data, buf.size) == 0
st = buf.reshape(NW, 4)
st = st[st[:, 1] > 0]
latest = int(st[:, 1].max())
st = st[(st[:, 1].astype(np.int64) > latest - 200000)]          # this launch only (2 ms window)
t0 = st[:, 0].astype(np.int64); t1 = st[:, 1].astype(np.int64)
base = t0.min()
beg = (t0 - base) / 100.0; end = (t1 - base) / 100.0
npairs = (st[:, 2] >> np.uint64(32)).astype(np.int64)
life = end - beg
span = end.max()
print("waves:", st.shape[0], "span %.1f us" % span, "sum of wave lifetimes / (span x 5120 wave slots) = %.3f" % (life.sum() / (span * 5120)))
print("wave lifetime us: mean %.1f p5 %.1f p50 %.1f p95 %.1f max %.1f" % (life.mean(), *np.percentile(life, [5, 50, 95]), life.max()))
edges = np.linspace(0, span, 21)
res = [(np.minimum(end, b) - np.maximum(beg, a)).clip(min=0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
print("resident waves by twentieth of the span:", " ".join("%d" % r for r in res))
for lo, hi in ((1, 64), (65, 128), (129, 192), (193, 240), (241, 256)):
    m = (npairs >= lo) & (npairs <= hi)
    if m.any():
        print("items of %3d-%3d pairs: %6d waves, lifetime mean %.1f us, %.3f us per pair" % (lo, hi, m.sum(), life[m].mean(), (life[m] / npairs[m]).mean()))
hw = st[:, 2].astype(np.int64) & 0xFFFFFFFF
xcc = st[:, 3].astype(np.int64) & 0xF
for x in range(8):
    m = xcc == x
    print("XCC %d: %5d waves, pairs %8d, last end %.1f us" % (x, m.sum(), npairs[m].sum(), end[m].max() if m.any() else 0))
