#!/usr/bin/env python3
"""Lifetimes of the waves of k_knn2_u8_direct (50k x 50k): needs a library built with -DSFM_MATCH_STAMPS=1
(tools/exp_matcher_lifetimes.sh builds it and sets SFM_AMD_LIB).  Prints, per launch, the spread of begin / end stamps
(100 MHz clock) and how lifetimes depend on the order in which the two workgroups of a CU arrived."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sfm_amd import synth, matcher, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
d1, d2 = synth.make_descriptors(n, n, seed=1002)
q = torch.from_numpy(d1.astype(np.uint8)).cuda(); t = torch.from_numpy(d2.astype(np.uint8)).cuda()
for _ in range(3):
    matcher.knn2(q, t, "l2")
torch.cuda.synchronize()
lib = _lib.get_handle(0).lib
nwg = 4096
buf = np.zeros(nwg * 4 * 4, dtype=np.uint64)
lib.sfm_debug_match_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.sfm_debug_match_stamps(buf.ctypes.data, buf.size)
assert rc == 0, rc
st = buf.reshape(nwg, 4, 4)
used = st[:, 0, 1] > 0
blk = np.nonzero(used)[0]
st = st[used]
# a wave that left before its first stamp (q0 >= nq) leaves an old begin stamp: keep waves of THIS launch only
latest = st[:, :, 1].max()
fresh = (st[:, :, 1] > latest - 100000) & (st[:, :, 0] > latest - 100000)          # within 1 ms of the last end stamp
keep = fresh.all(axis=1)
print("workgroups with a stale wave stamp (partly outside the query range):", int((~keep).sum()))
st, blk = st[keep], blk[keep]
print("workgroups with stamps:", st.shape[0])
t0 = st[:, :, 0].astype(np.int64); t1 = st[:, :, 1].astype(np.int64)
base = t0.min()
beg = (t0 - base) / 100.0; end = (t1 - base) / 100.0            # us
life = end - beg
print("kernel span %.1f us; wave begin: min %.1f max %.1f; wave end: p5 %.1f p50 %.1f p95 %.1f max %.1f" % (
    end.max(), beg.min(), beg.max(), *np.percentile(end, [5, 50, 95]), end.max()))
print("wave lifetime us: mean %.1f  p5 %.1f p25 %.1f p50 %.1f p75 %.1f p95 %.1f max %.1f -> mean / span = %.3f" % (
    life.mean(), *np.percentile(life, [5, 25, 50, 75, 95]), life.max(), life.mean() / end.max()))
hw = st[:, 0, 2].astype(np.int64); xcc = st[:, 0, 3].astype(np.int64) & 0xF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
place = xcc * 1000 + se * 100 + sh * 10 + cu
wg_end = end.max(axis=1); wg_beg = beg.min(axis=1)
from collections import defaultdict
per = defaultdict(list)
for i, pl in enumerate(place):
    per[pl].append((wg_beg[i], wg_end[i], int(blk[i])))
# hypothesis: position in the XCD's dispatch sequence (blockIdx >> 3) below the XCD's 32 CUs <=> first workgroup on its CU
hit = tot = 0
for v in per.values():
    v.sort()
    if len(v) == 2:
        tot += 2
        hit += int((v[0][2] >> 3) < 32) + int((v[1][2] >> 3) >= 32)
print("first / second on a CU predicted by (blockIdx >> 3) < 32: %d of %d" % (hit, tot))
cnt = np.bincount([len(v) for v in per.values()])
print("CUs used:", len(per), "workgroups per CU histogram:", cnt.tolist())
first, second, alone = [], [], []
for v in per.values():
    v.sort()
    if len(v) == 1:
        alone.append(v[0][1] - v[0][0])
    else:
        first.append(v[0][1] - v[0][0]); second.append(v[1][1] - v[1][0])
for name, a in (("alone on its CU", alone), ("first of two", first), ("second of two", second)):
    if a:
        a = np.array(a); print("%-16s n=%3d lifetime mean %.1f p5 %.1f p95 %.1f" % (name, a.size, a.mean(), *np.percentile(a, [5, 95])))
for x in range(8):
    m = xcc == x
    print("XCC %d: %3d workgroups, end mean %.1f max %.1f" % (x, m.sum(), wg_end[m].mean() if m.any() else 0, wg_end[m].max() if m.any() else 0))
