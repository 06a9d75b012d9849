#!/usr/bin/env python3
"""One trust-region case of tools/stress_ba.py looked at outer iteration by outer iteration: where the library's loop and the
dense Python oracle part ways, how fast, and whether the difference matters to the cost.  The oracle against ITSELF (dense
solver against point elimination - the same arithmetic in another order) is the yardstick for what rounding alone does.
usage: trf_drift_case.py <seed> <n_cases> <trf case>      (the arguments of the stress run that reported the case)"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import stress_ba
from oracle import ba_oracle as bo
from sfm_amd import synth
from sfm_amd.ba import GpuBA

seed, n_cases, want = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
for c in range(n_cases):                                  # replay the draws of the stage cases
    C = int(rng.choice([rng.integers(2, 9), rng.integers(9, 40), rng.integers(40, 130)]))
    P = int(rng.integers(max(20, 2 * C), 4000))
    d = int(rng.choice([6, 10]))
    stress_ba.ragged_scene(C, P, rng)
for c in range(want + 1):
    C = int(rng.integers(3, 14)); P = int(rng.integers(40, 400)); d = int(rng.choice([6, 10]))
    order = ["aligned", "reference"][c % 2]
    sc, cam_idx, pt_idx, uv = stress_ba.ragged_scene(C, P, rng)
    cams0 = sc.cams0[:, :d].copy()
    cams0[:, :6] += rng.normal(0, 0.004, size=(C, 6))
x0 = np.concatenate([cams0.ravel(), sc.pts0.ravel()])
uv_eff = bo.effective_uv(uv, cam_idx, order)
prob = bo.BAProblem(C, P, d, cam_idx, pt_idx, uv_eff, np.array(synth.K_REF))
print(f"case: C={C} P={P} N={len(cam_idx)} d={d} {order}")


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-3)))


full = bo.trf(prob, x0, solver="dense")
print("oracle (dense): nfev/njev/status", full.nfev, full.njev, full.status, "cost %.12e" % full.cost)
be = GpuBA(cams0, sc.pts0, cam_idx, pt_idx, uv_eff, synth.K_REF)
st = be.trf_begin()
k = 0
while True:
    k += 1
    more = st.outer()
    cams, pts = be.params()
    x = np.concatenate([cams.ravel(), pts.ravel()])
    rd = bo.trf(prob, x0, solver="dense", max_outer=k)
    try:
        rs = bo.trf(prob, x0, solver="schur", max_outer=k)
        own = f"schur vs dense {rel(rs.x, rd.x):.1e} (nfev {rs.nfev} / {rd.nfev})"
    except np.linalg.LinAlgError:
        own = "its point-elimination solve finds the system singular"
    r = st.result()
    print(f"outer {k:2d}: library vs dense oracle {rel(x, rd.x):.1e} (nfev {r.nfev} / {rd.nfev}, cost rel diff {abs(r.cost - rd.cost) / rd.cost:.1e});"
          f"  oracle: {own}", flush=True)
    if not more:
        break
res = st.result()
i = int(np.argmax(np.abs(x - full.x) / np.maximum(np.abs(full.x), 1e-3)))
what = f"camera {i // d} parameter {i % d}" if i < C * d else f"point {(i - C * d) // 3} coordinate {(i - C * d) % 3}"
print(f"end: status {res.status} / {full.status}, worst entry: {what}: {x[i]:.9g} vs {full.x[i]:.9g};"
      f" cost {res.cost:.12e} vs {full.cost:.12e} (rel {abs(res.cost - full.cost) / full.cost:.1e})")
