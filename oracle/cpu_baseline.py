"""bench.py's cpu_baseline leg: the C restatement (oracle/ba_oracle.c, OpenMP) timed on the host's cores
on a BOUNDED sample of the benchmark workload.  kind = "port" (the reference's own CPU path cannot run
this size at all: its dense Jacobian would need 4.8 TB)."""
import time

import numpy as np

from . import ba_c


def ba_baseline(scene, d, warmup=2, timed=3):
    """The SAME fixed schedule as the GPU run (termination tests off) on the CPU: the C restatement runs
    `warmup` outer iterations, then again `warmup + timed` from the same start; the difference is the time
    of `timed` outer iterations after the same warm-up (about 10-30 s of CPU work at 200 cams / 100k points)."""
    C, P = scene.cams0.shape[0], scene.pts0.shape[0]
    cb = ba_c.CBA(C, P, d, scene.cam_idx, scene.pt_idx, scene.uv, (1228.0, 1228.0, 512.0, 384.0))
    x0 = np.concatenate([scene.cams0[:, :d].ravel(), scene.pts0.ravel()])
    t0 = time.perf_counter()
    _, r1 = cb.trf(x0, max_nfev=10 ** 9, max_outer=warmup, check_tolerances=False)
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, r2 = cb.trf(x0, max_nfev=10 ** 9, max_outer=warmup + timed, check_tolerances=False)
    t2 = time.perf_counter() - t0
    dt = max(t2 - t1, 1e-9)
    return {"value": timed / dt, "unit": "LM-iterations/s", "cores": ba_c.threads(), "kind": "port",
            "sample": (f"oracle/ba_oracle.c (gcc -O3 -fopenmp, fp64, same algorithm; a plain restatement, not a tuned CPU solver: "
                       f"its dense Cholesky is a 64-column blocked loop nest without BLAS) on the same scene and fixed schedule: "
                       f"{timed} outer iterations after {warmup} warm-up iterations took {dt:.2f} s "
                       f"({r2['n_solves'] - r1['n_solves']} damped solves, {r2['nfev'] - r1['nfev']} trial steps); "
                       f"total CPU time spent {t1 + t2:.1f} s"),
            "seconds_per_iteration": dt / timed, "cost_end": r2["cost"]}


def matcher_baseline(d1_u8, d2_u8, n_queries=2000):
    q = d1_u8[:n_queries]
    t0 = time.perf_counter()
    ba_c.knn2_u8(q, d2_u8)
    dt = time.perf_counter() - t0
    return {"value": q.shape[0] * d2_u8.shape[0] / dt, "unit": "pairs/s", "cores": ba_c.threads(), "kind": "port",
            "sample": f"{q.shape[0]} of the query rows against all {d2_u8.shape[0]} train rows, mo_knn2_u8 (gcc -O3 -fopenmp)"}
