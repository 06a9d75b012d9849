#!/usr/bin/env python3
"""Device-time measurement of the driver-row kernels with inputs resident in HBM:
association (pair tests/s), two-view triangulation (tracks/s), epipolar verification (matches/s).  Prints one
JSON line per kernel.  bench.py calls measure() and, in its cpu_baseline leg, hands in the NumPy oracle's
functions to time on a bounded sample of the same inputs (this tool itself never imports oracle/).
usage: python tools/bench_driver.py [--reps 20]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps):
    import torch
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def measure(reps=20, tracks=100000, corr=20000, cpu_fns=None, emit=None):
    """[assoc, triangulate, epipolar] result dicts; `emit(dict)` is called as each becomes available.
    cpu_fns: optional {"associate", "triangulate_point", "symmetric_epipolar_errors"} callables (the oracle's)
    timed on a bounded sample beside each kernel."""
    import types
    a = types.SimpleNamespace(reps=reps, tracks=tracks, corr=corr)
    results = []

    def out(d):
        results.append(d)
        if emit:
            emit(d)
    import torch
    from sfm_amd import _lib
    from sfm_amd.driver import _p
    cpu = cpu_fns is not None
    h = _lib.get_handle(0)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    vp = C.c_void_p

    # ---- association: T tracks x M correspondences, ~40 % of tracks have a partner
    T, M = a.tracks, a.corr
    c = (rng.random((M, 2)) * [1024, 768]).astype(np.float32)
    t = (rng.random((T, 2)) * [1024, 768]).astype(np.float32)
    k = int(0.4 * T)
    t[rng.permutation(T)[:k]] = c[rng.integers(0, M, k)] + (rng.normal(size=(k, 2)) * 0.7).astype(np.float32)
    d_t = torch.from_numpy(t.astype(np.float64)).to(dev); d_c = torch.from_numpy(c.astype(np.float64)).to(dev)
    t_ptr = torch.tensor([0, T], dtype=torch.int64, device=dev); m_ptr = torch.tensor([0, M], dtype=torch.int64, device=dev)
    need = C.c_int64(); h.lib.sfm_assoc_workspace_bytes(T, C.byref(need))
    ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
    cap = 4 * T
    o_r = torch.empty(cap, dtype=torch.int32, device=dev); o_c = torch.empty(cap, dtype=torch.int32, device=dev)
    tot = torch.zeros(1, dtype=torch.int64, device=dev)

    def assoc():
        h.call("sfm_assoc_radius", _p(d_t), _p(t_ptr), _p(d_c), _p(m_ptr), 1, T, C.c_double(2.0), _p(o_r), _p(o_c),
               cap, _p(tot), _p(ws), need.value)
    sec = timed(assoc, a.reps)
    ns = min(T, 2000)
    r = {"kernel": "assoc_radius", "tracks": T, "correspondences": M, "hits": int(tot.item()),
         "ms": sec * 1e3, "pair_tests_per_s": T * M / sec,
         "fp64_flop_per_s": 5.0 * 2 * T * M / sec}      # 2 passes (count, fill) x 5 flop per test
    if cpu:
        t0 = time.perf_counter(); cpu_fns["associate"](t[:ns].astype(np.float64), c); sec_cpu = time.perf_counter() - t0
        r.update(cpu_numpy_pair_tests_per_s=ns * M / sec_cpu, cpu_sample=f"{ns} x {M}")
    out(r)

    # ---- triangulation: n two-view tracks over 64 cameras
    n = 1_000_000
    from sfm_amd import synth
    sc = synth.make_scene(64, 20000, obs_per_point=2, seed=3, noise_px=0.5)
    poses, pts, tracks, K = sc.state()
    ids = list(poses)
    proj = np.stack([K @ np.hstack([poses[i][0], np.asarray(poses[i][1]).reshape(3, 1)]) for i in ids])
    ci = sc.cam_idx.reshape(-1, 2); uv = sc.uv.reshape(-1, 2, 2)
    rep = -(-n // len(ci))
    c0 = np.tile(ci[:, 0], rep)[:n].astype(np.int32); c1 = np.tile(ci[:, 1], rep)[:n].astype(np.int32)
    x0 = np.tile(uv[:, 0], (rep, 1))[:n]; x1 = np.tile(uv[:, 1], (rep, 1))[:n]
    d = [torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (proj.reshape(-1, 12), c0, c1, x0, x1)]
    X = torch.empty((n, 3), dtype=torch.float64, device=dev); valid = torch.empty(n, dtype=torch.int32, device=dev)

    def tri():
        h.call("sfm_triangulate2", _p(d[0]), proj.shape[0], _p(d[1]), _p(d[2]), _p(d[3]), _p(d[4]), n,
               C.c_double(4.0), _p(X), _p(valid), vp(0))
    sec = timed(tri, a.reps)
    ns = 2000
    r = {"kernel": "triangulate2", "tracks": n, "valid_frac": float(valid.float().mean().item()),
         "ms": sec * 1e3, "tracks_per_s": n / sec, "hbm_GBps_algorithmic": n * 68 / sec / 1e9}
    if cpu:
        t0 = time.perf_counter()
        for i in range(ns):
            cpu_fns["triangulate_point"]([proj[c0[i]], proj[c1[i]]], [x0[i], x1[i]])
        sec_cpu = time.perf_counter() - t0
        r.update(cpu_numpy_tracks_per_s=ns / sec_cpu, cpu_sample=f"{ns} tracks")
    out(r)

    # ---- epipolar verification: n matches in 1,000 pairs
    n, n_seg = 10_000_000, 1000
    F = rng.normal(size=(n_seg, 9)) * np.array([1e-6, 1e-6, 1e-3, 1e-6, 1e-6, 1e-3, 1e-3, 1e-3, 1.0])
    p1 = (rng.random((n, 2)) * 1000).astype(np.float32); p2 = (rng.random((n, 2)) * 1000).astype(np.float32)
    seg = torch.from_numpy(np.linspace(0, n, n_seg + 1).astype(np.int64)).to(dev)
    dF, d1, d2 = (torch.from_numpy(v).to(dev) for v in (F, p1, p2))
    err = torch.empty(n, dtype=torch.float32, device=dev); mask = torch.empty(n, dtype=torch.uint8, device=dev)

    def epi():
        h.call("sfm_epipolar_errors", _p(dF), _p(seg), n_seg, _p(d1), _p(d2), n, C.c_float(3.0), _p(err), _p(mask))
    sec = timed(epi, a.reps)
    ns = 1_000_000
    r = {"kernel": "epipolar_errors", "matches": n, "pairs": n_seg, "ms": sec * 1e3,
         "matches_per_s": n / sec, "hbm_GBps_algorithmic": n * 21 / sec / 1e9,
         "hbm_frac_of_8TBps": n * 21 / sec / 8e12}
    if cpu:
        t0 = time.perf_counter(); cpu_fns["symmetric_epipolar_errors"](p1[:ns], p2[:ns], F[0].reshape(3, 3))
        sec_cpu = time.perf_counter() - t0
        r.update(cpu_numpy_matches_per_s=ns / sec_cpu, cpu_sample=f"{ns} matches")
    out(r)
    return results


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--tracks", type=int, default=100000)
    ap.add_argument("--corr", type=int, default=20000)
    a = ap.parse_args()
    measure(a.reps, a.tracks, a.corr, emit=lambda d: print(json.dumps(d), flush=True))


if __name__ == "__main__":
    main()
