#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X: `python bench.py --gpus N --steps K --warmup W`.

Metric (BASELINE.json): BA LM-iterations/sec + descriptor-pairs/sec, 200 cams / 100k pts.
A *step* is one outer iteration of the reference's trust-region solve (scipy trf.py:450-545 as driven
by /root/reference/utils/sfm_reconstruction.py:506-514) on the 200-camera / 100k-point / 1M-observation
synthetic scene: the trial steps of one linearisation (each trial = More' iteration of damped Schur
solves + one cost evaluation) up to the accepted step, then the next residual/Jacobian linearisation.
Termination tests are off (fixed schedule) so exactly K outer iterations are timed.  With N > 1 the
points (with their observations) are sharded over the ranks (strong scaling), cameras are replicated
and the reduced camera system is all-reduced over RCCL.  The matcher leg times the 50k x 50k x 128
brute-force L2 + ratio test (queries sharded over ranks).  Inputs are resident in HBM before timing.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (6.29 TB/s measured copy)
I8_MFMA_PEAK_TOPS = 5000.0   # dense i8 MFMA = 2x bf16 (~2.5 PF)
FP64_MFMA_PEAK_TFLOPS = 78.6 # v_mfma_f64_16x16x4_f64: 64 cycles per issue measured = 32 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cams", type=int, default=200)
    ap.add_argument("--pts", type=int, default=100000)
    ap.add_argument("--obs-per-point", type=int, default=10)
    ap.add_argument("--cam-dim", type=int, default=10, choices=(6, 10))
    ap.add_argument("--match-n", type=int, default=50000)
    ap.add_argument("--match-reps", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-matcher", action="store_true")
    ap.add_argument("--no-d6", action="store_true")
    ap.add_argument("--no-driver-rows", action="store_true")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from sfm_amd import synth, _lib
    from sfm_amd.ba import GpuBA
    from sfm_amd.comm import DistComm, LocalComm
    from sfm_amd.structure import build_structure, partition_points, shard_arrays
    from sfm_amd.trf import TRFState

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    # SFM_FORCE_DIST=1 runs the RCCL code path (process group, all-reduces of the workspace views) with a
    # single rank too - used to rehearse the multi-GPU path on a one-GPU box
    use_dist = world > 1 or os.environ.get("SFM_FORCE_DIST") == "1"
    if use_dist:
        if "RANK" not in os.environ:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        comm = DistComm()
        comm.force = True
    else:
        comm = LocalComm()

    def barrier_sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(v):
        if not use_dist:
            return v
        t = torch.tensor([v], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ------------------------------------------------------------------ BA workload (cfg4 by default)
    C, P, Lobs, d = args.cams, args.pts, args.obs_per_point, args.cam_dim
    sc = synth.make_scene(C, P, obs_per_point=Lobs, seed=1004, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002)
    full = build_structure(sc.cam_idx, sc.pt_idx, C, P) if world == 1 else None
    lo, hi = (0, P)
    if world > 1:
        pt_ptr = np.zeros(P + 1, dtype=np.int64)
        np.cumsum(np.bincount(sc.pt_idx, minlength=P), out=pt_ptr[1:])
        lo, hi = partition_points(pt_ptr, world)[rank]
    ci, pi, uv, pts0 = shard_arrays(sc.cam_idx, sc.pt_idx, sc.uv, sc.pts0, lo, hi)
    be = GpuBA(sc.cams0[:, :d], pts0, ci, pi, uv, synth.K_REF, device=local_rank, comm=comm,
               structure=full)
    n_obs_total = sc.n_obs
    st = TRFState(be, max_nfev=10 ** 9, check_tolerances=False)
    cost0 = st.cost
    be.h.set_profiling(True)
    for _ in range(args.warmup):
        st.outer()
    be.h.profile()
    solves0, nfev0 = st.n_solves, st.nfev
    barrier_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st.outer()
    barrier_sync()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    prof = be.h.profile()
    be.h.set_profiling(False)
    n_solves = st.n_solves - solves0
    n_trials = st.nfev - nfev0
    value = args.steps / elapsed

    # roofline of the Jacobian kernel: algorithmic bytes per observation:
    # idx 8 + uv 16 + residual 16 + Jc 2*d*8 + Jp 48
    bytes_per_obs = 8 + 16 + 16 + 2 * d * 8 + 48
    lin_ms, lin_cnt = prof["lin_obs"]
    roofline = None
    if lin_cnt > 0:
        per_launch_s = lin_ms / lin_cnt * 1e-3
        achieved = bytes_per_obs * be.N / per_launch_s / 1e9
        roofline = {"kernel": "k_lin_obs (residual + 2x(%d+3) Jacobian + Huber scaling)" % d, "bound": "hbm",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                    "bytes_per_launch": bytes_per_obs * be.N, "avg_us": round(per_launch_s * 1e6, 2),
                    "launches": lin_cnt}
        # HBM bytes per launch from the PMC counters (rocprofv3 cannot run inside this process): taken from
        # the committed summary of the same workload, collected and corrected as MI355X_MICROARCH.md prescribes
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
            wl = pmc["workload"]
            if world == 1 and (wl["cams"], wl["pts"], wl["obs"], wl["cam_dim"]) == (C, P, n_obs_total, d):
                roofline["traffic"] = pmc["k_lin_obs"]["hbm_bytes_per_launch"]
                roofline["traffic_source"] = "profiles/r01_pmc_summary.json (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)"
        except (OSError, KeyError, ValueError):
            pass
    kernels = {k: {"ms_total": round(v[0], 3), "launches": v[1],
                   "share": round(v[0] / (elapsed * 1e3), 4)} for k, v in prof.items() if v[1] > 0}
    # the two kernels that dominate the TIME are not HBM- or MFMA-bound; their rates are
    # reported for completeness: the Schur gather in bytes of G blocks it pulls through L2/MALL, the Cholesky
    # in fp64 FLOP/s against the dense MFMA peak (it is a latency chain: n/64 dependent steps)
    n_sys = C * d
    if "schur" in kernels:
        sec = kernels["schur"]["ms_total"] / kernels["schur"]["launches"] * 1e-3
        gb = 2.0 * int(be.st.n_pairs) * 3 * d * 8
        kernels["schur"].update(gather_bytes_per_launch=gb, gather_GBps=round(gb / sec / 1e9, 1), bound="gather latency")
    if "chol" in kernels:
        sec = kernels["chol"]["ms_total"] / kernels["chol"]["launches"] * 1e-3
        fl = n_sys ** 3 / 3.0
        kernels["chol"].update(flop_per_launch=fl, TFLOPs=round(fl / sec / 1e12, 2),
                               frac_of_fp64_mfma_peak=round(fl / sec / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
                               bound=("latency (%d dependent 64-column steps)" % ((n_sys + 63) // 64)) if n_sys < 4096
                               else "mfma (256-column strips + rank-256 trailing updates)")

    # ------------------------------------------------------------------ secondary: the north_star's 2x(6+3) shape
    ba_d6 = None
    if d == 10 and not args.no_d6:
        be.h.set_profiling(False)
        be6 = GpuBA(sc.cams0[:, :6], pts0, ci, pi, uv, synth.K_REF, device=local_rank, comm=comm, structure=be.st)
        st6 = TRFState(be6, max_nfev=10 ** 9, check_tolerances=False)
        for _ in range(args.warmup):
            st6.outer()
        s0 = st6.n_solves
        barrier_sync()
        t6 = time.perf_counter()
        for _ in range(args.steps):
            st6.outer()
        barrier_sync()
        e6 = max_over_ranks(time.perf_counter() - t6)
        ba_d6 = {"value": args.steps / e6, "unit": "LM-iterations/s", "ms_per_step": e6 / args.steps * 1e3,
                 "damped_solves": st6.n_solves - s0, "cost_end": st6.cost,
                 "workload": "same scene, camera block [rvec,t] with fixed K (cam_dim 6, n = %d)" % (6 * C)}
        del be6, st6
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ matcher workload (cfg2)
    matcher = None
    if not args.no_matcher:
        from sfm_amd import matcher as mt
        n = args.match_n
        d1, d2 = synth.make_descriptors(n, n, seed=1002)
        q_lo, q_hi = (rank * n) // world, ((rank + 1) * n) // world
        q = torch.from_numpy(d1[q_lo:q_hi].astype(np.uint8)).cuda()
        t = torch.from_numpy(d2.astype(np.uint8)).cuda()
        be.h.set_profiling(True)
        for _ in range(2):
            i1, i2, a, b = mt.knn2(q, t, "l2", device=local_rank)
            mq, mtr, md = mt.ratio_filter(i1, a, b, 0.75, device=local_rank)
        be.h.profile()
        barrier_sync()
        tm0 = time.perf_counter()
        for _ in range(args.match_reps):
            i1, i2, a, b = mt.knn2(q, t, "l2", device=local_rank)
            mq, mtr, md = mt.ratio_filter(i1, a, b, 0.75, device=local_rank)
        barrier_sync()
        tm = max_over_ranks(time.perf_counter() - tm0)
        mprof = be.h.profile()
        be.h.set_profiling(False)
        pairs = float(n) * float(n) * args.match_reps
        knn_ms, knn_cnt = mprof["knn"]
        mroof = None
        if knn_cnt > 0:
            ops = 2.0 * 128 * (q_hi - q_lo) * n            # i8 multiply-adds of the distance GEMM, per launch
            ach = ops / (knn_ms / knn_cnt * 1e-3) / 1e12
            mroof = {"kernel": "k_knn2_u8<4,2> (i8 MFMA distance + top-2)", "bound": "mfma", "achieved": round(ach, 1),
                     "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s", "frac": round(ach / I8_MFMA_PEAK_TOPS, 4),
                     "avg_us": round(knn_ms / knn_cnt * 1e3, 1), "traffic": None}
        matcher = {"metric": "descriptor-pairs/sec", "value": pairs / tm, "unit": "pairs/s",
                   "workload": f"{n} x {n} x 128 uint8 SIFT-like, brute-force L2 kNN(2) + ratio 0.75 + compaction",
                   "ms_per_pair_of_images": tm / args.match_reps * 1e3, "n_matches": int(mq.shape[0]),
                   "dtype": "u8/i32", "roofline": mroof}

    # ------------------------------------------------------------------ driver-row kernels, N = 1 only
    driver_rows = None
    if rank == 0 and world == 1 and not args.no_driver_rows:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_driver
            cpu_fns = None
            if not args.no_cpu_baseline:          # cpu_baseline leg: the NumPy oracle on a bounded sample
                from oracle import driver_oracle as dro
                cpu_fns = {"associate": dro.associate, "triangulate_point": dro.triangulate_point,
                           "symmetric_epipolar_errors": dro.symmetric_epipolar_errors}
            driver_rows = bench_driver.measure(reps=10, cpu_fns=cpu_fns)
        except Exception as e:       # secondary figures; never fail the main measurement on them
            driver_rows = {"error": repr(e)}

    # ------------------------------------------------------------------ CPU baseline (rank 0, N = 1 only)
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import cpu_baseline as cb
            cpu_baseline = cb.ba_baseline(sc, d, warmup=args.warmup, timed=min(3, args.steps))
            if matcher is not None:
                cpu_baseline["matcher"] = cb.matcher_baseline(d1.astype(np.uint8), d2.astype(np.uint8))
        except Exception as e:       # the baseline is a reported extra; never fail the GPU measurement on it
            cpu_baseline = {"error": repr(e)}

    if rank == 0:
        out = {
            "metric": "BA LM-iterations/sec + descriptor-pairs/sec, 200 cams / 100k pts, 1->8 MI355X",
            "value": value, "unit": "LM-iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BA: {C} cams / {P} pts / {n_obs_total} obs, cam block {d} "
                                   f"({'reference 10-parameter block + regulariser' if d == 10 else 'fixed K'}), "
                                   "aligned residual order, Huber, SciPy-TRF control flow, fixed schedule",
                       "parallelism": f"points sharded over {world} rank(s), cameras replicated, RCCL all-reduce of [S|r]",
                       "seed": 1004},
            "ba": {"damped_solves": n_solves, "trial_steps": n_trials, "cost_start": cost0, "cost_end": st.cost,
                   "solves_per_s": n_solves / elapsed, "kernels": kernels},
            "ba_cam_dim6": ba_d6, "roofline": roofline, "cpu_baseline": cpu_baseline, "matcher": matcher,
            "driver_rows": driver_rows,
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
