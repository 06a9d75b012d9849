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
    # 20 timed outer iterations after 5 (the schedule the round driver runs): the chip raises its clock only under sustained
    # load - at 10 after 2 (7 ms of warm-up) the same build reads ~9 % lower
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cams", type=int, default=200)
    ap.add_argument("--pts", type=int, default=100000)
    ap.add_argument("--obs-per-point", type=int, default=10)
    ap.add_argument("--cam-dim", type=int, default=10, choices=(6, 10))
    ap.add_argument("--match-n", type=int, default=50000)
    ap.add_argument("--match-reps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-matcher", action="store_true")
    ap.add_argument("--no-d6", action="store_true")
    ap.add_argument("--no-mixed", action="store_true")
    ap.add_argument("--no-dropin", action="store_true")
    ap.add_argument("--no-pcg", action="store_true")
    ap.add_argument("--camera-solver", default="auto", choices=("auto", "cholesky", "cg"),
                    help="how the formed camera system is solved in the headline run")
    ap.add_argument("--no-alt-camera-solver", action="store_true")
    ap.add_argument("--no-driver-rows", action="store_true")
    ap.add_argument("--visibility", default="random", choices=("random", "nearest"),
                    help="random: the BASELINE scene (L cameras per track drawn uniformly); nearest: spatially coherent scene "
                         "(each point seen by its L nearest cameras, cameras numbered along the hemisphere)")
    ap.add_argument("--no-coherent", action="store_true", help="skip the secondary row on the spatially coherent scene")
    ap.add_argument("--no-reference-order", action="store_true",
                    help="skip the secondary row on the reference's own (mis-paired) objective, the drop-in's default")
    ap.add_argument("--dry-launch", action="store_true",
                    help="--gpus N > 1 without WORLD_SIZE: print the launch plan (one JSON line) and exit, starting nothing")
    ap.add_argument("--child-cmd", default=None,
                    help="(tests) JSON list: the command every rank runs instead of this script")
    ap.add_argument("--launch-timeout", type=float, default=float(os.environ.get("SFM_BENCH_LAUNCH_TIMEOUT", "2400")),
                    help="seconds the group of ranks may take before all of it is killed")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: THIS process starts the N ranks.

    It is a pure parent: nothing here touches the GPU (no torch.cuda call that initialises HIP, no library load), so starting
    children is not an exec out of a GPU process.  Every rank is a fresh `python bench.py <same arguments>` with RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, in its own session (so exactly the process groups started here
    can be signalled).  Rank 0's stdout is captured and its single JSON line relayed; the other ranks' stdout goes to stderr.
    Exit code: 0 with the line; the first failing rank's code if a rank fails (the rest are killed); 124 if the group is
    still running after --launch-timeout seconds; 2 if fewer than N devices are visible."""
    import signal
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:]]
    for flag in ("--dry-launch",):
        argv = [a for a in argv if a != flag]
    if args.child_cmd is not None:                      # drop "--child-cmd <json>" / "--child-cmd=<json>" from the children
        out, skip = [], False
        for a in argv:
            if skip:
                skip = False
            elif a == "--child-cmd":
                skip = True
            elif not a.startswith("--child-cmd="):
                out.append(a)
        argv = out
    cmd = json.loads(args.child_cmd) if args.child_cmd else [sys.executable, os.path.abspath(__file__)] + argv
    plan = {"launcher": "bench.py", "world_size": n, "master_addr": "127.0.0.1", "master_port": port, "command": cmd,
            "ranks": [{"RANK": r, "LOCAL_RANK": r, "WORLD_SIZE": n} for r in range(n)], "timeout_s": args.launch_timeout}
    if args.dry_launch:
        print(json.dumps(plan), flush=True)
        return 0
    if args.child_cmd is None:
        import torch                                     # device_count() does not initialise the runtime on this image
        have = torch.cuda.device_count()
        if have < n:
            print("bench.py: --gpus %d asked for, %d device(s) visible" % (n, have), file=sys.stderr, flush=True)
            return 2
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, start_new_session=True))

    def kill_all():
        for pr in procs:
            if pr.poll() is None:
                try:
                    os.killpg(pr.pid, signal.SIGTERM)
                except (ProcessLookupError, PermissionError):
                    pass
        t_end = time.monotonic() + 10.0
        for pr in procs:
            try:
                pr.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
                pr.wait()

    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout.read().decode(errors="replace").splitlines()), daemon=True)
    reader.start()
    t0 = time.monotonic()
    rc = None
    try:
        while rc is None:
            codes = [pr.poll() for pr in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                print("bench.py: rank %d exited with %d; stopping the other ranks" % bad[0], file=sys.stderr, flush=True)
                rc = bad[0][1] if bad[0][1] > 0 else 1
            elif all(c == 0 for c in codes):
                rc = 0
            elif time.monotonic() - t0 > args.launch_timeout:
                print("bench.py: ranks still running after %.0f s; killing them" % args.launch_timeout, file=sys.stderr, flush=True)
                rc = 124
            else:
                time.sleep(0.2)
    finally:
        kill_all()
    reader.join(timeout=5.0)
    if rc != 0:
        return rc
    for ln in reversed(lines):
        try:
            if isinstance(json.loads(ln), dict):
                print(ln, flush=True)
                return 0
        except ValueError:
            continue
    print("bench.py: rank 0 printed no JSON line", file=sys.stderr, flush=True)
    return 1


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))          # before anything below touches the GPU
    if args.dry_launch:
        print(json.dumps({"launcher": "bench.py", "world_size": 1, "note": "single rank: nothing to launch"}), flush=True)
        return
    import torch
    import torch.distributed as dist
    from sfm_amd import synth, _lib
    from sfm_amd.ba import GpuBA
    from sfm_amd.comm import DistComm, LocalComm
    from sfm_amd.structure import partition_points, shard_arrays

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
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
        # stdout carries ONE JSON line: keep RCCL's version banner (printed to stdout at NCCL_DEBUG=VERSION) out of it
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            os.environ["NCCL_DEBUG"] = "WARN"
        # stdout carries ONE JSON line, but RCCL writes its warnings and version banner there while communicators come up
        # (torch creates its own lazily, at the first collective): file descriptor 1 points at stderr until both exist
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            # the exchanges run inside the library (sfm_comm_*: ncclAllReduce on the handle's stream, no host round trip per
            # exchange); torch.distributed only ships the 128-byte RCCL id and takes the timing barrier.  Checked against
            # torch.distributed's own all-reduce before use; SFM_BENCH_COMM=torch keeps the Python-callback route.
            comm, comm_kind = None, "torch.distributed all_reduce from Python callbacks (DistComm)"
            if os.environ.get("SFM_BENCH_COMM", "rccl") != "torch":
                # every rank runs the same two torch collectives whatever happens locally, so a local failure cannot
                # leave the other ranks waiting in a collective
                rc_, local_ok = None, 1.0
                try:
                    from sfm_amd.comm import RcclComm
                    rc_ = RcclComm(device=local_rank)
                except Exception as e:
                    local_ok = 0.0
                    print("bench: in-library RCCL unavailable (%r); using torch.distributed callbacks" % (e,), file=sys.stderr)
                probe = torch.arange(1, 1025, dtype=torch.float64, device="cuda") * (rank + 1)
                want = probe.clone()
                dist.all_reduce(want)
                if rc_ is not None:
                    try:
                        rc_.allreduce_sum(probe)
                        torch.cuda.synchronize()
                        local_ok = 1.0 if torch.equal(probe, want) else 0.0
                    except Exception as e:
                        local_ok = 0.0
                        print("bench: in-library all-reduce failed (%r)" % (e,), file=sys.stderr)
                ok = torch.tensor([local_ok], dtype=torch.float64, device="cuda")
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if float(ok.item()) == 1.0:
                    comm, comm_kind = rc_, "RCCL inside libsfm_amd.so (sfm_comm_reduce_hook: ncclAllReduce on the handle's stream)"
                elif rc_ is not None:
                    rc_.close()
            if comm is None:
                warm_t = torch.ones(8, dtype=torch.float64, device="cuda")
                dist.all_reduce(warm_t)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
        if comm is None:
            comm = DistComm()
        comm.force = True
    else:
        comm, comm_kind = LocalComm(), "single rank"

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
    sc = synth.make_scene(C, P, obs_per_point=Lobs, seed=1004, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002,
                          visibility=args.visibility)
    lo, hi = (0, P)
    if world > 1:
        pt_ptr = np.zeros(P + 1, dtype=np.int64)
        np.cumsum(np.bincount(sc.pt_idx, minlength=P), out=pt_ptr[1:])
        lo, hi = partition_points(pt_ptr, world)[rank]
    ci, pi, uv, pts0 = shard_arrays(sc.cam_idx, sc.pt_idx, sc.uv, sc.pts0, lo, hi)
    n_obs_total = sc.n_obs
    n_sys = C * d

    def shard_scene(scene):
        """This rank's shard of a scene: (cams0, pts0, cam_idx, pt_idx, uv)."""
        a, b = 0, P
        if world > 1:
            pp = np.zeros(P + 1, dtype=np.int64)
            np.cumsum(np.bincount(scene.pt_idx, minlength=P), out=pp[1:])
            a, b = partition_points(pp, world)[rank]
        ci_, pi_, uv_, pts_ = shard_arrays(scene.cam_idx, scene.pt_idx, scene.uv, scene.pts0, a, b)
        return scene.cams0, pts_, ci_, pi_, uv_

    def fixed_schedule_run(cam_dim, precision, profile, solver="dense", camera_solver=None, shard=None):
        """W warm-up + K timed outer iterations of the trust-region loop (library side, termination tests off).
        The profiled pass (never the timed one) also reads the cost after every outer iteration."""
        cams0_, pts0_, ci_, pi_, uv_ = shard if shard is not None else (sc.cams0, pts0, ci, pi, uv)
        be = GpuBA(cams0_[:, :cam_dim], pts0_, ci_, pi_, uv_, synth.K_REF, device=local_rank, comm=comm, precision=precision,
                   solver=solver, camera_solver=camera_solver or args.camera_solver)
        st = be.trf_begin(max_nfev=2 ** 31 - 1, check_tolerances=False)
        cost0 = st.result().cost
        cost_trace = []
        for _ in range(args.warmup):
            st.outer()
            if profile:
                cost_trace.append(st.result().cost)
        r0 = st.result()
        cg0, pcg0 = be.solver_stats(), be.cg_iters          # counters are cumulative: the timed pass is the difference
        if profile:
            be.h.set_profiling(True)
            be.h.profile()
        barrier_sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            st.outer()
            if profile:
                cost_trace.append(st.result().cost)
        barrier_sync()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        prof = None
        if profile:
            prof = be.h.profile()
            be.h.set_profiling(False)
        r1 = st.result()
        cg1 = be.solver_stats()
        st.close()
        out = {"cg_iters": be.cg_iters - pcg0, "camera_cg": (cg1[0] - cg0[0], cg1[1] - cg0[1]), "camera_cg_warmup": cg0,
               "pcg_stats": be.pcg_stats() if solver == "pcg" else None,
               "elapsed": elapsed, "value": args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
               "damped_solves": r1.n_solves - r0.n_solves, "trial_steps": r1.nfev - r0.nfev, "cost_start": cost0,
               "cost_end": r1.cost, "n_pairs": be.n_pairs, "n_obs_local": be.N, "prof": prof, "cost_trace": cost_trace}
        del st, be
        torch.cuda.empty_cache()
        return out

    def timed_and_profiled(cam_dim, precision, **kw):
        """A secondary row measured like the headline: the rate from a pass WITHOUT per-kernel events or cost reads, the kernel
        table and the solver counts of a second, identical pass attached to it.  (Until late in round 4 these rows were timed
        with the events and a read of the cost after every outer iteration in the way: 4-7 % low.)"""
        r = fixed_schedule_run(cam_dim, precision, False, **kw)
        rp = fixed_schedule_run(cam_dim, precision, True, **kw)
        r["prof"], r["cost_trace"] = rp["prof"], rp["cost_trace"]
        return r

    # the timed run carries NO per-kernel events; the kernel table comes from a second, identical pass
    main = fixed_schedule_run(d, "fp64", profile=False)
    elapsed, value = main["elapsed"], main["value"]
    profd = fixed_schedule_run(d, "fp64", profile=True)
    prof, prof_elapsed = profd["prof"], profd["elapsed"]
    kernels = {k: {"ms_total": round(v[0], 3), "launches": v[1], "us_per_launch": round(v[0] / v[1] * 1e3, 2),
                   "share": round(v[0] / (prof_elapsed * 1e3), 4)} for k, v in prof.items() if v[1] > 0}

    # ---- rooflines (MI355X_MICROARCH.md peaks).  Algorithmic work per launch:
    #   k_lin_obs     idx 8 + uv 16 + residual 16 + Jc 2*d*8 + Jp 48 bytes per observation
    #   Cholesky      n^3 / 3 fp64 FLOP per factorisation (all k_chol_* launches of one factorisation together)
    #   k_schur_items 2 G blocks (3*d doubles each) gathered per camera pair - bytes through the L2/Infinity-Cache fabric
    def pmc_traffic(kernel_key):
        """HBM bytes per launch from the committed PMC summary of the same workload (rocprofv3 cannot run inside
        this process); collected and corrected as MI355X_MICROARCH.md prescribes."""
        for name in ("r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json", "r01_pmc_summary.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                wl = pmc["workload"]
                if world == 1 and (wl["cams"], wl["pts"], wl["obs"], wl["cam_dim"]) == (C, P, n_obs_total, d) and kernel_key in pmc:
                    return pmc[kernel_key]["hbm_bytes_per_launch"], "profiles/" + name
            except (OSError, KeyError, ValueError):
                continue
        return None, None

    def roof(name, bound, work, unit_scale, peak, unit, slot, kernel_key=None, note=None):
        if slot not in kernels:
            return None
        sec = kernels[slot]["us_per_launch"] * 1e-6
        ach = work / sec / unit_scale
        r = {"kernel": name, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
             "frac": round(ach / peak, 4), "traffic": None, "work_per_launch": work,
             "avg_us": kernels[slot]["us_per_launch"], "launches": kernels[slot]["launches"],
             "time_share": kernels[slot]["share"]}
        if kernel_key:
            r["traffic"], src = pmc_traffic(kernel_key)
            if src:
                r["traffic_source"] = src + " (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)"
        if note:
            r["note"] = note
        return r

    bytes_per_obs = 8 + 16 + 16 + 2 * d * 8 + 48
    big_cg = n_sys >= int(os.environ.get("SFM_CGS_BIG_FROM", "2049")) and os.environ.get("SFM_CGS_BIG", "1") != "0"
    cg_on = args.camera_solver != "cholesky" and (n_sys <= 4096 or big_cg)
    cam_roofs = []
    if cg_on and "chol" in kernels:
        # both slots of the camera solve hold CG solves here (chol: the step system incl. its scaling kernels, trsv: the system
        # of the q term).  n <= 2048: ONE persistent launch per system (k_cgs_persist: the matrix rows live in registers, so S~
        # is read from memory once per system; what bounds it is the all-gather of S~ p between workgroups, once per iteration)
        its, fb = profd["camera_cg"]                     # iterations / fallbacks of the profiled pass alone
        n_systems = kernels["chol"]["launches"] + kernels.get("trsv", {}).get("launches", 0)
        per_system = its / max(n_systems, 1)
        persistent = n_sys <= 2048 and os.environ.get("SFM_CGS_PERSIST", "1") != "0"
        # n > 2048: every iteration streams the 128 x 128 tiles of the LOWER triangle of S~ (diagonal tiles whole)
        nb_t = (n_sys + 127) // 128
        edge = n_sys - (nb_t - 1) * 128
        tri_bytes = 8.0 * sum((edge if I == nb_t - 1 else 128) * (edge if J == nb_t - 1 else 128) for I in range(nb_t) for J in range(I + 1))
        cam_us = kernels["chol"]["ms_total"] * 1e3 + kernels.get("trsv", {}).get("ms_total", 0.0) * 1e3
        name = ("k_cgs_big_symv" if big_cg else "k_cgs_persist" if persistent else "k_cgs_iter") + \
               ": CG on the block-scaled camera system, n = %d" % n_sys
        if big_cg:
            # an iteration is one pass over the lower triangle of S~ (405 MB at n = 10,000: beyond the 256 MiB Infinity Cache, so
            # an HBM stream): priced per iteration over the WHOLE slot of the second system (prologue, both kernels of every
            # iteration, the host's looks at the residual) with the iteration count of this pass
            slot = "trsv" if "trsv" in kernels else "chol"
            sec_it = kernels[slot]["us_per_launch"] * 1e-6 / max(per_system, 1.0)
            ach = tri_bytes / sec_it / 1e9
            r = {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "work_per_launch": per_system * tri_bytes,
                 "avg_us": kernels[slot]["us_per_launch"], "launches": kernels[slot]["launches"],
                 "note": "two launches per iteration (Chronopoulos-Gear arrangement of the recurrences): k_cgb_symv forms the new "
                         "residual on its two 128-entry ranges and streams its tile of the lower triangle of S~ (%.0f MB per iteration: "
                         "every 128 x 128 tile serves both products it takes part in), k_cgb_reduce adds the per-tile partial sums in "
                         "fixed order and leaves the two dot products.  `achieved` = triangle bytes / (slot of the second system / its "
                         "iterations)" % (tri_bytes / 1e6)}
        else:
            r = roof(name, "hbm", (1.0 if persistent else per_system) * n_sys * n_sys * 8.0, 1e9, HBM_PEAK_GBS, "GB/s",
                     "trsv" if "trsv" in kernels else "chol",
                     note=("Exchange-latency bound, not bandwidth bound: one launch per system, S~ (n^2 doubles) is read "
                           "once into registers, every iteration all-gathers the n entries of S~ p between the n / 8 workgroups "
                           "through 8-byte granules; `achieved` prices that single read of S~ against HBM and is small by "
                           "construction - see us_per_iteration" if persistent else
                           "one launch per iteration, each streams S~ (n^2 doubles, L2 / Infinity-Cache resident) once: "
                           "launch-latency bound"))
        if r:
            r["iterations"] = its
            r["systems"] = n_systems
            r["fallbacks_to_the_factorisation"] = fb
            r["iterations_per_system"] = round(per_system, 2)
            # both slots of the camera solve (scaling kernels, prologue, iterations, epilogue) over the iterations they ran
            r["us_per_iteration"] = round(cam_us / max(its, 1), 2)
            r["counting"] = ("iterations = sfm_ba_solver_stats after the profiled pass minus before it (warm-up excluded); "
                             "systems = launches of the two camera-solve slots in that pass; us_per_iteration = both slots' "
                             "HIP-event time / iterations")
            r["time_share"] = kernels["chol"]["share"] + kernels.get("trsv", {}).get("share", 0.0)
        cam_roofs.append(r)
    else:
        cam_roofs.append(roof("k_chol_diag + k_chol_step (+ k_syrk_lower): bordered Cholesky of the reduced camera system, n = %d" % n_sys,
                              "mfma", n_sys ** 3 / 3.0, 1e12, FP64_MFMA_PEAK_TFLOPS, "TFLOP/s", "chol",
                              note=("latency chain of %d dependent 64-column steps" % ((n_sys + 63) // 64)) if n_sys < 4096
                              else "256-column strips + rank-256 trailing updates"))
    # k_schur_items is a gather: its roofline is priced on the UNIQUE bytes it has to touch per launch (every G block once +
    # the pair ids), which is what an ideal cache hierarchy would move from HBM.  The rate at which blocks are pulled through
    # L2 (every block ~10 times) is reported beside it as l2_gather_rate, never as `frac`.
    n_loc, n_pairs = main["n_obs_local"], main["n_pairs"]
    schur_unique = n_loc * 3 * d * 8 + n_pairs * 8.0
    g_block = 256 if (d == 10 and os.environ.get("SFM_G_PAD", "1") != "0") else 3 * d * 8      # bytes a pulled G block occupies
    schur_gathered = (2.0 * n_pairs - n_loc) * g_block + n_loc * 24.0
    roofs = cam_roofs + [
        roof("k_schur_items (S blocks = -sum G_k G_k2^T per camera pair, f64 MFMA)", "hbm", schur_unique, 1e9, HBM_PEAK_GBS, "GB/s",
             "schur_items", "k_schur_items",
             note="a gather: `achieved` / `frac` price the unique data of a launch (%.0f MB of G + %.0f MB of pair indices) "
                  "against the HBM peak; `traffic` is what leaves the L2s for the Infinity Cache / HBM (PMC), `wasted_traffic` = "
                  "traffic / unique bytes, `l2_gather_rate` the rate at which G blocks are pulled through L2 (2 x %d B per camera "
                  "pair: blocks are stored at a 256-byte stride for d = 10 so that one is exactly two 128-byte lines), `flop_frac` the "
                  "useful 2 x 3 x d x d FLOP per pair against the fp64 MFMA rate" % (
                      n_loc * 3 * d * 8 / 1e6, n_pairs * 8 / 1e6, g_block)),
        roof("k_lin_obs (residual + 2x(%d+3) Jacobian + Huber scaling)" % d, "hbm", float(bytes_per_obs) * main["n_obs_local"],
             1e9, HBM_PEAK_GBS, "GB/s", "lin_obs", "k_lin_obs"),
    ]
    roofs = [r for r in roofs if r]
    for r in roofs:
        if r["kernel"].startswith("k_schur_items"):
            sec = r["avg_us"] * 1e-6
            r["unique_bytes"] = schur_unique
            r["l2_gather_rate"] = {"value": round(schur_gathered / sec / 1e9, 1), "unit": "GB/s", "bytes_per_launch": schur_gathered}
            r["flop_frac"] = round(2.0 * 3 * d * d * n_pairs / sec / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4)
            if r["traffic"]:
                r["wasted_traffic"] = round(r["traffic"] / r["unique_bytes"], 2)
            # the whole Schur stage (items + assemble + right-hand side) shares this entry's time share
            r["time_share"] = kernels["schur"]["share"]
            r["stage_us"] = kernels["schur"]["us_per_launch"]
    # headline roofline object = the kernel (group) with the largest share of the step time
    roofline = max(roofs, key=lambda r: r["time_share"]) if roofs else None

    # ------------------------------------------------------------------ secondary BA figures on the same scene
    def brief(r, what):
        return {"value": r["value"], "unit": "LM-iterations/s", "ms_per_step": r["ms_per_step"],
                "damped_solves": r["damped_solves"], "cost_end": r["cost_end"], "workload": what}
    ba_d6 = ba_mixed = None
    if d == 10 and not args.no_d6:
        ba_d6 = brief(fixed_schedule_run(6, "fp64", False),
                      "same scene, camera block [rvec,t] with fixed K (cam_dim 6, n = %d)" % (6 * C))
    if not args.no_mixed:
        rm = timed_and_profiled(d, "mixed")
        ba_mixed = brief(rm, "same scene and schedule, SFM_BA_MIXED: Jacobian rows stored in float32, every sum / W L^-T / S / "
                             "solve in float64 (opt-in; the headline stays float64)")
        ba_mixed["kernels_us"] = {k: round(v[0] / v[1] * 1e3, 2) for k, v in rm["prof"].items() if v[1] > 0}

    ba_alt = None
    if not args.no_alt_camera_solver and (n_sys <= 4096 or big_cg):
        other = "cg" if args.camera_solver == "cholesky" else "cholesky"       # auto = cg
        ra = timed_and_profiled(d, "fp64", camera_solver=other)
        ba_alt = brief(ra, f"same scene and schedule, formed camera system solved by {other} instead of {args.camera_solver}")
        ba_alt["camera_solver"] = other
        ba_alt["cg_iterations_and_fallbacks_timed_pass"] = ra["camera_cg"]
        ba_alt["kernels_us"] = {k: round(v[0] / v[1] * 1e3, 2) for k, v in ra["prof"].items() if v[1] > 0}
    ba_pcg = None
    if not args.no_pcg:
        rp = timed_and_profiled(d, "fp64", solver="pcg")
        ba_pcg = brief(rp, "same scene and schedule, camera system solved by PCG on the implicit Schur complement "
                           "(sfm_ba_solve_pcg: S never formed; rtol 1e-13): the multi-rank / many-camera route")
        ba_pcg["pcg_iterations_timed_pass"] = rp["cg_iters"]
        ba_pcg["pcg_solves_redone_densely_and_worst_residual"] = rp["pcg_stats"]
        ba_pcg["kernels_us"] = {k: round(v[0] / v[1] * 1e3, 2) for k, v in rp["prof"].items() if v[1] > 0}

    # ---- the same sizes on a spatially coherent scene (what a real capture is: the reference's shipped bunny set has 36 views
    # around the object and tracks between neighbouring views only): each point seen by its L nearest cameras, cameras
    # numbered along the hemisphere.  Secondary row: the headline stays the uniformly random BASELINE scene.
    ba_coherent = None
    if args.visibility == "random" and not args.no_coherent:
        sc2 = synth.make_scene(C, P, obs_per_point=Lobs, seed=1004, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002, visibility="nearest")
        rc2 = timed_and_profiled(d, "fp64", shard=shard_scene(sc2))
        ba_coherent = brief(rc2, "same sizes, spatially coherent visibility (synth.make_scene(visibility='nearest')): each point seen by "
                                 "its L nearest cameras, cameras numbered along the hemisphere")
        ba_coherent["camera_cg_iterations_and_fallbacks_timed_pass"] = rc2["camera_cg"]
        ba_coherent["camera_cg_iterations_and_fallbacks_warmup"] = rc2["camera_cg_warmup"]
        ba_coherent["kernels_us"] = {k: round(v[0] / v[1] * 1e3, 2) for k, v in rc2["prof"].items() if v[1] > 0}
        ba_coherent["n_pairs"] = rc2["n_pairs"]
        uq = rc2["n_obs_local"] * 3 * d * 8 + rc2["n_pairs"] * 8.0
        si = ba_coherent["kernels_us"].get("schur_items")
        if si:
            ba_coherent["k_schur_items"] = {"avg_us": si, "unique_bytes": uq, "frac": round(uq / (si * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
            for name in ("r04_pmc_summary_coherent.json", "r03_pmc_summary_coherent.json"):
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
                    wl = pmc["workload"]
                    if world == 1 and (wl["cams"], wl["pts"], wl["obs"], wl["cam_dim"]) == (C, P, sc2.n_obs, d):
                        t = pmc["k_schur_items"]["hbm_bytes_per_launch"]
                        ba_coherent["k_schur_items"].update(traffic=t, wasted_traffic=round(t / uq, 2), traffic_source="profiles/" + name)
                except (OSError, KeyError, ValueError):
                    pass
        del sc2

    # ---- the objective the reference itself evaluates and the drop-in ships with by default (order="reference": projections
    # stacked camera by camera against point-major pixels, sfm_reconstruction.py:480-486): every reprojection row an outlier
    # but a handful, curvature scaled by EPS, p ~ -g / alpha.  Same scene, sizes and fixed schedule.
    ba_reference = None
    if not args.no_reference_order:
        from sfm_amd.reconstruction import reference_pairing
        uv_ref = reference_pairing(sc.uv, sc.cam_idx)            # the pairing is global: permute first, shard afterwards
        pp_ = np.zeros(P + 1, dtype=np.int64)
        np.cumsum(np.bincount(sc.pt_idx, minlength=P), out=pp_[1:])
        ci_r, pi_r, uv_r, pts_r = shard_arrays(sc.cam_idx, sc.pt_idx, uv_ref, sc.pts0, lo, hi)
        rr_ = timed_and_profiled(d, "fp64", shard=(sc.cams0, pts_r, ci_r, pi_r, uv_r))
        ba_reference = brief(rr_, "same scene, sizes and schedule with the reference's own residual pairing (the drop-in's default, "
                                  "order='reference'): all-outlier regime; parity: test_full_size_cfg4_reference_order_against_c_oracle")
        ba_reference["cost_start"] = rr_["cost_start"]
        ba_reference["trial_steps"] = rr_["trial_steps"]
        ba_reference["camera_cg_iterations_and_fallbacks_timed_pass"] = rr_["camera_cg"]
        ba_reference["kernels_us"] = {k: round(v[0] / v[1] * 1e3, 2) for k, v in rr_["prof"].items() if v[1] > 0}
        del uv_ref

    # ------------------------------------------------------------------ end-to-end drop-in call (N = 1 only)
    # wall clock of ONE StructureFromMotion.bundle_adjust() with the reference's settings (ftol = xtol = 1e-4,
    # max_nfev = 100) on the same scene held in the reference's Python containers: dict walk + packing, problem
    # construction on the device, the solve, write-back.  Not part of `value`.
    dropin = None
    if rank == 0 and world == 1 and not args.no_dropin:
        def dropin_call(**kw):
            from sfm_amd.reconstruction import StructureFromMotion
            s = StructureFromMotion(cam_dim=d, device=local_rank, **kw)        # kw empty = the class defaults (order="reference")
            s.poses, s.points3D, s.point_tracks, s.K = sc.state()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ret = s.bundle_adjust()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            r = s.last_ba_result
            out = {"bundle_adjust_wall_s": wall, "returned": None if ret is None else bool(ret), "order": s.ba_order,
                   "phases_s": {k: round(v, 4) for k, v in (s.last_ba_timing or {}).items()},
                   "nfev": r.nfev, "njev": r.njev, "status": r.status, "n_solves": r.n_solves, "cost": r.cost,
                   "workload": f"StructureFromMotion.bundle_adjust() on {C} cams / {P} pts / {n_obs_total} obs held as "
                               "dict / list state, reference solver settings (ftol = xtol = 1e-4, max_nfev = 100)"}
            t0 = time.perf_counter()
            s.compute_reconstruction_stats()
            out["compute_reconstruction_stats_wall_s"] = time.perf_counter() - t0
            return out
        try:
            dropin = dropin_call()                       # as shipped: the reference's own residual pairing
            dropin["aligned_pairing"] = dropin_call(order="aligned")
        except Exception as e:       # a reported extra; never fail the main measurement on it
            dropin = {"error": repr(e)}

    # ------------------------------------------------------------------ matcher workload (cfg2)
    matcher = None
    if not args.no_matcher:
        from sfm_amd import matcher as mt
        n = args.match_n
        d1, d2 = synth.make_descriptors(n, n, seed=1002)
        q_lo, q_hi = (rank * n) // world, ((rank + 1) * n) // world
        q = torch.from_numpy(d1[q_lo:q_hi].astype(np.uint8)).cuda()
        t = torch.from_numpy(d2.astype(np.uint8)).cuda()
        hd = _lib.get_handle(local_rank)

        def match_pass():
            i1, i2, a, b = mt.knn2(q, t, "l2", device=local_rank)
            return mt.ratio_filter(i1, a, b, 0.75, device=local_rank)
        # 20 untimed passes (~9 ms): the chip raises its clock only under sustained load - measured on the same build, the
        # distance kernel's HIP-event time is 371 us over 5 passes after 2, 341 over 20, 328 over 40
        for _ in range(20):
            mq, mtr, md = match_pass()
        barrier_sync()
        tm0 = time.perf_counter()
        for _ in range(args.match_reps):          # timed without per-kernel events
            mq, mtr, md = match_pass()
        barrier_sync()
        tm = max_over_ranks(time.perf_counter() - tm0)
        hd.set_profiling(True)
        hd.profile()
        for _ in range(args.match_reps):          # second pass for the kernel's own duration
            match_pass()
        mprof = hd.profile()
        hd.set_profiling(False)
        pairs = float(n) * float(n) * args.match_reps
        knn_ms, knn_cnt = mprof["knn"]
        mroof = None
        if knn_cnt > 0:
            ops = 2.0 * 128 * (q_hi - q_lo) * n            # i8 multiply-adds of the distance GEMM, per launch
            ach = ops / (knn_ms / knn_cnt * 1e-3) / 1e12
            mroof = {"kernel": "k_knn2_u8_direct<4> (i8 MFMA distance + filtered top-2, operands straight from L2)", "bound": "mfma", "achieved": round(ach, 1),
                     "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s", "frac": round(ach / I8_MFMA_PEAK_TOPS, 4),
                     "avg_us": round(knn_ms / knn_cnt * 1e3, 1), "traffic": None}
        matcher = {"metric": "descriptor-pairs/sec", "value": pairs / tm, "unit": "pairs/s",
                   "workload": f"{n} x {n} x 128 uint8 SIFT-like, brute-force L2 kNN(2) + ratio 0.75 + compaction",
                   "ms_per_pair_of_images": tm / args.match_reps * 1e3, "n_matches": int(mq.shape[0]),
                   "dtype": "u8/i32", "roofline": mroof}
        # the in-tree alternative feature type: ORB, 256-bit strings, Hamming distance (find_matches.py:144) - exact uint8 L2
        # over the unpacked bits on the same int8 kernels
        try:
            if rank != 0:
                raise StopIteration
            n_orb = min(n, 30000)
            rng_o = np.random.default_rng(1006)
            oq = torch.from_numpy(rng_o.integers(0, 256, size=(n_orb, 32), dtype=np.uint8)).cuda()
            ot = torch.from_numpy(rng_o.integers(0, 256, size=(n_orb, 32), dtype=np.uint8)).cuda()
            for _ in range(2):
                mt.knn2(oq, ot, "hamming", device=local_rank)
            torch.cuda.synchronize()
            t_o = time.perf_counter()
            for _ in range(5):
                mt.knn2(oq, ot, "hamming", device=local_rank)
            torch.cuda.synchronize()
            matcher["orb_hamming_256"] = {"value": float(n_orb) * n_orb * 5 / (time.perf_counter() - t_o), "unit": "pairs/s",
                                          "workload": f"{n_orb} x {n_orb} x 256-bit ORB-like, kNN(2), rank 0 only"}
        except StopIteration:
            pass
        except Exception as e:      # a secondary row must not take the line down
            matcher["orb_hamming_256"] = {"error": repr(e)}

    # ---- the drop-in AS CALLED (find_matches.py:272): ImageMatcher.match_features(desc1, desc2) with the descriptors as host
    # float32 arrays (what cv2.SIFT hands over), the returned sequence's len() taken and every match's three attributes read
    # once (what :274-279 does).  Wall clock, uploads / integer check / kernels / download / objects included; N = 1 only.
    if matcher is not None and rank == 0 and world == 1:
        try:
            im = mt.ImageMatcher(device=local_rank)
            f1, f2 = d1.astype(np.float32), d2.astype(np.float32)

            def wall(fn, reps):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    out = fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / reps, out
            t_call, ms_ = wall(lambda: im.match_features(f1, f2), 5)
            t_len, _ = wall(lambda: len(im.match_features(f1, f2)), 5)

            def consume():
                m = im.match_features(f1, f2)
                return sum(1 for x in m if x.queryIdx >= 0 and x.trainIdx >= 0 and x.distance >= 0.0)
            t_iter, n_seen = wall(consume, 3)
            t_eager, _ = wall(lambda: [mt.DMatch(a, b, c) for a, b, c in zip(*mt.match_arrays(f1, f2, 0.75, "auto", local_rank))], 3)
            matcher["match_features_wall_s"] = {
                "call": t_call, "call_and_len": t_len, "call_and_read_every_match": t_iter, "matches": int(n_seen),
                "eager_list_of_DMatch_objects": t_eager,
                "workload": f"ImageMatcher.match_features on two host float32 [{n}, 128] arrays (cfg2): upload 2 x {n * 512 / 1e6:.1f} MB, "
                            "integer check + uint8 conversion, kNN(2), ratio test, download; returns a DMatchList (objects on demand); "
                            "`eager` builds the Python list of DMatch objects up front, as rounds 1-3 did"}
            rng_w = np.random.default_rng(11)
            base_w, _ = synth.make_descriptors(2200, 2, seed=78)
            imgs_w = [np.clip(base_w[rng_w.permutation(2200)[:2000]] + np.rint(rng_w.normal(0, 5.0, size=(2000, 128))), 0, 255).astype(np.float32)
                      for _ in range(18)]
            pairs_w = [(i, j) for i in range(18) for j in range(i + 1, 18)][:148]
            t_loop, _ = wall(lambda: [len(im.match_features(imgs_w[i], imgs_w[j])) for i, j in pairs_w], 2)
            t_batch, _ = wall(lambda: [len(m) for m in im.match_features_batched(imgs_w, pairs_w)], 2)
            matcher["match_features_wall_s"]["image_pairs_148x2000"] = {
                "per_pair_calls_s": t_loop, "one_batched_call_s": t_batch,
                "workload": "148 image pairs of 2,000 float32 SIFT-like descriptors each (the bunny set's pair count): "
                            "match_features per pair as find_matches.py:329-350 calls it / match_features_batched once"}
        except Exception as e:
            matcher["match_features_wall_s"] = {"error": repr(e)}

    # ---- the reference's real call pattern: one match_features per image pair of a preprocessing step
    # (find_matches.py:329-350), a few hundred to a few thousand descriptors per image: all pairs in ONE launch
    # (match_pairs) beside the per-pair loop, host arrays in, match arrays out (uploads included on both sides)
    if matcher is not None and rank == 0 and world == 1:
        try:
            rows = []
            rng = np.random.default_rng(7)
            for per_image in (500, 2000):
                n_img = 18
                base, _ = synth.make_descriptors(per_image + 200, 2, seed=77)
                imgs = [np.clip(base[rng.permutation(base.shape[0])[:per_image]] +
                                np.rint(rng.normal(0, 5.0, size=(per_image, 128))), 0, 255).astype(np.uint8) for _ in range(n_img)]
                pairs = [(i, j) for i in range(n_img) for j in range(i + 1, n_img)][:148]      # the bunny set verified 148 pairs
                mt.match_pairs(imgs, pairs); [mt.match_arrays(imgs[i], imgs[j]) for i, j in pairs[:4]]
                torch.cuda.synchronize(); t0 = time.perf_counter()
                got = mt.match_pairs(imgs, pairs)
                torch.cuda.synchronize(); tb = time.perf_counter() - t0
                t0 = time.perf_counter()
                loop = [mt.match_arrays(imgs[i], imgs[j]) for i, j in pairs]
                torch.cuda.synchronize(); tl = time.perf_counter() - t0
                same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
                           for a, b in zip(got, loop))
                rows.append({"pairs": len(pairs), "descriptors_per_image": per_image, "batched_ms": tb * 1e3,
                             "per_pair_loop_ms": tl * 1e3, "speedup": tl / tb, "identical_results": bool(same),
                             "pairs_per_s_batched": len(pairs) * per_image * per_image / tb})
            matcher["batched_image_pairs"] = rows
        except Exception as e:
            matcher["batched_image_pairs"] = {"error": repr(e)}

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
            cpu_timed = min(3, args.steps)
            cpu_baseline = cb.ba_baseline(sc, d, warmup=args.warmup, timed=cpu_timed)
            # the two legs walk the same trajectory: the GPU's cost after the same number of outer iterations (profiled pass)
            k_cpu = args.warmup + cpu_timed
            if len(profd["cost_trace"]) >= k_cpu:
                cpu_baseline["iterations_run"] = k_cpu
                cpu_baseline["gpu_cost_at_cpu_baseline_iterations"] = profd["cost_trace"][k_cpu - 1]
                cpu_baseline["cost_rel_diff"] = abs(profd["cost_trace"][k_cpu - 1] - cpu_baseline["cost_end"]) / cpu_baseline["cost_end"]
            if matcher is not None:
                cpu_baseline["matcher"] = cb.matcher_baseline(d1.astype(np.uint8), d2.astype(np.uint8))
        except Exception as e:       # the baseline is a reported extra; never fail the GPU measurement on it
            cpu_baseline = {"error": repr(e)}

    # ---- what more GPUs can and cannot buy, written down BEFORE any curve is measured (N = 1): the measured kernel slots of
    # the profiled pass split into what shards with the points, what every rank repeats, and what is exchanged
    scaling_model = None
    if world == 1 and prof:
        per = lambda slot: kernels.get(slot, {}).get("ms_total", 0.0) * 1e3 / args.steps        # us per outer iteration
        shard_us = sum(per(k_) for k_ in ("lin_obs", "lin_rest", "build_G", "schur", "backsub", "step"))
        repl_us = per("chol") + per("trsv")
        step_us = prof_elapsed / args.steps * 1e6
        other_us = max(step_us - shard_us - repl_us, 0.0)
        solves = profd["damped_solves"] / args.steps
        trials = profd["trial_steps"] / args.steps
        big_bytes = (n_sys * (n_sys + 1) // 2 + n_sys) * 8
        BUS_GBS, LAT_US = 150.0, 20.0
        def ar_us(nbytes, N):          # ring all-reduce: 2 (N - 1) / N of the buffer over one link's rate + a latency term
            return 0.0 if N == 1 else LAT_US + 2.0 * (N - 1) / N * nbytes / (BUS_GBS * 1e3)
        proj = {}
        for N in (2, 4, 8):
            t = shard_us / N + repl_us + other_us + solves * (ar_us(big_bytes, N) + ar_us((n_sys + 2) * 8, N)) + \
                ar_us((2 * n_sys + 4) * 8, N) + trials * ar_us(40, N)
            proj[str(N)] = {"us_per_outer_iteration": round(t, 1), "speedup": round(step_us / t, 2)}
        scaling_model = {
            "per_outer_iteration_us": {"sharded_with_the_points": round(shard_us, 1), "replicated_camera_solve": round(repl_us, 1),
                                       "host_and_unattributed": round(other_us, 1), "measured_total": round(step_us, 1)},
            "damped_solves_per_outer_iteration": round(solves, 2), "trial_steps_per_outer_iteration": round(trials, 2),
            "exchanged_per_damped_solve_bytes": big_bytes + (n_sys + 2) * 8,
            "assumed": {"allreduce_bus_GBps": BUS_GBS, "allreduce_latency_us": LAT_US,
                        "note": "one xGMI link ~153 GB/s per direction (MI355X_MICROARCH.md); ring all-reduce of the packed "
                                "lower triangle of [S | r]; small exchanges priced at the latency term alone"},
            "projected": proj,
            "reading": "strong scaling of a fixed scene: only the first term shrinks with N; DESIGN.md section 5"}
        if ba_pcg and ba_pcg.get("pcg_iterations_timed_pass"):
            # the S-free route (sfm_ba_solve_pcg): nothing n x n is formed or exchanged; per PCG iteration two rank-local passes over
            # G (they shard with the points) and ONE all-reduce of an n-vector.  Measured here on one rank: iterations and the time
            # of the two solve slots of the profiled pass; projected: the local part / N but never below a floor of launch-bound
            # time (five small launches per iteration), plus the all-reduce
            kp = ba_pcg["kernels_us"]
            n_sys_pcg = 2 * profd["damped_solves"] - (profd["damped_solves"] - kernels.get("trsv", {}).get("launches", 0))
            its_pcg = ba_pcg["pcg_iterations_timed_pass"]
            solve_us = (kp.get("chol", 0.0) * profd["damped_solves"] + kp.get("trsv", 0.0) * kernels.get("trsv", {}).get("launches", 0))
            it_us = solve_us / max(its_pcg, 1)
            FLOOR_US = 30.0
            other_pcg = ba_pcg["ms_per_step"] * 1e3 - solve_us / args.steps          # linearisation, G, back-substitution, trial: shards
            projp = {}
            for N in (2, 4, 8):
                per_it = max(it_us / N, FLOOR_US) + ar_us(n_sys * 8, N)
                t = other_pcg / N + its_pcg / args.steps * per_it
                projp[str(N)] = {"us_per_outer_iteration": round(t, 1), "speedup_vs_one_gpu_dense_route": round(step_us / t, 2)}
            scaling_model["s_free_route"] = {
                "measured_one_rank": {"pcg_iterations_per_outer_iteration": round(its_pcg / args.steps, 1),
                                      "us_per_pcg_iteration": round(it_us, 1), "systems": n_sys_pcg,
                                      "us_per_outer_iteration": round(ba_pcg["ms_per_step"] * 1e3, 1)},
                "assumed": {"launch_bound_floor_us_per_iteration": FLOOR_US, "exchanged_per_iteration_bytes": n_sys * 8},
                "projected": projp}

    rccl_seen = None
    if getattr(comm, "in_library", False):
        import ctypes
        a_, b_ = _lib.i32(0), _lib.i32(0)
        if comm.h.lib.sfm_comm_info(comm.h._h, ctypes.byref(a_), ctypes.byref(b_)) == 0:
            rccl_seen = int(a_.value)          # ncclCommCount of the communicator the exchanges ran on

    if rank == 0:
        out = {
            "metric": "BA LM-iterations/sec + descriptor-pairs/sec, 200 cams / 100k pts, 1->8 MI355X",
            "value": value, "unit": "LM-iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BA: {C} cams / {P} pts / {n_obs_total} obs, cam block {d} "
                                   f"({'reference 10-parameter block + regulariser' if d == 10 else 'fixed K'}), "
                                   "aligned residual order, Huber, SciPy-TRF control flow, fixed schedule"
                                   + ("" if args.visibility == "random" else ", spatially coherent visibility"),
                       "parallelism": f"points sharded over {world} rank(s), cameras replicated, RCCL all-reduce of [S|r]",
                       "comm": comm_kind, "rccl_ranks_seen": rccl_seen,
                       "seed": 1004},
            "ba": {"damped_solves": main["damped_solves"], "trial_steps": main["trial_steps"], "cost_start": main["cost_start"],
                   "cost_end": main["cost_end"], "solves_per_s": main["damped_solves"] / elapsed, "kernels": kernels,
                   "kernels_note": "HIP-event times from a second pass of the same schedule; the timed pass carries no events",
                   "loop": "sfm_ba_trf_outer (library-side trust-region loop)", "camera_solver": args.camera_solver,
                   "camera_cg_iterations_and_fallbacks_timed_pass": main["camera_cg"],
                   "camera_cg_iterations_and_fallbacks_warmup": main["camera_cg_warmup"]},
            "ba_cam_dim6": ba_d6, "ba_mixed_precision": ba_mixed, "ba_other_camera_solver": ba_alt, "ba_pcg_solver": ba_pcg,
            "ba_coherent_scene": ba_coherent, "ba_reference_order": ba_reference,
            "dropin": dropin, "scaling_model": scaling_model,
            "roofline": roofline, "rooflines": roofs, "cpu_baseline": cpu_baseline, "matcher": matcher,
            "driver_rows": driver_rows,
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
