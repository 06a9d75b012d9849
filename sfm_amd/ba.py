"""GPU backend of the bundle-adjustment solve: owns the parameters in HBM and runs the stages of
include/sfm_amd.h through ctypes.  No CPU fallback - constructing it without the HIP library
or without a GPU raises."""
from __future__ import annotations

import ctypes as C
import math
import weakref

import numpy as np

from . import _lib
from .comm import LocalComm
from .trf import trf, TRFResult

_STRUCT_FIELDS = {  # name -> length as a function of (n_cams, n_pts, view)
    "pt_ptr": lambda c, p, v: p + 1, "cam_ptr": lambda c, p, v: c + 1, "cam_obs": lambda c, p, v: v.n_obs,
    "blk_ptr": lambda c, p, v: c * (c + 1) // 2 + 1, "pair_k": lambda c, p, v: v.n_pairs,
    "pair_k2": lambda c, p, v: v.n_pairs, "item_ptr": lambda c, p, v: c * (c + 1) // 2 + 1,
    "item_beg": lambda c, p, v: v.n_items, "item_end": lambda c, p, v: v.n_items, "xcd_ptr": lambda c, p, v: 9,
    "xcd_items": lambda c, p, v: v.n_items, "cch_ptr": lambda c, p, v: c + 1, "cch_beg": lambda c, p, v: v.n_cchunks,
    "cch_end": lambda c, p, v: v.n_cchunks,
}


class GpuBA:
    """One rank's shard of a BA problem (cameras replicated, points/observations local).

    cams [C,d] float64 (d = 10: rvec,t,fx,fy,cx,cy - the reference's block,
    /root/reference/utils/sfm_reconstruction.py:416-427; d = 6: rvec,t with shared K0),
    pts [P,3], observations point-major as the reference packs them (:430-435).  Only these packed
    arrays go in: the index structure of the Schur complement is built on the device by
    sfm_ba_create_problem.  precision="mixed" stores the Jacobian rows in float32 (every sum, W L^-T, S and
    the solve stay float64).  solver="dense" forms the reduced camera system and factors it (default);
    solver="pcg" solves it by preconditioned conjugate gradients on the implicit Schur complement
    (sfm_ba_solve_pcg): no n x n matrix, one n-vector exchanged per iteration between ranks.
    uv_pairing="reference": the library pairs observations and pixels as the reference's objective does (sfm_reconstruction.py:
    480-486: the q-th observation in camera-sorted order meets uv[q]) on the device; "as_given": observation k meets uv[k].
    camera_solver (solver="dense" only): how the formed system is solved - "auto" (default): conjugate gradients on
    the block-scaled system (n <= 2048: one persistent launch per system; up to 4096: one launch per iteration; beyond:
    tile-streaming over the symmetric half of the matrix), with the bordered Cholesky as fallback; "cholesky" / "cg"
    force one.
    """

    def __init__(self, cams, pts, cam_idx, pt_idx, uv, K0, width=1024.0, height=768.0,
                 reg_weight=0.1, device=0, comm=None, precision="fp64", solver="dense", pcg_rtol=1e-13,
                 pcg_max_iter=None, camera_solver="auto", uv_pairing="as_given"):
        import torch
        self.torch = torch
        self.comm = comm or LocalComm()
        self.h = _lib.get_handle(device)
        self.dev = torch.device("cuda", device)
        cams = np.ascontiguousarray(cams, dtype=np.float64)
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
        self.C, self.d = cams.shape
        self.P = pts.shape[0]
        if self.d not in (6, 10):
            raise ValueError("camera block must have 6 or 10 parameters")
        if precision not in ("fp64", "mixed"):
            raise ValueError(f"unknown precision {precision!r}")
        self.precision = precision
        if solver not in ("dense", "pcg"):
            raise ValueError(f"unknown solver {solver!r}")
        self.solver, self.pcg_rtol, self.pcg_max_iter = solver, float(pcg_rtol), pcg_max_iter
        if camera_solver not in ("auto", "cholesky", "cg"):
            raise ValueError(f"unknown camera_solver {camera_solver!r}")
        self.camera_solver = camera_solver
        self.cg_iters = 0
        ci = np.ascontiguousarray(cam_idx, dtype=np.int32)
        pi = np.ascontiguousarray(pt_idx, dtype=np.int32)
        uvh = np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2)
        self.N = int(ci.shape[0])
        if self.N == 0:
            raise ValueError("no observations")
        if pi.shape[0] != self.N or uvh.shape[0] != self.N:
            raise ValueError("cam_idx, pt_idx and uv must have one entry per observation")
        self.n = self.C * self.d
        d = _lib.BADesc()
        d.n_cams, d.n_pts, d.cam_dim = self.C, self.P, self.d
        d.apply_reg = 1 if (self.d == 10 and self.comm.rank == 0) else 0
        d.n_obs = self.N
        # host pointers: sfm_ba_create_problem copies them into device memory it owns
        d.cam_idx, d.pt_idx, d.uv = ci.ctypes.data, pi.ctypes.data, uvh.ctypes.data
        d.fx0, d.fy0, d.cx0, d.cy0 = (float(v) for v in K0)
        d.width, d.height, d.reg_weight = float(width), float(height), float(reg_weight)
        d.precision = _lib.PREC_MIXED if precision == "mixed" else _lib.PREC_FP64
        d.camera_solver = {"auto": _lib.CAMERA_AUTO, "cholesky": _lib.CAMERA_CHOLESKY, "cg": _lib.CAMERA_CG}[camera_solver]
        if uv_pairing not in ("as_given", "reference"):
            raise ValueError(f"unknown uv_pairing {uv_pairing!r}")
        if uv_pairing == "reference" and self.comm.world_size > 1:
            raise ValueError("the reference pairing permutes ALL observations: apply it before sharding (reference_pairing)")
        d.uv_pairing = _lib.UV_REFERENCE_PAIRING if uv_pairing == "reference" else _lib.UV_AS_GIVEN
        self.desc = d
        self._pp = _lib.vp()
        rc = self.h.lib.sfm_ba_create_problem(self.h._h, C.byref(d), C.byref(self._pp))
        if rc != 0:
            msg = self.h.lib.sfm_last_error(self.h._h).decode()
            raise ValueError(msg) if rc == -1 else _lib.SfmError(f"sfm_ba_create_problem failed ({rc}): {msg}")
        lay = _lib.BALayout()
        self.h.lib.sfm_ba_get_layout(self._pp, C.byref(lay))
        self.lay = lay
        self.ws = torch.zeros(lay.total_bytes, dtype=torch.uint8, device=self.dev)
        self.h.call("sfm_ba_bind_workspace", self._pp, C.c_void_p(self.ws.data_ptr()), lay.total_bytes)
        if self.comm.world_size > 1:           # one rank's shard: the camera solve must take the same route on every rank
            self.h.call("sfm_ba_set_sharded", self._pp, 1)
        sv = _lib.BAStructureView()
        self.h.lib.sfm_ba_get_structure(self._pp, C.byref(sv))
        self.sview = sv
        self.n_pairs, self.n_items = int(sv.n_pairs), int(sv.n_items)
        self.x = torch.from_numpy(np.concatenate([cams.ravel(), pts.ravel()])).to(self.dev)
        self.x_new = torch.empty_like(self.x)
        self._sc = (C.c_double * _lib.SC_COUNT)()
        self.n_solves = 0
        self._reduce_cb = None
        self._open_trf = weakref.WeakSet()

    def close(self):
        """Release the problem.  Loop states begun on it are ended first: when a cycle (a kept traceback, say) is collected the
        finalizers of backend and loop state run in no particular order, and sfm_ba_trf_end on a destroyed problem - or
        sfm_ba_destroy_problem on a destroyed handle - is a use after free."""
        for st in list(getattr(self, "_open_trf", ())):
            st.close()
        if getattr(self, "_pp", None):
            if self.h._h:                      # the handle outlives its problems unless the interpreter is tearing down
                self.h.lib.sfm_synchronize(self.h._h)
                self.h.lib.sfm_ba_destroy_problem(self._pp)
            self._pp = _lib.vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solver_stats(self):
        """(CG iterations, fallbacks to the factorisation) of the camera_solver="cg" route so far."""
        a, b = C.c_int64(0), C.c_int64(0)
        self.h.lib.sfm_ba_solver_stats(self._pp, C.byref(a), C.byref(b))
        return a.value, b.value

    def pcg_stats(self):
        """solver="pcg": (damped solves redone by the formed-S route because PCG ran out of iterations, worst relative
        residual PCG had reached in one of them)."""
        a, b = C.c_int64(0), C.c_double(0.0)
        self.h.lib.sfm_ba_pcg_stats(self._pp, C.byref(a), C.byref(b))
        return a.value, b.value

    # ---- structure (inspection / tests)
    def structure(self):
        """The index structure built on the device, as a dict of int32 NumPy arrays (names of sfm_ba_structure)."""
        out = {}
        for name, length in _STRUCT_FIELDS.items():
            n = int(length(self.C, self.P, self.sview))
            a = np.empty(n, dtype=np.int32)
            if n:
                self.h.call("sfm_copy_to_host", C.c_void_p(a.ctypes.data), C.c_void_p(getattr(self.sview, name)), n * 4)
            out[name] = a
        return out

    # ---- workspace views
    def view(self, off, count, dtype=None):
        dtype = dtype or self.torch.float64
        return self.ws[off:off + dtype.itemsize * count].view(dtype)

    @property
    def rec_dtype(self):
        return self.torch.float32 if self.precision == "mixed" else self.torch.float64

    def scalars(self):
        self.h.call("sfm_ba_read_scalars", self._pp, self._sc)
        return self._sc

    @property
    def _dist(self):
        return self.comm.world_size > 1 or getattr(self.comm, 'force', False)

    def x_norm(self):
        part = (self.x[self.n:] ** 2).sum().reshape(1)
        self.comm.allreduce_sum(part)
        return math.sqrt(float(part.item()) + float((self.x[:self.n] ** 2).sum().item()))

    # ---- TRF backend protocol
    def linearize(self):
        L = self.lay
        self.h.call("sfm_ba_linearize", self._pp, C.c_void_p(self.x.data_ptr()))
        if self._dist:
            self.comm.allreduce_sum(self.view(L.reduce_lin_off, L.reduce_lin_count))
            self.comm.allreduce_max(self.view(L.gmax_off, 2))
        self.h.call("sfm_ba_finish_linearize", self._pp)
        s = self.scalars()
        return s[_lib.SC_COST], math.sqrt(s[_lib.SC_GNORM2]), s[_lib.SC_GINF], s[_lib.SC_HDIAG]

    def _pcg_max_iter(self):
        return int(self.pcg_max_iter) if self.pcg_max_iter else 400      # then the formed-S fallback inside sfm_ba_solve_pcg

    def solve(self, alpha, want_q):
        L = self.lay
        wq = 1 if want_q else 0
        if self.solver == "pcg":
            fn, _keep, user = self._reduce_hook()
            its = C.c_int32(0)
            self.h.call("sfm_ba_solve_pcg", self._pp, C.c_double(alpha), wq, C.c_double(self.pcg_rtol),
                        self._pcg_max_iter(), fn, user, C.byref(its))
            self.cg_iters += its.value
            return self._solve_scalars(alpha)
        self.h.call("sfm_ba_schur_build", self._pp, C.c_double(alpha))
        if self._dist:
            # the factorisation reads only the lower triangle of S: ranks exchange n(n+1)/2 + n doubles, not n^2 + n
            self.h.call("sfm_ba_pack_system", self._pp)
            self.comm.allreduce_sum(self.view(L.reduce_Sp_off, L.reduce_Sp_count))
            self.h.call("sfm_ba_unpack_system", self._pp)
        self.h.call("sfm_ba_schur_solve", self._pp, C.c_double(alpha), wq)
        if self._dist:
            self.comm.allreduce_sum(self.view(L.reduce_q_off, L.reduce_q_count))
        self.h.call("sfm_ba_finish_solve", self._pp, wq)
        return self._solve_scalars(alpha)

    def _solve_scalars(self, alpha):
        s = self.scalars()
        self.n_solves += 1
        fail = s[_lib.SC_CHOL_FAIL]
        p2, pq = s[_lib.SC_PNORM2], s[_lib.SC_PQ]
        if fail != 0.0 or not (math.isfinite(p2) and math.isfinite(pq)):
            why = {1.0: "reduced camera system not positive definite", 2.0: "triangular solve stalled"}.get(
                fail, "damped step is not finite")
            raise _lib.SfmNumericError(f"{why} at alpha={alpha}")
        return math.sqrt(p2), pq

    def step(self, scale):
        L = self.lay
        xp, xn = C.c_void_p(self.x.data_ptr()), C.c_void_p(self.x_new.data_ptr())
        self.h.call("sfm_ba_step", self._pp, xp, C.c_double(scale), xn)
        if self._dist:
            self.comm.allreduce_sum(self.view(L.reduce_step_off, L.reduce_step_count))
        self.h.call("sfm_ba_finish_step", self._pp, xp, C.c_double(scale), xn)
        s = self.scalars()
        return (s[_lib.SC_JS2], s[_lib.SC_GTS], s[_lib.SC_COST_NEW], math.sqrt(s[_lib.SC_SNORM2]),
                math.sqrt(s[_lib.SC_XNEW_NORM2]))

    def accept(self):
        self.x, self.x_new = self.x_new, self.x

    # ---- the trust-region loop inside the library (sfm_ba_trf_*): same state machine as sfm_amd.trf
    def _reduce_hook(self):
        """ctypes callback the C loop calls between stages when the problem is sharded over ranks."""
        if not self._dist:
            return _lib.REDUCE_FN(), None, None
        if getattr(self.comm, "in_library", False):       # RCCL inside the library: no Python between the stages
            fn, user = self.comm.reduce_hook()
            return fn, fn, user
        base = self.ws.data_ptr()

        def cb(_user, ptr, count, op):
            try:
                t = self.view(int(ptr) - base, int(count))
                (self.comm.allreduce_max if op == 1 else self.comm.allreduce_sum)(t)
                return 0
            except Exception:          # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1

        fn = _lib.REDUCE_FN(cb)
        return fn, fn, None

    def trf_begin(self, ftol=1e-4, xtol=1e-4, gtol=1e-8, max_nfev=100, max_outer=None, check_tolerances=True):
        return CTrf(self, ftol, xtol, gtol, max_nfev, max_outer, check_tolerances)

    def run_trf(self, **kw):
        """sfm_ba_run_trf semantics: the whole solve with the loop in the library.  Returns TRFResult."""
        st = self.trf_begin(**kw)
        try:
            while st.outer():
                pass
            return st.result()
        finally:
            st.close()

    # ---- extras
    def cost(self, x=None):
        L = self.lay
        x = self.x if x is None else x
        self.h.call("sfm_ba_cost", self._pp, C.c_void_p(x.data_ptr()))
        red = self.view(L.reduce_step_off, L.reduce_step_count)
        if self._dist:
            self.comm.allreduce_sum(red)
        return float(red[2].item())

    def reproj_errors(self, x=None, shared_k=True):
        """||proj - uv|| per observation; shared_k=False uses each camera's own intrinsics (d=10)."""
        x = self.x if x is None else x
        out = self.torch.empty(self.N, dtype=self.torch.float64, device=self.dev)
        self.h.call("sfm_ba_reproj_errors", self._pp, C.c_void_p(x.data_ptr()), 1 if shared_k else 0,
                    C.c_void_p(out.data_ptr()))
        return out

    def residual_norm2(self, x=None, shared_k=False):
        """sum ||proj - uv||^2 over this rank's observations at x (sfm_ba_residual_norm2), as a Python float."""
        x = self.x if x is None else x
        out = C.c_double(0.0)
        self.h.call("sfm_ba_residual_norm2", self._pp, C.c_void_p(x.data_ptr()), 1 if shared_k else 0, C.byref(out))
        return out.value

    def params(self):
        """(cams [C,d], pts [P,3]) of this rank as NumPy arrays."""
        x = self.x.cpu().numpy()
        return x[:self.n].reshape(self.C, self.d).copy(), x[self.n:].reshape(self.P, 3).copy()


class CTrf:
    """Steppable handle on the library's trust-region loop (sfm_ba_trf_begin / _outer / _result / _end)."""

    def __init__(self, be, ftol, xtol, gtol, max_nfev, max_outer, check_tolerances):
        self.be = be
        o = _lib.TRFOptions()
        o.ftol, o.xtol, o.gtol = float(ftol), float(xtol), float(gtol)
        o.max_nfev = int(min(max_nfev, 2 ** 31 - 1))
        o.max_outer = -1 if max_outer is None else int(max_outer)
        o.check_tolerances = 1 if check_tolerances else 0
        o.solver = _lib.SOLVER_PCG if be.solver == "pcg" else _lib.SOLVER_DENSE
        o.pcg_rtol, o.pcg_max_iter = be.pcg_rtol, be._pcg_max_iter()
        self._fn, self._keep, user = be._reduce_hook()
        self._st = _lib.vp()
        be.h.call("sfm_ba_trf_begin", be._pp, C.c_void_p(be.x.data_ptr()), C.byref(o), self._fn, user, C.byref(self._st))
        be._open_trf.add(self)

    def outer(self):
        more = C.c_int(0)
        rc = self.be.h.lib.sfm_ba_trf_outer(self._st, C.byref(more))
        if rc != 0:
            self.be.h.check(rc, "sfm_ba_trf_outer")
        return bool(more.value)

    def result(self):
        r = _lib.TRFResultC()
        self.be.h.lib.sfm_ba_trf_result(self._st, C.byref(r))
        n = self.be.h.lib.sfm_ba_trf_trace(self._st, None, 0)
        buf = (C.c_double * (4 * max(n, 1)))()
        self.be.h.lib.sfm_ba_trf_trace(self._st, buf, n)
        trace = [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3] != 0.0) for i in range(n)]
        self.be.n_solves = r.n_solves
        self.be.cg_iters = r.cg_iters
        return TRFResult(r.cost, r.nfev, r.njev, r.status, r.optimality, r.n_solves, trace)

    def close(self):
        if getattr(self, "_st", None):
            if self.be._pp and self.be.h._h:   # never after the problem or the handle is gone (GpuBA.close ends us first)
                self.be.h.lib.sfm_ba_trf_end(self._st)
            self._st = _lib.vp()
            self.be._open_trf.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def reproj_errors(cams, pts, cam_idx, pt_idx, uv, K, shared_k=True, device=0):
    """Per-observation reprojection error from the packed arrays alone (sfm_reproj_errors): what
    compute_reconstruction_stats needs - no Schur structure, no workspace.  Returns a CUDA tensor."""
    import torch
    h = _lib.get_handle(device)
    dev = torch.device("cuda", device)
    cams = np.ascontiguousarray(cams, dtype=np.float64)
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
    ci = np.ascontiguousarray(cam_idx, dtype=np.int32)
    pi = np.ascontiguousarray(pt_idx, dtype=np.int32)
    if ci.size and (ci.min() < 0 or ci.max() >= cams.shape[0] or pi.min() < 0 or pi.max() >= pts.shape[0]):
        raise ValueError("index out of range")
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    t_ci, t_pi, t_uv = up(ci), up(pi), up(np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2))
    x = up(np.concatenate([cams.ravel(), pts.ravel()]))
    out = torch.empty(ci.shape[0], dtype=torch.float64, device=dev)
    h.call("sfm_reproj_errors", cams.shape[0], cams.shape[1], ci.shape[0], C.c_void_p(t_ci.data_ptr()),
           C.c_void_p(t_pi.data_ptr()), C.c_void_p(t_uv.data_ptr()), C.c_void_p(x.data_ptr()),
           float(K[0]), float(K[1]), float(K[2]), float(K[3]), 1 if shared_k else 0, C.c_void_p(out.data_ptr()))
    return out


def solve_ba(cams, pts, cam_idx, pt_idx, uv, K0, width=1024.0, height=768.0, ftol=1e-4, xtol=1e-4,
             gtol=1e-8, max_nfev=100, device=0, comm=None, **trf_kw):
    """Run the reference's solver settings (sfm_reconstruction.py:506-514) on one shard.
    Returns (TRFResult, cams, pts, backend)."""
    be = GpuBA(cams, pts, cam_idx, pt_idx, uv, K0, width, height, device=device, comm=comm)
    res = trf(be, ftol=ftol, xtol=xtol, gtol=gtol, max_nfev=max_nfev, **trf_kw)
    c, p = be.params()
    return res, c, p, be
