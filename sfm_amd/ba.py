"""GPU backend of the bundle-adjustment solve: owns the parameters in HBM and runs the stages of
include/sfm_amd.h through ctypes.  No CPU fallback - constructing it without the HIP library
or without a GPU raises."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib
from .comm import LocalComm
from .structure import build_structure
from .trf import trf, TRFResult


class GpuBA:
    """One rank's shard of a BA problem (cameras replicated, points/observations local).

    cams [C,d] float64 (d = 10: rvec,t,fx,fy,cx,cy - the reference's block,
    /root/reference/utils/sfm_reconstruction.py:416-427; d = 6: rvec,t with shared K0),
    pts [P,3], observations point-major as the reference packs them (:430-435).
    """

    def __init__(self, cams, pts, cam_idx, pt_idx, uv, K0, width=1024.0, height=768.0,
                 reg_weight=0.1, device=0, comm=None, structure=None):
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
        st = structure or build_structure(cam_idx, pt_idx, self.C, self.P)
        self.st = st
        self.N = st.n_obs
        self.n = self.C * self.d
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        self.t_cam_idx, self.t_pt_idx = up(st.cam_idx), up(st.pt_idx)
        self.t_uv = up(np.ascontiguousarray(uv, dtype=np.float64).reshape(-1, 2))
        self.t_pt_ptr, self.t_cam_ptr, self.t_cam_obs = up(st.pt_ptr), up(st.cam_ptr), up(st.cam_obs)
        self.t_blk_ptr, self.t_pair_k, self.t_pair_k2 = up(st.blk_ptr), up(st.pair_k), up(st.pair_k2)
        self.t_item_ptr, self.t_item_beg, self.t_item_end = up(st.item_ptr), up(st.item_beg), up(st.item_end)
        self.t_cch_ptr, self.t_cch_beg, self.t_cch_end = up(st.cch_ptr), up(st.cch_beg), up(st.cch_end)
        self.t_xcd_ptr, self.t_xcd_items = up(st.xcd_ptr), up(st.xcd_items)
        self.xcd_max = int(np.max(np.diff(st.xcd_ptr)))
        lay = _lib.BALayout()
        rc = self.h.lib.sfm_ba_get_layout(self.C, self.P, self.N, self.d, st.n_items, st.n_cchunks, C.byref(lay))
        if rc != 0:
            raise _lib.SfmError(f"sfm_ba_get_layout failed ({rc})")
        self.lay = lay
        self.ws = torch.zeros(lay.total_bytes, dtype=torch.uint8, device=self.dev)
        x0 = np.concatenate([cams.ravel(), pts.ravel()])
        self.x = up(x0)
        self.x_new = torch.empty_like(self.x)
        p = _lib.BAProblem()
        p.n_cams, p.n_pts, p.cam_dim = self.C, self.P, self.d
        p.apply_reg = 1 if (self.d == 10 and self.comm.rank == 0) else 0
        p.n_obs = self.N
        for name, t in (("cam_idx", self.t_cam_idx), ("pt_idx", self.t_pt_idx), ("uv", self.t_uv),
                        ("pt_ptr", self.t_pt_ptr), ("cam_ptr", self.t_cam_ptr), ("cam_obs", self.t_cam_obs),
                        ("blk_ptr", self.t_blk_ptr), ("pair_k", self.t_pair_k), ("pair_k2", self.t_pair_k2),
                        ("item_ptr", self.t_item_ptr), ("item_beg", self.t_item_beg), ("item_end", self.t_item_end),
                        ("cch_ptr", self.t_cch_ptr), ("cch_beg", self.t_cch_beg), ("cch_end", self.t_cch_end),
                        ("xcd_ptr", self.t_xcd_ptr), ("xcd_items", self.t_xcd_items)):
            setattr(p, name, t.data_ptr())
        p.n_pairs, p.n_items, p.n_cchunks = st.n_pairs, st.n_items, st.n_cchunks
        p.xcd_max_items = self.xcd_max
        p.fx0, p.fy0, p.cx0, p.cy0 = (float(v) for v in K0)
        p.width, p.height, p.reg_weight = float(width), float(height), float(reg_weight)
        p.workspace, p.workspace_bytes = self.ws.data_ptr(), lay.total_bytes
        self.prob = p
        self._pp = C.byref(p)
        self._sc = (C.c_double * _lib.SC_COUNT)()
        self.n_solves = 0

    # ---- workspace views
    def view(self, off, count):
        return self.ws[off:off + 8 * count].view(self.torch.float64)

    def scalars(self):
        self.h.call("sfm_ba_read_scalars", self._pp, self._sc)
        return self._sc

    def x_norm(self):
        t = self.torch
        part = (self.x[self.n:] ** 2).sum().reshape(1)
        self.comm.allreduce_sum(part)
        return math.sqrt(float(part.item()) + float((self.x[:self.n] ** 2).sum().item()))

    # ---- TRF backend protocol
    def linearize(self):
        L = self.lay
        self.h.call("sfm_ba_linearize", self._pp, C.c_void_p(self.x.data_ptr()))
        if self.comm.world_size > 1 or getattr(self.comm, 'force', False):
            self.comm.allreduce_sum(self.view(L.reduce_lin_off, L.reduce_lin_count))
            self.comm.allreduce_max(self.view(L.gmax_off, 2))
        self.h.call("sfm_ba_finish_linearize", self._pp)
        s = self.scalars()
        return s[_lib.SC_COST], math.sqrt(s[_lib.SC_GNORM2]), s[_lib.SC_GINF], s[_lib.SC_HDIAG]

    def solve(self, alpha, want_q):
        L = self.lay
        wq = 1 if want_q else 0
        self.h.call("sfm_ba_schur_build", self._pp, C.c_double(alpha))
        if self.comm.world_size > 1 or getattr(self.comm, 'force', False):
            # the factorisation reads only the lower triangle of S: ranks exchange n(n+1)/2 + n doubles, not n^2 + n
            self.h.call("sfm_ba_pack_system", self._pp)
            self.comm.allreduce_sum(self.view(L.reduce_Sp_off, L.reduce_Sp_count))
            self.h.call("sfm_ba_unpack_system", self._pp)
        self.h.call("sfm_ba_schur_solve", self._pp, C.c_double(alpha), wq)
        if self.comm.world_size > 1 or getattr(self.comm, 'force', False):
            self.comm.allreduce_sum(self.view(L.reduce_q_off, L.reduce_q_count))
        self.h.call("sfm_ba_finish_solve", self._pp, wq)
        s = self.scalars()
        self.n_solves += 1
        if s[_lib.SC_CHOL_FAIL] != 0.0:
            raise _lib.SfmError(f"reduced camera system not positive definite at alpha={alpha}")
        return math.sqrt(s[_lib.SC_PNORM2]), s[_lib.SC_PQ]

    def step(self, scale):
        L = self.lay
        xp, xn = C.c_void_p(self.x.data_ptr()), C.c_void_p(self.x_new.data_ptr())
        self.h.call("sfm_ba_step", self._pp, xp, C.c_double(scale), xn)
        if self.comm.world_size > 1 or getattr(self.comm, 'force', False):
            self.comm.allreduce_sum(self.view(L.reduce_step_off, L.reduce_step_count))
        self.h.call("sfm_ba_finish_step", self._pp, xp, C.c_double(scale), xn)
        s = self.scalars()
        return (s[_lib.SC_JS2], s[_lib.SC_GTS], s[_lib.SC_COST_NEW], math.sqrt(s[_lib.SC_SNORM2]),
                math.sqrt(s[_lib.SC_XNEW_NORM2]))

    def accept(self):
        self.x, self.x_new = self.x_new, self.x

    # ---- extras
    def cost(self, x=None):
        L = self.lay
        x = self.x if x is None else x
        self.h.call("sfm_ba_cost", self._pp, C.c_void_p(x.data_ptr()))
        red = self.view(L.reduce_step_off, L.reduce_step_count)
        if self.comm.world_size > 1 or getattr(self.comm, 'force', False):
            self.comm.allreduce_sum(red)
        return float(red[2].item())

    def reproj_errors(self, x=None, shared_k=True):
        """||proj - uv|| per observation; shared_k=False uses each camera's own intrinsics (d=10)."""
        x = self.x if x is None else x
        out = self.torch.empty(self.N, dtype=self.torch.float64, device=self.dev)
        self.h.call("sfm_ba_reproj_errors", self._pp, C.c_void_p(x.data_ptr()), 1 if shared_k else 0,
                    C.c_void_p(out.data_ptr()))
        return out

    def params(self):
        """(cams [C,d], pts [P,3]) of this rank as NumPy arrays."""
        x = self.x.cpu().numpy()
        return x[:self.n].reshape(self.C, self.d).copy(), x[self.n:].reshape(self.P, 3).copy()


def solve_ba(cams, pts, cam_idx, pt_idx, uv, K0, width=1024.0, height=768.0, ftol=1e-4, xtol=1e-4,
             gtol=1e-8, max_nfev=100, device=0, comm=None, **trf_kw):
    """Run the reference's solver settings (sfm_reconstruction.py:506-514) on one shard.
    Returns (TRFResult, cams, pts, backend)."""
    be = GpuBA(cams, pts, cam_idx, pt_idx, uv, K0, width, height, device=device, comm=comm)
    res = trf(be, ftol=ftol, xtol=xtol, gtol=gtol, max_nfev=max_nfev, **trf_kw)
    c, p = be.params()
    return res, c, p, be
