"""ctypes wrapper of the C restatement (oracle/ba_oracle.c).  TEST INFRASTRUCTURE / cpu_baseline only."""
import ctypes as C

import numpy as np

from . import build_c

_libs = {}


def lib(variant=None):
    """The compiled restatement; variant="reverse_sums" is the same source with another summation order (build_c.VARIANTS)."""
    if variant not in _libs:
        l = C.CDLL(build_c.build(variant=variant))
        vp, i32, i64, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_double
        l.bao_create.restype = vp
        l.bao_create.argtypes = [i32, i32, i32, i64, vp, vp, vp, vp, f64, f64, f64, i32]
        l.bao_destroy.argtypes = [vp]
        l.bao_cost.restype = f64
        l.bao_cost.argtypes = [vp, vp]
        l.bao_linearize.argtypes = [vp, vp]
        l.bao_solve.restype = i32
        l.bao_solve.argtypes = [vp, f64, i32, C.POINTER(f64), C.POINTER(f64)]
        l.bao_step.argtypes = [vp, vp, f64, vp, vp]
        l.bao_trf.restype = i32
        l.bao_trf.argtypes = [vp, vp, f64, f64, f64, i32, i32, i32, vp]
        l.bao_get.argtypes = [vp, vp]
        l.bao_get_step.argtypes = [vp, vp, vp]
        l.bao_threads.restype = i32
        l.bao_set_threads.argtypes = [i32]
        # a GPU box shows every hardware thread but grants a share of them (16 per GPU): oversubscribing
        # OpenMP there runs the baseline slower than it deserves
        import os
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        l.bao_set_threads(max(1, min(avail, int(os.environ.get("SFM_ORACLE_THREADS", "16")))))
        l.mo_knn2_u8.argtypes = [vp, i64, vp, i64, i32, vp, vp, vp, vp]
        _libs[variant] = l
    return _libs[variant]


def _p(a):
    return C.c_void_p(a.ctypes.data)


class CBA:
    """C oracle instance for one problem (arrays as in oracle.ba_oracle.BAProblem)."""

    def __init__(self, n_cams, n_pts, d, cam_idx, pt_idx, uv, K0, width=1024.0, height=768.0, reg_weight=0.1,
                 apply_reg=True, variant=None):
        self.l = lib(variant)
        self.ci = np.ascontiguousarray(cam_idx, dtype=np.int32)
        self.pi = np.ascontiguousarray(pt_idx, dtype=np.int32)
        self.uv = np.ascontiguousarray(uv, dtype=np.float64)
        self.K0 = np.ascontiguousarray(K0, dtype=np.float64)
        self.C, self.P, self.d = int(n_cams), int(n_pts), int(d)
        self.h = self.l.bao_create(self.C, self.P, self.d, len(self.ci), _p(self.ci), _p(self.pi), _p(self.uv),
                                   _p(self.K0), width, height, reg_weight, 1 if apply_reg else 0)

    def __del__(self):
        try:
            self.l.bao_destroy(self.h)
        except Exception:
            pass

    def cost(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return self.l.bao_cost(self.h, _p(x))

    def linearize(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        self.l.bao_linearize(self.h, _p(x))
        out = np.zeros(4)
        self.l.bao_get(self.h, _p(out))
        return tuple(out)            # cost, ||g||, ||g||inf, max diag H

    def solve(self, alpha, want_q=True):
        pn, pq = C.c_double(), C.c_double()
        rc = self.l.bao_solve(self.h, alpha, 1 if want_q else 0, C.byref(pn), C.byref(pq))
        if rc:
            raise np.linalg.LinAlgError("reduced camera system not positive definite")
        return pn.value, pq.value

    def step_vector(self):
        pc, pp = np.zeros(self.C * self.d), np.zeros(self.P * 3)
        self.l.bao_get_step(self.h, _p(pc), _p(pp))
        return np.concatenate([pc, pp])

    def step(self, x, scale):
        x = np.ascontiguousarray(x, dtype=np.float64)
        xn = np.empty_like(x)
        out = np.zeros(5)
        self.l.bao_step(self.h, _p(x), scale, _p(xn), _p(out))
        return xn, tuple(out)

    def trf(self, x0, ftol=1e-4, xtol=1e-4, gtol=1e-8, max_nfev=100, max_outer=-1, check_tolerances=True):
        x = np.ascontiguousarray(x0, dtype=np.float64).copy()
        res = np.zeros(6)
        rc = self.l.bao_trf(self.h, _p(x), ftol, xtol, gtol, max_nfev, max_outer, 1 if check_tolerances else 0, _p(res))
        if rc:
            raise np.linalg.LinAlgError("bao_trf failed")
        return x, dict(cost=res[0], nfev=int(res[1]), njev=int(res[2]), status=int(res[3]), n_solves=int(res[4]),
                       optimality=res[5])


def knn2_u8(q, t):
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    nq, nt = q.shape[0], t.shape[0]
    i1, i2 = np.empty(nq, np.int32), np.empty(nq, np.int32)
    d1, d2 = np.empty(nq, np.float32), np.empty(nq, np.float32)
    lib().mo_knn2_u8(_p(q), nq, _p(t), nt, q.shape[1], _p(i1), _p(i2), _p(d1), _p(d2))
    return i1, i2, d1, d2


def threads():
    return lib().bao_threads()
