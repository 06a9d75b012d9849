"""ctypes binding of libsfm_amd.so (include/sfm_amd.h).  There is no CPU fallback: if the
library or a GPU is missing, calls fail loudly."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsfm_amd.so")

METRIC_L2_U8, METRIC_L2_F32, METRIC_HAMMING = 0, 1, 2
(SC_COST, SC_GNORM2, SC_GINF, SC_PNORM2, SC_PQ, SC_JS2, SC_GTS, SC_COST_NEW, SC_SNORM2,
 SC_XNEW_NORM2, SC_CHOL_FAIL, SC_HDIAG) = range(12)
SC_COUNT = 16
PROF_SLOTS = ("lin_obs", "lin_rest", "build_G", "schur", "chol", "trsv", "backsub", "step", "knn")

i32, i64, f64, vp = C.c_int32, C.c_int64, C.c_double, C.c_void_p


class BAProblem(C.Structure):
    _fields_ = [("n_cams", i32), ("n_pts", i32), ("cam_dim", i32), ("apply_reg", i32),
                ("n_obs", i64),
                ("cam_idx", vp), ("pt_idx", vp), ("uv", vp), ("pt_ptr", vp), ("cam_ptr", vp),
                ("cam_obs", vp), ("blk_ptr", vp), ("pair_k", vp), ("pair_k2", vp),
                ("n_pairs", i64),
                ("item_ptr", vp), ("item_beg", vp), ("item_end", vp), ("n_items", i64),
                ("xcd_ptr", vp), ("xcd_items", vp), ("xcd_max_items", i64),
                ("cch_ptr", vp), ("cch_beg", vp), ("cch_end", vp), ("n_cchunks", i64),
                ("fx0", f64), ("fy0", f64), ("cx0", f64), ("cy0", f64),
                ("width", f64), ("height", f64), ("reg_weight", f64),
                ("workspace", vp), ("workspace_bytes", i64)]


class BALayout(C.Structure):
    _fields_ = [(n, i64) for n in (
        "total_bytes", "rec_off", "rec_stride", "recB_off", "B_off", "gc_off", "Cp_off", "gp_off",
        "reduce_lin_off", "reduce_lin_count", "gmax_off", "reduce_S_off", "reduce_S_count", "reduce_Sp_off", "reduce_Sp_count",
        "reduce_q_off", "reduce_q_count", "reduce_step_off", "reduce_step_count",
        "pc_off", "pp_off", "scalars_off", "G_off")]


# name -> (restype, argtypes); every symbol include/sfm_amd.h declares
SIGNATURES = {
    "sfm_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "sfm_destroy": (None, [vp]),
    "sfm_last_error": (C.c_char_p, [vp]),
    "sfm_set_stream": (C.c_int, [vp, vp]),
    "sfm_synchronize": (C.c_int, [vp]),
    "sfm_version": (C.c_char_p, []),
    "sfm_set_profiling": (C.c_int, [vp, C.c_int]),
    "sfm_profile_read": (C.c_int, [vp, C.c_int, C.POINTER(f64), C.POINTER(i64)]),
    "sfm_match_workspace_bytes": (C.c_int, [C.c_int, i64, i64, C.c_int, C.POINTER(i64)]),
    "sfm_match_knn2": (C.c_int, [vp, C.c_int, vp, i64, vp, i64, C.c_int, vp, vp, vp, vp, vp, i64]),
    "sfm_match_ratio": (C.c_int, [vp, i64, vp, vp, vp, f64, vp, vp, vp, vp, vp, i64]),
    "sfm_match_f32_to_u8": (C.c_int, [vp, vp, i64, vp, vp]),
    "sfm_ba_get_layout": (C.c_int, [i32, i32, i64, i32, i64, i64, C.POINTER(BALayout)]),
    "sfm_ba_cost": (C.c_int, [vp, C.POINTER(BAProblem), vp]),
    "sfm_ba_reproj_errors": (C.c_int, [vp, C.POINTER(BAProblem), vp, C.c_int, vp]),
    "sfm_ba_linearize": (C.c_int, [vp, C.POINTER(BAProblem), vp]),
    "sfm_ba_finish_linearize": (C.c_int, [vp, C.POINTER(BAProblem)]),
    "sfm_ba_schur_build": (C.c_int, [vp, C.POINTER(BAProblem), f64]),
    "sfm_ba_pack_system": (C.c_int, [vp, C.POINTER(BAProblem)]),
    "sfm_ba_unpack_system": (C.c_int, [vp, C.POINTER(BAProblem)]),
    "sfm_ba_schur_solve": (C.c_int, [vp, C.POINTER(BAProblem), f64, C.c_int]),
    "sfm_ba_finish_solve": (C.c_int, [vp, C.POINTER(BAProblem), C.c_int]),
    "sfm_ba_step": (C.c_int, [vp, C.POINTER(BAProblem), vp, f64, vp]),
    "sfm_ba_finish_step": (C.c_int, [vp, C.POINTER(BAProblem), vp, f64, vp]),
    "sfm_ba_read_scalars": (C.c_int, [vp, C.POINTER(BAProblem), C.POINTER(f64)]),
    "sfm_dense_cholesky": (C.c_int, [vp, vp, i32, vp]),
    "sfm_dense_trsv": (C.c_int, [vp, vp, i32, vp, C.c_int]),
    "sfm_assoc_workspace_bytes": (C.c_int, [i64, C.POINTER(i64)]),
    "sfm_assoc_radius": (C.c_int, [vp, vp, vp, vp, vp, i32, i64, f64, vp, vp, i64, vp, vp, i64]),
    "sfm_triangulate2": (C.c_int, [vp, vp, i32, vp, vp, vp, vp, i64, f64, vp, vp, vp]),
    "sfm_epipolar_errors": (C.c_int, [vp, vp, vp, i32, vp, vp, i64, C.c_float, vp, vp]),
}

_lib = None


class SfmError(RuntimeError):
    pass


def load():
    """dlopen the in-tree library and declare every prototype.  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SfmError(f"{LIB_PATH} not built: run `python -m sfm_amd.build` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class Handle:
    """One sfm_handle on `device`, bound to torch's current stream of that device."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise SfmError("sfm_amd needs a ROCm GPU (gfx950); none is visible and there is no CPU fallback")
        self.lib = load()
        self.device = int(device)
        self._h = vp()
        rc = self.lib.sfm_create(self.device, C.byref(self._h))
        if rc != 0:
            raise SfmError(f"sfm_create(device={device}) failed with {rc}")
        self.bind_stream()

    def bind_stream(self):
        import torch
        s = torch.cuda.current_stream(self.device).cuda_stream
        self.check(self.lib.sfm_set_stream(self._h, vp(s)), "sfm_set_stream")

    def check(self, rc, what):
        if rc != 0:
            msg = self.lib.sfm_last_error(self._h)
            raise SfmError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def call(self, name, *args):
        self.check(getattr(self.lib, name)(self._h, *args), name)

    def set_profiling(self, on):
        self.call("sfm_set_profiling", 1 if on else 0)

    def profile(self):
        """{slot: (total_ms, launches)} since the last read (synchronises); resets the counters."""
        out = {}
        for i, name in enumerate(PROF_SLOTS):
            ms, cnt = f64(), i64()
            self.call("sfm_profile_read", i, C.byref(ms), C.byref(cnt))
            out[name] = (ms.value, cnt.value)
        return out

    def __del__(self):
        try:
            if self._h:
                self.lib.sfm_destroy(self._h)
                self._h = vp()
        except Exception:
            pass


_handles = {}


def get_handle(device=0):
    h = _handles.get(device)
    if h is None:
        h = _handles[device] = Handle(device)
    else:
        h.bind_stream()
    return h
