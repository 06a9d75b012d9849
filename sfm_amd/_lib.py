"""ctypes binding of libsfm_amd.so (include/sfm_amd.h).  There is no CPU fallback: if the
library or a GPU is missing, calls fail loudly."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SFM_AMD_LIB: a diagnostic build of the same library (tools/exp_*.sh); there is no other library to fall back to
LIB_PATH = os.environ.get("SFM_AMD_LIB") or os.path.join(_HERE, "lib", "libsfm_amd.so")

METRIC_L2_U8, METRIC_L2_F32, METRIC_HAMMING = 0, 1, 2
(SC_COST, SC_GNORM2, SC_GINF, SC_PNORM2, SC_PQ, SC_JS2, SC_GTS, SC_COST_NEW, SC_SNORM2,
 SC_XNEW_NORM2, SC_CHOL_FAIL, SC_HDIAG) = range(12)
SC_COUNT = 16
PROF_SLOTS = ("lin_obs", "lin_rest", "build_G", "schur", "chol", "trsv", "backsub", "step", "knn", "schur_items")

i32, i64, f64, vp = C.c_int32, C.c_int64, C.c_double, C.c_void_p


PREC_FP64, PREC_MIXED = 0, 1
SOLVER_DENSE, SOLVER_PCG = 0, 1
CAMERA_AUTO, CAMERA_CHOLESKY, CAMERA_CG = 0, 1, 2
UV_AS_GIVEN, UV_REFERENCE_PAIRING = 0, 1


class BADesc(C.Structure):
    """sfm_ba_desc: the arrays bundle_adjust packs (sfm_reconstruction.py:409-451) and the options."""
    _fields_ = [("n_cams", i32), ("n_pts", i32), ("cam_dim", i32), ("apply_reg", i32),
                ("n_obs", i64),
                ("cam_idx", vp), ("pt_idx", vp), ("uv", vp),
                ("fx0", f64), ("fy0", f64), ("cx0", f64), ("cy0", f64),
                ("width", f64), ("height", f64), ("reg_weight", f64),
                ("precision", i32), ("camera_solver", i32), ("uv_pairing", i32), ("reserved", i32)]


class BAStructureView(C.Structure):
    """sfm_ba_structure: device pointers of the index structure a problem owns."""
    _fields_ = ([(n, i64) for n in ("n_obs", "n_pairs", "n_items", "n_cchunks", "xcd_max_items")] +
                [(n, vp) for n in ("pt_ptr", "cam_ptr", "cam_obs", "blk_ptr", "pair_k", "pair_k2", "item_ptr",
                                   "item_beg", "item_end", "xcd_ptr", "xcd_items", "cch_ptr", "cch_beg", "cch_end")])


class TRFOptions(C.Structure):
    _fields_ = [("ftol", f64), ("xtol", f64), ("gtol", f64), ("max_nfev", i32), ("max_outer", i32),
                ("check_tolerances", i32), ("solver", i32), ("pcg_rtol", f64), ("pcg_max_iter", i32), ("reserved", i32)]


class TRFResultC(C.Structure):
    _fields_ = [("cost", f64), ("optimality", f64), ("nfev", i32), ("njev", i32), ("status", i32),
                ("n_solves", i32), ("n_outer", i32), ("cg_iters", i32)]


REDUCE_FN = C.CFUNCTYPE(C.c_int, vp, vp, i64, C.c_int)


class BALayout(C.Structure):
    _fields_ = [(n, i64) for n in (
        "total_bytes", "rec_off", "rec_stride", "recB_off", "B_off", "gc_off", "Cp_off", "gp_off",
        "reduce_lin_off", "reduce_lin_count", "gmax_off", "reduce_S_off", "reduce_S_count", "reduce_Sp_off", "reduce_Sp_count",
        "reduce_q_off", "reduce_q_count", "reduce_step_off", "reduce_step_count",
        "pc_off", "pp_off", "scalars_off", "G_off", "cg_Ap_off", "cg_M_off")]


# name -> (restype, argtypes); every symbol include/sfm_amd.h declares
SIGNATURES = {
    "sfm_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "sfm_destroy": (None, [vp]),
    "sfm_last_error": (C.c_char_p, [vp]),
    "sfm_set_stream": (C.c_int, [vp, vp]),
    "sfm_synchronize": (C.c_int, [vp]),
    "sfm_version": (C.c_char_p, []),
    "sfm_cgs_persist_enable": (C.c_int, [vp, C.c_int]),
    "sfm_cgs_persist_enabled": (C.c_int, [vp]),
    "sfm_set_profiling": (C.c_int, [vp, C.c_int]),
    "sfm_profile_read": (C.c_int, [vp, C.c_int, C.POINTER(f64), C.POINTER(i64)]),
    "sfm_match_workspace_bytes": (C.c_int, [C.c_int, i64, i64, C.c_int, C.POINTER(i64)]),
    "sfm_match_knn2": (C.c_int, [vp, C.c_int, vp, i64, vp, i64, C.c_int, vp, vp, vp, vp, vp, i64]),
    "sfm_match_ratio": (C.c_int, [vp, i64, vp, vp, vp, f64, vp, vp, vp, vp, vp, i64]),
    "sfm_match_f32_to_u8": (C.c_int, [vp, vp, i64, vp, vp]),
    "sfm_match_batched_workspace_bytes": (C.c_int, [C.c_int, i32, vp, vp, vp, vp, i64, i64, C.POINTER(i64), C.POINTER(i64)]),
    "sfm_match_knn2_batched": (C.c_int, [vp, C.c_int, vp, i64, vp, i64, C.c_int, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64]),
    "sfm_match_ratio_batched": (C.c_int, [vp, i64, i32, vp, vp, vp, vp, f64, vp, vp, vp, vp, vp, i64]),
    "sfm_copy_to_host": (C.c_int, [vp, vp, vp, i64]),
    "sfm_comm_unique_id": (C.c_int, [vp, vp]),
    "sfm_comm_init_rank": (C.c_int, [vp, vp, i32, i32]),
    "sfm_comm_adopt": (C.c_int, [vp, vp, i32, i32]),
    "sfm_comm_destroy": (C.c_int, [vp]),
    "sfm_comm_info": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32)]),
    "sfm_comm_allreduce": (C.c_int, [vp, vp, i64, C.c_int]),
    "sfm_comm_reduce_hook": (C.c_int, [vp, vp, i64, C.c_int]),
    "sfm_ba_create_problem": (C.c_int, [vp, C.POINTER(BADesc), C.POINTER(vp)]),
    "sfm_ba_destroy_problem": (None, [vp]),
    "sfm_ba_get_structure": (C.c_int, [vp, C.POINTER(BAStructureView)]),
    "sfm_ba_get_layout": (C.c_int, [vp, C.POINTER(BALayout)]),
    "sfm_ba_set_sharded": (C.c_int, [vp, vp, C.c_int]),
    "sfm_ba_solver_stats": (C.c_int, [vp, C.POINTER(i64), C.POINTER(i64)]),
    "sfm_ba_pcg_stats": (C.c_int, [vp, C.POINTER(i64), C.POINTER(f64)]),
    "sfm_ba_bind_workspace": (C.c_int, [vp, vp, vp, i64]),
    "sfm_reproj_errors": (C.c_int, [vp, i32, i32, i64, vp, vp, vp, vp, f64, f64, f64, f64, C.c_int, vp]),
    "sfm_ba_solve_pcg": (C.c_int, [vp, vp, f64, C.c_int, f64, i32, REDUCE_FN, vp, C.POINTER(i32)]),
    "sfm_ba_trf_begin": (C.c_int, [vp, vp, vp, C.POINTER(TRFOptions), REDUCE_FN, vp, C.POINTER(vp)]),
    "sfm_ba_trf_outer": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "sfm_ba_trf_result": (C.c_int, [vp, C.POINTER(TRFResultC)]),
    "sfm_ba_trf_trace": (C.c_int, [vp, vp, i32]),
    "sfm_ba_trf_end": (None, [vp]),
    "sfm_ba_run_trf": (C.c_int, [vp, vp, vp, C.POINTER(TRFOptions), REDUCE_FN, vp, C.POINTER(TRFResultC)]),
    "sfm_ba_cost": (C.c_int, [vp, vp, vp]),
    "sfm_ba_reproj_errors": (C.c_int, [vp, vp, vp, C.c_int, vp]),
    "sfm_ba_residual_norm2": (C.c_int, [vp, vp, vp, C.c_int, C.POINTER(f64)]),
    "sfm_ba_linearize": (C.c_int, [vp, vp, vp]),
    "sfm_ba_finish_linearize": (C.c_int, [vp, vp]),
    "sfm_ba_schur_build": (C.c_int, [vp, vp, f64]),
    "sfm_ba_pack_system": (C.c_int, [vp, vp]),
    "sfm_ba_unpack_system": (C.c_int, [vp, vp]),
    "sfm_ba_schur_solve": (C.c_int, [vp, vp, f64, C.c_int]),
    "sfm_ba_finish_solve": (C.c_int, [vp, vp, C.c_int]),
    "sfm_ba_step": (C.c_int, [vp, vp, vp, f64, vp]),
    "sfm_ba_finish_step": (C.c_int, [vp, vp, vp, f64, vp]),
    "sfm_ba_read_scalars": (C.c_int, [vp, vp, C.POINTER(f64)]),
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


class SfmNumericError(SfmError):
    """SFM_ERR_NUMERIC: a damped system was not positive definite, a solve stalled or a step was not finite."""


def load():
    """dlopen the in-tree library and declare every prototype.  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SfmError(f"{LIB_PATH} not built: run `python -m sfm_amd.build` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        # PyTorch bundles its own HIP runtime (libamdhip64 under torch/lib, found through its RPATH).  Whichever copy is
        # mapped first serves the whole process: if this library came first it would bind the system copy, torch would
        # then bring its own, and the second runtime to initialise sees no device ("no ROCm-capable device is detected"
        # from sfm_create).  So torch - the owner of device memory and streams here - goes first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
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
            raise (SfmNumericError if rc == -4 else SfmError)(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

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


def _release_handles():
    """At interpreter exit: the per-device handles go LAST, after a collection has finalised every problem / loop state that is
    still alive (a failed test keeps its frame, and with it a backend, until the very end)."""
    import gc
    gc.collect()
    _handles.clear()


import atexit
atexit.register(_release_handles)


def get_handle(device=0):
    h = _handles.get(device)
    if h is None:
        h = _handles[device] = Handle(device)
    else:
        h.bind_stream()
    return h
