import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


LARGE_GOLDENS = ("bunny", "cfg1")


def golden_files(pattern="ba_", include_large=True):
    """Reference-generated BA goldens.  ba_bunny_* (35 cams / 2,555 pts: the reconstruction the reference ships)
    and ba_cfg1_* (BASELINE.json configs[0] at its stated size: 10 cams / 1,000 pts, n = 3,100) are too slow for
    the dense NumPy oracle and are checked with the C oracle and on the GPU."""
    out = sorted(f for f in os.listdir(GOLDEN) if f.startswith(pattern) and f.endswith(".npz"))
    return [f for f in out if include_large or not any(t in f for t in LARGE_GOLDENS)]


def golden_x_tolerance(name):
    """Relative tolerance on the parameters against the reference's run: north_star's 1e-4, except for the runs of
    the LITERAL (mis-paired, sfm_reconstruction.py:480-486) objective at cfg1 size, where SciPy's own result is not
    determined to that level: every residual is an outlier (||f|| 2.4e4 -> 1.6e4), so every reprojection row sits in
    the Huber-linear regime with its curvature scaled by sqrt(EPS) and the steps are set by round-off-sized
    curvature; the 1e-8 difference between SciPy's finite-difference Jacobian and an analytic one grows to 2e-3 in
    the parameters after 8 Jacobian evaluations (ba_cfg1_c10p1000_reference_nfev12.npz: the same run cut at
    max_nfev = 12) and to 1.6e-2 after 22 (the full run), while the evaluation counts, the status and the cost
    (5e-9 / 2.5e-7 relative) agree.  SURVEY.md section 0 fact 8 describes the mechanism.  The correctly paired run
    of the same scene (ba_cfg1_c10p1000_aligned.npz) agrees to 5e-7 and is held to 1e-4 like every other golden."""
    return {"ba_cfg1_c10p1000_reference.npz": 5e-2, "ba_cfg1_c10p1000_reference_nfev12.npz": 5e-3}.get(name, 1e-4)


def load_golden_problem(name):
    """(golden dict, oracle BAProblem, x0) for a tests/golden/ba_*.npz file."""
    from oracle import ba_oracle as bo
    g = dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))
    K = g["K"]
    prob = bo.BAProblem(g["R0"].shape[0], g["pts0"].shape[0], 10, g["cam_idx"], g["pt_idx"],
                        bo.effective_uv(g["uv"], g["cam_idx"], str(g["order"])),
                        np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]))
    return g, prob, g["x0"].copy()


@pytest.fixture(scope="session")
def gpu_ready():
    """Build/load the HIP library and make sure a GPU is visible; GPU tests fail loudly otherwise."""
    import torch
    from sfm_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    _lib.load()
    return True
