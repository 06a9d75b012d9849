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


def golden_files(pattern="ba_", include_large=True):
    """Reference-generated BA goldens.  ba_bunny_* (35 cams / 2,555 pts: the reconstruction the reference ships)
    is too large for the dense NumPy oracle (8015 x 8015) and is checked with the C oracle and on the GPU."""
    out = sorted(f for f in os.listdir(GOLDEN) if f.startswith(pattern) and f.endswith(".npz"))
    return [f for f in out if include_large or "bunny" not in f]


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
