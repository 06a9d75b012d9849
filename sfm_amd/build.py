"""Build libsfm_amd.so (gfx950) in-tree with hipcc.  `python -m sfm_amd.build [--force]`."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", f) for f in ("ctx.hip", "ba.hip", "dense.hip", "match.hip", "driver.hip")]
HDR = [os.path.join(HERE, "csrc", "common.h"), os.path.join(HERE, "csrc", "dense.h"), os.path.join(ROOT, "include", "sfm_amd.h")]
LIB = os.path.join(HERE, "lib", "libsfm_amd.so")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SRC + HDR)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (gfx950 has one unified register file); with the default
    # AGPR form every accumulator value costs a v_accvgpr_read before the VALU can use it - in the matcher that
    # was a quarter of all VALU work (profiles/r01_pmc_matcher.txt)
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
           "-mllvm", "-amdgpu-mfma-vgpr-form",
           "-I" + os.path.join(ROOT, "include"), "-o", LIB] + SRC
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
