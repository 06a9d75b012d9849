"""Build libsfm_amd.so (gfx950) in-tree with hipcc.  `python -m sfm_amd.build [--force]`."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", f) for f in ("ctx.hip", "ba.hip", "problem.hip", "trf.hip", "dense.hip", "match.hip", "driver.hip", "comm_rccl.hip")]
import glob
# every header under csrc/ (match_plan.h and trf_loop.h were missing from a hand-kept list once: a stale library then
# ran under the GPU tests while the CPU sanitizer tests compiled the new header)
HDR = sorted(glob.glob(os.path.join(HERE, "csrc", "*.h"))) + sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
LIB = os.path.join(HERE, "lib", "libsfm_amd.so")


OBJ_DIR = os.path.join(HERE, "lib", "obj")


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(p) > t for p in deps)


def needs_build():
    return _stale(LIB, SRC + HDR)


def build(force=False, verbose=True):
    """One object per source (compiled in parallel, only when stale), then one link."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (gfx950 has one unified register file); with the default
    # AGPR form every accumulator value costs a v_accvgpr_read before the VALU can use it - in the matcher that
    # was a quarter of all VALU work (profiles/r01_pmc_matcher.txt)
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form",
             "-I" + os.path.join(ROOT, "include")]

    def compile_one(src):
        obj = _obj(src)
        if force or _stale(obj, [src] + HDR):
            cmd = [hipcc] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SRC), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SRC))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
