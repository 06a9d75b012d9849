import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sfm_amd import synth
from sfm_amd.ba import GpuBA
sc = synth.make_scene(6, 60, obs_per_point=4, seed=9, cam_sigma=0.005)
keep = []
def f():
    be = GpuBA(sc.cams0, sc.pts0, sc.cam_idx, sc.pt_idx, sc.uv, synth.K_REF)
    st = be.trf_begin(max_nfev=2 ** 31 - 1, check_tolerances=False)
    st.outer()
    raise RuntimeError("kept")
try:
    f()
except RuntimeError as e:
    keep.append(e)          # the traceback keeps the frame (be, st) alive until interpreter exit
print("exiting with a live backend + loop state")
