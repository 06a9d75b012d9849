import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
from sfm_amd import synth
from sfm_amd.ba import GpuBA
for vis in ("random","nearest"):
    sc = synth.make_scene(200, 100000, obs_per_point=10, seed=1004, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002, visibility=vis)
    be = GpuBA(sc.cams0, sc.pts0, sc.cam_idx, sc.pt_idx, sc.uv, synth.K_REF)
    st = be.trf_begin(max_nfev=2**31-1, check_tolerances=False)
    prev=(0,0); ps=0
    for it in range(12):
        st.outer()
        r = st.result(); its, fb = be.solver_stats()
        print(vis, it, "solves", r.n_solves-ps, "cg its", its-prev[0], "fallbacks", fb, "cost %.6g"%r.cost, flush=True)
        prev=(its,fb); ps=r.n_solves
    st.close()
