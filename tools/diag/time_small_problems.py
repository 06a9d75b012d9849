import sys, time, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from conftest import load_golden_problem, golden_files
from sfm_amd.ba import GpuBA
for name in [f for f in golden_files() if 'bunny' in f or 'cfg1' in f]:
    g, prob, x0 = load_golden_problem(name)
    C_, d = prob.n_cams, prob.d
    def run():
        be = GpuBA(x0[:C_*d].reshape(C_, d), x0[C_*d:].reshape(-1, 3), prob.cam_idx, prob.pt_idx, prob.uv, prob.K0, prob.width, prob.height)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = be.run_trf(max_nfev=100)
        torch.cuda.synchronize(); return time.perf_counter() - t0, res
    run()
    dt, res = run()
    print(name, "C", C_, "P", prob.n_pts, "N", prob.n_obs, "nfev", res.nfev, "njev", res.njev, "n_solves", res.n_solves, "solve wall %.1f ms" % (dt * 1e3), "= %.0f us per damped solve" % (dt * 1e6 / max(res.n_solves, 1)))
