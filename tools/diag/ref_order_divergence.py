"""How fast do the product (per camera-system route) and the C oracle part ways on the reference-order (all-outlier) objective?
Prints, per outer iteration, counts and the largest relative parameter difference between: oracle (T threads) and oracle (1 thread),
product auto (CG), product cholesky, product pcg - each against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ba_c, ba_oracle as bo
from sfm_amd import synth
from sfm_amd.ba import GpuBA

C_, P_, seed, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
sc = synth.make_scene(C_, P_, obs_per_point=10, seed=seed, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002)
uv = bo.effective_uv(sc.uv, sc.cam_idx, "reference")
x0 = np.concatenate([sc.cams0.ravel(), sc.pts0.ravel()])
cb = ba_c.CBA(C_, P_, 10, sc.cam_idx, sc.pt_idx, uv, synth.K_REF)
ref = []
for k in range(1, K + 1):
    xr, rr = cb.trf(x0, max_nfev=10 ** 9, max_outer=k, check_tolerances=False)
    ref.append((xr, rr))
ba_c.lib().bao_set_threads(1)
ref1 = []
for k in range(1, K + 1):
    xr, rr = cb.trf(x0, max_nfev=10 ** 9, max_outer=k, check_tolerances=False)
    ref1.append((xr, rr))


def rel(a, b):
    e = np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
    i = int(np.argmax(e))
    return e[i], i


n = C_ * 10
for k in range(K):
    e, i = rel(ref1[k][0], ref[k][0])
    print("oracle 1 thread vs T threads, iteration %d: %.2e at %d  counts %s / %s" % (
        k + 1, e, i, (ref1[k][1]["nfev"], ref1[k][1]["n_solves"]), (ref[k][1]["nfev"], ref[k][1]["n_solves"])))
for route in ("auto", "cholesky", "pcg"):
    kw = dict(solver="pcg") if route == "pcg" else dict(camera_solver=route)
    be = GpuBA(sc.cams0, sc.pts0, sc.cam_idx, sc.pt_idx, uv, synth.K_REF, **kw)
    st = be.trf_begin(max_nfev=2 ** 31 - 1, check_tolerances=False)
    for k in range(K):
        st.outer()
        r = st.result()
        cams, pts = be.params()
        x = np.concatenate([cams.ravel(), pts.ravel()])
        e, i = rel(x, ref[k][0])
        ec, _ = rel(x[:n], ref[k][0][:n])
        print("%-8s iteration %2d: nfev %d njev %d n_solves %d (oracle %d %d %d) cost rel %.1e  x max rel %.2e at %d (%s; value %.3e), "
              "cameras only %.2e, alpha %.3e CG %s" % (route, k + 1, r.nfev, r.njev, r.n_solves, ref[k][1]["nfev"], ref[k][1]["njev"],
              ref[k][1]["n_solves"], abs(r.cost - ref[k][1]["cost"]) / ref[k][1]["cost"], e, i, "camera" if i < n else "point",
              ref[k][0][i], ec, r.trace[-1][0], be.solver_stats()))
    st.close()
    del st, be
