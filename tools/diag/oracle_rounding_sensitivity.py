"""How far does the C oracle determine ITSELF on the reference-order objective?  Same source, two builds that differ only in
summation order (every camera's observation list walked forwards / backwards in B_c, g_c and the rows of S): parameters after k outer iterations, fixed schedule.  CPU only."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
C_, P_, seed, K = (int(v) for v in sys.argv[1:5])
order = sys.argv[5] if len(sys.argv) > 5 else "reference"
if len(sys.argv) > 6:
    variant = None if sys.argv[6] == 'std' else sys.argv[6]
    from oracle import ba_c, ba_oracle as bo
    from sfm_amd import synth
    sc = synth.make_scene(C_, P_, obs_per_point=10, seed=seed, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002)
    uv = bo.effective_uv(sc.uv, sc.cam_idx, order)
    x0 = np.concatenate([sc.cams0.ravel(), sc.pts0.ravel()])
    cb = ba_c.CBA(C_, P_, 10, sc.cam_idx, sc.pt_idx, uv, synth.K_REF, variant=variant)
    out = {}
    for k in range(1, K + 1):
        x, r = cb.trf(x0, max_nfev=10 ** 9, max_outer=k, check_tolerances=False)
        out["x%d" % k] = x; out["c%d" % k] = np.array([r["cost"], r["nfev"], r["njev"], r["n_solves"]])
    np.savez(sys.argv[7], **out)
else:
    for lib, out in (("std", "/tmp/sens_std.npz"), ("reverse_sums", "/tmp/sens_fma.npz")):
        subprocess.run([sys.executable, __file__] + sys.argv[1:5] + [order, lib, out], check=True)
    a, b = np.load("/tmp/sens_std.npz"), np.load("/tmp/sens_fma.npz")
    n = C_ * 10
    for k in range(1, K + 1):
        xa, xb = a["x%d" % k], b["x%d" % k]
        e = np.abs(xa - xb) / np.maximum(np.abs(xa), 1e-3)
        print("iteration %2d: counts %s / %s  cost rel %.1e  x max rel %.2e (cameras %.2e)" % (
            k, a["c%d" % k][1:].astype(int).tolist(), b["c%d" % k][1:].astype(int).tolist(),
            abs(a["c%d" % k][0] - b["c%d" % k][0]) / a["c%d" % k][0], e.max(), e[:n].max()))
