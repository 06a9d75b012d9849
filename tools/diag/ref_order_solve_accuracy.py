"""Which side is off?  One damped solve of the reference-order problem at x0: product (each route), C oracle, NumPy oracle
(independent code: explicit 3x3 inverses, LAPACK Cholesky).  Prints relative differences of p (cameras / points), ||p||, p^T q."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ba_c, ba_oracle as bo
from sfm_amd import synth
from sfm_amd.ba import GpuBA

C_, P_, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
order = sys.argv[4] if len(sys.argv) > 4 else "reference"
sc = synth.make_scene(C_, P_, obs_per_point=10, seed=seed, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002)
uv = bo.effective_uv(sc.uv, sc.cam_idx, order)
x0 = np.concatenate([sc.cams0.ravel(), sc.pts0.ravel()])
n = C_ * 10
prob = bo.BAProblem(C_, P_, 10, sc.cam_idx, sc.pt_idx, uv, np.array(synth.K_REF))
t = time.time(); lin = bo.linearize(x0, prob); print("numpy linearize %.1f s" % (time.time() - t), flush=True)
cb = ba_c.CBA(C_, P_, 10, sc.cam_idx, sc.pt_idx, uv, synth.K_REF)
rc, rg, ri, rh = cb.linearize(x0)
print("cost C vs numpy rel %.1e; gnorm rel %.1e" % (abs(rc - lin.cost) / lin.cost, abs(rg - np.linalg.norm(lin.g)) / rg))
be = {r: GpuBA(sc.cams0, sc.pts0, sc.cam_idx, sc.pt_idx, uv, synth.K_REF, **(dict(solver="pcg") if r == "pcg" else dict(camera_solver=r)))
      for r in ("auto", "cholesky", "pcg")}
for r, b in be.items():
    c, g, gi, hd = b.linearize()
    gc = b.view(b.lay.gc_off, n).cpu().numpy(); gp = b.view(b.lay.gp_off, P_ * 3).cpu().numpy()
    gg = np.concatenate([gc, gp])
    print("%s: cost rel %.1e, g vs numpy: max abs diff %.2e (max |g| %.2e), cams rel %.1e pts rel %.1e" % (
        r, abs(c - lin.cost) / lin.cost, np.max(np.abs(gg - lin.g)), np.max(np.abs(lin.g)),
        np.linalg.norm(gc - lin.g[:n]) / np.linalg.norm(lin.g[:n]), np.linalg.norm(gp - lin.g[n:]) / np.linalg.norm(lin.g[n:])))
a0 = rg / np.linalg.norm(x0)
for alpha in (a0, 48.38, 1.686, 1e-3 * a0):
    t = time.time()
    p_np = bo.schur_solve(lin, prob, alpha)
    q_np = bo.schur_solve(lin, prob, alpha, p_np)
    pn_np, pq_np = np.linalg.norm(p_np), float(p_np @ q_np)
    print("alpha %.4e: numpy %.1f s  ||p|| %.6e  pq %.6e" % (alpha, time.time() - t, pn_np, pq_np), flush=True)
    rpn, rpq = cb.solve(alpha, True)
    pc_ = cb.step_vector()
    def show(name, p, pn, pq):
        print("   %-9s p cams rel %.2e  pts rel %.2e  ||p|| rel %.2e  pq rel %.2e" % (
            name, np.linalg.norm(p[:n] - p_np[:n]) / np.linalg.norm(p_np[:n]), np.linalg.norm(p[n:] - p_np[n:]) / np.linalg.norm(p_np[n:]),
            abs(pn - pn_np) / pn_np, abs(pq - pq_np) / pq_np))
    show("C oracle", pc_, rpn, rpq)
    for r, b in be.items():
        pn, pq = b.solve(alpha, True)
        p = np.concatenate([b.view(b.lay.pc_off, n).cpu().numpy(), b.view(b.lay.pp_off, P_ * 3).cpu().numpy()])
        show(r, p, pn, pq)
