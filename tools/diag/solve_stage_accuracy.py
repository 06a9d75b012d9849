"""Per-stage accuracy of ONE damped solve against an 80-bit (np.longdouble) evaluation of the same formulas from the product's own
linearisation: M = L^-1, e, G, S, r, p_c, p_p.  Also the C oracle's p.  Shows which stage carries the product's error."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ba_c, ba_oracle as bo
from sfm_amd import synth
from sfm_amd.ba import GpuBA

C_, P_, seed, alpha = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
order = sys.argv[5] if len(sys.argv) > 5 else "reference"
route = sys.argv[6] if len(sys.argv) > 6 else "cholesky"
d = 10
sc = synth.make_scene(C_, P_, obs_per_point=10, seed=seed, noise_px=0.5, pt_sigma=0.02, cam_sigma=0.002)
uv = bo.effective_uv(sc.uv, sc.cam_idx, order)
x0 = np.concatenate([sc.cams0.ravel(), sc.pts0.ravel()])
n, N = C_ * d, sc.n_obs
be = GpuBA(sc.cams0, sc.pts0, sc.cam_idx, sc.pt_idx, uv, synth.K_REF, camera_solver=route)
be.linearize()
L = be.lay
ld = np.longdouble
get = lambda off, cnt: be.view(off, cnt).cpu().numpy()
al = lambda v: (v + 31) // 32 * 32
recA = get(L.rec_off, N * 2 * d).reshape(N, 2, d); recB = get(L.recB_off, N * 8).reshape(N, 8)
B = get(L.B_off, C_ * d * d).reshape(C_, d, d); gc = get(L.gc_off, n)
Cp = get(L.Cp_off, P_ * 6).reshape(P_, 6); gp = get(L.gp_off, P_ * 3).reshape(P_, 3)
linv_off = al(L.gp_off // 8 + P_ * 3) * 8
e_off = al(linv_off // 8 + P_ * 6) * 8
be.h.call("sfm_ba_schur_build", be._pp, C.c_double(alpha))
Linv = get(linv_off, P_ * 6).reshape(P_, 6); e = get(e_off, P_ * 3).reshape(P_, 3)
GS = 32
G = get(L.G_off, N * GS).reshape(N, GS)[:, :3 * d].reshape(N, 3, d)
Sr = get(L.reduce_S_off, n * n + n)
S, r = Sr[:n * n].reshape(n, n).copy(), Sr[n * n:].copy()
pn, pq = None, None
be.h.call("sfm_ba_schur_solve", be._pp, C.c_double(alpha), 1)
be.h.call("sfm_ba_finish_solve", be._pp, 1)
pn, pq = be._solve_scalars(alpha)
pc = get(L.pc_off, n); pp = get(L.pp_off, P_ * 3).reshape(P_, 3)

# ---- 80-bit reference from the product's linearisation
c = Cp.astype(ld)
a00, a10, a20, a11, a21, a22 = c[:, 0] + alpha, c[:, 1], c[:, 2], c[:, 3] + alpha, c[:, 4], c[:, 5] + alpha
l00 = np.sqrt(a00); l10 = a10 / l00; l20 = a20 / l00; l11 = np.sqrt(a11 - l10 * l10); l21 = (a21 - l20 * l10) / l11
l22 = np.sqrt(a22 - l20 * l20 - l21 * l21)
m00, m11, m22 = 1 / l00, 1 / l11, 1 / l22
m10 = -l10 * m00 * m11; m21 = -l21 * m11 * m22; m20 = -(l20 * m00 + l21 * m10) * m22
M_ref = np.stack([m00, m10, m11, m20, m21, m22], 1)
g_ = gp.astype(ld)
e_ref = np.stack([m00 * g_[:, 0], m10 * g_[:, 0] + m11 * g_[:, 1], m20 * g_[:, 0] + m21 * g_[:, 1] + m22 * g_[:, 2]], 1)
pj = sc.pt_idx
Mm = np.zeros((P_, 3, 3), ld); Mm[:, 0, 0] = m00; Mm[:, 1, 0] = m10; Mm[:, 1, 1] = m11; Mm[:, 2, 0] = m20; Mm[:, 2, 1] = m21; Mm[:, 2, 2] = m22
Jp = recB[:, :6].reshape(N, 2, 3).astype(ld); Jc = recA.astype(ld)
V = np.einsum("nrq,nmq->nrm", Jp, Mm[pj])                   # [N,2,3]
G_ref = np.einsum("nra,nrm->nma", Jc, V)                    # [N,3,d]
rel = lambda a, b: float(np.linalg.norm((a.astype(ld) - b).ravel()) / np.linalg.norm(b.ravel()))
print("alpha %.4e route %s" % (alpha, route))
print("M    rel %.2e" % rel(Linv, M_ref)); print("e    rel %.2e" % rel(e, e_ref)); print("G    rel %.2e" % rel(G, G_ref))
# S = B + alpha I - sum_j Z_j^T Z_j, Z_j [3, n]; r = gc - sum G_k e_j
S_ref = np.zeros((n, n), ld)
for cix in range(C_):
    S_ref[cix * d:(cix + 1) * d, cix * d:(cix + 1) * d] = B[cix].astype(ld) + alpha * np.eye(d, dtype=ld)
r_ref = gc.astype(ld).copy()
np.subtract.at(r_ref.reshape(C_, d), sc.cam_idx, np.einsum("nma,nm->na", G_ref, e_ref[pj]))
ptr = np.searchsorted(pj, np.arange(P_ + 1))
assert np.all(np.diff(pj) >= 0)
Lmax = int(np.max(np.diff(ptr)))
for i in range(Lmax):            # pair (k, k2) = (ptr[j] + i, ptr[j] + i2) over all points at once
    for i2 in range(Lmax):
        sel = np.nonzero((ptr[:-1] + max(i, i2)) < ptr[1:])[0]
        k, k2 = ptr[sel] + i, ptr[sel] + i2
        blk = np.einsum("nma,nmb->nab", G_ref[k], G_ref[k2])       # [sel, d, d]
        ck, ck2 = sc.cam_idx[k], sc.cam_idx[k2]
        rows = (ck[:, None] * d + np.arange(d)[None, :])[:, :, None]
        cols = (ck2[:, None] * d + np.arange(d)[None, :])[:, None, :]
        np.subtract.at(S_ref, (np.broadcast_to(rows, blk.shape), np.broadcast_to(cols, blk.shape)), blk)
print("S    rel %.2e   max abs diff / max |S| %.2e" % (rel(S, S_ref), float(np.max(np.abs(S.astype(ld) - S_ref)) / np.max(np.abs(S_ref)))))
print("r    rel %.2e" % rel(r, r_ref))
# p_c = -S^-1 r with refinement in 80 bits
import scipy.linalg as sla
cf = sla.cho_factor(S_ref.astype(np.float64), lower=True)
xk = np.zeros(n, ld)
for _ in range(6):
    res = -r_ref - S_ref @ xk
    xk = xk + sla.cho_solve(cf, res.astype(np.float64)).astype(ld)
pc_ref = xk
print("p_c  rel %.2e   (residual of the reference %.1e)" % (rel(pc, pc_ref), float(np.linalg.norm(-r_ref - S_ref @ pc_ref) / np.linalg.norm(r_ref))))
u = e_ref.copy()
np.add.at(u, pj, np.einsum("nma,na->nm", G_ref, pc_ref.reshape(C_, d)[sc.cam_idx]))
pp_ref = -np.einsum("pqm,pq->pm", Mm, u)                     # M^T u
print("p_p  rel %.2e" % rel(pp, pp_ref))
pn_ref = np.sqrt(np.sum(pc_ref ** 2) + np.sum(pp_ref ** 2))
print("||p|| rel %.2e" % float(abs(pn - pn_ref) / pn_ref))
# the same with the PRODUCT's S and r (isolates the camera solve + back-substitution from the Schur build)
xk = np.zeros(n, ld); Sg, rg_ = S.astype(ld), r.astype(ld)
cf = sla.cho_factor(S, lower=True)
for _ in range(6):
    xk = xk + sla.cho_solve(cf, (-rg_ - Sg @ xk).astype(np.float64)).astype(ld)
print("p_c vs exact solve of the product's own S, r: rel %.2e" % rel(pc, xk))
# C oracle at the same point
cb = ba_c.CBA(C_, P_, d, sc.cam_idx, sc.pt_idx, uv, synth.K_REF)
cb.linearize(x0); cb.solve(alpha, True)
pco = cb.step_vector()
print("C oracle: p_c rel %.2e  p_p rel %.2e" % (rel(pco[:n], pc_ref), rel(pco[n:].reshape(P_, 3), pp_ref)))
