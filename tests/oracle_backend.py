"""CPU stand-in for sfm_amd.ba.GpuBA used ONLY by the tests: it implements the same backend protocol
(linearize / solve / step / accept) and the same staged, shardable algebra (partial [S | r] -> reduce ->
replicated camera solve -> local point back-substitution -> reduce of short vectors) with the CPU
oracle's blocks, so sfm_amd/trf.py, sfm_amd/structure.py and sfm_amd/comm.py can be exercised without a
GPU, including world_size > 1 over gloo."""
import math

import numpy as np
import torch

from oracle import ba_oracle as bo
from sfm_amd.comm import LocalComm
from sfm_amd.structure import build_structure


class OracleBackend:
    def __init__(self, prob, x0, comm=None, solver="dense"):
        self.solver = solver          # "pcg": the algebra of sfm_ba_solve_pcg (implicit Schur complement, block-Jacobi CG)
        self.cg_iters = 0
        self.prob = prob
        self.comm = comm or LocalComm()
        self.x = np.asarray(x0, dtype=np.float64).copy()
        self.x_new = None
        self.n = prob.n_cams * prob.d
        self.st = build_structure(prob.cam_idx, prob.pt_idx, prob.n_cams, prob.n_pts)
        self.n_solves = 0

    def _sum(self, arr):
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64))
        self.comm.allreduce_sum(t)
        return t.numpy()

    def _max(self, v):
        t = torch.tensor([float(v)], dtype=torch.float64)
        self.comm.allreduce_max(t)
        return float(t[0])

    def x_norm(self):
        pts2 = self._sum(np.array([np.sum(self.x[self.n:] ** 2)]))[0]
        return math.sqrt(pts2 + float(np.sum(self.x[:self.n] ** 2)))

    def linearize(self):
        lin = self.lin = bo.linearize(self.x, self.prob)
        n = self.n
        bdiag = np.einsum("cii->ci", lin.B).ravel()
        red = self._sum(np.concatenate([lin.g[:n], [lin.cost, np.sum(lin.g[n:] ** 2)], bdiag]))
        gmax = self._max(np.max(np.abs(lin.g[n:])) if lin.g.size > n else 0.0)
        cmax = self._max(np.max(np.einsum("pii->pi", lin.Cp)) if lin.Cp.size else 0.0)
        self.g_c = red[:n]
        return (red[n], math.sqrt(np.sum(red[:n] ** 2) + red[n + 1]), max(np.max(np.abs(red[:n])), gmax),
                max(float(np.max(red[n + 2:])), cmax))

    def solve(self, alpha, want_q):
        pr, lin, st, n, d = self.prob, self.lin, self.st, self.n, self.prob.d
        Ca = lin.Cp + alpha * np.eye(3)[None]
        Lc = np.linalg.cholesky(Ca)
        M = np.linalg.inv(Lc)                                        # L_j^-1
        W = np.einsum("nri,nrj->nij", lin.Jc, lin.Jp)
        G = np.einsum("nij,nkj->nik", W, M[pr.pt_idx])               # W L^-T   [N,d,3]
        g_p = lin.g[n:].reshape(-1, 3)
        e = np.einsum("pij,pj->pi", M, g_p)
        if self.solver == "pcg":
            return self._solve_pcg(alpha, want_q, G, M, e)
        S = np.zeros((n, n))
        for c in range(pr.n_cams):
            S[c * d:(c + 1) * d, c * d:(c + 1) * d] = lin.B[c]
        blk_of = np.repeat(np.arange(len(st.blk_ptr) - 1), np.diff(st.blk_ptr))
        # block id -> (c, c2)
        cs = np.concatenate([np.full(pr.n_cams - c, c) for c in range(pr.n_cams)])
        c2s = np.concatenate([np.arange(c, pr.n_cams) for c in range(pr.n_cams)])
        for p in range(st.n_pairs):
            k, k2, b = st.pair_k[p], st.pair_k2[p], blk_of[p]
            c, c2 = cs[b], c2s[b]
            blk = G[k] @ G[k2].T
            S[c * d:(c + 1) * d, c2 * d:(c2 + 1) * d] -= blk
            if c != c2:
                S[c2 * d:(c2 + 1) * d, c * d:(c + 1) * d] -= blk.T
        r = lin.g[:n].reshape(-1, d).copy()
        np.add.at(r, pr.cam_idx, -np.einsum("nij,nj->ni", G, e[pr.pt_idx]))
        red = self._sum(np.concatenate([S.ravel(), r.ravel()]))
        S = red[:n * n].reshape(n, n) + alpha * np.eye(n)
        r = red[n * n:]
        Ls = np.linalg.cholesky(S)
        y = np.linalg.solve(Ls, r)
        pc = -np.linalg.solve(Ls.T, y)
        u = e.copy()
        np.add.at(u, pr.pt_idx, np.einsum("nij,ni->nj", G, pc.reshape(-1, d)[pr.cam_idx]))
        pp = -np.einsum("pji,pj->pi", M, u)
        v = np.einsum("pij,pj->pi", M, pp)
        rhs2 = np.zeros((pr.n_cams, d))
        np.add.at(rhs2, pr.cam_idx, -np.einsum("nij,nj->ni", G, v[pr.pt_idx]))
        red = self._sum(np.concatenate([rhs2.ravel(), [np.sum(pp ** 2), np.sum(v ** 2)]]))
        self.pc, self.pp = pc, pp.ravel()
        self.n_solves += 1
        pnorm2 = float(pc @ pc) + red[n]
        pq = 0.0
        if want_q:
            yy = np.linalg.solve(Ls, pc + red[:n])
            pq = float(yy @ yy) + red[n + 1]
        return math.sqrt(pnorm2), pq

    def _solve_pcg(self, alpha, want_q, G, M, e):
        """Same staged algebra as sfm_ba_solve_pcg: per call one reduction of the right-hand side and of the diagonal
        blocks, per CG iteration one reduction of an n-vector; every rank runs the identical recurrence."""
        pr, lin, n, d = self.prob, self.lin, self.n, self.prob.d
        C = pr.n_cams

        def local_matvec(v):          # this rank's part of (S - alpha I) v
            vc = v.reshape(C, d)
            t = np.einsum("nij,ni->nj", G, vc[pr.cam_idx])                    # G_k^T v_cam(k)
            u = np.zeros((pr.n_pts, 3)); np.add.at(u, pr.pt_idx, t)           # per track
            out = np.einsum("cij,cj->ci", lin.B, vc)
            np.add.at(out, pr.cam_idx, -np.einsum("nij,nj->ni", G, u[pr.pt_idx]))
            return out.ravel()

        r = lin.g[:n].reshape(C, d).copy()
        np.add.at(r, pr.cam_idx, -np.einsum("nij,nj->ni", G, e[pr.pt_idx]))
        blocks = lin.B.copy()
        np.add.at(blocks, pr.cam_idx, -np.einsum("nim,njm->nij", G, G))
        red = self._sum(np.concatenate([r.ravel(), blocks.ravel()]))
        r, blocks = red[:n], red[n:].reshape(C, d, d) + alpha * np.eye(d)[None]
        Minv = np.linalg.inv(blocks)

        def cg(rhs):
            x = np.zeros(n); res = rhs.copy()
            z = np.einsum("cij,cj->ci", Minv, res.reshape(C, d)).ravel(); pv = z.copy()
            rz, rr0 = float(res @ z), float(res @ res)
            for _ in range(20 * n):
                Ap = self._sum(local_matvec(pv)) + alpha * pv
                a = rz / float(pv @ Ap)
                x += a * pv; res -= a * Ap
                self.cg_iters += 1
                if float(res @ res) <= 1e-26 * rr0:
                    break
                z = np.einsum("cij,cj->ci", Minv, res.reshape(C, d)).ravel()
                rz_new = float(res @ z)
                pv = z + (rz_new / rz) * pv
                rz = rz_new
            return x

        pc = -cg(r)
        u = e.copy()
        np.add.at(u, pr.pt_idx, np.einsum("nij,ni->nj", G, pc.reshape(-1, d)[pr.cam_idx]))
        pp = -np.einsum("pji,pj->pi", M, u)
        v = np.einsum("pij,pj->pi", M, pp)
        rhs2 = np.zeros((C, d))
        np.add.at(rhs2, pr.cam_idx, -np.einsum("nij,nj->ni", G, v[pr.pt_idx]))
        red = self._sum(np.concatenate([rhs2.ravel(), [np.sum(pp ** 2), np.sum(v ** 2)]]))
        self.pc, self.pp = pc, pp.ravel()
        self.n_solves += 1
        pnorm2 = float(pc @ pc) + red[n]
        pq = 0.0
        if want_q:
            b2 = pc + red[:n]
            pq = float(b2 @ cg(b2)) + red[n + 1]
        return math.sqrt(pnorm2), pq

    def step(self, scale):
        s = scale * np.concatenate([self.pc, self.pp])
        n = self.n
        self.x_new = self.x + s
        jp = bo.apply_J(self.lin, self.prob, s)
        red = self._sum(np.array([jp @ jp, float(self.lin.g[n:] @ s[n:]) + self._gc_local_dot(s[:n]),
                                  bo.huber_cost(bo.residuals(self.x_new, self.prob)),
                                  np.sum(s[n:] ** 2), np.sum(self.x_new[n:] ** 2)]))
        return (red[0], red[1], red[2], math.sqrt(red[3] + float(s[:n] @ s[:n])),
                math.sqrt(red[4] + float(self.x_new[:n] @ self.x_new[:n])))

    def _gc_local_dot(self, sc):
        return float(self.lin.g[:self.n] @ sc)       # this rank's partial g_c (sums to g_c . s_c over ranks)

    def accept(self):
        self.x = self.x_new
