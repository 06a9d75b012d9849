#!/usr/bin/env python3
"""Would a coarse space cut the camera CG's iterations on the spatially coherent scene?  Offline (NumPy) on the hardest damped
systems of a run, saved by tools/diag/dump_hard_systems.py nearest 25 3 (gpurun_out/hard_systems_nearest.npz: packed lower triangle
of S, right-hand side, alpha).  The block-scaled system's spectrum, plain CG to 1e-13, and additive two-level preconditioning
with SPECTRAL coarse spaces: aggregates of g consecutive cameras x the k lowest eigenvectors of the aggregate's own diagonal
block.  Result (round 4): eigenvalues 4e-3 .. 1.4, condition ~300, no gap anywhere - 141 plain iterations at alpha = 1e-4
max diag H against 145 / 125 / 150 / 146 / 156 with coarse spaces of 140 / 280 / 70 / 160 / 60 vectors: nothing to deflate."""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
z=np.load(os.path.join(ROOT, 'gpurun_out', 'hard_systems_nearest.npz'))
n=int(z['n']); d=10; C=n//d
def unpack(tri):
    S=np.zeros((n,n)); S[np.tril_indices(n)]=tri; return S+np.tril(S,-1).T
recs=[z['rec%d'%i] for i in range(3)]
print(recs)
S0=unpack(z['tri0']); rhs=z['rhs0']; a0=recs[0][1]
# S0 lacks alpha on the diagonal? (dump_hard: S from reduce_S before add alpha: exp tools add alpha*I) -> add
def scaled(S, alpha):
    A=S+alpha*np.eye(n)
    Einv=np.zeros_like(A)
    for i in range(0,n,d):
        Einv[i:i+d,i:i+d]=np.linalg.inv(np.linalg.cholesky(A[i:i+d,i:i+d]))
    return Einv@A@Einv.T, Einv
def cg(A,b,M=None,rtol=1e-13,maxit=600):
    x=np.zeros(n); r=b.copy(); z_=M(r) if M else r; p=z_.copy(); rz=r@z_; r0=np.sqrt(r@r)
    for it in range(1,maxit+1):
        Ap=A@p; a=rz/(p@Ap); x+=a*p; r-=a*Ap
        if np.sqrt(r@r)<=rtol*r0: return it
        z_=M(r) if M else r; rzn=r@z_; p=z_+(rzn/rz)*p; rz=rzn
    return maxit
hd=a0/recs[0][2]
for mult in (7e-6, 1e-4, 3e-4, 1e-3):
    alpha=mult*hd
    A,Einv=scaled(S0, alpha); b=Einv@rhs
    w=np.linalg.eigvalsh(A)
    base=cg(A,b)
    out=[base]
    # spectral coarse space: aggregates of g cameras, k lowest eigenvectors of the aggregate's diagonal block of A
    for g,k in ((10,7),(10,14),(20,7),(5,4),(10,3)):
        cols=[]
        for a in range(0,C,g):
            idx=np.arange(a*d,min((a+g)*d,n))
            ww,V=np.linalg.eigh(A[np.ix_(idx,idx)])
            for j in range(k):
                v=np.zeros(n); v[idx]=V[:,j]; cols.append(v)
        Z=np.array(cols).T
        Ac=Z.T@A@Z; Aci=np.linalg.inv(Ac)
        M=lambda r: r+Z@(Aci@(Z.T@r))        # additive two-level (block-Jacobi = identity on the scaled system)
        out.append((g,k,Z.shape[1],cg(A,b,M)))
    print("alpha/hd %.0e: eig %.2e..%.2e (cond %.1e); CG plain %d; two-level (aggregate size, modes, coarse dim, its):"%(mult,w[0],w[-1],w[-1]/w[0],base), out[1:], flush=True)
