"""The right-hand side of the marginalisation prior below the rank of its factor ("the last algorithmic gap"): on the ORACLE's kept systems of
GNSS windows (oracle_py.ba_marg_system: A_r, b_r), J^T r of
  * the reference's route: eigen-decomposition, eigenvalues <= 1e-8 dropped, r = S^-1/2 V^T b  ->  J^T r = projection of b onto the kept eigenvectors (marginalization_factor.cpp:294-302),
  * a rank-revealing pivoted Cholesky with r from forward substitution (the library until round 3): exact in the pivot rows, predicted in the others,
  * the same factor with the least-squares r (the library since round 4, csrc/gf_ba_marg.hpp): orthogonal projection onto the factor's range.
CPU only:  python scripts/marg_rhs_projection.py"""
import sys, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ground-fusion_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import oracle_py as O, synth_window as SW
def pivchol(A, b, eps=1e-8):
    A=A.copy(); n=len(A); perm=np.arange(n); L=np.zeros((n,n)); z=b.copy(); rank=n
    d=np.diag(A).copy()
    for k in range(n):
        p=k+np.argmax(np.diag(A)[k:])
        if not A[p,p]>eps: rank=k; break
        if p!=k:
            A[[k,p]]=A[[p,k]]; A[:,[k,p]]=A[:,[p,k]]; perm[[k,p]]=perm[[p,k]]; z[[k,p]]=z[[p,k]]; L[[k,p]]=L[[p,k]]
        l=np.sqrt(A[k,k]); L[k,k]=l; L[k+1:,k]=A[k+1:,k]/l
        A[k+1:,k+1:]-=np.outer(L[k+1:,k],L[k+1:,k])
    L=L[:,:rank]
    # J = L^T P^T : J[:, perm[pos]] = L[pos,:]
    J=np.zeros((rank,n)); J[:,perm]=L.T
    bt=b[perm]
    r_fs=np.linalg.solve(L[:rank,:rank], bt[:rank])      # forward substitution of the leading rows (library)
    r_ls=np.linalg.lstsq(L, bt, rcond=None)[0]           # least squares
    return J, r_fs, r_ls, rank
for seed in (1,2,3):
    w=SW.make_window(seed,O,gnss=True); O.ba_solve(w,8)
    s=O.ba_marg_system(w,0); Ar,br,n=s["Ar"],s["br"],s["n"]
    Ar=0.5*(Ar+Ar.T)
    lam,V=np.linalg.eigh(Ar); keep=lam>1e-8
    ref=V[:,keep]@(V[:,keep].T@br)                      # reference: projection of b onto the kept eigenvectors
    J,rf,rl,rank=pivchol(Ar,br)
    sc=np.abs(br).max()
    print("seed",seed,"n",n,"kept eig",keep.sum(),"rank",rank,"|b|max %.3g"%sc,
          " J^T r - ref: forward-subst %.3e  least-squares %.3e   (b - ref %.3e)"%(np.abs(J.T@rf-ref).max(), np.abs(J.T@rl-ref).max(), np.abs(br-ref).max()))
    # what the next solve sees: gradient in the solve's metric; compare also cost constants
    print("     |r|^2: ref %.6f fs %.6f ls %.6f"%( (br@(V[:,keep]/lam[keep])@V[:,keep].T@br), rf@rf, rl@rl))
