"""CPU simulation of ba_marg_finish's arithmetic (block elimination + rank-revealing pivoted Cholesky, eps 1e-8) under 1-ulp perturbations of
the assembled system, to locate the run-to-run spread of J^T r."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O, synth_window as SW

def block_elim(A, b, m, mp_cols):
    M = A.copy(); g = b.copy()
    d = np.diag(M)[mp_cols:m].copy()
    keep = np.r_[0:mp_cols, m:A.shape[0]]
    E = M[np.ix_(keep, range(mp_cols, m))]
    w = np.where(d > 1e-8, 1.0 / d, 0.0)
    M2 = M[np.ix_(keep, keep)] - (E * w) @ E.T
    g2 = g[keep] - (E * w) @ g[mp_cols:m]
    P = M2[:mp_cols, :mp_cols]; P = 0.5 * (P + P.T)
    Pinv = np.linalg.inv(P)
    K = M2[mp_cols:, :mp_cols]
    Ar = M2[mp_cols:, mp_cols:] - K @ Pinv @ K.T
    br = g2[mp_cols:] - K @ (Pinv @ g2[:mp_cols])
    return Ar, br

def piv_chol(A, b, eps=1e-8):
    n = A.shape[0]; A = A.copy(); perm = np.arange(n); zb = b.copy(); dg = np.diag(A).copy()
    L = np.zeros((n, n)); z = np.zeros(n); rank = n
    for k in range(n):
        cand = dg[perm[k:]]; bi = k + int(np.argmax(cand)); best = cand.max()
        if not best > eps: rank = k; break
        perm[[k, bi]] = perm[[bi, k]]; zb[[k, bi]] = zb[[bi, k]]
        pk = perm[k]; dinv = 1 / np.sqrt(best); rk = zb[k] * dinv; z[k] = rk
        L[pk, k] = np.sqrt(best)
        rest = perm[k + 1:]
        l = A[rest, pk] * dinv
        L[rest, k] = l
        zb[k + 1:] -= l * rk; dg[rest] -= l * l
        A[np.ix_(rest, rest)] -= np.outer(A[rest, pk], A[rest, pk]) / best
    J = L.T.copy(); J[rank:] = 0
    return J, z, rank, perm

rng = np.random.default_rng(0)
for seed, kw in [(4, {}), (12, {}), (1, {"gnss": True})]:
    w = SW.make_window(seed, O, **kw); O.ba_solve(w, 8)
    s = O.ba_marg_system(w, 0); A, b, m, n = s["A"], s["b"], s["m"], s["n"]; mpc = 20 if kw.get("gnss") else 15
    res = []
    for rep in range(6):
        Ap = A * (1 + 1.1e-16 * rng.standard_normal(A.shape)); Ap = np.tril(Ap) + np.tril(Ap, -1).T; bp = b * (1 + 1.1e-16 * rng.standard_normal(b.shape))
        if rep == 0: Ap, bp = A, b
        Ar, br = block_elim(Ap, bp, m, mpc); Ar = 0.5 * (Ar + Ar.T)
        J, z, rank, perm = piv_chol(Ar, br)
        res.append((Ar, br, J.T @ J, J.T @ z, rank))
    A0, b0, JJ0, Jr0, r0 = res[0]
    ev = np.linalg.eigvalsh(A0)
    print("seed %d %s: eig(A_r) smallest %s largest %.2e" % (seed, kw, np.array2string(ev[:6], precision=2), ev[-1]))
    for Ar, br, JJ, Jr, rank in res:
        print("   rank %d | spread of b_r %.1e A_r %.1e | J^T J vs A_r %.1e (abs %.1e) | J^T r vs b_r %.1e | J^Tr spread vs run0 %.1e" % (
            rank, np.abs(br - b0).max() / np.abs(b0).max(), np.abs(Ar - A0).max() / np.abs(A0).max(), np.abs(JJ - Ar).max() / np.abs(Ar).max(), np.abs(JJ - Ar).max(),
            np.abs(Jr - br).max() / np.abs(br).max(), np.abs(Jr - Jr0).max() / np.abs(Jr0).max()))
