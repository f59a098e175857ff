"""Where does the marginalisation's rounding noise come from?  CPU-only experiment (oracle + mpmath).
Exact Schur complement (100 digits) vs (a) the reference's eigen pseudo-inverse route in double (the oracle), (b) block elimination in double."""
import os, sys
import numpy as np
import mpmath as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O, synth_window as SW
mp.mp.dps = 60

def exact_schur(A, b, m):
    P = A.shape[0]
    Am = mp.matrix(A.tolist()); bm = mp.matrix(b.tolist())
    Amm = Am[:m, :m]; Amr = Am[:m, m:]; Arm = Am[m:, :m]; Arr = Am[m:, m:]
    Amm = (Amm + Amm.T) / 2
    Ai = mp.inverse(Amm); X = Ai * Amr; y = Ai * bm[:m]
    Ar = Arr - Arm * X; br = bm[m:] - Arm * y
    return np.array(Ar.tolist(), dtype=float), np.array([float(v) for v in br])

def block_elim(A, b, m, mp_cols):
    """double: eliminate the diagonal feature block first (columns mp_cols..m), then the small pose/speed-bias block"""
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

for seed, kw in [(4, {}), (9, {}), (11, {}), (12, {}), (1, {"gnss": True}), (2, {"gnss": True})]:
    w = SW.make_window(seed, O, **kw)
    O.ba_solve(w, 8)
    s = O.ba_marg_system(w, 0)
    A, b, m, n = s["A"], s["b"], s["m"], s["n"]
    # dropped non-feature columns come first among the dropped ones? find the feature columns: scalar blocks after the 15 (or 20) pose/sb/clock columns
    mpc = 20 if kw.get("gnss") else 15
    Ax, bx = exact_schur(A, b, m)
    Ab, bb = block_elim(A, b, m, mpc)
    sc = np.sqrt(np.outer(np.diag(Ax), np.diag(Ax))) + 1e-6 * np.abs(Ax).max()
    ev = np.linalg.eigvalsh(0.5 * (A[:m, :m] + A[:m, :m].T))
    print("seed %d %s m=%d n=%d cond(Amm)=%.1e | oracle(eigen route): dA %.2e db %.2e | block elim: dA %.2e db %.2e | |br| %.2e |b_k| %.2e" % (
        seed, kw, m, n, ev.max() / ev.min(), (np.abs(s["Ar"] - Ax) / sc).max(), np.abs(s["br"] - bx).max() / np.abs(bx).max(),
        (np.abs(Ab - Ax) / sc).max(), np.abs(bb - bx).max() / np.abs(bx).max(), np.abs(bx).max(), np.abs(b[m:]).max()))
