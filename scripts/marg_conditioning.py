"""CPU only: how well is the Schur complement of one dumped window's MARGIN_OLD marginalisation determined in double precision?  Condition of the dropped
block A_mm, and the kept system A_r through three routes: the C oracle (the reference's eigen pseudo-inverse, Jacobi), the same route with numpy's LAPACK, a plain
inverse.  usage: python scripts/marg_conditioning.py tmp_dump/win_k30.pkl"""
import os, sys, pickle
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import oracle_py as O, gfwindow as gw
d = pickle.load(open(sys.argv[1], "rb"))
w = gw.Window(); w.update(d); w.finalize()
O.ba_solve(w, 8)
s = O.ba_marg_system(w, 0)
A, b, m, n = s["A"], s["b"], s["m"], s["n"]
Amm = 0.5 * (A[:m, :m] + A[:m, :m].T)
ev = np.linalg.eigvalsh(Amm)
print("m", m, "n", n, "Amm eigen: min %.3e, <1e-8: %d, <1e-6: %d, <1e-3: %d, max %.3e" % (ev.min(), (ev < 1e-8).sum(), (ev < 1e-6).sum(), (ev < 1e-3).sum(), ev.max()))
print("smallest eigenvalues of Amm:", np.round(ev[:8], 12))
dg = np.diag(A)[:m]
print("diag of Amm: min %.3e ; entries < 1e-6: %d ; first 15 diag: %s" % (dg.min(), (dg < 1e-6).sum(), np.array2string(dg[:15], precision=3)))
# reference route: pseudo inverse by eigen with eps 1e-8
wv, V = np.linalg.eigh(Amm)
inv = V @ np.diag(np.where(wv > 1e-8, 1.0 / wv, 0.0)) @ V.T
Ar_ref = A[m:, m:] - A[m:, :m] @ inv @ A[:m, m:]
br_ref = b[m:] - A[m:, :m] @ inv @ b[:m]
sc = np.sqrt(np.maximum(np.diag(Ar_ref), 1e-300))
print("oracle Ar vs numpy eigen route: %.3e (scaled)" % np.abs((s["Ar"] - Ar_ref) / np.outer(sc, sc)).max())
# plain inverse
try:
    inv2 = np.linalg.inv(Amm)
    Ar2 = A[m:, m:] - A[m:, :m] @ inv2 @ A[:m, m:]
    print("plain inverse vs eigen route: dAr scaled %.3e" % np.abs((Ar2 - Ar_ref) / np.outer(sc, sc)).max())
except Exception as e:
    print("inv failed", e)
evr = np.linalg.eigvalsh(0.5 * (Ar_ref + Ar_ref.T))
print("Ar eigen: min %.3e <1e-8: %d <1e-6: %d" % (evr.min(), (evr < 1e-8).sum(), (evr < 1e-6).sum()))
