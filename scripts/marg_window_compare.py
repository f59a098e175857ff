"""one dumped window (scripts/replay_debug.py style pickle): solve it with the oracle, then marginalise the solved window with the oracle and with the library and
compare the two priors as J^T J, J^T r and as what they say about a unit step in every kept direction"""
import os, sys, pickle
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gfamd, oracle_py as O, gfwindow as gw
d = pickle.load(open(sys.argv[1], "rb"))
w = gw.Window(); w.update(d); w.finalize()
O.ba_solve(w, 8)
ba = gfamd.Estimator(10, 512, 4096)
for mode in (0, 1):
    po = O.ba_marginalize(w.copy(), mode)
    pp = ba.marginalize([w.copy()], mode)[0]
    n = po["n"]
    Jo, Jp = po["J"].reshape(n, n), pp["J"].reshape(pp["n"], pp["n"])
    Ao, Ap, bo, bp = Jo.T @ Jo, Jp.T @ Jp, Jo.T @ po["r"], Jp.T @ pp["r"]
    ev = np.linalg.eigvalsh(Ao)
    sc = np.sqrt(np.maximum(np.diag(Ao), 1e-300))
    print("mode %d: n %d/%d  |A| %.3e  dA %.3e (scaled %.3e)  |b| %.3e db %.3e (scaled %.3e)  eigenvalues of A: min %.3e, below 1e-6: %d, below 1e-3: %d, max %.3e" % (
        mode, n, pp["n"], np.abs(Ao).max(), np.abs(Ap - Ao).max(), np.abs((Ap - Ao) / np.outer(sc, sc)).max(), np.abs(bo).max(), np.abs(bp - bo).max(),
        np.abs((bp - bo) / sc).max(), ev.min(), int((ev < 1e-6).sum()), int((ev < 1e-3).sum()), ev.max()))
    # the Gauss-Newton step each prior alone would take (pseudo-inverse over eigenvalues above 1e-8, as the reference's truncation)
    wv, V = np.linalg.eigh(Ao)
    keep = wv > 1e-8
    step_o = V[:, keep] @ ((V[:, keep].T @ bo) / wv[keep])
    step_p = V[:, keep] @ ((V[:, keep].T @ bp) / wv[keep])
    print("        step of the prior alone: |x| %.3e, difference between the two priors' steps %.3e" % (np.abs(step_o).max(), np.abs(step_o - step_p).max()))
