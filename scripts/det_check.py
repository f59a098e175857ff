"""Determinism + parity diagnostics of the back end on the GPU: two fresh handles must give bit-identical solves and priors;
priors / chained poses against the oracle.  Prints one line per case (used to set the bounds in tests/test_backend_gpu.py)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gfamd as gf, oracle_py as O, synth_window as SW

def inv(p):
    n = p["n"]; J = p["J"].reshape(n, n); return J.T @ J, J.T @ p["r"]

def pose_diff(a, b):
    pa, pb = a["para_Pose"].reshape(-1, 7), b["para_Pose"].reshape(-1, 7)
    return np.abs(pa[:, :3] - pb[:, :3]).max(), 2 * min(np.abs(pa[:, 3:] - pb[:, 3:]).max(), np.abs(pa[:, 3:] + pb[:, 3:]).max())

cases = [(4, {}), (7, {"use_wheel": False}), (9, {}), (11, {}), (12, {}), (1, {"gnss": True}), (2, {"gnss": True})]
for seed, kw in cases:
    gn = kw.get("gnss")
    mk = (lambda: gf.Estimator(10, 150, 1500, 1, max_gnss=12 * 11)) if gn else (lambda: gf.Estimator())
    w = SW.make_window(seed, O, **kw)
    wo = w.copy(); so = O.ba_solve(wo, 8); po = O.ba_marginalize(wo, 0)
    runs = []
    for rep in range(3):
        est = mk(); wg = w.copy(); sg = est.solve([wg], 8)[0]; pg = est.marginalize([wg], 0)[0]; pgo = est.marginalize([wo.copy()], 0)[0]; est.close()
        runs.append((wg, sg, pg, pgo))
    same = all(np.array_equal(runs[0][0]["para_Pose"], r[0]["para_Pose"]) and np.array_equal(runs[0][2]["J"], r[2]["J"]) and np.array_equal(runs[0][2]["r"], r[2]["r"]) for r in runs[1:])
    wg, sg, pg, pgo = runs[0]
    dp, dr = pose_diff(wo, wg)
    Ao, bo = inv(po); Ag, bg = inv(pgo)     # same (oracle-solved) window marginalised on both sides
    sc = np.sqrt(np.outer(np.diag(Ao), np.diag(Ao))) + 1e-6 * np.abs(Ao).max()
    # chain: next window with each prior
    w2o = SW.make_window(seed, O, frame0=1, prior=po, **kw); w2g = SW.make_window(seed, O, frame0=1, prior=pg, **kw)
    est = mk(); O.ba_solve(w2o, 8); est.solve([w2g], 8); est.close()
    cp, cr = pose_diff(w2o, w2g)
    print("seed %2d %-22s bit-identical runs: %s | solve dP %.1e dR %.1e it %d/%d | prior (same window) dA %.1e dA/|A| %.1e db %.1e | all-HIP chain vs all-oracle dP %.1e dR %.1e" % (
        seed, kw, same, dp, dr, sg["iterations"], so["iterations"], (np.abs(Ao - Ag) / sc).max(), np.abs(Ao - Ag).max() / np.abs(Ao).max(), np.abs(bo - bg).max() / np.abs(bo).max(), cp, cr), flush=True)
