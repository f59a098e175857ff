"""Conditioning of the GNSS chain (CPU only, oracle only): the oracle's own prior perturbed in its last bit, fed to the oracle's own next solve."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O, synth_window as SW
rng = np.random.default_rng(1)
for seed, kw in [(12, {}), (1, {"gnss": True}), (2, {"gnss": True})]:
    w = SW.make_window(seed, O, **kw); O.ba_solve(w, 8); po = O.ba_marginalize(w, 0)
    ref = SW.make_window(seed, O, frame0=1, prior=po, **kw); O.ba_solve(ref, 8)
    out = []
    for rep in range(5):
        pp = dict(po); pp["J"] = po["J"] * (1 + 2.2e-16 * rng.integers(-1, 2, po["J"].shape)); pp["r"] = po["r"] * (1 + 2.2e-16 * rng.integers(-1, 2, po["r"].shape))
        w2 = SW.make_window(seed, O, frame0=1, prior=pp, **kw); O.ba_solve(w2, 8)
        P = w2["para_Pose"].reshape(-1, 7) - ref["para_Pose"].reshape(-1, 7)
        out.append((np.abs(P[:, :3]).max(), np.abs(P[:, 3:]).max() * 2, np.abs(w2["para_SpeedBias"] - ref["para_SpeedBias"]).max()))
    print(seed, kw, "last-bit perturbation of the oracle's own prior -> oracle's next solve: max |dP| %.1e |dR| %.1e |dSB| %.1e" % tuple(np.max(np.array(out), axis=0)))
    if kw.get("gnss"):
        P = w2["para_Pose"].reshape(-1, 7) - ref["para_Pose"].reshape(-1, 7)
        print("   dP per frame (last rep):", np.array2string(np.abs(P[:, :3]).max(axis=1), precision=1))

print("--- last-bit perturbation of the solved window state BEFORE the oracle's own marginalisation (all-oracle chain against itself)")
for seed, kw in [(12, {}), (9, {}), (1, {"gnss": True}), (2, {"gnss": True})]:
    w = SW.make_window(seed, O, **kw); O.ba_solve(w, 8); po = O.ba_marginalize(w, 0)
    ref = SW.make_window(seed, O, frame0=1, prior=po, **kw); O.ba_solve(ref, 8)
    n = po["n"]; Jo = po["J"].reshape(n, n); bo = Jo.T @ po["r"]
    out = []
    for rep in range(6):
        wp = w.copy()
        for k in ("para_Pose", "para_SpeedBias", "para_Feature"):
            wp[k] = wp[k] * (1 + 2.2e-16 * rng.integers(-1, 2, wp[k].shape))
        pp = O.ba_marginalize(wp, 0)
        Jp = pp["J"].reshape(n, n); bp = Jp.T @ pp["r"]
        w2 = SW.make_window(seed, O, frame0=1, prior=pp, **kw); O.ba_solve(w2, 8)
        P = w2["para_Pose"].reshape(-1, 7) - ref["para_Pose"].reshape(-1, 7)
        out.append((np.abs(bp - bo).max() / np.abs(bo).max(), np.abs(P[:, :3]).max(), np.abs(P[:, 3:]).max() * 2))
    o = np.array(out)
    print(seed, kw, "db (J^T r) %s | chained |dP| %s |dR| max %.1e" % (np.array2string(o[:, 0], precision=1), np.array2string(o[:, 1], precision=1), o[:, 2].max()))
