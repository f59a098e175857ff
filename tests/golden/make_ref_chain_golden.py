"""A two-window CHAIN at 60 digits: tests/golden/ref_chain_*.json.gz.

What `Estimator::optimization` does for two consecutive keyframes (estimator.cpp:2890-3631): window 1 is solved, its oldest frame is marginalised (MARGIN_OLD), the prior
is carried -- renamed by one frame -- into window 2, window 2 is solved.  Every piece is the 60-digit one:
  solve         make_ref_solve_golden.solve   (H, g, cost of every iterate from the reference's factor formulas; Ceres' trust-region loop; LU on the full system)
  marginalise   make_ref_golden.marg_golden   (the reference's route: eigen pseudo-inverse of the dropped block, eigen square root of the kept system, eps 1e-8)
with the hand-overs rounded to double where the reference stores doubles (the state vector, the prior's linearised_jacobians / linearised_residuals / keep_block_data).

Inputs of the fixture: both windows (window 2 without its prior).  Expected: the state of window 1 after its solve, the state of window 2 after its solve with the prior
of window 1.  A test runs an implementation's own chain (its solve, its marginalisation, its solve) and compares the end state: tests/test_golden.py (CPU oracle),
tests/test_backend_gpu.py (HIP).

  python tests/golden/make_ref_chain_golden.py            (~15 min)"""
import gzip
import json
import os
import sys
import time

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "ground-fusion_amd"), os.path.join(ROOT, "oracle"), HERE, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import gfwindow as gw  # noqa: E402
import make_ref_golden as G  # noqa: E402
import make_ref_solve_golden as S  # noqa: E402

mp.mp.dps = 60


def shifted(kept):
    """the ids the prior carries are those of the NEXT window (addr_shift, estimator.cpp:3471-3500)"""
    ids = []
    for b in kept:
        kind, i = b // 4096, b % 4096
        if kind in (gw.POSE, gw.SPEEDBIAS, gw.RCV_DDT):
            i -= 1
        elif kind == gw.RCV_DT:
            i -= 4
        ids.append(kind * 4096 + i)
    return np.array(ids, np.int32)


def exact_prior(x1):
    g = G.marg_golden(x1, 0)
    n = g["n"]
    return {"block_id": shifted(g["kept"]), "J": np.array([[float(g["J"][a, c]) for c in range(n)] for a in range(n)]).reshape(-1),
            "r": np.array([float(g["r"][a]) for a in range(n)]), "x0": np.array(g["x0"]), "m": g["m"], "n": n}, g


def pose_dev(a, b):
    P, Q = np.asarray(a["para_Pose"]).reshape(-1, 7), np.asarray(b["para_Pose"]).reshape(-1, 7)
    return (np.abs(P[:, :3] - Q[:, :3]).max(), np.abs((P[:, :3] - P[0, :3]) - (Q[:, :3] - Q[0, :3])).max(),
            2 * min(np.abs(P[:, 3:] - Q[:, 3:]).max(), np.abs(P[:, 3:] + Q[:, 3:]).max()))


def to_list(a):
    return np.asarray(a).reshape(-1).tolist()


CASES = {"ref_chain_wheel": dict(seed=21, max_features=8, n_landmarks=12),
         "ref_chain_gnss": dict(seed=1, max_features=8, n_landmarks=12, gnss=True)}


def main():
    import oracle_py as O     # synthesises the windows (input data) and gives the column order; its solver / marginalisation are what gets compared
    import synth_window as SW
    for name in (sys.argv[1:] or CASES):
        kw = dict(CASES[name])
        seed = kw.pop("seed")
        t0 = time.time()
        w1 = SW.make_window(seed, O, **kw)
        w1.finalize()
        ids1 = [int(x) for x in O.ba_linearize(w1.copy())["ids"]]
        print("%s: window 1, %d columns" % (name, len(ids1)), flush=True)
        x1, info1 = S.solve(w1, ids1, 8)
        pe, g = exact_prior(x1)
        print("   prior: %d dropped, %d kept columns; kept eigenvalues next to the 1e-8 cut: %s   (%.0f s)" % (g["m"], g["n"], ["%.1e" % v for v in g["eigenvalues_kept"] if 1e-12 < abs(v) < 1e-5],
                                                                                                            time.time() - t0), flush=True)
        w2 = SW.make_window(seed, O, frame0=1, **kw)
        w2.finalize()
        w2p = w2.copy().set_prior(pe)
        ids2 = [int(x) for x in O.ba_linearize(w2p.copy())["ids"]]
        print("   window 2, %d columns" % len(ids2), flush=True)
        x2, info2 = S.solve(w2p, ids2, 8)
        # the oracle's own chain next to it
        a1 = w1.copy()
        O.ba_solve(a1, 8)
        po = O.ba_marginalize(a1, 0)
        assert list(po["block_id"]) == list(pe["block_id"])
        a2 = w2.copy().set_prior(po)
        so = O.ba_solve(a2, 8)
        print("   oracle chain vs 60-digit chain: window 1 position %.2e m; window 2 position %.2e m, shape %.2e m, rotation %.2e rad; iterations %d/%d vs %d/%d   (%.0f s)"
              % (pose_dev(a1, x1)[0], *pose_dev(a2, x2), so["iterations"], so["successful_steps"], info2["iterations"], info2["successful_steps"], time.time() - t0), flush=True)
        fx = {"about": "two consecutive windows through solve -> MARGIN_OLD -> solve with every number at 60 digits (tests/golden/make_ref_chain_golden.py): inputs = window_1, "
                       "window_2 (without a prior); expected = state_1 after the first solve, state_2 after the second solve with the first window's prior",
              "window_1": {k: (to_list(v) if isinstance(v, np.ndarray) else v) for k, v in dict(w1).items()},
              "window_2": {k: (to_list(v) if isinstance(v, np.ndarray) else v) for k, v in dict(w2).items()},
              "state_1": {k: to_list(x1[k]) for k in gw.STATE_KEYS}, "summary_1": info1, "state_2": {k: to_list(x2[k]) for k in gw.STATE_KEYS}, "summary_2": info2,
              "prior_block_id": [int(b) for b in pe["block_id"]], "prior_m": int(pe["m"]), "prior_n": int(pe["n"]), "eigenvalues_kept": g["eigenvalues_kept"]}
        path = os.path.join(HERE, name + ".json.gz")
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(json.dumps(fx).encode())
        print("   wrote", path, os.path.getsize(path), "bytes", flush=True)


if __name__ == "__main__":
    main()
