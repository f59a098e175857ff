"""Which marginalisation is closer to the exact one?  The prior of a GNSS window (MARGIN_OLD) from the CPU oracle (the reference's route: eigen pseudo-inverse of the
dropped block, eigen square root of the kept system, both cut at 1e-8) and from the HIP library (block elimination, rank-revealing Cholesky, same threshold),
each measured against the same route evaluated with 60 digits on the oracle's assembled system (A, b).

  python scripts/marg_mpmath_check.py --dump gpurun_out/marg_dump.pkl [--seed 1] [--w20 | --novisual]   (on the GPU box: solves, marginalises on both sides, stores everything)
  python scripts/marg_mpmath_check.py gpurun_out/marg_dump.pkl                                 (CPU: the 60-digit evaluation and the comparison)

What is compared: J^T J and J^T r of the two priors (what the next solve sees), entry by entry, scaled by sqrt(A_ii A_jj) of the exact kept system."""
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O  # noqa: E402
import gfwindow as gw  # noqa: E402
import synth_window as SW  # noqa: E402


def dump(path, seed, w20):
    import gfamd
    W, F = (20, 150) if w20 else (10, 150)
    est = gfamd.Estimator(W, F, F * W, 1, max_gnss=12 * (W + 1))
    if "--novisual" in sys.argv:   # tests/test_backend_gpu.py::test_windows_without_a_factor_family[visual]: IMU + wheel + prior only
        est.close(); est = gfamd.Estimator()
        w = SW.make_window(seed, O)
        for k in [k for k in w if isinstance(w[k], np.ndarray) and k.startswith("vis_")]:
            w[k] = w[k][:0]
        w["para_Feature"], w["feature_fixed"] = w["para_Feature"][:0], w["feature_fixed"][:0]
    else:
        w = SW.make_window(seed, O, W=W, gnss=True) if w20 else SW.make_window(seed, O, gnss=True)
    O.ba_solve(w, 8)
    po = O.ba_marginalize(w, 0, cap_n=512)
    pg = est.marginalize([w], 0, cap_n=512)[0]
    state = {k: (np.array(v) if isinstance(v, np.ndarray) else v) for k, v in dict(w).items()}
    pickle.dump({"window": state, "po": po, "pg": pg}, open(path, "wb"))
    n = po["n"]
    Ao, Ag = po["J"].reshape(n, n).T @ po["J"].reshape(n, n), pg["J"].reshape(n, n).T @ pg["J"].reshape(n, n)
    sc = np.sqrt(np.maximum(np.diag(Ao), 1e-300))
    print("dumped %s: n %d; HIP vs oracle J^T J scaled %.2e, J^T r %.2e" % (path, n, np.abs((Ao - Ag) / np.outer(sc, sc)).max(),
                                                                            np.abs(po["J"].reshape(n, n).T @ po["r"] - pg["J"].reshape(n, n).T @ pg["r"]).max()))
    est.close()


def check(path, dps=60):
    import mpmath as mp
    mp.mp.dps = dps
    d = pickle.load(open(path, "rb"))
    w = gw.Window(); w.update(d["window"]); w.finalize()
    s = O.ba_marg_system(w, 0)
    A, b, m, n = s["A"], s["b"], s["m"], s["n"]
    po, pg = d["po"], d["pg"]
    assert po["n"] == n == pg["n"]
    print("system: %d dropped + %d kept columns" % (m, n))
    Amm = 0.5 * (A[:m, :m] + A[:m, :m].T)
    ev = np.linalg.eigvalsh(Amm)
    print("A_mm eigenvalues (double): min %.3e, %d below the 1e-8 cut, max %.3e  -> cond %.1e" % (ev.min(), int((ev < 1e-8).sum()), ev.max(), ev.max() / max(ev.min(), 1e-300)))

    def M(a):
        return mp.matrix([[mp.mpf(float(x)) for x in row] for row in np.atleast_2d(a)])
    # the reference's route with 60 digits (marginalization_factor.cpp:265-302): A_mm symmetrised, eigen pseudo-inverse (eigenvalues > 1e-8), Schur complement,
    # eigen-decomposition of the kept system, eigenvalues > 1e-8 kept: J^T J = V S V^T, J^T r = V V^T b_r
    Amm_m = M(Amm)
    E, Q = mp.eigsy(Amm_m)
    inv = mp.matrix(m, m)
    cut = mp.mpf("1e-8")
    for k in range(m):
        if E[k] > cut:
            for i in range(m):
                qi = Q[i, k] / E[k]
                if qi == 0:
                    continue
                for j in range(m):
                    inv[i, j] += qi * Q[j, k]
    Amr, Arm, Arr = M(A[:m, m:]), M(A[m:, :m]), M(A[m:, m:])
    bm, br = M(b[:m]).T, M(b[m:]).T
    Ar = Arr - Arm * inv * Amr
    brr = br - Arm * (inv * bm)
    E2, V2 = mp.eigsy((Ar + Ar.T) / 2)
    keep = [k for k in range(n) if E2[k] > cut]
    print("kept system eigenvalues (60 digits): %d of %d above the cut; those within a factor 100 of it: %s" % (len(keep), n, ", ".join("%.2e" % float(E2[k]) for k in range(n) if 1e-10 < float(E2[k]) < 1e-6)))
    AtA = mp.matrix(n, n)
    for k in keep:
        for i in range(n):
            vi = V2[i, k] * E2[k]
            for j in range(n):
                AtA[i, j] += vi * V2[j, k]
    Atb = mp.matrix(n, 1)
    for k in keep:
        c = sum(V2[i, k] * brr[i] for i in range(n))
        for i in range(n):
            Atb[i] += V2[i, k] * c
    exact_A = np.array([[float(AtA[i, j]) for j in range(n)] for i in range(n)])
    exact_b = np.array([float(Atb[i]) for i in range(n)])
    sc = np.sqrt(np.maximum(np.diag(exact_A), 1e-16 * np.abs(exact_A).max()))   # columns the prior says nothing about (zero rows of the exact system) scale with the top
    bs = np.abs(exact_b).max()
    for name, p in (("oracle", po), ("HIP   ", pg)):
        J = p["J"].reshape(n, n)
        Aj, bj = J.T @ J, J.T @ p["r"]
        eA = np.abs((Aj - exact_A) / np.outer(sc, sc))
        eb = np.abs(bj - exact_b)
        i, j = np.unravel_index(np.argmax(eA), eA.shape)
        print("%s vs 60 digits: J^T J scaled max %.2e (entry %d,%d: %.3e against %.3e), median %.1e, largest absolute %.2e of %.2e;  J^T r max %.2e (of %.2e), rank of J %d"
              % (name, eA.max(), i, j, Aj[i, j], exact_A[i, j], np.median(eA), np.abs(Aj - exact_A).max(), np.abs(exact_A).max(), eb.max(), bs, int((np.abs(J).max(axis=1) > 0).sum())))
    Jo, Jg = po["J"].reshape(n, n), pg["J"].reshape(n, n)
    print("oracle vs HIP:   J^T J scaled max %.2e;  J^T r max %.2e" % (np.abs((Jo.T @ Jo - Jg.T @ Jg) / np.outer(sc, sc)).max(), np.abs(Jo.T @ po["r"] - Jg.T @ pg["r"]).max()))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--dump" in sys.argv:
        seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 1
        dump(sys.argv[sys.argv.index("--dump") + 1], seed, "--w20" in sys.argv)
    else:
        check(args[0])
