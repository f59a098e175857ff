"""Adjudication of the bars on a FREED camera extrinsic (tests/test_estimator_gpu.py `subset_cam`: tic within 1e-4 relative; tests/test_backend_gpu.py seed 3: relative
1e-6 on a block that wanders by metres): how far is ONE trust-region step of each implementation from the same step evaluated with 60 digits?

Window: tests/golden/ref_window_free_ex_td (free camera extrinsic and td, no prior: the extrinsic translation is observed through parallax only).  The first step of Ceres'
dogleg from the initial radius 1e4 is the regularised Gauss-Newton step
    delta = - s o [ (s H s + mu D^2)^-1 (s g) ],   s = 1 / (1 + sqrt(diag H)),  D^2 = clamp(s^2 diag H, 1e-6, 1e32),  mu = 1e-8
(trust_region_minimizer.cc: Jacobi scaling; dogleg_strategy.cc: ComputeGaussNewtonStep with mu = min_mu, the step is taken whole when it lies inside the radius), followed
by x (+) delta with the reference's local parameterisations.  H and g come from the reference's factor formulas at 60 digits (tests/golden/make_ref_golden.py), the solve
is mpmath's LU at 60 digits: no Schur complement, no Cholesky, no double-precision anything.  Compared: the camera extrinsic (and the poses) after one iteration of the CPU
oracle and -- with the dump made on the GPU box -- of the HIP solver.

  python scripts/adjudicate_free_extrinsic_step.py --dump gpurun_out/free_ex_step_hip.pkl    (GPU box)
  python scripts/adjudicate_free_extrinsic_step.py [gpurun_out/free_ex_step_hip.pkl]        (CPU)"""
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ground-fusion_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import gfwindow as gw  # noqa: E402


def load():
    from test_golden import load_ref_window
    return load_ref_window("ref_window_free_ex_td")


def dump(path):
    import gfamd
    w, fx, _, _ = load()
    est = gfamd.Estimator(max_features=16, max_visual=256)
    a = w.copy()
    s = est.solve([a], 1)[0]
    pickle.dump({"state": {k: a[k] for k in gw.STATE_KEYS}, "summary": s}, open(path, "wb"))
    est.close()
    print("wrote", path, s)


def main():
    import mpmath as mp
    import make_ref_golden as G
    import oracle_py as O
    mp.mp.dps = 60
    w, fx, _, _ = load()
    ids = fx["ids"]
    n = len(ids)
    H, g, cost = G.window_normal_equations(w, ids)
    s = [1 / (1 + mp.sqrt(H[c, c])) for c in range(n)]
    S = mp.matrix(n, n)
    for a in range(n):
        for c in range(n):
            S[a, c] = s[a] * H[a, c] * s[c]
    mu = mp.mpf("1e-8")
    for c in range(n):
        S[c, c] += mu * min(max(s[c] * s[c] * H[c, c], mp.mpf("1e-6")), mp.mpf("1e32"))
    y = mp.lu_solve(S, mp.matrix([s[c] * g[c] for c in range(n)]))
    delta = [-s[c] * y[c] for c in range(n)]
    # the step inside the radius?  |gn| in dogleg space = |D y|
    gn_norm = mp.sqrt(sum((mp.sqrt(min(max(s[c] * s[c] * H[c, c], mp.mpf("1e-6")), mp.mpf("1e32"))) * y[c]) ** 2 for c in range(n)))
    print("exact Gauss-Newton step: |D y| = %s (radius 1e4: %s)" % (mp.nstr(gn_norm, 6), "taken whole" if gn_norm <= 1e4 else "CLIPPED -- the comparison below does not apply"))
    col0 = {}
    for c, b in enumerate(ids):
        col0.setdefault(int(b), c)
    # x (+) delta for the camera extrinsic (PoseLocalParameterization::Plus, pose_local_parameterization.cpp:12-28) and the poses
    def plus(p7, d6):
        P, Q = G.pose_of(p7)
        q = G.qnormalized(G.qmul(Q, G.delta_q(mp.matrix(d6[3:6]))))
        return [P[0] + d6[0], P[1] + d6[1], P[2] + d6[2], q[1], q[2], q[3], q[0]]
    c_ex = col0[gw.bid(gw.EX_POSE)]
    ex_exact = plus(w["para_Ex_Pose"], delta[c_ex:c_ex + 6])
    poses_exact = [plus(w["para_Pose"][7 * i:7 * i + 7], delta[col0[gw.bid(gw.POSE, i)]:col0[gw.bid(gw.POSE, i)] + 6]) for i in range(int(w["W"]) + 1)]
    runs = {}
    a = w.copy()
    so = O.ba_solve(a, 1)
    runs["oracle"] = ({k: a[k] for k in gw.STATE_KEYS}, so)
    if len(sys.argv) > 1 and os.path.exists(sys.argv[1]):
        d = pickle.load(open(sys.argv[1], "rb"))
        runs["hip"] = (d["state"], d["summary"])
    ex0 = np.array(w["para_Ex_Pose"])
    exE = np.array([float(v) for v in ex_exact])
    print("camera extrinsic: start tic %s; exact step moves it by %s m (a block of %.2f m)" % (np.round(ex0[:3], 4), np.round(exE[:3] - ex0[:3], 4), np.linalg.norm(exE[:3])))
    for name, (st, summ) in runs.items():
        ex = np.array(st["para_Ex_Pose"])
        PE = np.array([[float(v) for v in p] for p in poses_exact])
        P = np.array(st["para_Pose"]).reshape(-1, 7)
        moved = not np.array_equal(ex, ex0)
        print("   %-6s after one iteration (accepted: %s, successful steps %d): tic vs exact %.2e m (relative to the step %.2e), qic %.2e; poses: position %.2e m, rotation %.2e rad"
              % (name, moved, summ["successful_steps"], np.abs(ex[:3] - exE[:3]).max(), np.abs(ex[:3] - exE[:3]).max() / max(np.abs(exE[:3] - ex0[:3]).max(), 1e-300),
                 min(np.abs(ex[3:] - exE[3:]).max(), np.abs(ex[3:] + exE[3:]).max()), np.abs(P[:, :3] - PE[:, :3]).max(),
                 2 * min(np.abs(P[:, 3:] - PE[:, 3:]).max(), np.abs(P[:, 3:] + PE[:, 3:]).max())))
    if "hip" in runs:
        print("   hip vs oracle: tic %.2e m" % np.abs(np.array(runs["hip"][0]["para_Ex_Pose"])[:3] - np.array(runs["oracle"][0]["para_Ex_Pose"])[:3]).max())
    # conditioning of the scaled, damped system
    Sd = np.array([[float(S[a_, c]) for c in range(n)] for a_ in range(n)])
    ev = np.linalg.eigvalsh(Sd)
    print("scaled damped system: eigenvalues %.2e .. %.2e (cond %.1e): a double-precision solve may lose that many digits in the weakest direction" % (ev.min(), ev.max(), ev.max() / ev.min()))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--dump":
        dump(sys.argv[2])
    else:
        main()
