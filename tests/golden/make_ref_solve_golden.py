"""The trust-region LOOP of a window solve at 60 digits: tests/golden/ref_solve_*.json.gz.

make_ref_golden.py pins what ONE linearisation is (H, g, cost from the reference's factor formulas) and what the first step is.  This script runs the whole solve the
reference asks Ceres for (estimator.cpp:3299-3318: DENSE_SCHUR, DOGLEG, max_num_iterations = NUM_ITERATIONS = 8) with every number at 60 digits:

  * H, g and the cost of every iterate come from `make_ref_golden.window_normal_equations` (the reference's formulas, file:line cited there);
  * the loop is Ceres' (ceres-solver is a dependency of the reference that is not in its tree -- pinned by the reference's build to the 1.14 line; its published algorithm,
    restated): Jacobi scaling s = 1 / (1 + sqrt(diag H)) of the FIRST linearisation (trust_region_minimizer.cc), TRADITIONAL_DOGLEG (dogleg_strategy.cc: diagonal
    D = sqrt(clamp(s^2 diag H, 1e-6, 1e32)), Cauchy step length alpha, the regularised Gauss-Newton step (s H s + mu D^2) y = s g, mu from 1e-8, the three cases
    Gauss-Newton / scaled gradient / dogleg interpolation, radius 1e4, x 0.5 on a rejected or poor step, max(radius, 3 |step|) on a good one), model cost change
    -(g.d + d^T H d / 2), x (+) d by the reference's local parameterisations (pose_local_parameterization.cpp:12-28; masked components of the extrinsics dropped as
    pose_subset_parameterization.cpp:27-56 does), parameter tolerance 1e-8, function tolerance 1e-6, min_relative_decrease 1e-3, gradient tolerance 1e-10;
  * the linear solve is LU on the full system at 60 digits: no Schur complement, no Cholesky, no elimination order to agree on.
The iterate is rounded to double after every x (+) d, as Ceres' own state vector is: what is pinned is the arithmetic between two iterates and the decisions taken on it.

The only wall-clock rule of the reference's options (max_solver_time_in_seconds = SOLVER_TIME) is not part of the fixture: it cuts a solve short on a slow host, it does
not change an iterate.

Expected, per window: the state after the solve, the number of iterations / successful steps, the termination reason, and the cost after every iteration.
Tests: tests/test_golden.py (the CPU oracle against it), tests/test_backend_gpu.py (the HIP solver against it).

  python tests/golden/make_ref_solve_golden.py            (~6 min)"""
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

mp.mp.dps = 60
mpf = mp.mpf

SLOT = {gw.POSE: ("para_Pose", 7), gw.SPEEDBIAS: ("para_SpeedBias", 9), gw.EX_POSE: ("para_Ex_Pose", 0), gw.EX_WHEEL: ("para_Ex_Pose_wheel", 0), gw.TD: ("para_Td", 0),
        gw.TD_WHEEL: ("para_Td_wheel", 0), gw.FEATURE: ("para_Feature", 1), gw.RCV_DT: ("para_rcv_dt", 1), gw.RCV_DDT: ("para_rcv_ddt", 1), gw.YAW: ("para_yaw_enu_local", 0),
        gw.ANC: ("para_anc_ecef", 0)}


def slot(b):
    """block id -> (state key, offset, global size)"""
    kind, idx = b // 4096, b % 4096
    if kind in (gw.SX, gw.SY, gw.SW):
        return "para_Ix", kind - gw.SX, 1
    key, stride = SLOT[kind]
    return key, stride * idx, gw.gsize(kind)


def blocks(ids):
    out, seen = [], set()
    for c, b in enumerate(ids):
        if int(b) not in seen:
            seen.add(int(b))
            out.append((int(b), c))
    return out


def plus(w, ids, d):
    """x (+) d in 60 digits, rounded to double at the end (Ceres' state is a double vector)"""
    a = w.copy()
    for b, c0 in blocks(ids):
        kind = b // 4096
        key, off, gs = slot(b)
        ls = gw.lsize(kind)
        dd = [d[c0 + i] for i in range(ls)]
        if kind in (gw.EX_POSE, gw.EX_WHEEL):
            mask = int(w["ex_pose_mask" if kind == gw.EX_POSE else "ex_wheel_mask"])
            dd = [mpf(0) if (mask >> i) & 1 else dd[i] for i in range(6)]
        x = w[key][off:off + gs]
        if gs == 7:
            P, Q = G.pose_of(x)
            q = G.qnormalized(G.qmul(Q, G.delta_q(mp.matrix(dd[3:6]))))
            new = [P[0] + dd[0], P[1] + dd[1], P[2] + dd[2], q[1], q[2], q[3], q[0]]
        else:
            new = [mpf(x[i]) + dd[i] for i in range(gs)]
        a[key][off:off + gs] = [float(v) for v in new]
    return a


def gather(w, ids):
    x = []
    for b, _ in blocks(ids):
        key, off, gs = slot(b)
        x += [mpf(v) for v in w[key][off:off + gs]]
    return x


def solve(w, ids, max_iters=8, log=print):
    n = len(ids)
    x = w.copy()
    H, g, cost = G.window_normal_equations(x, ids)
    s = [1 / (1 + mp.sqrt(H[c, c])) for c in range(n)]
    radius, mu = mpf(10) ** 4, mpf("1e-8")
    reuse = False
    gmax = max(abs(g[c]) for c in range(n))
    info = {"initial_cost": mp.nstr(cost, 30), "iterations": 0, "successful_steps": 0, "termination": 0, "trace": []}
    last_ok, invalid_run = True, 0
    it = 0
    while True:
        it += 1
        if it - 1 >= max_iters:
            info["termination"] = 0
            break
        if last_ok and gmax <= mpf("1e-10"):
            info["termination"] = 3
            break
        if radius <= mpf("1e-32"):
            info["termination"] = 4
            break
        info["iterations"] = it
        if not reuse:
            reuse = True
            D = [mp.sqrt(min(max(s[c] * s[c] * H[c, c], mpf("1e-6")), mpf("1e32"))) for c in range(n)]
            grad = [s[c] * g[c] / D[c] for c in range(n)]                 # gradient in the dogleg's space
            sg = [s[c] * grad[c] / D[c] for c in range(n)]                # ... mapped back to the unscaled space: J_s (grad / D) = J (s grad / D)
            Hsg = H * mp.matrix(sg)
            alpha = sum(v * v for v in grad) / sum(sg[c] * Hsg[c] for c in range(n))
            A = mp.matrix(n, n)
            for a_ in range(n):
                for c in range(n):
                    A[a_, c] = s[a_] * H[a_, c] * s[c]
            for c in range(n):
                A[c, c] += mu * D[c] * D[c]
            y = mp.lu_solve(A, mp.matrix([s[c] * g[c] for c in range(n)]))
            gn = [-D[c] * y[c] for c in range(n)]
        gnorm = mp.sqrt(sum(v * v for v in grad))
        gnn = mp.sqrt(sum(v * v for v in gn))
        if gnn <= radius:
            step, case, sn = list(gn), "gauss-newton", gnn
        elif gnorm * alpha >= radius:
            step, case, sn = [-(radius / gnorm) * v for v in grad], "gradient", radius
        else:
            gdot = sum(grad[c] * gn[c] for c in range(n))
            b_dot_a, a_sq = -alpha * gdot, (alpha * gnorm) ** 2
            bma = a_sq - 2 * b_dot_a + gnn ** 2
            c_ = b_dot_a - a_sq
            d_ = mp.sqrt(c_ * c_ + bma * (radius ** 2 - a_sq))
            beta = (d_ - c_) / bma if c_ <= 0 else (radius * radius - a_sq) / (d_ + c_)
            step = [(-alpha * (1 - beta)) * grad[c] + beta * gn[c] for c in range(n)]
            case, sn = "dogleg", mp.sqrt(sum(v * v for v in step))
        d = [step[c] / D[c] * s[c] for c in range(n)]                       # the unscaled increment
        Hd = H * mp.matrix(d)
        model = -(sum(g[c] * d[c] for c in range(n)) + sum(d[c] * Hd[c] for c in range(n)) / 2)
        if not model > 0:
            last_ok = False
            invalid_run += 1
            if invalid_run >= 5:
                info["termination"] = 4
                break
            mu *= 10
            reuse = False
            continue
        invalid_run = 0
        cand = plus(x, ids, d)
        Hc, gc, cc = G.window_normal_equations(cand, ids)
        xv, cv = gather(x, ids), gather(cand, ids)
        step_norm = mp.sqrt(sum((a_ - b_) ** 2 for a_, b_ in zip(xv, cv)))
        x_norm = mp.sqrt(sum(a_ * a_ for a_ in xv))
        rec = {"iteration": it, "case": case, "radius": mp.nstr(radius, 8), "mu": mp.nstr(mu, 4), "cost_before": mp.nstr(cost, 30), "candidate_cost": mp.nstr(cc, 30),
               "model_cost_change": mp.nstr(model, 12)}
        info["trace"].append(rec)
        if step_norm <= mpf("1e-8") * (x_norm + mpf("1e-8")):
            info["termination"] = 2
            rec["accepted"] = False
            break
        if abs(cost - cc) <= mpf("1e-6") * cost:
            info["termination"] = 1
            rec["accepted"] = False
            break
        rel = (cost - cc) / model
        rec["relative_decrease"] = mp.nstr(rel, 8)
        if rel > mpf("1e-3"):
            x, H, g, cost = cand, Hc, gc, cc
            gmax = max(abs(g[c]) for c in range(n))
            if rel < mpf("0.25"):
                radius *= mpf("0.5")
            if rel > mpf("0.75"):
                radius = max(radius, 3 * sn)
            mu = max(mpf("1e-8"), 2 * mu / 10)
            reuse = False
            last_ok = True
            info["successful_steps"] += 1
            rec["accepted"] = True
        else:
            radius *= mpf("0.5")
            reuse = True
            last_ok = False
            rec["accepted"] = False
        log("      iteration %d: %-12s cost %s -> %s  %s" % (it, case, mp.nstr(mpf(rec["cost_before"]), 12), mp.nstr(cc, 12), "accepted" if rec["accepted"] else "REJECTED"))
    info["final_cost"] = mp.nstr(cost, 30)
    info["radius"] = mp.nstr(radius, 12)
    return x, info


CASES = ("ref_window_free_ex_td", "ref_window_with_prior", "ref_window_wheel", "ref_window_wheel_free_ix_td", "ref_window_gnss")


def main():
    import oracle_py as O
    from test_golden import load_ref_window
    names = sys.argv[1:] or CASES
    for name in names:
        t0 = time.time()
        w, fx, _, _ = load_ref_window(name)
        ids = fx["ids"]
        print("%s: %d columns" % (name, len(ids)), flush=True)
        x, info = solve(w, ids, 8)
        a = w.copy()
        so = O.ba_solve(a, 8)
        P, Pe = a["para_Pose"].reshape(-1, 7), x["para_Pose"].reshape(-1, 7)
        dq = 2 * min(np.abs(P[:, 3:] - Pe[:, 3:]).max(), np.abs(P[:, 3:] + Pe[:, 3:]).max())
        print("   60 digits: %d iterations, %d successful, termination %d, cost %s -> %s" % (info["iterations"], info["successful_steps"], info["termination"],
                                                                                           mp.nstr(mpf(info["initial_cost"]), 12), mp.nstr(mpf(info["final_cost"]), 12)))
        print("   oracle   : %d iterations, %d successful, termination %d, cost %.12g -> %.12g" % (so["iterations"], so["successful_steps"], so["termination"], so["initial_cost"], so["final_cost"]))
        print("   oracle vs 60 digits after the solve: position %.2e m, rotation %.2e rad, speed / bias %.2e, inverse depths %.2e, other blocks %.2e   (%.0f s)"
              % (np.abs(P[:, :3] - Pe[:, :3]).max(), dq, np.abs(a["para_SpeedBias"] - x["para_SpeedBias"]).max(), np.abs(a["para_Feature"] - x["para_Feature"]).max(),
                 max(np.abs(np.asarray(a[k]) - np.asarray(x[k])).max() for k in gw.STATE_KEYS if k not in ("para_Pose", "para_SpeedBias", "para_Feature") and len(a[k])), time.time() - t0), flush=True)
        out = {"about": "state of the window tests/golden/%s.json.gz after the reference's solve (DENSE_SCHUR, DOGLEG, 8 iterations) with every iterate's H, g, cost from the "
                        "reference's formulas at 60 digits and Ceres' trust-region loop restated at 60 digits (tests/golden/make_ref_solve_golden.py)" % name,
               "window": name, "state": {k: np.asarray(x[k]).reshape(-1).tolist() for k in gw.STATE_KEYS}, "summary": info}
        path = os.path.join(HERE, name.replace("ref_window_", "ref_solve_") + ".json.gz")
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(json.dumps(out).encode())
        print("   wrote", path, os.path.getsize(path), "bytes", flush=True)


if __name__ == "__main__":
    main()
