"""Which side is off when the two EPnP implementations disagree?  ("what's weak" 2(ii): SfM initialisation at W = 20 agrees to 2e-3 only.)

The recording of tests/test_init_sfm_host.py::test_ransac_hypotheses_of_near_degenerate_subsets_are_noise_driven is replayed up to initialStructure; the
correspondences solveRelativeRT_PNP hands to cv::solvePnPRansac are captured, the RANSAC subsets are drawn with the same generator, and every 5-point
subset goes through three implementations of the SAME algorithm (epnp.cpp as oracle/init_oracle.py restates it):
  * oracle/init_oracle.py  (numpy / LAPACK, double)
  * the library            (csrc/gf_init_sfm.hpp: Jacobi eigen-solver + Householder QR, double; debug op "epnp")
  * this file              (mpmath, 60 digits, same steps and the same canonical choices)
and the two double results are measured against the 60-digit one.  CPU only:  python scripts/epnp_mpmath_check.py [--window 20] [--seed 5]"""
import argparse
import os
import sys

import numpy as np
import mpmath as mp

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, "ground-fusion_amd"))
sys.path.insert(0, os.path.join(R, "oracle"))
sys.path.insert(0, os.path.join(R, "tests"))
import gfamd  # noqa: E402
import init_oracle as IO  # noqa: E402
import estimator_oracle as EO  # noqa: E402
import synth_stream as SS  # noqa: E402

mp.mp.dps = 60
_PAIRS = IO._PAIRS


def mp_sym_eig(A):
    """ascending eigenvalues, eigenvectors as columns, canonical sign (largest-magnitude entry positive)"""
    n = A.rows
    E, Q = mp.eigsy((A + A.T) * mp.mpf(1) / 2)
    order = sorted(range(n), key=lambda i: E[i])
    cols = []
    for i in order:
        v = [Q[r, i] for r in range(n)]
        k = max(range(n), key=lambda r: abs(v[r]))
        if v[k] < 0:
            v = [-x for x in v]
        cols.append(v)
    return [E[i] for i in order], cols


def mp_canonical_basis(cols, n):
    """the basis IO.canonical_subspace_basis picks inside span(cols)"""
    k = len(cols)
    out = []
    for j in range(n):
        v = [sum(c[r] * c[j] for c in cols) for r in range(n)]
        for u in out:
            d = sum(a * b for a, b in zip(u, v))
            v = [a - b * d for a, b in zip(v, u)]
        nv = mp.sqrt(sum(a * a for a in v))
        if nv > mp.mpf("0.1"):
            out.append([a / nv for a in v])
            if len(out) == k:
                break
    assert len(out) == k
    return out


def mp_lsq(A, b):
    """least squares (full column rank assumed)"""
    return mp.qr_solve(A, b)[0]


def mp_rodrigues_inv(Rm):
    r = [Rm[2, 1] - Rm[1, 2], Rm[0, 2] - Rm[2, 0], Rm[1, 0] - Rm[0, 1]]
    s = mp.sqrt(sum(x * x for x in r) / 4)
    c = min(mp.mpf(1), max(mp.mpf(-1), (Rm[0, 0] + Rm[1, 1] + Rm[2, 2] - 1) / 2))
    th = mp.acos(c)
    if s < mp.mpf("1e-5"):
        return [mp.mpf(0)] * 3 if c > 0 else None   # the half-turn branch does not occur here
    return [x * (th / (2 * s)) for x in r]


def mp_epnp(X, uv):
    """IO.epnp step by step at mp.dps digits; X (n x 3), uv (n x 2) as floats (the float32-rounded inputs, exact in mpf)"""
    n = len(X)
    Xm = [[mp.mpf(float(v)) for v in p] for p in X]
    um = [[mp.mpf(float(v)) for v in p] for p in uv]
    c0 = [sum(p[a] for p in Xm) / n for a in range(3)]
    C = mp.matrix(3, 3)
    for p in Xm:
        d = [p[a] - c0[a] for a in range(3)]
        for a in range(3):
            for b in range(3):
                C[a, b] += d[a] * d[b]
    w, V = mp_sym_eig(C)
    cws = [c0] + [[c0[a] + mp.sqrt(max(w[2 - i], mp.mpf(0)) / n) * V[2 - i][a] for a in range(3)] for i in range(3)]
    CC = mp.matrix(3, 3)
    for i in (1, 2, 3):
        for a in range(3):
            CC[a, i - 1] = cws[i][a] - cws[0][a]
    alphas = []
    for p in Xm:
        a123 = mp.lu_solve(CC, mp.matrix([p[a] - cws[0][a] for a in range(3)]))
        alphas.append([1 - (a123[0] + a123[1] + a123[2]), a123[0], a123[1], a123[2]])
    M = mp.matrix(2 * n, 12)
    for r in range(n):
        for j in range(4):
            M[2 * r, 3 * j] = alphas[r][j]
            M[2 * r, 3 * j + 2] = alphas[r][j] * (0 - um[r][0])
            M[2 * r + 1, 3 * j + 1] = alphas[r][j]
            M[2 * r + 1, 3 * j + 2] = alphas[r][j] * (0 - um[r][1])
    w12, V12 = mp_sym_eig(M.T * M)
    k = max(1, sum(1 for x in w12 if x < mp.mpf("1e-9") * w12[-1]))
    vs = [V12[i] for i in range(4)]
    if k >= 2:
        k = min(k, 4)
        Bc = mp_canonical_basis(V12[:k], 12)
        for i in range(k):
            vs[i] = Bc[i]
    dv = [[[vs[i][3 * a + q] - vs[i][3 * b + q] for q in range(3)] for (a, b) in _PAIRS] for i in range(4)]
    dot = lambda u, v: sum(x * y for x, y in zip(u, v))  # noqa: E731
    L = mp.matrix(6, 10)
    for p in range(6):
        d = [dv[i][p] for i in range(4)]
        row = [dot(d[0], d[0]), 2 * dot(d[0], d[1]), dot(d[1], d[1]), 2 * dot(d[0], d[2]), 2 * dot(d[1], d[2]), dot(d[2], d[2]), 2 * dot(d[0], d[3]),
               2 * dot(d[1], d[3]), 2 * dot(d[2], d[3]), dot(d[3], d[3])]
        for q in range(10):
            L[p, q] = row[q]
    rho = mp.matrix([dot([cws[a][q] - cws[b][q] for q in range(3)], [cws[a][q] - cws[b][q] for q in range(3)]) for a, b in _PAIRS])

    def cols(idx):
        A = mp.matrix(6, len(idx))
        for r in range(6):
            for q, c in enumerate(idx):
                A[r, q] = L[r, c]
        return A

    def approx(which):
        be = [mp.mpf(0)] * 4
        if which == 1:
            b4 = mp_lsq(cols([0, 1, 3, 6]), rho)
            if b4[0] < 0:
                be[0] = mp.sqrt(-b4[0]); be[1:] = [-b4[i] / be[0] for i in (1, 2, 3)]
            else:
                be[0] = mp.sqrt(b4[0]); be[1:] = [b4[i] / be[0] for i in (1, 2, 3)]
        else:
            bb = mp_lsq(cols([0, 1, 2]) if which == 2 else cols([0, 1, 2, 3, 4]), rho)
            if bb[0] < 0:
                be[0] = mp.sqrt(-bb[0]); be[1] = mp.sqrt(-bb[2]) if bb[2] < 0 else mp.mpf(0)
            else:
                be[0] = mp.sqrt(bb[0]); be[1] = mp.sqrt(bb[2]) if bb[2] > 0 else mp.mpf(0)
            if bb[1] < 0:
                be[0] = -be[0]
            if which == 3:
                be[2] = bb[3] / be[0]
        return be

    def gauss_newton(be):
        be = list(be)
        for _ in range(5):
            A = mp.matrix(6, 4)
            b = mp.matrix(6, 1)
            for r in range(6):
                l = [L[r, q] for q in range(10)]
                A[r, 0] = 2 * l[0] * be[0] + l[1] * be[1] + l[3] * be[2] + l[6] * be[3]
                A[r, 1] = l[1] * be[0] + 2 * l[2] * be[1] + l[4] * be[2] + l[7] * be[3]
                A[r, 2] = l[3] * be[0] + l[4] * be[1] + 2 * l[5] * be[2] + l[8] * be[3]
                A[r, 3] = l[6] * be[0] + l[7] * be[1] + l[8] * be[2] + 2 * l[9] * be[3]
                b[r] = rho[r] - (l[0] * be[0] * be[0] + l[1] * be[0] * be[1] + l[2] * be[1] * be[1] + l[3] * be[0] * be[2] + l[4] * be[1] * be[2] + l[5] * be[2] * be[2]
                                 + l[6] * be[0] * be[3] + l[7] * be[1] * be[3] + l[8] * be[2] * be[3] + l[9] * be[3] * be[3])
            dx = mp_lsq(A, b)
            be = [be[i] + dx[i] for i in range(4)]
        return be

    def pose(be):
        ccs = [[sum(be[i] * vs[i][3 * j + q] for i in range(4)) for q in range(3)] for j in range(4)]
        pcs = [[sum(alphas[r][j] * ccs[j][q] for j in range(4)) for q in range(3)] for r in range(n)]
        if pcs[0][2] < 0:
            ccs = [[-x for x in c] for c in ccs]
            pcs = [[-x for x in p] for p in pcs]
        pc0 = [sum(p[q] for p in pcs) / n for q in range(3)]
        pw0 = [sum(p[q] for p in Xm) / n for q in range(3)]
        ABt = mp.matrix(3, 3)
        for r in range(n):
            for a in range(3):
                for b in range(3):
                    ABt[a, b] += (pcs[r][a] - pc0[a]) * (Xm[r][b] - pw0[b])
        U, _, Vt = mp.svd_r(ABt)
        Rm = U * Vt
        if mp.det(Rm) < 0:
            for q in range(3):
                Rm[2, q] = -Rm[2, q]
        t = [pc0[a] - sum(Rm[a, b] * pw0[b] for b in range(3)) for a in range(3)]
        err = mp.mpf(0)
        for r in range(n):
            P = [sum(Rm[a, b] * Xm[r][b] for b in range(3)) + t[a] for a in range(3)]
            err += mp.sqrt((P[0] / P[2] - um[r][0]) ** 2 + (P[1] / P[2] - um[r][1]) ** 2)
        return err / n, Rm, t

    best = None
    for which in (1, 2, 3):
        try:
            r = pose(gauss_newton(approx(which)))
        except (ZeroDivisionError, ValueError):
            continue
        if best is None or r[0] < best[0]:
            best = r + (which,)
    rv = mp_rodrigues_inv(best[1])
    cond = w12[-1] / max(w12[k], mp.mpf("1e-300")) if k < 12 else mp.inf   # spread of M^T M above its null space
    return np.array([float(x) for x in rv]), np.array([float(x) for x in best[2]]), int(best[3]), k, float(cond)


def capture_correspondences(seed, window):
    """the (X, uv) solve_pnp_ransac receives at the SfM initialisation of the recording, captured from the oracle pipeline"""
    st = SS.Stream(seed, t_still=0.0, t_move=2.0 if window <= 10 else 3.2, v_max=0.5, v_start=0.5, yaw_turn=0.4)
    eo = EO.Estimator(dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, window_size=window))
    eo.optimization = lambda: None
    eo.slideWindow = lambda: None
    got = {}
    orig = IO.solve_pnp_ransac

    def spy(X, uv, *a, **k):
        got.setdefault("X", np.array(X, float)); got.setdefault("uv", np.array(uv, float))
        return orig(X, uv, *a, **k)
    IO.solve_pnp_ransac = spy
    try:
        tp, k = -1.0, 0
        while eo.solver_flag == EO.INITIAL and k < len(st.cam_t):
            tp = st.feed(eo, k, tp)
            eo.inputFeature(float(st.cam_t[k]), st.feature_frame(k))
            k += 3
    finally:
        IO.solve_pnp_ransac = orig
    return got["X"], got["uv"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--window", type=int, default=20)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--subsets", type=int, default=40)
    args = ap.parse_args()
    X, uv = capture_correspondences(args.seed, args.window)
    X, uv = IO.f32(X), IO.f32(uv)
    n = len(X)
    print("W = %d, seed %d: %d correspondences handed to solvePnPRansac" % (args.window, args.seed, n))
    est = gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, window_size=args.window))
    rng = IO.CvRNG()
    rows = []
    for it in range(args.subsets):
        idx = []
        while len(idx) < 5:
            i = rng.uniform(0, n)
            if i not in idx:
                idx.append(i)
        with np.errstate(all="ignore"):
            mo = IO.epnp(X[idx], uv[idx])
        out = est.debug("epnp", np.concatenate([[5.0], np.concatenate([X[idx], uv[idx]], axis=1).ravel()]))
        ml = (out[1:4], out[4:7]) if out[0] else None
        rv, tv, which, k, cond = mp_epnp(X[idx], uv[idx])
        eo_ = None if mo is None else max(np.abs(mo[0] - rv).max(), np.abs(mo[1] - tv).max())
        el_ = None if ml is None else max(np.abs(ml[0] - rv).max(), np.abs(ml[1] - tv).max())
        eol = None if (mo is None or ml is None) else max(np.abs(mo[0] - ml[0]).max(), np.abs(mo[1] - ml[1]).max())
        rows.append((it, idx, k, cond, which, eo_, el_, eol))
        print("subset %2d %-18s null space %d, cond(M^T M) %.1e, winner %d: |oracle - mp| %s  |library - mp| %s  |oracle - library| %s"
              % (it, " ".join(map(str, idx)), k, cond, which, "%.1e" % eo_ if eo_ is not None else "fail", "%.1e" % el_ if el_ is not None else "fail",
                 "%.1e" % eol if eol is not None else "-"))
    ok = [r for r in rows if r[5] is not None and r[6] is not None]
    eo_ = np.array([r[5] for r in ok]); el_ = np.array([r[6] for r in ok]); eol = np.array([r[7] for r in ok])
    print("\n%d subsets, %d solved by both double implementations" % (len(rows), len(ok)))
    print("oracle  vs 60 digits: median %.1e, max %.1e;  closer than the library on %d subsets" % (np.median(eo_), eo_.max(), int((eo_ < el_).sum())))
    print("library vs 60 digits: median %.1e, max %.1e;  closer than the oracle  on %d subsets" % (np.median(el_), el_.max(), int((el_ < eo_).sum())))
    print("oracle vs library:    median %.1e, max %.1e" % (np.median(eol), eol.max()))
    big = [r for r in ok if r[7] > 1e-6]
    print("subsets on which the two differ by more than 1e-6: %d; on those, oracle / library error vs 60 digits: %s"
          % (len(big), ", ".join("%.0e / %.0e" % (r[5], r[6]) for r in big[:12])))
    est.close()


if __name__ == "__main__":
    main()
