"""Committed golden vectors (tests/golden/oracle_golden.npz, made by tests/golden/make_golden.py): the oracle must reproduce them on the CPU,
the HIP path must meet them on the GPU.  They pin this repo's oracle, not the reference (see the generator's header)."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = dict(np.load(os.path.join(HERE, "golden", "oracle_golden.npz")))


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_oracle_reproduces_golden(oracle):
    now = _gen().compute()
    assert sorted(now) == sorted(GOLD)
    for k, v in GOLD.items():
        if k.startswith(("trk_", "lk_", "corners", "marg_ids", "init_ransac_inliers", "init_sfm_l_points")):
            assert np.array_equal(now[k], v), k                                   # integer / float front-end results: bit-exact
        else:
            np.testing.assert_allclose(now[k], v, rtol=1e-9, atol=1e-9, err_msg=k)  # FP64 back end: libm differences only


def test_library_initialisation_meets_golden():
    """initialisation while moving is host code in the library (csrc/gf_init_sfm.hpp): it meets the committed vectors without a GPU"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "ground-fusion_amd"))
    import gfamd as gf
    import synth_stream as SS
    rng = np.random.default_rng(31)
    X = np.stack([rng.uniform(-2, 2, 64), rng.uniform(-1.5, 1.5, 64), rng.uniform(2.5, 7.0, 64)], axis=1)
    rvec, tvec = np.array([0.05, -0.08, 0.03]), np.array([0.06, -0.02, -0.45])
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    import init_oracle as IO
    import estimator_oracle as EO
    uv = IO.project(rvec, tvec, X) + rng.normal(0, 0.2 / 460.0, (64, 2))
    uv[::4] += 0.08
    inl = set(int(i) for i in GOLD["init_ransac_inliers"])
    assert inl <= set(range(64)) - set(range(0, 64, 4)) and len(inl) >= 40                    # the gross mismatches are out (and a few points the best 5-point model misses by a pixel)
    kw = dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1)
    ep = gf.SlidingWindowEstimator(gf.default_estimator_cfg(**kw))
    d2 = np.full(64, 3.5)
    out = ep.debug("solveRelativeRT_PNP", np.concatenate([X, uv * d2[:, None], d2[:, None]], axis=1).reshape(-1))
    rota = EO.ypr2R_deg_free(GOLD["init_ransac_rvec"])
    np.testing.assert_allclose(out[1:10].reshape(3, 3), rota.T, atol=1e-9)
    np.testing.assert_allclose(out[10:13], -rota.T @ GOLD["init_ransac_tvec"], atol=1e-9)
    st = SS.Stream(5, t_still=0.0, t_move=2.0, v_max=0.5, v_start=0.5, yaw_turn=0.4)
    ep.debug("skip_solve", [1.0])
    tp, k = -1.0, 0
    while ep.state()["solver_flag"] == 0 and k < 45:
        tp = st.feed(ep, k, tp)
        ep.inputFeature(float(st.cam_t[k]), st.feature_frame(k))
        k += 3
    s, info = ep.state(), ep.debug("init_info")
    assert [int(info[0]), int(info[1])] == list(GOLD["init_sfm_l_points"]) and abs(info[2] - GOLD["init_sfm_s"][0]) < 1e-9
    np.testing.assert_allclose(info[3:6], GOLD["init_sfm_g_c0"], atol=1e-9)
    for key in ("Ps", "Rs", "Vs"):
        np.testing.assert_allclose(s[key], GOLD["init_sfm_" + key], atol=1e-9, err_msg=key)
    ep.close()


@pytest.mark.gpu
def test_hip_front_end_meets_golden(gf):
    import synth
    frames = synth.tracker_sequence(4242, 3)
    depth = np.full(frames[0].shape, 1500, np.uint16)
    trk = gf.FeatureTracker(gf.default_cfg())
    for k, f in enumerate(frames):
        ids, obs = trk.trackImage(0.0666 * k, f, depth)
        assert np.array_equal(ids, GOLD["trk_ids_%d" % k]) and np.array_equal(obs.view(np.uint64), GOLD["trk_obs_%d" % k].view(np.uint64)), k
    trk.close()
    pts = np.array([[100.25, 80.5], [320.0, 240.0], [500.75, 400.125]], np.float32)
    nxt, st, _ = gf.lk_track(frames[0], frames[1], pts)
    assert np.array_equal(nxt.view(np.uint32), GOLD["lk_next"].view(np.uint32)) and np.array_equal(st, GOLD["lk_status"])
    assert np.array_equal(gf.good_features(frames[0], 40, min_dist=30), GOLD["corners"])


@pytest.mark.gpu
def test_hip_back_end_meets_golden(gf, oracle):
    import synth_window as SW
    w = SW.make_window(77, oracle, max_features=40, n_landmarks=60)
    est = gf.Estimator(max_features=40, max_visual=400)
    s = est.solve([w], 4)[0]
    assert [s["iterations"], s["successful_steps"], s["termination"]] == [int(v) for v in GOLD["ba_summary"][:3]]
    assert np.abs(w["para_Pose"] - GOLD["ba_pose"]).max() < 1e-6 and np.abs(w["para_SpeedBias"] - GOLD["ba_speedbias"]).max() < 1e-6
    p = est.marginalize([w], 0)[0]
    J = p["J"].reshape(p["n"], p["n"])
    assert np.array_equal(p["block_id"], GOLD["marg_ids"])
    assert np.abs(J.T @ J - GOLD["marg_JtJ"]).max() <= 1e-6 * np.abs(GOLD["marg_JtJ"]).max()   # the linearisation points already differ by the solver bar
    est.close()
    g = SW.make_window(78, oracle, max_features=30, n_landmarks=45, gnss=True, anchor=True)
    eg = gf.Estimator(max_features=30, max_visual=300, max_gnss=12 * 11)
    sg = eg.solve([g], 4)[0]
    assert [sg["iterations"], sg["successful_steps"]] == [int(v) for v in GOLD["gnss_summary"][:2]]
    assert np.abs(g["para_Pose"] - GOLD["gnss_pose"]).max() < 1e-6 and np.abs(g["para_rcv_dt"] - GOLD["gnss_rcv_dt"]).max() < 1e-6
    assert np.abs(g["para_anc_ecef"] - GOLD["gnss_anc"]).max() < 1e-6
    eg.close()


# ---------------------------------------------------------------- fixtures NOT made by this repo's oracle: the reference's formulas at 60 digits
def load_ref_window(name):
    """tests/golden/ref_*.json.gz (tests/golden/make_ref_golden.py: ProjectionTwoFrameOneCamFactor, IMUFactor, WheelFactor, MarginalizationFactor, HuberLoss + Corrector transcribed
    from the reference / Ceres into mpmath, independent of oracle/ and of the product): the window and its normal equations"""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "ground-fusion_amd"))
    import gfwindow as gw
    import gzip
    with gzip.open(os.path.join(HERE, "golden", name + ".json.gz"), "rt") as f:
        fx = json.load(f)
    w = gw.Window()
    for k, v in fx["window"].items():
        w[k] = np.array(v) if isinstance(v, list) else v
    w.finalize()
    n = len(fx["ids"])
    H = np.zeros((n, n))
    H[np.tril_indices(n)] = fx["H_lower"]
    H = H + np.tril(H, -1).T
    return w, fx, H, np.array(fx["g"])


def check_against_ref(lin, fx, H, g, tol=1e-11):
    assert [int(x) for x in lin["ids"]] == fx["ids"] and lin["n_f"] == fx["n_f"] and lin["n_e"] == fx["n_e"]
    assert abs(lin["cost"] - fx["cost"]) <= tol * fx["cost"]
    hs = np.sqrt(np.outer(np.abs(np.diag(H)), np.abs(np.diag(H)))) + 1e-300
    dev_h, dev_g = float(np.abs((lin["H"] - H) / hs).max()), float(np.abs(lin["g"] - g).max() / np.abs(g).max())
    assert dev_h < tol and dev_g < tol, (dev_h, dev_g)
    return dev_h, dev_g


REF_WINDOWS = ["ref_window_free_ex_td", "ref_window_with_prior", "ref_window_wheel", "ref_window_wheel_free_ix_td", "ref_window_gnss"]


@pytest.mark.parametrize("name", REF_WINDOWS)
def test_oracle_meets_the_reference_formulas_at_60_digits(oracle, name):
    """rows F1 (visual), F2 (IMU), F3 (wheel), F5 (prior), L1 (Huber corrector) of SURVEY.md 8a: the oracle's H, g, cost of a whole small window against numbers it did not produce"""
    w, fx, H, g = load_ref_window(name)
    dev = check_against_ref(oracle.ba_linearize(w), fx, H, g)
    print(name, "oracle vs reference formulas at 60 digits: H scaled %.1e, g %.1e" % dev)


REF_MARG = ["ref_marg_old_first_window", "ref_marg_old_with_prior", "ref_marg_second_new", "ref_marg_old_gnss"]


def load_ref_marg(name):
    """tests/golden/ref_marg_*.json.gz: the marginalisation prior by the reference's route (eigen pseudo-inverse of the dropped block, eigen square root of the kept
    system, eps 1e-8) at 60 digits: the window, the mode, and J^T J / J^T r over the kept blocks"""
    import gzip
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "ground-fusion_amd"))
    import gfwindow as gw
    with gzip.open(os.path.join(HERE, "golden", name + ".json.gz"), "rt") as f:
        fx = json.load(f)
    w = gw.Window()
    for k, v in fx["window"].items():
        w[k] = np.array(v) if isinstance(v, list) else v
    w.finalize()
    n = fx["n"]
    A = np.zeros((n, n))
    A[np.tril_indices(n)] = fx["JtJ_lower"]
    A = A + np.tril(A, -1).T
    # the ids the prior carries are those of the NEXT window (addr_shift, estimator.cpp:3471-3500 / :3583-3626)
    W, ids = int(w["W"]), []
    for b in fx["kept"]:
        kind, i = b // 4096, b % 4096
        if kind in (gw.POSE, gw.SPEEDBIAS, gw.RCV_DDT):
            i = i - 1 if fx["mode"] == 0 else (W - 1 if i == W else i)
        elif kind == gw.RCV_DT:   # four clock biases per frame
            i = i - 4 if fx["mode"] == 0 else (i - 4 if i // 4 == W else i)
        ids.append(kind * 4096 + i)
    return w, fx, A, np.array(fx["Jtr"]), ids


def check_prior_against_ref(p, fx, A, b, ids, tol_a=2e-6, tol_b=2e-8):
    if fx["window"].get("gnss_enabled"):   # kept eigenvalues 1.9e-11, 2.6e-10 | 5.8e-8, 2.1e-7 on either side of the 1e-8 cut, ECEF magnitudes: the reference's own route in
        tol_a, tol_b = 2e-5, 5e-7          # double precision (= the oracle) sits 4e-6 / 6e-8 from its 60-digit evaluation
    n = fx["n"]
    assert p["n"] == n and p["m"] == fx["m"] and [int(x) for x in p["block_id"]] == ids
    J = p["J"].reshape(n, n)
    sc = np.sqrt(np.maximum(np.diag(A), 1e-300))
    dev_a = float(np.abs((J.T @ J - A) / np.outer(sc, sc)).max())
    dev_b = float(np.abs(J.T @ p["r"] - b).max() / np.abs(b).max())
    assert dev_a < tol_a and dev_b < tol_b, (dev_a, dev_b)
    return dev_a, dev_b


@pytest.mark.parametrize("name", REF_MARG)
def test_oracle_marginalisation_meets_the_reference_route_at_60_digits(oracle, name):
    """rows M1 / M2 / M3: what the next window's solver sees of the prior (J^T J, J^T r -- independent of the eigenvector basis) against the reference's algorithm evaluated
    with 60 digits.  The bars are the double-precision noise of that algorithm itself (its eigen pseudo-inverse of the 25-column dropped block carries ~1e-7 of the
    kept system's scale), not a property of the oracle: observed 3e-8 .. 6e-7 (J^T J, scaled by its diagonal) and 2e-10 .. 2e-9 (J^T r)."""
    w, fx, A, b, ids = load_ref_marg(name)
    dev = check_prior_against_ref(oracle.ba_marginalize(w, fx["mode"]), fx, A, b, ids)
    print(name, "oracle vs reference route at 60 digits: J^T J scaled %.1e, J^T r %.1e" % dev)


def check_first_step(solve_one, name="ref_window_free_ex_td", tol=1e-7):
    """the state after ONE trust-region iteration against the exact first step stored in the fixture (make_ref_golden.first_step_golden: Jacobi scaling, the dogleg's
    regularised Gauss-Newton step with mu = 1e-8 inside the initial radius, x (+) delta; LU on the full system at 60 digits -- no Schur complement, no Cholesky)"""
    w, fx, _, _ = load_ref_window(name)
    a = w.copy()
    s = solve_one(a)
    assert s["successful_steps"] == 1 and s["iterations"] == 1
    dev = {k: float(np.abs(np.array(v) - a[k]).max()) for k, v in fx["first_step"].items()}
    assert max(dev.values()) < tol, dev
    return dev


def test_oracle_first_step_meets_the_exact_step(oracle):
    """row S1 (trust region / dogleg / Schur / Cholesky), first iteration: observed 1e-9 m on a system of condition 3e8"""
    print("oracle vs exact first step:", check_first_step(lambda a: oracle.ba_solve(a, 1)))


REF_SOLVES = ["ref_solve_free_ex_td", "ref_solve_with_prior", "ref_solve_wheel", "ref_solve_wheel_free_ix_td", "ref_solve_gnss"]
# bars of the whole solve against the loop run at 60 digits: 1e-6 m / 1e-6 rad (2 x quaternion components) / 1e-6 in its own unit for every other block -- BASELINE.json's bar,
# for every block of every window; an entry here would widen one (none needed: the oracle sits at 2e-10 ... 6e-10 m, <= 4e-8 on the free camera extrinsic)
SOLVE_BARS = {}


def check_solve(solve, name):
    """the state after the reference's whole solve (8 iterations of DOGLEG on DENSE_SCHUR) against tests/golden/ref_solve_*.json.gz: every iterate's H, g and cost from the
    reference's formulas and Ceres' loop, all at 60 digits (tests/golden/make_ref_solve_golden.py).  Also: the same number of iterations and of accepted steps, the same
    termination, the final cost to 1e-8."""
    import gzip
    import json
    with gzip.open(os.path.join(HERE, "golden", name + ".json.gz"), "rt") as f:
        fx = json.load(f)
    w = load_ref_window(fx["window"])[0]
    a = w.copy()
    s = solve(a)
    e, ex = fx["summary"], fx["state"]
    assert [s["iterations"], s["successful_steps"], s["termination"]] == [e["iterations"], e["successful_steps"], e["termination"]], (s, e["iterations"], e["successful_steps"], e["termination"])
    # the cost: 1e-8 relative (a sum of squares of pseudorange residuals formed from 2e7 m ranges carries ~1e-9 of its value as rounding in double precision)
    assert abs(s["final_cost"] - float(e["final_cost"])) <= 1e-8 * float(e["final_cost"]), (s["final_cost"], e["final_cost"])
    bars = dict(dict(pos=1e-6, rot=1e-6, other=1e-6), **SOLVE_BARS.get(name, {}))
    dev = state_dev(a, ex)
    assert dev["para_Pose.p"] < bars["pos"] and dev["para_Pose.q"] < bars["rot"], dev
    assert max(v for k, v in dev.items() if not k.startswith("para_Pose.")) < bars["other"], dev
    return {k: "%.1e" % v for k, v in dev.items() if v > 0}


def state_dev(a, ex):
    """largest deviation per state block: positions [m], 2 x quaternion components [rad], everything else in its own unit"""
    dev = {}
    for k in ("para_Pose", "para_Ex_Pose", "para_Ex_Pose_wheel"):
        P, E = np.asarray(a[k]).reshape(-1, 7), np.asarray(ex[k]).reshape(-1, 7)
        dev[k + ".p"] = float(np.abs(P[:, :3] - E[:, :3]).max())
        dev[k + ".q"] = 2 * float(min(np.abs(P[:, 3:] - E[:, 3:]).max(), np.abs(P[:, 3:] + E[:, 3:]).max()))
    for k in ex:
        if k not in ("para_Pose", "para_Ex_Pose", "para_Ex_Pose_wheel") and len(ex[k]):
            dev[k] = float(np.abs(np.asarray(a[k]) - np.asarray(ex[k])).max())
    return dev


@pytest.mark.parametrize("name", REF_SOLVES)
def test_oracle_solve_meets_the_loop_at_60_digits(oracle, name):
    """row S1, the whole trust-region loop: the CPU oracle's double-precision solve (Schur complement, Cholesky) against the loop at 60 digits (LU on the full system)"""
    print(name, "oracle vs the loop at 60 digits:", check_solve(lambda a: oracle.ba_solve(a, 8), name))


# ---------------------------------------------------------------- two windows: solve -> MARGIN_OLD -> solve, every number at 60 digits
# Bars of the END state of window 2, per block; what is not listed is held to BASELINE.json's 1e-6 (m / rad / the block's own unit).  Listed are the directions the chain
# observes weakly, where the reference's own double-precision route (eigen pseudo-inverse + eigen square root with a 1e-8 cut) is not defined to 1e-6 -- measured, not
# assumed: the bar is what the CPU oracle, a line-by-line restatement of that route, needs against the exact chain (observed value in brackets):
#   * the wheel extrinsic's translation, whose vertical component a near-planar drive does not observe [oracle 1.5e-6 m without GNSS, 3.5e-4 m with];
#   * with GNSS: the window's absolute position, the anchor and the receiver clocks, which hang on kept eigenvalues of the prior next to the cut (3.4e-8, 1.6e-7, 6.1e-7 in
#     the fixture's `eigenvalues_kept`) [3.5e-5 m, 2.7e-5 m, 3.9e-6 m].  The window's SHAPE and its rotations meet 1e-6 [1.1e-7 m, 9.4e-8 rad].
CHAIN_BARS = {"ref_chain_wheel": {"para_Ex_Pose_wheel.p": 1e-5},
              "ref_chain_gnss": {"para_Pose.p": 1e-4, "para_Ex_Pose_wheel.p": 2e-3, "para_Ex_Pose_wheel.q": 2e-5, "para_anc_ecef": 1e-4, "para_rcv_dt": 1e-4}}


# The library's marginalisation (block elimination, rank-revealing Cholesky, least-squares right-hand side) is not the eigen route and does not inherit its noise: measured
# against the exact chain it sits at 9e-8 m / 8e-10 rad where the oracle sits at 3.5e-5 m, so its test keeps 1e-6 on every block but the unobserved vertical lever arm [2.0e-6 m].
CHAIN_BARS_HIP = {"ref_chain_wheel": {}, "ref_chain_gnss": {"para_Ex_Pose_wheel.p": 1e-5}}


def check_chain(solve, marginalize, name, bars=None):
    """an implementation's own chain (its solve of window 1, its MARGIN_OLD prior, its solve of window 2 with that prior) against tests/golden/ref_chain_*.json.gz"""
    import gzip
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "ground-fusion_amd"))
    import gfwindow as gw
    with gzip.open(os.path.join(HERE, "golden", name + ".json.gz"), "rt") as f:
        fx = json.load(f)
    wins = []
    for key in ("window_1", "window_2"):
        w = gw.Window()
        for k, v in fx[key].items():
            w[k] = np.array(v) if isinstance(v, list) else v
        wins.append(w.finalize())
    bars = CHAIN_BARS[name] if bars is None else bars
    a1 = wins[0].copy()
    s1 = solve(a1)
    assert [s1["iterations"], s1["successful_steps"], s1["termination"]] == [fx["summary_1"][k] for k in ("iterations", "successful_steps", "termination")]
    d1 = state_dev(a1, fx["state_1"])
    assert max(d1.values()) < 1e-6, d1
    p = marginalize(a1)
    assert [int(b) for b in p["block_id"]] == fx["prior_block_id"] and (int(p["m"]), int(p["n"])) == (fx["prior_m"], fx["prior_n"])
    a2 = wins[1].copy().set_prior(p)
    s2 = solve(a2)
    assert [s2["iterations"], s2["successful_steps"], s2["termination"]] == [fx["summary_2"][k] for k in ("iterations", "successful_steps", "termination")]
    d2 = state_dev(a2, fx["state_2"])
    P, E = a2["para_Pose"].reshape(-1, 7), np.asarray(fx["state_2"]["para_Pose"]).reshape(-1, 7)
    d2["shape"] = shape = float(np.abs((P[:, :3] - P[0, :3]) - (E[:, :3] - E[0, :3])).max())
    over = {k: (v, bars.get(k, 1e-6)) for k, v in d2.items() if not v < bars.get(k, 1e-6)}
    assert not over, over
    return {"window 1": "%.1e m" % d1["para_Pose.p"], "window 2": {k: "%.1e" % v for k, v in d2.items() if v > 0}}


@pytest.mark.parametrize("name", sorted(CHAIN_BARS))
def test_oracle_chain_meets_the_chain_at_60_digits(oracle, name):
    """rows S1 + M1-M3 chained: the CPU oracle's solve -> marginalise -> solve against the same chain with every number at 60 digits"""
    print(name, "oracle chain vs 60 digits:", check_chain(lambda a: oracle.ba_solve(a, 8), lambda a: oracle.ba_marginalize(a, 0), name))


# ---------------------------------------------------------------- front end, the in-tree floating point of row T8: camera model and velocities
IDC_CAM = dict(fx=6.2097277909374247e+02, fy=6.2212293397677581e+02, cx=3.1175896455154810e+02, cy=2.4718077836114819e+02,
               k1=1.4865749308203452e-01, k2=-4.6815685578576460e-01, p1=1.6205585303208318e-03, p2=-8.9101576735577930e-03)   # config/realsense/idc_cam.yaml


def check_t8_against_the_reference_formulas(frames_out, times, K):
    """`frames_out`: per frame (ids, observations [x, y, z, u, v, vx, vy, depth]) of ANY implementation of trackImage.  Checked against the reference's own formulas, evaluated
    independently of oracle/ and of the library:
      * x, y = float(PinholeCamera::liftProjective(u, v)) (camera_models/src/camera_models/PinholeCamera.cc:449-510: m_inv_K of :292-295, the recursive model with n = 8 and
        `distortion` :646-662), transcribed into mpmath at 60 digits; FeatureTracker::undistortedPts (feature_tracker.cpp:797-808) rounds it to float.  The double-precision
        evaluation may land on the other side of a float rounding boundary once in ~1e8 points: at most one float ulp is tolerated, and the count of exact hits is returned;
      * vx, vy = float((x - x_prev) / dt), the subtraction in float, the division in double (FeatureTracker::ptsVelocity, feature_tracker.cpp:810-847), 0 for a new id or the
        first frame: IEEE operations restated in numpy, compared bit for bit."""
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 60
    f = mp.mpf
    k1, k2, p1, p2 = (f(K[k]) for k in ("k1", "k2", "p1", "p2"))
    # the class stores the inverse intrinsics as doubles (PinholeCamera.cc:292-295): the same doubles here
    iK11, iK13, iK22, iK23 = 1.0 / K["fx"], -K["cx"] / K["fx"], 1.0 / K["fy"], -K["cy"] / K["fy"]

    def distortion(x, y):
        mx2, my2, mxy = x * x, y * y, x * y
        rho2 = mx2 + my2
        rad = k1 * rho2 + k2 * rho2 * rho2
        return x * rad + 2 * p1 * mxy + p2 * (rho2 + 2 * mx2), y * rad + 2 * p2 * mxy + p1 * (rho2 + 2 * my2)

    def lift(u, v):
        mxd, myd = f(iK11) * f(u) + f(iK13), f(iK22) * f(v) + f(iK23)
        dx, dy = distortion(mxd, myd)
        mxu, myu = mxd - dx, myd - dy
        for _ in range(1, 8):
            dx, dy = distortion(mxu, myu)
            mxu, myu = mxd - dx, myd - dy
        return mxu, myu
    exact = total = 0
    prev = {}
    for k, (ids, o) in enumerate(frames_out):
        cur = {}
        for i, ob in zip(ids, o):
            x, y = lift(float(ob[3]), float(ob[4]))
            for got, want in ((ob[0], x), (ob[1], y)):
                w32 = np.float32(float(want))      # float(mpf) rounds to nearest double; the value is nowhere near a double-rounding tie of a float boundary at 60 digits
                assert abs(np.float32(got) - w32) <= abs(np.spacing(w32)), (k, int(i), got, want)
                exact += int(np.float32(got) == w32)
                total += 1
            assert ob[2] == 1.0
            cur[int(i)] = (np.float32(ob[0]), np.float32(ob[1]))
            if k > 0 and int(i) in prev:
                dt = times[k] - times[k - 1]
                want_v = [np.float32(np.float64(np.float32(cur[int(i)][q] - prev[int(i)][q])) / dt) for q in (0, 1)]
            else:
                want_v = [np.float32(0), np.float32(0)]
            assert np.float32(ob[5]) == want_v[0] and np.float32(ob[6]) == want_v[1], (k, int(i), ob[5], ob[6], want_v)
        prev = cur
    assert total > 0
    return exact, total


def test_oracle_camera_model_and_velocities_meet_the_reference_formulas(oracle):
    """row T8 on the CPU oracle (its tracker restates FT:797-847 and camodocal's pinhole model; this test restates them a second time, in other arithmetic)"""
    import sys
    sys.path.insert(0, HERE)
    import synth
    cfg = oracle.default_cfg()
    for k, v in IDC_CAM.items():
        setattr(cfg, k, v)
    tr = oracle.Tracker(cfg)
    frames = synth.tracker_sequence(1005, 4, 640, 480)
    depth = np.full(frames[0].shape, 2100, np.uint16)
    times = [0.0666 * k for k in range(len(frames))]
    out = [tr.track(times[k], f, depth) for k, f in enumerate(frames)]
    exact, total = check_t8_against_the_reference_formulas(out, times, IDC_CAM)
    print("oracle undistorted coordinates: %d of %d floats equal the 60-digit value rounded to float (the rest within one ulp)" % (exact, total))
    assert exact >= 0.999 * total


# ---------------------------------------------------------------- row B2: IMU / wheel pre-integration against the reference's formulas at 60 digits
def check_preint_against_ref(imu_fn, wheel_fn, tol=1e-13):
    """tests/golden/ref_preint.json.gz (make_ref_preint_golden.py: IntegrationBase / WheelIntegrationBase::midPointIntegration + propagate transcribed into mpmath): every
    output within `tol` of the 60-digit value, relative to the largest entry of its array (observed 1e-15)"""
    import gzip
    import json
    with gzip.open(os.path.join(HERE, "golden", "ref_preint.json.gz"), "rt") as f:
        fx = json.load(f)
    worst = 0.0
    for kind, fn in (("imu", imu_fn), ("wheel", wheel_fn)):
        if fn is None:
            continue
        for c in fx[kind]:
            got = fn(**c["input"])
            for k, v in c["expected"].items():
                e = np.asarray(v, np.float64).reshape(-1)
                dev = float(np.abs(np.asarray(got[k], np.float64).reshape(-1) - e).max() / np.abs(e).max())
                assert dev < tol, (kind, len(c["input"]["dt"]), k, dev)
                worst = max(worst, dev)
    return worst


def test_oracle_preintegration_meets_the_reference_formulas_at_60_digits(oracle):
    print("oracle pre-integration vs 60 digits: %.1e" % check_preint_against_ref(oracle.imu_preintegrate, oracle.wheel_preintegrate))
