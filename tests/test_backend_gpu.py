"""Back-end parity tests: HIP window solver (through the C-ABI) vs the CPU oracle. Run with -m gpu."""
import numpy as np
import pytest
import gfwindow as gw
import synth_window as SW

pytestmark = pytest.mark.gpu


def _pose_diff(a, b):
    pa, pb = a["para_Pose"].reshape(-1, 7), b["para_Pose"].reshape(-1, 7)
    dp = np.abs(pa[:, :3] - pb[:, :3]).max()
    dq = min(np.abs(pa[:, 3:] - pb[:, 3:]).max(), np.abs(pa[:, 3:] + pb[:, 3:]).max())
    return dp, 2 * dq  # rad ~ 2 * |dq_vec|


def test_preintegration_matches_oracle(gf, oracle):
    w_o = SW.make_window(5, oracle)
    w_g = SW.make_window(5, gf)
    for k in ("imu_delta_p", "imu_delta_q", "imu_delta_v", "imu_jacobian", "imu_covariance", "wh_delta_p", "wh_delta_q", "wh_jacobian", "wh_covariance"):
        scale = max(1.0, np.abs(w_o[k]).max())
        assert np.abs(w_o[k] - w_g[k]).max() <= 1e-12 * scale, k


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, {"use_wheel": False}), (3, {"fix_ex_pose": 0}), (4, {"fix_td": 0, "fix_ex_wheel": 1})])
def test_normal_equations_match_oracle(gf, oracle, seed, kw):
    w = SW.make_window(seed, oracle, **kw)
    if seed == 4:
        w["para_Td"][0] = 0.004
    est = gf.Estimator()
    a = oracle.ba_linearize(w)
    b = est.linearize(w)
    assert a["n_f"] == b["n_f"] and a["n_e"] == b["n_e"] and np.array_equal(a["ids"], b["ids"])
    assert abs(a["cost"] - b["cost"]) <= 1e-10 * a["cost"]
    hs = np.sqrt(np.outer(np.diag(a["H"]), np.diag(a["H"]))) + 1e-300
    assert (np.abs(a["H"] - b["H"]) / hs).max() < 1e-9
    assert np.abs(a["g"] - b["g"]).max() <= 1e-9 * np.abs(a["g"]).max()
    est.close()


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, {"use_wheel": False}), (3, {"fix_ex_pose": 0}), (6, {"max_features": 60})])
def test_solve_matches_oracle_poses(gf, oracle, seed, kw):
    """poses within 1e-6 m / 1e-6 rad of the CPU path after the same iteration count (BASELINE.json north_star)"""
    w0 = SW.make_window(seed, oracle, **kw)
    wo, wg = w0.copy(), w0.copy()
    so = oracle.ba_solve(wo, 8)
    est = gf.Estimator()
    sg = est.solve([wg], 8)[0]
    assert sg["iterations"] == so["iterations"] and sg["successful_steps"] == so["successful_steps"] and sg["termination"] == so["termination"]
    assert abs(sg["final_cost"] - so["final_cost"]) <= 1e-8 * so["final_cost"]
    dp, dr = _pose_diff(wo, wg)
    assert dp < 1e-6 and dr < 1e-6, (dp, dr)
    for k in ("para_SpeedBias", "para_Feature", "para_Ex_Pose", "para_Ex_Pose_wheel"):
        # (seed 3 frees the camera extrinsic without a prior: its translation is barely observable and wanders to ~10 m; relative bound)
        assert (np.abs(wo[k] - wg[k]) / np.maximum(1.0, np.abs(wo[k]))).max() < 1e-6, k
    est.close()


def test_batched_solve_matches_single(gf, oracle):
    wins = [SW.make_window(10 + b, oracle) for b in range(5)]
    ref = [w.copy() for w in wins]
    for w in ref:
        oracle.ba_solve(w, 6)
    est = gf.Estimator(batch=8)
    sums = est.solve(wins, 6)
    for w, r in zip(wins, ref):
        dp, dr = _pose_diff(w, r)
        assert dp < 1e-6 and dr < 1e-6
    assert all(s["iterations"] == 6 for s in sums)
    est.close()


def _prior_invariants(p):
    n = p["n"]
    J = p["J"].reshape(n, n)
    return J.T @ J, J.T @ p["r"], float(p["r"] @ p["r"])


@pytest.mark.parametrize("seed,kw", [(4, {}), (7, {"use_wheel": False})])
def test_marginalisation_matches_oracle(gf, oracle, seed, kw):
    """prior built on the GPU == prior built by the oracle, compared through the basis-independent J^T J, J^T r
    (the eigenvector basis of marginalization_factor.cpp:294-302 is not unique)"""
    est = gf.Estimator()
    w = SW.make_window(seed, oracle, **kw)
    oracle.ba_solve(w, 4)
    po = oracle.ba_marginalize(w, 0)
    pg = est.marginalize([w], 0)[0]
    assert pg is not None and np.array_equal(po["block_id"], pg["block_id"]) and po["n"] == pg["n"] and po["m"] == pg["m"]
    assert np.array_equal(po["x0"], pg["x0"])
    Ao, bo, co = _prior_invariants(po)
    Ag, bg, cg = _prior_invariants(pg)
    # The Schur complement through the dropped pose/speed-bias block is ill-conditioned (cond ~1e6-1e7: biases vs positions), so
    # rounding-level differences of the accumulation order (atomics) show up at ~1e-6..1e-5 in a few weak entries; the bound that
    # matters — poses of the next solve within 1e-6 — is asserted below and in test_prior_chain_solve_with_gpu_prior.
    sc = np.sqrt(np.outer(np.diag(Ao), np.diag(Ao))) + 1e-6 * np.abs(Ao).max()
    assert (np.abs(Ao - Ag) / sc).max() < 1e-4
    assert np.abs(bo - bg).max() <= 1e-5 * np.abs(bo).max()
    # second window: solve with that prior on both sides, then both marginalisation modes
    w2 = SW.make_window(seed, oracle, frame0=1, prior=po, **kw)
    wo, wg = w2.copy(), w2.copy()
    oracle.ba_solve(wo, 8)
    est.solve([wg], 8)
    dp, dr = _pose_diff(wo, wg)
    assert dp < 1e-6 and dr < 1e-6
    for mode in (0, 1):
        p1o = oracle.ba_marginalize(wo, mode)
        p1g = est.marginalize([wo], mode)[0]
        assert np.array_equal(p1o["block_id"], p1g["block_id"]) and p1o["m"] == p1g["m"]
        Ao, bo, co = _prior_invariants(p1o)
        Ag, bg, cg = _prior_invariants(p1g)
        sc = np.sqrt(np.outer(np.diag(Ao), np.diag(Ao))) + 1e-6 * np.abs(Ao).max()
        assert (np.abs(Ao - Ag) / sc).max() < 1e-4, mode
        assert np.abs(bo - bg).max() <= 1e-5 * np.abs(bo).max(), mode
    est.close()


@pytest.mark.parametrize("seed", [9, 11, 12])
def test_prior_chain_solve_with_gpu_prior(gf, oracle, seed):
    """a prior produced on the GPU, fed back into the next window, gives the same poses as the all-oracle chain"""
    est = gf.Estimator()
    w = SW.make_window(seed, oracle)
    wg = w.copy()
    oracle.ba_solve(w, 8); est.solve([wg], 8)
    po = oracle.ba_marginalize(w, 0)
    pg = est.marginalize([wg], 0)[0]
    w2o = SW.make_window(seed, oracle, frame0=1, prior=po)
    w2g = SW.make_window(seed, oracle, frame0=1, prior=pg)
    w2x = SW.make_window(seed, oracle, frame0=1, prior=pg)
    oracle.ba_solve(w2o, 8); est.solve([w2g], 8); oracle.ba_solve(w2x, 8)
    # same (GPU-made) prior, HIP solve vs oracle solve: the solver bar
    dp, dr = _pose_diff(w2x, w2g)
    assert dp < 1e-6 and dr < 1e-6, (dp, dr)
    # all-HIP chain vs all-oracle chain: additionally carries the prior's own rounding.  The Schur complement of the dropped block has a
    # condition number of 1e6..1e7 on these windows, so the two priors agree to ~1e-10 relative only, and that difference moves with the
    # order of the atomic accumulation (scripts/chain_flaky.py: 1e-7 .. 3e-5 on seed 12 from run to run) -- conditioning, not a defect
    dp, dr = _pose_diff(w2o, w2g)
    assert dp < 1e-4 and dr < 1e-4, (dp, dr)
    est.close()


@pytest.mark.parametrize("W,F", [(20, 500), (12, 200)])
def test_large_window_matches_oracle(gf, oracle, W, F):
    """BASELINE.json config 5 sizes (20-frame window, 500 features, ~9000 visual factors; without GNSS): the reduced system (332 columns)
    no longer fits LDS, ba_step<true> / ba_marg_finish<true> keep it in global memory."""
    est = gf.Estimator(W, F, F * W)
    w = SW.make_window(1, oracle, W=W, n_landmarks=int(F * 1.5), max_features=F)
    assert w["n_feature"] == F and w["n_visual"] > 8 * F
    lo, lg = oracle.ba_linearize(w.copy(), cap=1024), est.linearize(w.copy(), cap=1024)
    assert lo["n_f"] == lg["n_f"] == 15 * (W + 1) + 6 and lo["n_e"] == lg["n_e"]
    assert abs(lo["cost"] - lg["cost"]) <= 1e-12 * lo["cost"]
    assert np.abs(lo["H"] - lg["H"]).max() <= 1e-13 * np.abs(lo["H"]).max() and np.abs(lo["g"] - lg["g"]).max() <= 1e-13 * np.abs(lo["g"]).max()
    wo, wg = w.copy(), w.copy()
    so, sg = oracle.ba_solve(wo, 8), est.solve([wg], 8)[0]
    assert (so["iterations"], so["successful_steps"]) == (sg["iterations"], sg["successful_steps"])
    dp, dr = _pose_diff(wo, wg)
    assert dp < 1e-6 and dr < 1e-6, (dp, dr)
    po, pg = oracle.ba_marginalize(wo, 0, cap_n=512), est.marginalize([wo], 0, cap_n=512)[0]
    assert pg["n"] == po["n"] == 6 * W + 9 + 17 and list(pg["block_id"]) == list(po["block_id"])
    n = po["n"]
    Ao, Ag = po["J"].reshape(n, n).T @ po["J"].reshape(n, n), pg["J"].reshape(n, n).T @ pg["J"].reshape(n, n)
    assert np.abs(Ao - Ag).max() <= 1e-9 * np.abs(Ao).max()          # conditioning of the dropped block, see test_marginalization_matches_oracle
    w2o = SW.make_window(1, oracle, W=W, n_landmarks=int(F * 1.5), max_features=F, frame0=1, prior=pg)
    w2g = w2o.copy()
    oracle.ba_solve(w2o, 8); est.solve([w2g], 8)
    dp, dr = _pose_diff(w2o, w2g)
    assert dp < 1e-6 and dr < 1e-6, (dp, dr)
    est.close()


def test_lds_and_global_reduced_system_agree(gf, oracle, monkeypatch):
    """the global-memory variant of the step / marginalisation kernels on a window that also fits LDS"""
    w = SW.make_window(4, oracle)
    a, b = w.copy(), w.copy()
    e1 = gf.Estimator(); e1.solve([a], 8); p1 = e1.marginalize([a], 0)[0]; e1.close()
    monkeypatch.setenv("GF_BA_FORCE_GLOBAL", "1")
    e2 = gf.Estimator(); e2.solve([b], 8); p2 = e2.marginalize([b], 0)[0]; e2.close()
    dp, dr = _pose_diff(a, b)
    assert dp < 1e-8 and dr < 1e-8
    A1, A2 = p1["J"].reshape(p1["n"], -1), p2["J"].reshape(p2["n"], -1)
    assert np.abs(A1.T @ A1 - A2.T @ A2).max() <= 1e-9 * np.abs(A1.T @ A1).max()


@pytest.mark.parametrize("drop", ["visual", "inertial"])
def test_windows_without_a_factor_family(gf, oracle, drop):
    """ragged inputs: a window without a single visual factor (vision failure: IMU + wheel only) and one without IMU / wheel factors
    (sum_dt > 10 skips them, estimator.cpp:3114-3132): same solve and marginalisation as the oracle"""
    w = SW.make_window(5, oracle)
    keys = [k for k in w if isinstance(w[k], np.ndarray) and (k.startswith("vis_") if drop == "visual" else (k.startswith("imu_") or k.startswith("wh_")))]
    for k in keys:
        w[k] = w[k][:0]
    if drop == "visual":
        w["para_Feature"], w["feature_fixed"] = w["para_Feature"][:0], w["feature_fixed"][:0]
    wo, wg = w.copy(), w.copy()
    so = oracle.ba_solve(wo, 6)
    est = gf.Estimator()
    sg = est.solve([wg], 6)[0]
    assert (sg["iterations"], sg["successful_steps"]) == (so["iterations"], so["successful_steps"])
    assert abs(sg["final_cost"] - so["final_cost"]) <= 1e-7 * max(so["final_cost"], 1.0)
    dp, dr = _pose_diff(wo, wg)
    assert dp < 1e-6 and dr < 1e-6, (dp, dr)
    po, pg = oracle.ba_marginalize(wo, 0), est.marginalize([wo.copy()], 0)[0]
    assert pg is not None and list(pg["block_id"]) == list(po["block_id"]) and pg["n"] == po["n"]
    Ao, Ag = po["J"].reshape(po["n"], -1), pg["J"].reshape(pg["n"], -1)
    assert np.abs(Ao.T @ Ao - Ag.T @ Ag).max() <= 1e-6 * np.abs(Ao.T @ Ao).max()
    est.close()


def test_pair_tiles_in_lds_and_in_global_memory_agree(gf, oracle, monkeypatch):
    """ba_linearize_visual_win keeps its per-pair tiles in LDS at W = 10 and in global memory for longer windows; the global variant on a
    window that also fits LDS: the same sums in the same order, bit-identical normal equations and solve"""
    w = SW.make_window(9, oracle)
    e1 = gf.Estimator(); l1 = e1.linearize(w); a = w.copy(); e1.solve([a], 8); e1.close()
    monkeypatch.setenv("GF_BA_GLOBAL_TILES", "1")
    e2 = gf.Estimator(); l2 = e2.linearize(w); b = w.copy(); e2.solve([b], 8); e2.close()
    assert l1["cost"] == l2["cost"] and np.array_equal(l1["H"], l2["H"]) and np.array_equal(l1["g"], l2["g"])
    assert np.array_equal(a["para_Pose"], b["para_Pose"])


# ---------------------------------------------------------------- GNSS residual blocks on the device
def _gnss_est(gf, W=10, F=150):
    return gf.Estimator(W, F, F * W, 1, max_gnss=12 * (W + 1))


@pytest.mark.parametrize("kw", [dict(), dict(anchor=True), dict(gnss_lowspeed=1)])
def test_gnss_normal_equations_and_solve_match_oracle(gf, oracle, kw):
    est = _gnss_est(gf)
    w = SW.make_window(1, oracle, gnss=True, **kw)
    lo, lg = oracle.ba_linearize(w.copy(), cap=1024), est.linearize(w.copy(), cap=1024)
    assert list(lo["ids"]) == list(lg["ids"]) and lo["n_f"] == lg["n_f"]
    if not kw.get("gnss_lowspeed"):
        assert gw.bid(gw.RCV_DT, 7) in lo["ids"] and gw.bid(gw.RCV_DDT, 3) in lo["ids"] and gw.bid(gw.ANC) in lo["ids"] and gw.bid(gw.YAW) not in lo["ids"]
    assert abs(lo["cost"] - lg["cost"]) <= 1e-12 * lo["cost"]
    assert np.abs(lo["H"] - lg["H"]).max() <= 1e-12 * np.abs(lo["H"]).max() and np.abs(lo["g"] - lg["g"]).max() <= 1e-12 * np.abs(lo["g"]).max()
    wo, wg = w.copy(), w.copy()
    so, sg = oracle.ba_solve(wo, 8), est.solve([wg], 8)[0]
    assert (so["iterations"], so["successful_steps"]) == (sg["iterations"], sg["successful_steps"])
    dp, dr = _pose_diff(wo, wg)
    assert dp < 1e-6 and dr < 1e-6, (dp, dr)
    assert np.abs(wo["para_rcv_dt"] - wg["para_rcv_dt"]).max() < 1e-6 and np.abs(wo["para_rcv_ddt"] - wg["para_rcv_ddt"]).max() < 1e-6      # metres, m/s
    assert np.abs(wo["para_anc_ecef"] - wg["para_anc_ecef"]).max() < 1e-6 and wg["para_yaw_enu_local"][0] == w["para_yaw_enu_local"][0]
    est.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_gnss_marginalization_and_chain(gf, oracle, seed):
    """MARGIN_OLD drops pose 0, speed-bias 0, the four frame-0 clock biases and the frame-0 drift (20 columns); the prior keeps the frame-1 clocks,
    yaw_enu_local and the anchor; MARGIN_SECOND_NEW on the next window only renames the newest clocks."""
    est = _gnss_est(gf)
    w = SW.make_window(seed, oracle, gnss=True)
    oracle.ba_solve(w, 8)
    po, pg = oracle.ba_marginalize(w, 0), est.marginalize([w], 0)[0]
    assert pg["m"] == po["m"] and pg["n"] == po["n"] == 95 and list(pg["block_id"]) == list(po["block_id"])
    assert np.abs(pg["x0"] - po["x0"]).max() == 0
    n = po["n"]
    Ao, Ag = po["J"].reshape(n, n).T @ po["J"].reshape(n, n), pg["J"].reshape(n, n).T @ pg["J"].reshape(n, n)
    bo, bg = po["J"].reshape(n, n).T @ po["r"], pg["J"].reshape(n, n).T @ pg["r"]
    # the dropped block now also holds five receiver-clock columns: its pseudo-inverse is conditioned ~1e7, b carries that (see test_marginalisation_matches_oracle)
    # b_r = b_k - M_kp P^+ b_p cancels ~3 digits on these windows: the prior's right-hand side agrees to ~1e-4 relative from run to run
    assert np.abs(Ao - Ag).max() <= 1e-9 * np.abs(Ao).max() and np.abs(bo - bg).max() <= 1e-3 * max(1.0, np.abs(bo).max())
    w2 = SW.make_window(seed, oracle, gnss=True, frame0=1, prior=pg)
    a, b = w2.copy(), w2.copy()
    so, sg = oracle.ba_solve(a, 8), est.solve([b], 8)[0]
    assert (so["iterations"], so["successful_steps"]) == (sg["iterations"], sg["successful_steps"])
    dp, dr = _pose_diff(a, b)
    assert dp < 1e-6 and dr < 1e-6 and np.abs(a["para_rcv_dt"] - b["para_rcv_dt"]).max() < 1e-6
    p1o, p1g = oracle.ba_marginalize(a, 1), est.marginalize([a], 1)[0]
    assert p1g["n"] == p1o["n"] == 89 and list(p1g["block_id"]) == list(p1o["block_id"])
    n1 = p1o["n"]
    A1o, A1g = p1o["J"].reshape(n1, n1).T @ p1o["J"].reshape(n1, n1), p1g["J"].reshape(n1, n1).T @ p1g["J"].reshape(n1, n1)
    assert np.abs(A1o - A1g).max() <= 1e-9 * np.abs(A1o).max()
    est.close()


def test_config5_window_with_gnss(gf, oracle):
    """BASELINE.json config 5: 20-frame window, 500 features, RGB-D + IMU + wheel + GNSS factors: 440 reduced columns, global-memory Cholesky."""
    W, F = 20, 500
    est = _gnss_est(gf, W, F)
    w = SW.make_window(1, oracle, W=W, n_landmarks=750, max_features=F, gnss=True)
    assert w["n_gnss"] == 12 * (W + 1) and w["n_feature"] == F
    a, b = w.copy(), w.copy()
    so, sg = oracle.ba_solve(a, 8), est.solve([b], 8)[0]
    assert (so["iterations"], so["successful_steps"]) == (sg["iterations"], sg["successful_steps"])
    dp, dr = _pose_diff(a, b)
    assert dp < 1e-6 and dr < 1e-6, (dp, dr)
    po, pg = oracle.ba_marginalize(a, 0, cap_n=512), est.marginalize([a], 0, cap_n=512)[0]
    assert pg["n"] == po["n"] == 6 * W + 9 + 17 + 9 and list(pg["block_id"]) == list(po["block_id"])
    est.close()
