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


@pytest.mark.parametrize("block,mask", [("ex_pose", 0x04), ("ex_pose", 0x07), ("ex_pose", 0x38), ("ex_pose", 0x3c), ("ex_wheel", 0x04), ("ex_wheel", 0x07),
                                        ("ex_wheel", 0x38), ("ex_wheel", 0x3c)])
def test_subset_parameterisation_masks_match_oracle(gf, oracle, block, mask):
    """PoseSubsetParameterization (pose_subset_parameterization.cpp:27-64) on the camera extrinsic (EST:2969-2985, `estimate_extrinsic: 1` with extrinsic_type 3 NO_Z,
    2 ROTATION, 1 TRANSLATION, 4 NO_ROTATION_NO_Z) and on the wheel extrinsic (EST:3010-3026): the block keeps six columns with full Jacobians, Plus drops the masked
    increments.  Same iteration / step counts as the oracle, poses 1e-6, and the masked components never move."""
    kw = {"fix_ex_pose": 0} if block == "ex_pose" else {"fix_ex_wheel": 0}
    w0 = SW.make_window(21 if block == "ex_pose" else 22, oracle, **kw)
    w0["%s_mask" % block] = mask
    key = "para_Ex_Pose" if block == "ex_pose" else "para_Ex_Pose_wheel"
    wo, wg, wfree = w0.copy(), w0.copy(), w0.copy()
    wfree["%s_mask" % block] = 0
    so = oracle.ba_solve(wo, 8)
    est = gf.Estimator()
    sg = est.solve([wg], 8)[0]
    est.solve([wfree], 8)
    est.close()
    assert sg["iterations"] == so["iterations"] and sg["successful_steps"] == so["successful_steps"] and sg["termination"] == so["termination"]
    dp, dr = _pose_diff(wo, wg)
    print("subset", block, hex(mask), "cost rel", abs(sg["final_cost"] - so["final_cost"]) / so["final_cost"], "dp", dp, "dr", dr)
    # The camera extrinsic's translation has no prior in these windows and is barely observable (seed 3 of test_solve_matches_oracle_poses: it wanders by metres).
    # With its translation masked (0x07) the solver still computes that component of the step -- a ratio of small numbers that the dogleg scales the whole step
    # by -- and then drops it: the cost is held to 1e-6 relative there (observed 2e-7), 1e-8 elsewhere.
    assert abs(sg["final_cost"] - so["final_cost"]) <= (1e-6 if (block, mask) == ("ex_pose", 0x07) else 1e-8) * so["final_cost"]
    assert dp < 1e-6 and dr < 1e-6, (dp, dr)
    for k in ("para_SpeedBias", "para_Feature", "para_Ex_Pose", "para_Ex_Pose_wheel"):
        assert (np.abs(wo[k] - wg[k]) / np.maximum(1.0, np.abs(wo[k]))).max() < 1e-6, k
    e0, eg = w0[key], wg[key]
    if mask & 0x07:
        held = [i for i in range(3) if (mask >> i) & 1]
        assert np.array_equal(e0[held], eg[held])                 # x + 0 is exact
    if (mask & 0x38) == 0x38:
        assert np.abs(e0[3:] - eg[3:]).max() < 1e-15              # q * deltaQ(0), normalised: the same quaternion up to rounding
    assert np.abs(wfree[key] - eg).max() > 1e-6                   # and the mask does change the result (the unmasked block moves in those components)


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


def _assert_prior_close(Ao, bo, Ag, bg, b_tol=1e-8):
    """J^T J within 1e-9 of its largest entry and J^T r within 1e-8 (relative).  Entry by entry, scaled by sqrt(A_ii A_jj), the bound is
    5e-7: that is the rounding noise of the ORACLE's own route (eigen pseudo-inverse of the whole dropped block, marginalization_factor.cpp:278-283;
    scripts/marg_precision.py measures 4e-8 .. 2e-7 for it against a 60-digit Schur complement, and 3e-9 .. 8e-9 for the block elimination the
    kernel uses).  The HIP side itself is reproducible to the bit (test_runs_are_bit_identical)."""
    assert np.abs(Ao - Ag).max() <= 1e-9 * np.abs(Ao).max()
    sc = np.sqrt(np.outer(np.diag(Ao), np.diag(Ao))) + 1e-6 * np.abs(Ao).max()
    assert (np.abs(Ao - Ag) / sc).max() < 5e-7
    assert np.abs(bo - bg).max() <= b_tol * np.abs(bo).max()


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
    _assert_prior_close(Ao, bo, Ag, bg)
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
        _assert_prior_close(Ao, bo, Ag, bg)
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
    # all-HIP chain (HIP solve -> HIP prior -> HIP solve) vs all-oracle chain: the bar of BASELINE.json (observed 3e-8 .. 5e-8; the oracle moves
    # by 3e-8 .. 1.3e-7 against itself when its solved state is perturbed in the last bit, scripts/gnss_chain_sensitivity.py)
    dp, dr = _pose_diff(w2o, w2g)
    assert dp < 1e-6 and dr < 1e-6, (dp, dr)
    est.close()


@pytest.mark.parametrize("kw", [dict(), dict(gnss=True)])
def test_runs_are_bit_identical(gf, oracle, kw):
    """no floating-point atomics: every sum of the back end has a fixed order, so two handles give the same bits -- states after the solve,
    both kinds of prior, and a batch slot gives what a single-window handle gives"""
    mk = (lambda b=1: _gnss_est(gf, batch=b)) if kw else (lambda b=1: gf.Estimator(batch=b))
    w = SW.make_window(12, oracle, **kw)
    outs = []
    for rep in range(3):
        est = mk(4 if rep == 2 else 1)
        a = w.copy()
        wins = [SW.make_window(20 + q, oracle, **kw) for q in range(2)] + [a] if rep == 2 else [a]
        s = est.solve(wins, 8)[-1]
        p0 = est.marginalize(wins, 0)[-1]
        p1 = None
        if not kw:
            w2 = SW.make_window(12, oracle, frame0=1, prior=p0); est.solve([w2], 8); p1 = est.marginalize([w2], 1)[0]
        outs.append((a, s, p0, p1))
        est.close()
    for a, s, p0, p1 in outs[1:]:
        for k in gw.STATE_KEYS:
            assert np.array_equal(a[k], outs[0][0][k]), k
        assert s == outs[0][1]
        assert np.array_equal(p0["J"], outs[0][2]["J"]) and np.array_equal(p0["r"], outs[0][2]["r"])
        if p1 is not None:
            assert np.array_equal(p1["J"], outs[0][3]["J"]) and np.array_equal(p1["r"], outs[0][3]["r"])


def test_batch_with_different_gravity_and_flags(gf, oracle):
    """gravity and the visual sqrt_info are per window (each Estimator has its own `g` after initialStructure, estimator.cpp:1630-1650), as are the
    constant-block decisions: a batch of windows that differ in all of them against the oracle's solo solves"""
    kws = [dict(), dict(use_wheel=False), dict(fix_td=0, fix_ex_wheel=1), dict()]
    wins = [SW.make_window(30 + q, oracle, **kw) for q, kw in enumerate(kws)]
    for q, g in enumerate([[0.0, 0.0, 9.805], [0.35, -0.2, 9.7966], [-0.5, 0.1, 9.7917], [0.0, 0.0, 9.81]]):
        wins[q]["G"] = np.array(g)
    wins[3]["vis_sqrt_info"] = 460.0 / 1.5
    ref = [w.copy() for w in wins]
    sums_o = [oracle.ba_solve(w, 8) for w in ref]
    est = gf.Estimator(batch=4)
    sums = est.solve(wins, 8)
    for w, r, so, sg in zip(wins, ref, sums_o, sums):
        assert (sg["iterations"], sg["successful_steps"]) == (so["iterations"], so["successful_steps"])
        dp, dr = _pose_diff(w, r)
        assert dp < 1e-6 and dr < 1e-6, (dp, dr)
    pg = est.marginalize(ref, 0)
    for w, p in zip(ref, pg):
        po = oracle.ba_marginalize(w, 0)
        Ao, bo, _ = _prior_invariants(po)
        Ag, bg, _ = _prior_invariants(p)
        _assert_prior_close(Ao, bo, Ag, bg)
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


def test_config3_300_features_matches_oracle(gf, oracle):
    """BASELINE.json configs[2]: 10-frame window, 300 features (3000 visual factors): solve, MARGIN_OLD, next solve, MARGIN_SECOND_NEW"""
    F = 300
    est = gf.Estimator(10, F, F * 10)
    w = SW.make_window(3, oracle, n_landmarks=int(F * 1.5), max_features=F)
    assert w["n_feature"] == F and w["n_visual"] > 2000
    wo, wg = w.copy(), w.copy()
    so, sg = oracle.ba_solve(wo, 8), est.solve([wg], 8)[0]
    assert (so["iterations"], so["successful_steps"], so["termination"]) == (sg["iterations"], sg["successful_steps"], sg["termination"])
    dp, dr = _pose_diff(wo, wg)
    assert dp < 1e-6 and dr < 1e-6, (dp, dr)
    po, pg = oracle.ba_marginalize(wo, 0, cap_n=512), est.marginalize([wo], 0, cap_n=512)[0]
    assert list(pg["block_id"]) == list(po["block_id"]) and pg["m"] == po["m"]
    Ao, bo, _ = _prior_invariants(po)
    Ag, bg, _ = _prior_invariants(pg)
    _assert_prior_close(Ao, bo, Ag, bg)
    w2o = SW.make_window(3, oracle, n_landmarks=int(F * 1.5), max_features=F, frame0=1, prior=po)
    w2g = SW.make_window(3, oracle, n_landmarks=int(F * 1.5), max_features=F, frame0=1, prior=pg)
    oracle.ba_solve(w2o, 8); est.solve([w2g], 8)
    dp, dr = _pose_diff(w2o, w2g)
    assert dp < 1e-6 and dr < 1e-6, (dp, dr)
    p1o, p1g = oracle.ba_marginalize(w2o, 1, cap_n=512), est.marginalize([w2o], 1, cap_n=512)[0]
    Ao, bo, _ = _prior_invariants(p1o)
    Ag, bg, _ = _prior_invariants(p1g)
    _assert_prior_close(Ao, bo, Ag, bg)
    est.close()


def test_lds_and_global_reduced_system_agree(gf, oracle, monkeypatch):
    """the global-memory variant of the step / marginalisation kernels on a window that also fits LDS"""
    monkeypatch.delenv("GF_BA_CHAIN", raising=False)   # two placements of the DENSE form's system are compared here (a suite run under GF_BA_CHAIN=1 would put the chain form on one side)
    w = SW.make_window(4, oracle)
    a, b = w.copy(), w.copy()
    e1 = gf.Estimator(); e1.solve([a], 8); p1 = e1.marginalize([a], 0)[0]; e1.close()
    monkeypatch.setenv("GF_BA_FORCE_GLOBAL", "1")
    e2 = gf.Estimator(); e2.solve([b], 8); p2 = e2.marginalize([b], 0)[0]; e2.close()
    dp, dr = _pose_diff(a, b)
    assert dp < 1e-9 and dr < 1e-9
    A1, A2 = p1["J"].reshape(p1["n"], -1), p2["J"].reshape(p2["n"], -1)
    assert np.abs(A1.T @ A1 - A2.T @ A2).max() <= 1e-11 * np.abs(A1.T @ A1).max()


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
    assert np.abs(Ao.T @ Ao - Ag.T @ Ag).max() <= 2e-8 * np.abs(Ao.T @ Ao).max()   # no visual information: the oracle's eigen route carries the 1e9 dynamic range of IMU vs wheel blocks
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
def _gnss_est(gf, W=10, F=150, batch=1):
    return gf.Estimator(W, F, F * W, batch, max_gnss=12 * (W + 1))


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
    # With GNSS the kept system has eigenvalues right at the truncation threshold (1e-8 .. 1e-6 against 1e8 at the top: yaw_enu_local, the ECEF anchor)
    # whose right-hand-side components are rounding noise of the 6.4e6-m ECEF arithmetic: J^T r is defined to ~1e-7 only -- the oracle moves by
    # 2e-8 .. 9e-8 against itself when its input state is perturbed in the last bit (scripts/gnss_chain_sensitivity.py); observed here 6e-9 .. 4e-8 with the
    # least-squares right-hand side of round 4 (2e-8 .. 4e-7 with the forward-substituted one before it): bar 2e-7
    print("gnss prior seed %d: J^T r rel %.3e" % (seed, np.abs(bo - bg).max() / np.abs(bo).max()))
    _assert_prior_close(Ao, bo, Ag, bg, b_tol=2e-7)
    w2 = SW.make_window(seed, oracle, gnss=True, frame0=1, prior=pg)
    a, b = w2.copy(), w2.copy()
    so, sg = oracle.ba_solve(a, 8), est.solve([b], 8)[0]
    assert (so["iterations"], so["successful_steps"]) == (sg["iterations"], sg["successful_steps"])
    dp, dr = _pose_diff(a, b)
    assert dp < 1e-6 and dr < 1e-6 and np.abs(a["para_rcv_dt"] - b["para_rcv_dt"]).max() < 1e-6
    # all-HIP chain against the all-oracle chain.  Rotations and the window's shape meet the bar; its absolute position is observed through
    # pseudoranges only and inherits the prior's noise above: the oracle's own chain moves by 2e-6 .. 4e-5 m under last-bit perturbations
    wgpu = SW.make_window(seed, oracle, gnss=True); est.solve([wgpu], 8); pgg = est.marginalize([wgpu], 0)[0]
    w2g = SW.make_window(seed, oracle, gnss=True, frame0=1, prior=pgg); est.solve([w2g], 8)
    w2o = SW.make_window(seed, oracle, gnss=True, frame0=1, prior=po); oracle.ba_solve(w2o, 8)
    Pg, Po = w2g["para_Pose"].reshape(-1, 7), w2o["para_Pose"].reshape(-1, 7)
    dp, dr = _pose_diff(w2o, w2g)
    shape = np.abs((Pg[:, :3] - Pg[0, :3]) - (Po[:, :3] - Po[0, :3])).max()
    print("gnss chain seed %d: dp %.3e dr %.3e shape %.3e" % (seed, dp, dr, shape))
    # Round 5, adjudicated at 60 digits (scripts/adjudicate_gnss_chain.py, output in profiles/r05_adjudicate_gnss_chain.txt): window 2 solved by ONE solver with the exact
    # prior (the reference's route from the reference's factor formulas in mpmath), the oracle's and the library's -- the ORACLE's chain sits 3.5e-5 m (seed 1) / 3.5e-7 m
    # (seed 2) from the exact one, the library's 1.5e-5 / 3.5e-7 m: the 1e-4 between the two is the double-precision noise of the reference's own algorithm in the
    # directions next to its 1e-8 cut (kept eigenvalues 3.4e-8, 1.6e-7 at seed 1), with the oracle the farther of the two.
    # Later in round 5 the whole chain was run at 60 digits (tests/golden/ref_chain_gnss.json.gz, this seed): the library's chain ends 9.1e-8 m from it, the oracle's 3.5e-5 m --
    # this bar measures the oracle; the library's own bar against the exact chain is 1e-6 (test_chain_meets_the_chain_at_60_digits below).
    assert dr < 1e-6 and shape < 1e-6 and dp < 1e-4, (dp, dr, shape)      # observed 5e-7 and 3.5e-5
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


def test_max_solver_time_is_optional_and_stops_the_schedule(gf):
    """ceres::Solver::Options::max_solver_time_in_seconds (estimator.cpp:3312-3315): off by default (fixed schedule, no host round trip); with a generous cap
    the result is bit-identical to the uncapped solve; with a cap that is already spent after the first iteration the solve ends early -- fewer iterations,
    a higher final cost -- and still hands back a consistent, accepted state"""
    import ctypes as C
    est = gf.Estimator(batch=1)
    base = SW.make_window(1004, gf)
    free, capped, cut = base.copy(), base.copy(), base.copy()
    s_free = est.solve([free], 8)[0]
    gf._chk(gf.lib().gf_ba_set_max_solver_time(est.h, C.c_double(30.0)))
    s_cap = est.solve([capped], 8)[0]
    assert s_cap == s_free and all(np.array_equal(free[k], capped[k]) for k in gw.STATE_KEYS if k in free)
    gf._chk(gf.lib().gf_ba_set_max_solver_time(est.h, C.c_double(1e-9)))
    s_cut = est.solve([cut], 8)[0]
    assert 1 <= s_cut["iterations"] < s_free["iterations"] and s_cut["final_cost"] > s_free["final_cost"] and s_cut["final_cost"] < s_cut["initial_cost"]
    gf._chk(gf.lib().gf_ba_set_max_solver_time(est.h, C.c_double(0.0)))
    again = base.copy()
    assert est.solve([again], 8)[0] == s_free
    with pytest.raises(gf.GfError):
        gf._chk(gf.lib().gf_ba_set_max_solver_time(est.h, C.c_double(-1.0)))


def test_split_jtj_formulation_gives_the_same_bits(gf, oracle):
    """north_star's formulation of the visual sweep -- block rows [J | r] of every factor through HBM, a contraction-only MFMA kernel behind it
    (gf_ba_set_split_jtj) -- against the fused kernel: same products in the same order, so states, summaries and the next prior are identical to the bit"""
    wins = [SW.make_window(300 + b, oracle) for b in range(3)]
    a, c = [w.copy() for w in wins], [w.copy() for w in wins]
    ea, ec = gf.Estimator(batch=4), gf.Estimator(batch=4)
    ec.set_split_jtj(True)
    sa, sc = ea.solve(a, 8), ec.solve(c, 8)
    pa, pc = ea.marginalize(a, 0), ec.marginalize(c, 0)
    for wa, wc, x, y, p, q in zip(a, c, sa, sc, pa, pc):
        assert x == y
        for k in gw.STATE_KEYS:
            assert np.array_equal(wa[k], wc[k]), k
        assert np.array_equal(p["J"], q["J"]) and np.array_equal(p["r"], q["r"])
    st = ec.stats()
    assert st["jtj_contract_launches"] > 0 and st["ms_jtj_contract"] > 0
    ea.close(); ec.close()


@pytest.mark.parametrize("name", ["ref_window_free_ex_td", "ref_window_with_prior", "ref_window_wheel", "ref_window_wheel_free_ix_td", "ref_window_gnss"])
def test_normal_equations_meet_the_reference_formulas_at_60_digits(gf, name):
    """H, g, cost of a whole small window from the HIP sweeps (gf_ba_linearize) against tests/golden/ref_*.json.gz: the reference's ProjectionTwoFrameOneCamFactor, IMUFactor, WheelFactor,
    MarginalizationFactor and Ceres' Huber corrector evaluated with 60 digits by tests/golden/make_ref_golden.py -- numbers neither the oracle nor the library produced"""
    from test_golden import load_ref_window, check_against_ref
    w, fx, H, g = load_ref_window(name)
    est = gf.Estimator(max_features=16, max_visual=256, max_gnss=132 if w["gnss_enabled"] else 0)
    dev = check_against_ref(est.linearize(w, cap=1024), fx, H, g, tol=1e-11)
    print(name, "HIP vs reference formulas at 60 digits: H scaled %.1e, g %.1e" % dev)
    est.close()


@pytest.mark.parametrize("name", ["ref_marg_old_first_window", "ref_marg_old_with_prior", "ref_marg_second_new", "ref_marg_old_gnss"])
def test_marginalisation_meets_the_reference_route_at_60_digits(gf, name):
    """the HIP marginalisation (block elimination + rank-revealing Cholesky + least-squares right-hand side) against the reference's eigen route evaluated with 60 digits
    (tests/golden/ref_marg_*.json.gz): J^T J and J^T r of the prior, the same bars as the oracle's own test"""
    from test_golden import load_ref_marg, check_prior_against_ref
    w, fx, A, b, ids = load_ref_marg(name)
    est = gf.Estimator(max_features=16, max_visual=256, max_gnss=132 if w["gnss_enabled"] else 0)
    dev = check_prior_against_ref(est.marginalize([w], fx["mode"])[0], fx, A, b, ids)
    print(name, "HIP vs reference route at 60 digits: J^T J scaled %.1e, J^T r %.1e" % dev)
    est.close()


def test_first_step_meets_the_exact_step(gf):
    """ba_step's first iteration (Jacobi scaling, Schur complement on the matrix cores, blocked Cholesky, back-substitution, candidate (+)) against the same step by LU on
    the full system at 60 digits (tests/golden/ref_window_free_ex_td.json.gz, `first_step`): free camera extrinsic, condition 3e8"""
    from test_golden import check_first_step
    est = gf.Estimator(max_features=16, max_visual=256)
    print("HIP vs exact first step:", check_first_step(lambda a: est.solve([a], 1)[0]))
    est.close()


@pytest.mark.parametrize("name", ["ref_solve_free_ex_td", "ref_solve_with_prior", "ref_solve_wheel", "ref_solve_wheel_free_ix_td", "ref_solve_gnss"])
def test_solve_meets_the_loop_at_60_digits(gf, name):
    """the whole HIP solve -- 8 trust-region iterations: linearisation kernels, ba_step's Schur complement / blocked Cholesky / dogleg, candidate evaluation, step
    acceptance -- against the same loop with every number at 60 digits (tests/golden/ref_solve_*.json.gz, made by tests/golden/make_ref_solve_golden.py from the
    reference's factor formulas and Ceres' loop; LU on the full system): iterations, accepted steps, termination, final cost, final state"""
    from test_golden import check_solve
    est = gf.Estimator(max_features=16, max_visual=256, max_gnss=132 if "gnss" in name else 0)
    print(name, "HIP vs the loop at 60 digits:", check_solve(lambda a: est.solve([a], 8)[0], name))
    est.close()


@pytest.mark.parametrize("name", ["ref_chain_gnss", "ref_chain_wheel"])
def test_chain_meets_the_chain_at_60_digits(gf, name):
    """the library's own chain through the C-ABI -- gf_ba_solve, gf_ba_marginalize (MARGIN_OLD), gf_ba_solve with that prior -- against the chain with every number at 60
    digits (tests/golden/ref_chain_*.json.gz; bars and their reasons: tests/test_golden.py CHAIN_BARS_HIP -- 1e-6 on every block but one)"""
    from test_golden import check_chain, CHAIN_BARS_HIP
    est = gf.Estimator(max_features=16, max_visual=256, max_gnss=132 if "gnss" in name else 0)
    print(name, "HIP chain vs 60 digits:", check_chain(lambda a: est.solve([a], 8)[0], lambda a: est.marginalize([a], 0)[0], name, CHAIN_BARS_HIP[name]))
    est.close()


def test_upload_refuses_a_repeated_observation_and_a_second_start_frame(gf):
    """Round-5 advisor: the fixed-extrinsic sweep STORES a factor's Jd^T Jj block into its feature's E^T F row at frame j, so a window with two factors of one
    (feature, j) -- or a feature whose factors name different start frames -- would lose a term silently.  The reference cannot build such a window
    (estimator.cpp:3269-3297: one factor per later observation, all from feature_per_frame[0]); a caller of the C-ABI can, and is told so at upload."""
    est = gf.Estimator(batch=1)
    base = SW.make_window(1004, gf)
    ok = base.copy()
    est.solve([ok], 1)
    rep = base.copy()
    rep["vis_j"] = rep["vis_j"].copy()
    f0 = int(rep["vis_feature"][0])
    same = np.nonzero(rep["vis_feature"] == f0)[0]
    assert len(same) >= 2
    rep["vis_j"][same[1]] = rep["vis_j"][same[0]]
    with pytest.raises(gf.GfError, match="repeats the observation"):
        est.solve([rep], 1)
    two = base.copy()
    two["vis_i"] = two["vis_i"].copy()
    k = same[-1]
    assert two["vis_j"][k] - two["vis_i"][k] >= 2
    two["vis_i"][k] += 1
    with pytest.raises(gf.GfError, match="names start frame"):
        est.solve([two], 1)
    again = base.copy()
    assert est.solve([again], 1)[0] == est.solve([base.copy()], 1)[0]
    est.close()


@pytest.mark.parametrize("seed,kw", [(1, {}), (2, {"use_wheel": False}), (3, {"fix_ex_pose": 0}), (4, {"fix_td": 0, "fix_ex_wheel": 1}), (6, {"max_features": 60}), (21, {"prior_chain": True})])
def test_chain_form_of_the_step_matches_the_dense_form_and_the_oracle(gf, oracle, monkeypatch, seed, kw):
    """Round 6: ba_step in its chain form (GF_BA_CHAIN=1: the speed-bias blocks eliminated first, block by block, along the chain the IMU factors make of them; only the
    dense part -- poses + trailing blocks -- is an LDS-resident triangle, so that two windows share a CU) solves the same systems in another pivot order.  Against the dense
    form: the same iteration counts, accepted steps and termination, states within 1e-9 (relative; two orders of summation of the same sums); against the oracle: the
    bars of test_solve_matches_oracle_poses.  Windows with a prior (made by the library's own marginalisation), with free camera extrinsic / td (a larger dense part), without
    wheel factors, and a batch of them in one launch."""
    kw = dict(kw)
    chain_prior = kw.pop("prior_chain", False)
    w0 = SW.make_window(seed, oracle, **kw)
    if chain_prior:
        e0 = gf.Estimator()
        wa = w0.copy()
        e0.solve([wa], 8)
        pr = e0.marginalize([wa], 0)[0]
        e0.close()
        w0 = SW.make_window(seed, oracle, frame0=1, prior=pr, **kw)
    wo, wd, wc = w0.copy(), w0.copy(), w0.copy()
    so = oracle.ba_solve(wo, 8)
    est_d = gf.Estimator()
    sd = est_d.solve([wd], 8)[0]
    monkeypatch.setenv("GF_BA_CHAIN", "1")
    est_c = gf.Estimator(batch=4)
    sc = est_c.solve([wc], 8)[0]
    for k_ in ("iterations", "successful_steps", "termination"):
        assert sc[k_] == sd[k_] == so[k_], (k_, sc, sd, so)
    assert abs(sc["final_cost"] - sd["final_cost"]) <= (1e-9 if seed == 3 else 1e-10) * sd["final_cost"]   # (seed 3: a barely observable free extrinsic without a prior)
    for k_ in gw.STATE_KEYS:
        if k_ in wd and np.size(wd[k_]):
            assert (np.abs(wc[k_] - wd[k_]) / np.maximum(1.0, np.abs(wd[k_]))).max() < (1e-7 if seed == 3 else 1e-9), k_
    dp, dr = _pose_diff(wo, wc)
    assert dp < 1e-6 and dr < 1e-6, (dp, dr)
    # the same window three times in one launch next to itself: every block computes what the lone window computed, to the bit
    ws = [w0.copy() for _ in range(3)]
    est_c.solve(ws, 8)
    for w_ in ws:
        assert all(np.array_equal(w_[k_], wc[k_]) for k_ in gw.STATE_KEYS if k_ in wc)
    # and the marginalisation behind a chain-form solve is the one behind a dense-form solve (it reads the solved state only)
    if seed != 3:   # (seed 3 frees the camera extrinsic without a prior: barely observable directions, the kept system of its marginalisation amplifies the last bits of the state)
        pd_, pc_ = est_d.marginalize([wd], 0)[0], est_c.marginalize([wc], 0)[0]
        Ad, bd, _ = _prior_invariants(pd_); Ac, bc, _ = _prior_invariants(pc_)
        _assert_prior_close(Ad, bd, Ac, bc)
    est_d.close(); est_c.close()


def test_chain_form_in_a_mixed_batch(gf, oracle, monkeypatch):
    """A batch may hold windows the chain form takes (standard column layout) next to windows it leaves to the dense form (here: a stationary window, every pose and
    speed-bias block constant: estimator.cpp:3233-3246): both kernels are launched, each takes its own windows, and every window comes out as it does alone."""
    monkeypatch.setenv("GF_BA_CHAIN", "1")
    a, c = SW.make_window(31, oracle), SW.make_window(33, oracle, use_wheel=False)   # (no free camera extrinsic in the batch: one such window switches the whole batch's visual sweep to its extrinsic-column variant, another order of the same sums)
    b = SW.make_window(32, oracle)
    b["fix_poses"] = 1
    est = gf.Estimator(batch=4)
    solo = []
    for w_ in (a, b, c):
        x = w_.copy()
        solo.append((x, est.solve([x], 8)[0]))
    mixed = [a.copy(), b.copy(), c.copy(), a.copy()]
    sums = est.solve(mixed, 8)
    for k, (x, s_) in enumerate(solo):
        assert sums[k] == s_, (k, sums[k], s_)
        assert all(np.array_equal(mixed[k][key], x[key]) for key in gw.STATE_KEYS if key in x), k
    assert sums[3] == solo[0][1] and all(np.array_equal(mixed[3][key], solo[0][0][key]) for key in gw.STATE_KEYS if key in a)
    # and the stationary window is the dense form's result: the same bits as a handle without the switch
    monkeypatch.delenv("GF_BA_CHAIN")
    est_d = gf.Estimator(batch=1)
    y = b.copy()
    assert est_d.solve([y], 8)[0] == solo[1][1] and all(np.array_equal(y[key], solo[1][0][key]) for key in gw.STATE_KEYS if key in y)
    est.close(); est_d.close()


def test_cost_only_last_linearisation_changes_nothing_but_the_last_bits_of_the_cost(gf, oracle, monkeypatch):
    """Round 6: the candidate of a solve's last iteration is linearised cost-only (the closing step only accepts or rejects it).  Against GF_BA_COST_ONLY=0 -- every candidate
    in full -- the states are the same to the bit (no state depends on that linearisation), counts and termination are the same, and the costs agree to rounding (the residuals
    come out of a different instantiation of the same arithmetic)."""
    for seed, kw in ((1, {}), (3, {"fix_ex_pose": 0}), (5, {"use_wheel": False})):
        w0 = SW.make_window(seed, oracle, **kw)
        wa, wb = w0.copy(), w0.copy()
        est_a = gf.Estimator(batch=1)
        sa = est_a.solve([wa], 8)[0]
        monkeypatch.setenv("GF_BA_COST_ONLY", "0")
        est_b = gf.Estimator(batch=1)
        sb = est_b.solve([wb], 8)[0]
        monkeypatch.delenv("GF_BA_COST_ONLY")
        for k_ in ("iterations", "successful_steps", "termination"):
            assert sa[k_] == sb[k_], (seed, k_, sa, sb)
        assert abs(sa["final_cost"] - sb["final_cost"]) <= 1e-13 * sb["final_cost"]
        assert all(np.array_equal(wa[key], wb[key]) for key in gw.STATE_KEYS if key in wa), seed
        est_a.close(); est_b.close()
