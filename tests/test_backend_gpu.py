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
    oracle.ba_solve(w2o, 8); est.solve([w2g], 8)
    dp, dr = _pose_diff(w2o, w2g)
    assert dp < 1e-6 and dr < 1e-6, (dp, dr)
    est.close()
