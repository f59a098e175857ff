"""Initialisation while moving: the SfM branch of Estimator::initialStructure (estimator.cpp:1684-1847) -- solveRelativeRT_PNP
(initial/solve_5pts.cpp:244-277), GlobalSFM::constructWithDepth (initial/initial_sfm.cpp:379-594), the per-frame cv::solvePnP, visualInitialAlign
(estimator.cpp:1849-1926) with the depth variants of VisualIMUAlignment (initial/initial_aligment.cpp:427-653) -- in libgroundfusion_hip.so
(ground-fusion_amd/csrc/gf_init_sfm.hpp, gf_estimator.hip) against the numpy restatement (oracle/init_oracle.py, oracle/estimator_oracle.py).
Host code on both sides: no GPU needed (the optimisation behind initialStructure is switched off through the debug entry).  Parity is unpinned: the
OpenCV / Ceres pieces are restated from the published algorithms on both sides (oracle/init_oracle.py header); what is checked here is that two
independent implementations of that restatement agree, that the pieces recover known poses, and that the result is physically sane."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gfamd  # noqa: E402
import synth_stream as SS  # noqa: E402
import estimator_oracle as EO  # noqa: E402
import init_oracle as IO  # noqa: E402


def scene(seed, n=60, noise_px=0.0, planar=False):
    rng = np.random.default_rng(seed)
    X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(2.5, 7.0, n)], axis=1)
    if planar:
        X[:, 2] = 4.0
    rvec = rng.normal(0, 0.08, 3)
    tvec = np.array([0.05, -0.02, -0.45]) + rng.normal(0, 0.02, 3)
    uv = IO.project(rvec, tvec, X) + rng.normal(0, noise_px / 460.0, (n, 2))
    return X, uv, rvec, tvec


def product_pnp(est, X, uv, guess=None):
    g = np.zeros(6) if guess is None else np.concatenate(guess)
    out = est.debug("solvePnP", np.concatenate([[len(X), 0.0 if guess is None else 1.0], g, np.concatenate([X, uv], axis=1).reshape(-1)]))
    return bool(out[0]), out[1:4], out[4:7]


@pytest.fixture(scope="module")
def est():
    e = gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(tio=SS.TIO, rio=SS.RIO))
    yield e
    e.close()


def test_cv_rng_is_the_multiply_with_carry_generator():
    """cv::RNG (core/operations.hpp): state <- (uint32)state * 4164903690 + (state >> 32), output = low word; uniform(a, b) = a + next() % (b - a)"""
    r = IO.CvRNG()
    s = 0xFFFFFFFFFFFFFFFF
    for _ in range(5):
        s = ((s & 0xFFFFFFFF) * 4164903690 + (s >> 32)) & 0xFFFFFFFFFFFFFFFF
        assert r.next() == s & 0xFFFFFFFF
    r2 = IO.CvRNG()
    draws = [r2.uniform(0, 37) for _ in range(200)]
    assert min(draws) >= 0 and max(draws) < 37 and len(set(draws)) > 25


def test_epnp_and_iterative_pnp_recover_a_known_pose(est):
    X, uv, rvec, tvec = scene(1)
    r = IO.epnp(X[:5], uv[:5])                          # the minimal solver of the RANSAC: exact data, 5 points
    assert r is not None and np.abs(r[0] - rvec).max() < 1e-6 and np.abs(r[1] - tvec).max() < 1e-6
    r = IO.epnp(X, uv)
    assert np.abs(r[0] - rvec).max() < 1e-8 and np.abs(r[1] - tvec).max() < 1e-8
    ro = IO.solve_pnp_iterative(X, uv)                  # DLT start + Levenberg-Marquardt; points pass through float like cv::Point3f
    assert np.abs(ro[0] - rvec).max() < 1e-6 and np.abs(ro[1] - tvec).max() < 1e-6
    ok, rv, tv = product_pnp(est, X, uv)
    assert ok and np.abs(rv - ro[0]).max() < 1e-10 and np.abs(tv - ro[1]).max() < 1e-10
    guess = (rvec + 0.05, tvec + np.array([0.1, -0.1, 0.2]))
    rg = IO.solve_pnp_iterative(X, uv, guess)           # useExtrinsicGuess: Levenberg-Marquardt only
    ok, rv, tv = product_pnp(est, X, uv, guess)
    assert ok and np.abs(rv - rg[0]).max() < 1e-10 and np.abs(tv - rg[1]).max() < 1e-10
    assert np.abs(rg[0] - rvec).max() < 1e-6 and np.abs(rg[1] - tvec).max() < 1e-6


def test_analytic_jacobians_against_finite_differences():
    """independent of both implementations: the projection Jacobian of the pose refinement (d/d rvec through the derivative of the exponential map, d/d tvec)
    and the local Jacobian of the bundle adjustment's residual (quaternion perturbed on the left, as ceres::QuaternionParameterization::Plus does)"""
    X, uv, rvec, tvec = scene(6, n=12)
    _, J = IO.project(rvec, tvec, X, jac=True)
    h = 1e-6
    for k in range(6):
        d = np.zeros(6); d[k] = h
        num = (IO.project(rvec + d[:3], tvec + d[3:], X) - IO.project(rvec - d[:3], tvec - d[3:], X)).reshape(-1) / (2 * h)
        np.testing.assert_allclose(J[:, k], num, rtol=1e-6, atol=1e-8)
    q = np.asarray(EO.R_to_quat(IO.rodrigues(rvec)), float)
    p = X[0]

    def res(qq, tt, pp):
        v = IO.quat_rot(qq) @ pp + tt
        return np.array([v[0] / v[2], v[1] / v[2]])

    Rx = IO.quat_rot(q) @ p
    v = Rx + tvec
    D = np.array([[1 / v[2], 0, -v[0] / v[2] ** 2], [0, 1 / v[2], -v[1] / v[2] ** 2]])
    for k in range(3):
        d = np.zeros(3); d[k] = h
        num_r = (res(IO.quat_plus(q, d), tvec, p) - res(IO.quat_plus(q, -d), tvec, p)) / (2 * h)
        np.testing.assert_allclose((D @ (-2.0 * IO.skew(Rx)))[:, k], num_r, rtol=1e-6, atol=1e-8)
        num_t = (res(q, tvec + d, p) - res(q, tvec - d, p)) / (2 * h)
        np.testing.assert_allclose(D[:, k], num_t, rtol=1e-6, atol=1e-8)
        num_p = (res(q, tvec, p + d) - res(q, tvec, p - d)) / (2 * h)
        np.testing.assert_allclose((D @ IO.quat_rot(q))[:, k], num_p, rtol=1e-6, atol=1e-8)


def test_planar_point_sets_start_from_a_homography(est):
    X, uv, rvec, tvec = scene(2, planar=True)           # a wall: cvFindExtrinsicCameraParams2 starts from a homography instead of the DLT
    ro = IO.solve_pnp_iterative(X, uv)
    assert np.abs(ro[0] - rvec).max() < 1e-6 and np.abs(ro[1] - tvec).max() < 1e-6
    ok, rv, tv = product_pnp(est, X, uv)
    assert ok and np.abs(rv - ro[0]).max() < 1e-10 and np.abs(tv - ro[1]).max() < 1e-10
    X2 = X @ IO.rodrigues(np.array([0.3, -0.2, 0.1])).T + np.array([0.2, 0.1, 0.5])     # the same wall, tilted
    uv2 = IO.project(rvec, tvec, X2)
    ro = IO.solve_pnp_iterative(X2, uv2)
    assert np.abs(ro[0] - rvec).max() < 1e-6 and np.abs(ro[1] - tvec).max() < 1e-6
    ok, rv, tv = product_pnp(est, X2, uv2)
    assert ok and np.abs(rv - ro[0]).max() < 1e-10 and np.abs(tv - ro[1]).max() < 1e-10


@pytest.mark.parametrize("seed", [3, 4, 11])
def test_pnp_ransac_drops_outliers_and_matches_the_oracle(est, seed):
    """solveRelativeRT_PNP: correspondences of frame l (3-D, from depth) and the newest frame (3-D, used as 2-D), 1/460 reprojection threshold; a quarter of
    the pairs are gross mismatches"""
    X, uv, rvec, tvec = scene(seed, n=64, noise_px=0.2)
    rng = np.random.default_rng(100 + seed)
    bad = rng.choice(64, 16, replace=False)
    uv[bad] += rng.uniform(0.03, 0.2, (16, 2)) * rng.choice([-1, 1], (16, 2))
    r = IO.solve_pnp_ransac(X, uv)
    assert r is not None and set(r[2]) == set(range(64)) - set(bad)
    assert np.abs(r[0] - rvec).max() < 2e-3 and np.abs(r[1] - tvec).max() < 5e-3
    d2 = 3.0 + rng.uniform(0, 2, 64)                     # depth of the second observation (only its ratio is used)
    corres = np.concatenate([X, uv * d2[:, None], d2[:, None]], axis=1)
    out = est.debug("solveRelativeRT_PNP", corres.reshape(-1))
    rota = EO.ypr2R_deg_free(r[0])                       # Sophus::SO3(rx, ry, rz) of the reference: Rx Ry Rz of the Rodrigues components
    assert out[0] == 1.0
    np.testing.assert_allclose(out[1:10].reshape(3, 3), rota.T, atol=1e-9)
    np.testing.assert_allclose(out[10:13], -rota.T @ r[1], atol=1e-9)


def test_bundle_adjustment_restores_perturbed_cameras_and_points():
    """ceres::Solve of GlobalSFM: rotation of camera l, translations of cameras l and last held; everything else returns to the exact structure"""
    rng = np.random.default_rng(8)
    nf, l = 6, 2
    pts = np.stack([rng.uniform(-2, 2, 40), rng.uniform(-1.5, 1.5, 40), rng.uniform(3, 7, 40)], axis=1)
    qs, ts = [], []
    for i in range(nf):
        rv = np.array([0.0, 0.03 * (i - l), 0.0])
        R = IO.rodrigues(rv)
        qs.append(np.asarray(EO.R_to_quat(R), float)); ts.append(np.array([-0.1 * (i - l), 0.0, 0.0]))
    obs = []
    for i in range(nf):
        R = IO.quat_rot(qs[i])
        for j in range(len(pts)):
            p = R @ pts[j] + ts[i]
            obs.append((i, j, p[0] / p[2], p[1] / p[2]))
    q0 = [IO.quat_plus(q, rng.normal(0, 0.01, 3)) if i != l else q.copy() for i, q in enumerate(qs)]
    t0 = [t + rng.normal(0, 0.02, 3) if i not in (l, nf - 1) else t.copy() for i, t in enumerate(ts)]
    p0 = [p + rng.normal(0, 0.05, 3) for p in pts]
    q1, t1, p1, conv, cost = IO.sfm_bundle_adjust(q0, t0, p0, obs, const_rot={l}, const_trans={l, nf - 1})
    assert conv and cost < 1e-12
    assert np.array_equal(q1[l], qs[l]) and np.array_equal(t1[l], ts[l]) and np.array_equal(t1[nf - 1], ts[nf - 1])
    assert max(np.abs(IO.quat_rot(a) - IO.quat_rot(b)).max() for a, b in zip(q1, qs)) < 1e-6
    assert max(np.abs(a - b).max() for a, b in zip(t1, ts)) < 1e-5 and max(np.abs(a - b).max() for a, b in zip(p1, pts)) < 1e-4


def test_initialisation_from_nearly_coplanar_correspondences_w20():
    """W = 20: after 2 s of driving only the frame next to the newest one shares more than 20 tracks with it (l = 19, 21 correspondences, on the two walls of the
    scene), and the 5-point subsets RANSAC draws from such a set are close to coplanar: EPnP's 3 x 3 correlation matrix has a tiny singular value, and the rotation
    U V^T has to come from an SVD that resolves it.  Rounds 1-2 took it from the eigen-decomposition of M^T M in the library (error ~ cond^2): hypotheses 1e-8 ... 4e-2
    off, poses 3e-4 from the oracle, a bar of 2e-3 and a paragraph about noise.  scripts/epnp_mpmath_check.py compares both implementations with a 60-digit
    evaluation: the oracle's LAPACK route was right to 1e-10, the library was off; with a one-sided Jacobi SVD there (gf_init_sfm.hpp svd_rotation) both sit within
    4e-9 of the 60-digit hypotheses and the initialised windows agree to 1e-11."""
    st, eo, ep = run_to_init(5, 0.4, window_size=20)
    s, info, d = ep.state(), ep.debug("init_info"), eo.init_debug
    assert s["solver_flag"] == EO.NON_LINEAR and (int(info[0]), int(info[6])) == (d["l"], 0) and d["l"] == eo.W - 1
    assert int(info[1]) == d["n_tracked"]
    np.testing.assert_allclose(info[3:6], d["g_c0"], atol=1e-9)
    np.testing.assert_allclose(s["Rs"], np.array(eo.Rs), atol=1e-9)
    np.testing.assert_allclose(s["Ps"], np.array(eo.Ps), atol=1e-9)
    ep.close()


def test_epnp_on_nearly_coplanar_subsets_matches_the_oracle(est):
    """the hypotheses themselves: 5-point subsets of points on two walls meeting at a shallow angle, library (debug op "epnp") against the oracle's LAPACK route"""
    rng = np.random.default_rng(77)
    worst = 0.0
    for trial in range(20):
        u = rng.uniform(-1.5, 1.5, 5); v = rng.uniform(-1.0, 1.0, 5)
        X = np.stack([u, v, 4.0 + 0.02 * np.abs(u) + rng.normal(0, 1e-4, 5)], axis=1)   # two planes 1 degree apart
        rv, tv = np.array([0.02, -0.05, 0.01]), np.array([0.1, -0.02, 0.05])
        P = X @ IO.rodrigues(rv).T + tv
        uv = P[:, :2] / P[:, 2:3]
        X, uv = IO.f32(X), IO.f32(uv)
        with np.errstate(all="ignore"):
            mo = IO.epnp(X, uv)
        out = est.debug("epnp", np.concatenate([[5.0], np.concatenate([X, uv], axis=1).ravel()]))
        assert (mo is None) == (out[0] == 0)
        if mo is not None:
            worst = max(worst, float(np.abs(out[1:4] - mo[0]).max()), float(np.abs(out[4:7] - mo[1]).max()))
    assert worst < 1e-6, worst


def run_to_init(seed, yaw_turn, **kw):
    """a recording that begins in motion at constant speed: neither the stationary nor the wheel-activated shortcut fires, the window fills, initialStructure
    takes the SfM branch.  Both pipelines run it with the optimisation behind it switched off."""
    st = SS.Stream(seed, t_still=0.0, t_move=2.0 if kw.get("window_size", 10) <= 10 else 3.2, v_max=0.5, v_start=0.5, yaw_turn=yaw_turn)
    cfg = dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, **kw)
    eo = EO.Estimator(dict(cfg))
    ep = gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(**cfg))
    eo.optimization = lambda: None
    eo.slideWindow = lambda: None
    ep.debug("skip_solve", [1.0])
    tp, k = -1.0, 0
    while eo.solver_flag == EO.INITIAL:
        t1 = tp
        for e in (eo, ep):
            t1 = st.feed(e, k, tp)
        tp = t1
        frame = st.feature_frame(k)
        for e in (eo, ep):
            e.inputFeature(float(st.cam_t[k]), frame)
        k += 3
        assert k < 4.5 * (eo.W + 1) and k < len(st.cam_t), "initialStructure never succeeded"
    return st, eo, ep


@pytest.mark.parametrize("seed,yaw_turn,kw", [(5, 0.4, {}), (7, -0.6, {}), (9, 0.3, dict(use_wheel=0, wdetect=0))])
def test_initial_structure_sfm_branch_matches_oracle(seed, yaw_turn, kw):
    st, eo, ep = run_to_init(seed, yaw_turn, **kw)
    s, info, d = ep.state(), ep.debug("init_info"), eo.init_debug
    assert s["solver_flag"] == EO.NON_LINEAR and not eo.is_imu_excited and not eo.systemstationary      # reached through the SfM branch, not a shortcut
    assert (int(info[0]), int(info[1]), int(info[6])) == (d["l"], d["n_tracked"], 0) and d["n_tracked"] > 60
    np.testing.assert_allclose(info[3:6], d["g_c0"], atol=1e-9)             # gravity in the frame of camera l after the refinement
    assert abs(info[2] - d["x"][-1]) < 1e-9                                   # the `scale' of estimator.cpp:1871
    np.testing.assert_allclose(s["Ps"], np.array(eo.Ps), atol=1e-9)
    np.testing.assert_allclose(s["Rs"], np.array(eo.Rs), atol=1e-9)
    np.testing.assert_allclose(s["Vs"], np.array(eo.Vs), atol=1e-9)
    np.testing.assert_allclose(s["Bas"], np.array(eo.Bas), atol=1e-12)
    np.testing.assert_allclose(s["Bgs"], np.array(eo.Bgs), atol=1e-10)
    fo, fp = eo.f_manager.feature, ep.features()
    assert [f.feature_id for f in fo] == list(fp["id"])
    np.testing.assert_allclose(fp["estimated_depth"], np.array([f.estimated_depth for f in fo]), atol=1e-9)
    # physics: gravity has the configured norm and, seen from the body, the direction the accelerometer measures; the SfM track is the driven one
    np.testing.assert_allclose(eo.g, [0, 0, eo.cfg["g_norm"]], atol=1e-9)
    W = eo.W
    up_body = eo.Rs[W].T @ np.array([0, 0, 1.0])
    up_true = st.R_wb(eo.Headers[W]).T @ np.array([0, 0, 1.0])
    assert np.degrees(np.arccos(np.clip(up_body @ up_true, -1, 1))) < 3.0      # one second of data, accelerometer bias 0.02: observed 0.5 ... 1.3 degrees
    chord = [np.linalg.norm(d["T"][i] - d["T"][0]) for i in range(W + 1)]     # camera positions of the SfM, metric through the depth
    true = [np.linalg.norm(st.p_wb(eo.Headers[i]) - st.p_wb(eo.Headers[0])) for i in range(W + 1)]
    assert np.abs(np.array(chord) - np.array(true)).max() < 0.02
    v_true = np.linalg.norm(st._at(st._vw, eo.Headers[W]))
    if not eo.cfg["use_wheel"]:
        assert abs(np.linalg.norm(eo.Vs[W]) - v_true) < 0.1 * v_true         # LinearAlignmentWithDepth carries the SfM displacement in its right-hand side
    else:
        # LinearAlignmentWithWD has the displacement column of the IMU rows commented out (initial_aligment.cpp:527): at constant speed its velocities come
        # out near zero and the optimisation that follows has to find them (scripts/sfm_init_replay.py: 0.46 of 0.50 m/s after the first solve)
        assert np.linalg.norm(eo.Vs[W]) < 0.1 * v_true
    ep.close()
