"""Host half of the back end: FeatureManager and the processImage bookkeeping of libgroundfusion_hip.so
(ground-fusion_amd/csrc/gf_estimator.hip) against the numpy restatement oracle/estimator_oracle.py.  None of these paths touches the GPU:
the INITIAL fill phase and the single-step debug entry points never reach Estimator::optimization."""
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


def make_pair(seed, **kw):
    st = SS.Stream(seed, t_still=kw.pop("t_still", 0.15), t_move=kw.pop("t_move", 1.6), v_max=0.8)
    ocfg = dict(tio=SS.TIO, rio=SS.RIO, **kw)
    est_o = EO.Estimator(ocfg)
    est_p = gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(tio=SS.TIO, rio=SS.RIO, **kw))
    return st, est_o, est_p


STRIDE = 3   # every third 30 Hz camera frame reaches the estimator: ~1 s of motion inside one window


def feed(st, ests, k, tp):
    k = k * STRIDE
    t1 = tp
    for e in ests:
        t1 = st.feed(e, k, tp)
    frame = st.feature_frame(k)
    for e in ests:
        e.inputFeature(float(st.cam_t[k]), frame)
    return t1


def compare_features(est_o, est_p, tol=1e-12):
    fo, fp = est_o.f_manager.feature, est_p.features()
    assert [f.feature_id for f in fo] == list(fp["id"])
    assert [f.start_frame for f in fo] == list(fp["start_frame"])
    assert [len(f.feature_per_frame) for f in fo] == list(fp["n_obs"])
    assert [f.estimate_flag for f in fo] == list(fp["estimate_flag"])
    assert [f.solve_flag for f in fo] == list(fp["solve_flag"])
    do = np.array([f.estimated_depth for f in fo])
    np.testing.assert_allclose(fp["estimated_depth"], do, rtol=tol, atol=tol)


def compare_state(est_o, est_p, tol=1e-12):
    s = est_p.state()
    assert s["frame_count"] == est_o.frame_count and s["solver_flag"] == est_o.solver_flag and s["marginalization_flag"] == est_o.marginalization_flag
    assert bool(s["systemstationary"]) == bool(est_o.systemstationary)
    np.testing.assert_allclose(s["Ps"], np.array(est_o.Ps), atol=tol)
    np.testing.assert_allclose(s["Rs"], np.array(est_o.Rs), atol=tol)
    np.testing.assert_allclose(s["Vs"], np.array(est_o.Vs), atol=tol)
    np.testing.assert_allclose(s["Headers"], np.array(est_o.Headers), atol=0)
    return s


def fill_window(seed, **kw):
    st, est_o, est_p = make_pair(seed, **kw)
    tp = -1.0
    k = 0
    while est_o.frame_count < est_o.W:
        tp = feed(st, (est_o, est_p), k, tp)
        k += 1
    # observations of the newest frame (index W) without running processImage, which would call the (GPU) optimisation
    frame = st.feature_frame(k * STRIDE)
    flat = [float(est_o.W), 0.0]
    for fid in sorted(frame):
        flat += [float(fid)] + list(frame[fid])
    kf_p = est_p.debug("addFeature", flat)[0]
    kf_o = est_o.f_manager.addFeatureCheckParallax(est_o.W, frame, 0.0)
    assert bool(kf_p) == bool(kf_o)
    st.window_times = list(est_o.Headers[:est_o.W]) + [float(st.cam_t[k * STRIDE])]
    return st, est_o, est_p, k, tp


@pytest.mark.parametrize("seed", [1, 2])
def test_fill_phase_matches_oracle(seed):
    """INITIAL phase (estimator.cpp:1064-1073): keyframe votes, wheel dead-reckoning of Ps/Rs/Vs, feature list, stationarity votes."""
    st, est_o, est_p = make_pair(seed)
    tp = -1.0
    for k in range(10):
        tp = feed(st, (est_o, est_p), k, tp)
        s = compare_state(est_o, est_p)
        compare_features(est_o, est_p)
        assert s["last_track_num"] == est_o.f_manager.last_track_num and s["long_track_num"] == est_o.f_manager.long_track_num
        assert s["new_feature_num"] == est_o.f_manager.new_feature_num
        assert abs(s["last_average_parallax"] - est_o.f_manager.last_average_parallax) < 1e-9
    assert est_o.frame_count == 10 and np.linalg.norm(est_o.Ps[10]) > 0.05   # the vehicle moved: dead reckoning is exercised


def _seed_truth(st, est_o, est_p):
    """poses of the window frames from ground truth (what initialisation would deliver), NON_LINEAR"""
    W = est_o.W
    Ps = np.array([st.p_wb(t) for t in st.window_times])
    Rs = np.array([st.R_wb(t) for t in st.window_times])
    est_p.set_state(W, 1, Ps, Rs)
    est_o.solver_flag = 1
    est_o.Ps = [p.copy() for p in Ps]
    est_o.Rs = [r.copy() for r in Rs]


def test_triangulation_and_depth_bookkeeping():
    """FeatureManager::triangulateWithDepth / triangulate / setDepth / getDepthVector / removeFailures (FM:249-302, :669-799)."""
    st, est_o, est_p, k, tp = fill_window(3)
    _seed_truth(st, est_o, est_p)
    est_o.f_manager.triangulateWithDepth(est_o.Ps, est_o.Rs, est_o.tic, est_o.ric)
    est_p.debug("triangulateWithDepth")
    compare_features(est_o, est_p)
    flags = np.array([f.estimate_flag for f in est_o.f_manager.feature])
    assert (flags == 1).sum() > 20                     # near-wall points got a depth-camera depth
    est_o.f_manager.triangulate(est_o.Ps, est_o.Rs, est_o.tic, est_o.ric)
    est_p.debug("triangulate")
    compare_features(est_o, est_p, tol=1e-8)           # SVD (LAPACK) vs one-sided Jacobi: agree to the conditioning of the 4-column system
    flags = np.array([f.estimate_flag for f in est_o.f_manager.feature])
    assert (flags == 2).sum() > 5                      # far-wall points were triangulated
    dep_o = est_o.f_manager.getDepthVector()
    dep_p = est_p.debug("getDepthVector")
    np.testing.assert_allclose(dep_p, dep_o, rtol=1e-8)
    assert int(est_p.debug("getFeatureCount")[0]) == est_o.f_manager.getFeatureCount() == len(dep_o)
    x = dep_o.copy()
    x[::7] *= -1.0                                     # some negative inverse depths -> solve_flag 2
    est_o.f_manager.setDepth(x)
    est_p.debug("setDepth", x)
    compare_features(est_o, est_p, tol=1e-8)
    est_o.f_manager.removeFailures()
    est_p.debug("removeFailures")
    compare_features(est_o, est_p, tol=1e-8)


# ---- the two per-feature sweeps against the reference's formulas in other arithmetic (mpmath, 60 digits): used here on the host loops, in test_featsweep_gpu.py on the kernels
def sweep_window(est_o, est_p):
    """the estimator's window as plain arrays: poses from the library's state, observations from the (identical) feature list of the numpy oracle"""
    s, fp = est_p.state(), est_p.features()
    fo = est_o.f_manager.feature
    assert [f.feature_id for f in fo] == list(fp["id"])
    obs = [np.array([[fr.point[0], fr.point[1], fr.point[2], fr.depth] for fr in f.feature_per_frame]) for f in fo]
    return dict(Rs=s["Rs"].copy(), Ps=s["Ps"].copy(), tic=est_o.tic.copy(), ric=est_o.ric.copy(), start_frame=fp["start_frame"].copy(), obs=obs,
                estimated_depth=fp["estimated_depth"].copy(), estimate_flag=fp["estimate_flag"].copy(), ids=fp["id"].copy())


def _mp():
    mp = pytest.importorskip("mpmath")
    mp.mp.dps = 60
    return mp, (lambda a: mp.matrix([[mp.mpf(float(v)) for v in row] for row in np.atleast_2d(a)])), (lambda a: mp.matrix([mp.mpf(float(v)) for v in np.asarray(a).reshape(-1)]))


def exact_triangulate_with_depth(w, depth_threshold, init_depth):
    """FeatureManager::triangulateWithDepth (feature_manager.cpp:726-799) transcribed: per feature (depth, flag, margin) -- `margin` is how far the closest accept / reject
    decision (|residual| < 10 / 460, depth in [0.1, threshold], average < 0.1) sits from its threshold, so that a caller can tell a rounding tie from a wrong result"""
    mp, Mx, Vx = _mp()
    tic, ric = Vx(np.asarray(w["tic"]).reshape(-1)[:3]), Mx(np.asarray(w["ric"]).reshape(-1)[:9].reshape(3, 3))
    Rs, Ps = [Mx(R) for R in w["Rs"]], [Vx(P) for P in w["Ps"]]
    out = []
    for f, obs in enumerate(w["obs"]):
        dep, flag = float(w["estimated_depth"][f]), int(w["estimate_flag"][f])
        if len(obs) < 4 or dep > 0:
            out.append((dep, flag, np.inf))
            continue
        s = int(w["start_frame"][f])
        tr, Rr = Ps[s] + Rs[s] * tic, Rs[s] * ric
        ver, margin = [], mp.inf
        for i in range(len(obs)):
            d = float(obs[i][3])
            if d < 0.1 or d > depth_threshold:
                continue
            t0, R0 = Ps[s + i] + Rs[s + i] * tic, Rs[s + i] * ric
            point0 = Vx(obs[i][:3]) * mp.mpf(d)
            t2r, R2r = Rr.T * (t0 - tr), Rr.T * R0
            for j in range(len(obs)):
                if i == j:
                    continue
                t1, R1 = Ps[s + j] + Rs[s + j] * tic, Rs[s + j] * ric
                t20, R20 = R0.T * (t1 - t0), R0.T * R1
                pp = R20.T * point0 - R20.T * t20
                rx, ry = mp.mpf(float(obs[j][0])) - pp[0] / pp[2], mp.mpf(float(obs[j][1])) - pp[1] / pp[2]
                res = mp.sqrt(rx * rx + ry * ry)
                margin = min(margin, abs(res - mp.mpf(10) / 460))
                if res < mp.mpf(10) / 460:
                    ver.append((R2r * point0 + t2r)[2])
        if not ver:
            out.append((dep, flag, float(margin)))
            continue
        ave = sum(ver) / len(ver)
        margin = min(margin, abs(ave - mp.mpf("0.1")))
        out.append((float(ave), 1, float(margin)) if ave >= mp.mpf("0.1") else (float(init_depth), 0, float(margin)))
    return out


def exact_moving_consistency(w, focal_length):
    """Estimator::movingConsistencyCheckW with reprojectionError / reprojectionError3D (estimator.cpp:3888-3907, :3968-4012) transcribed: per feature (removed, margin)"""
    mp, Mx, Vx = _mp()
    W = len(w["Ps"]) - 1
    tic, ric = Vx(np.asarray(w["tic"]).reshape(-1)[:3]), Mx(np.asarray(w["ric"]).reshape(-1)[:9].reshape(3, 3))
    Rs, Ps = [Mx(R) for R in w["Rs"]], [Vx(P) for P in w["Ps"]]
    out = []
    for f, obs in enumerate(w["obs"]):
        s, depth = int(w["start_frame"][f]), mp.mpf(float(w["estimated_depth"][f]))
        if not (len(obs) >= 2 and s < W - 2) or depth < 0:
            out.append((False, np.inf))
            continue
        uvi = Vx(obs[0][:3])
        err = err3 = mp.mpf(0)
        cnt = 0
        for k in range(1, len(obs)):
            j = s + k
            uvj = Vx(obs[k][:3])
            pts_w = Rs[s] * (ric * (depth * uvi) + tic) + Ps[s]
            pc = ric.T * (Rs[j].T * (pts_w - Ps[j]) - tic)
            rx, ry = pc[0] / pc[2] - uvj[0], pc[1] / pc[2] - uvj[1]
            err += mp.sqrt(rx * rx + ry * ry)
            dv = pc - uvj
            err3 += mp.sqrt(dv[0] ** 2 + dv[1] ** 2 + dv[2] ** 2) / depth
            cnt += 1
        if cnt == 0:
            out.append((False, np.inf))
            continue
        a, b = mp.mpf(float(focal_length)) * err / cnt, err3 / cnt
        out.append((bool(a > 10 or b > 2), float(min(abs(a - 10), abs(b - 2)))))
    return out


def check_depth_sweep(w, got_depth, got_flag, depth_threshold, init_depth, tol=1e-12):
    ex = exact_triangulate_with_depth(w, depth_threshold, init_depth)
    worst = 0.0
    for f, (d, fl, margin) in enumerate(ex):
        if margin < 1e-9:       # a decision within rounding of its threshold: either answer is the reference's
            continue
        assert int(got_flag[f]) == fl, (f, int(got_flag[f]), fl, margin)
        dev = abs(float(got_depth[f]) - d) / max(1.0, abs(d))
        assert dev < tol, (f, float(got_depth[f]), d)
        worst = max(worst, dev)
    return worst, sum(1 for e in ex if e[1] == 1)


def test_host_sweeps_meet_the_reference_formulas_at_60_digits():
    """triangulateWithDepth and movingConsistencyCheckW of the library's host code (no GPU, no oracle in the comparison) against the transcriptions above"""
    c = gfamd.default_estimator_cfg()
    for seed in (3, 5):
        st, est_o, est_p, k, tp = fill_window(seed)
        _seed_truth(st, est_o, est_p)
        w0 = sweep_window(est_o, est_p)
        est_p.debug("triangulateWithDepth")
        fp = est_p.features()
        worst, n1 = check_depth_sweep(w0, fp["estimated_depth"], fp["estimate_flag"], c.depth_threshold, c.init_depth)
        assert n1 > 20
        est_p.debug("triangulate")
        est_o.f_manager.triangulateWithDepth(est_o.Ps, est_o.Rs, est_o.tic, est_o.ric); est_o.f_manager.triangulate(est_o.Ps, est_o.Rs, est_o.tic, est_o.ric)
        w2 = sweep_window(est_o, est_p)
        removed = set(int(i) for i in est_p.debug("movingConsistencyCheckW"))
        ex = exact_moving_consistency(w2, c.focal_length)
        for f, (rem, margin) in enumerate(ex):
            if margin > 1e-9:
                assert (int(w2["ids"][f]) in removed) == rem, (f, rem, margin)
        print("seed %d: %d depths within %.1e of the 60-digit average; movingConsistencyCheckW decisions equal on %d tracks" % (seed, n1, worst, len(ex)))


@pytest.mark.parametrize("flag", [0, 1])
def test_slide_window_and_feedback(flag):
    """slideWindow (EST:3638-3837) with removeBackShiftDepth / removeFront, movingConsistencyCheckW and predictPtsInNextFrame (EST:3862-3995)."""
    st, est_o, est_p, k, tp = fill_window(4)
    _seed_truth(st, est_o, est_p)
    for e in (est_o.f_manager,):
        e.triangulateWithDepth(est_o.Ps, est_o.Rs, est_o.tic, est_o.ric)
        e.triangulate(est_o.Ps, est_o.Rs, est_o.tic, est_o.ric)
    est_p.debug("triangulateWithDepth")
    est_p.debug("triangulate")
    # corrupt a few depths so the consistency check has something to reject
    x = est_o.f_manager.getDepthVector()
    x[::9] *= 6.0
    est_o.f_manager.setDepth(x)
    est_p.debug("setDepth", x)
    rem_o = set()
    est_o.movingConsistencyCheckW(rem_o)
    rem_p = est_p.debug("movingConsistencyCheckW")
    assert sorted(rem_o) == [int(v) for v in rem_p] and len(rem_o) > 3
    est_o.predictPts = {}
    est_o.predictPtsInNextFrame()
    pp = est_p.debug("predictPtsInNextFrame").reshape(-1, 4)
    assert [int(v) for v in pp[:, 0]] == sorted(est_o.predictPts) and len(pp) > 50
    np.testing.assert_allclose(pp[:, 1:], np.array([est_o.predictPts[i] for i in sorted(est_o.predictPts)]), rtol=1e-9, atol=1e-9)
    est_o.marginalization_flag = flag
    est_o.slideWindow()
    est_p.debug("slideWindow", [flag])
    compare_features(est_o, est_p, tol=1e-8)
    s = est_p.state()
    np.testing.assert_allclose(s["Ps"], np.array(est_o.Ps), atol=1e-12)
    np.testing.assert_allclose(s["Headers"], np.array(est_o.Headers), atol=0)
    assert (s["sum_of_back"], s["sum_of_front"]) == (est_o.sum_of_back, est_o.sum_of_front)


def test_remove_back_shift_depth_meets_the_reference_formula_at_60_digits():
    """FeatureManager::removeBackShiftDepth (feature_manager.cpp:818-856) of the library's host code against the formula in mpmath: a track that starts in the dropped frame
    moves its depth into its next frame's camera; a track left with one observation goes; every other track only renumbers its start frame"""
    mp, Mx, Vx = _mp()
    c = gfamd.default_estimator_cfg()
    st, est_o, est_p, k, tp = fill_window(4)
    _seed_truth(st, est_o, est_p)
    for name in ("triangulateWithDepth", "triangulate"):
        est_p.debug(name)
        getattr(est_o.f_manager, name)(est_o.Ps, est_o.Rs, est_o.tic, est_o.ric)
    w = sweep_window(est_o, est_p)
    ric, tic = np.asarray(w["ric"]).reshape(-1)[:9].reshape(3, 3), np.asarray(w["tic"]).reshape(-1)[:3]
    # slideWindowOld hands in the camera poses of frames 0 and 1 (estimator.cpp:3721-3740); this recording stands still there, so the second pose is taken from the middle
    # of the window (the vehicle has moved and turned by then): the formula does not care which two poses it gets
    m = len(w["Ps"]) // 2
    R0, P0, R1, P1 = w["Rs"][0] @ ric, w["Ps"][0] + w["Rs"][0] @ tic, w["Rs"][m] @ ric, w["Ps"][m] + w["Rs"][m] @ tic
    assert np.linalg.norm(P1 - P0) > 0.05
    est_p.debug("removeBackShiftDepth", list(R0.reshape(-1)) + list(P0) + list(R1.reshape(-1)) + list(P1))
    fp = est_p.features()
    want = []
    for f, obs in enumerate(w["obs"]):
        if int(w["start_frame"][f]) != 0:
            want.append((int(w["ids"][f]), int(w["start_frame"][f]) - 1, len(obs), float(w["estimated_depth"][f])))
        elif len(obs) - 1 >= 2:
            pts_i = Vx(obs[0][:3]) * mp.mpf(float(w["estimated_depth"][f]))
            pts_j = Mx(R1).T * (Mx(R0) * pts_i + Vx(P0) - Vx(P1))
            want.append((int(w["ids"][f]), 0, len(obs) - 1, float(pts_j[2]) if pts_j[2] > 0 else float(c.init_depth)))
    assert [int(i) for i in fp["id"]] == [t[0] for t in want] and [int(v) for v in fp["start_frame"]] == [t[1] for t in want] and [int(v) for v in fp["n_obs"]] == [t[2] for t in want]
    dev = np.abs(fp["estimated_depth"] - np.array([t[3] for t in want])) / np.maximum(1.0, np.abs([t[3] for t in want]))
    shifted = sum(1 for f, obs in enumerate(w["obs"]) if int(w["start_frame"][f]) == 0 and len(obs) >= 3)
    moved = np.abs(fp["estimated_depth"] - np.array([w["estimated_depth"][f] for f, obs in enumerate(w["obs"]) if int(w["start_frame"][f]) != 0 or len(obs) >= 3])).max()
    assert shifted > 10 and dev.max() < 1e-13 and moved > 0.01, (shifted, dev.max(), moved)
    print("removeBackShiftDepth: %d shifted depths within %.1e of the 60-digit value, %d tracks dropped" % (shifted, dev.max(), len(w["obs"]) - len(want)))


def test_predict_pts_in_next_frame_meets_the_reference_formula_at_60_digits():
    """Estimator::predictPtsInNextFrame (estimator.cpp:3839-3886: constant-velocity next pose curT (prevT^-1 curT), the track's first observation at its depth carried into the
    predicted camera) of the library's host code against the formula in mpmath (the 4 x 4 inverse exact)"""
    mp, Mx, Vx = _mp()
    st, est_o, est_p, k, tp = fill_window(4)
    _seed_truth(st, est_o, est_p)
    for name in ("triangulateWithDepth", "triangulate"):
        est_p.debug(name)
        getattr(est_o.f_manager, name)(est_o.Ps, est_o.Rs, est_o.tic, est_o.ric)
    w = sweep_window(est_o, est_p)
    W = len(w["Ps"]) - 1
    got = est_p.debug("predictPtsInNextFrame").reshape(-1, 4)

    def T4(R, P):
        T = mp.eye(4)
        for r in range(3):
            for c in range(3):
                T[r, c] = mp.mpf(float(R[r, c]))
            T[r, 3] = mp.mpf(float(P[r]))
        return T
    curT, prevT = T4(w["Rs"][W], w["Ps"][W]), T4(w["Rs"][W - 1], w["Ps"][W - 1])
    nextT = curT * (prevT ** -1 * curT)
    nR, nP = nextT[0:3, 0:3], nextT[0:3, 3]
    ric, tic = Mx(np.asarray(w["ric"]).reshape(-1)[:9].reshape(3, 3)), Vx(np.asarray(w["tic"]).reshape(-1)[:3])
    want = {}
    for f, obs in enumerate(w["obs"]):
        s0, dep = int(w["start_frame"][f]), float(w["estimated_depth"][f])
        if dep > 0 and len(obs) >= 2 and s0 + len(obs) - 1 == W:
            pts_j = ric * (Vx(obs[0][:3]) * mp.mpf(dep)) + tic
            pts_w = Mx(w["Rs"][s0]) * pts_j + Vx(w["Ps"][s0])
            pc = ric.T * (nR.T * (pts_w - nP) - tic)
            want[int(w["ids"][f])] = [float(v) for v in pc]
    assert [int(v) for v in got[:, 0]] == sorted(want) and len(want) > 50
    E = np.array([want[i] for i in sorted(want)])
    dev = np.abs(got[:, 1:] - E).max() / np.abs(E).max()
    assert dev < 1e-13
    moved = float(mp.sqrt(sum((nP[r] - mp.mpf(float(w["Ps"][W][r]))) ** 2 for r in range(3))))
    assert moved > 0.01          # the predicted pose is not the current one: the formula is exercised
    print("predictPtsInNextFrame: %d predictions within %.1e of the 60-digit value (next pose %.3f m ahead)" % (len(want), dev, moved))


def test_triangulate_meets_the_exact_singular_vector_at_60_digits():
    """FeatureManager::triangulate (feature_manager.cpp:669-723): the depth is v[2] / v[3] of the right singular vector of the smallest singular value of the 2n x 4 system
    of all observations of a track.  Here that vector is the eigenvector of A^T A by mpmath's eigsy at 60 digits -- no Jacobi sweeps, no LAPACK -- and both the library's
    one-sided Jacobi SVD and the oracle's numpy SVD are measured against it (they agree with each other to 1e-8 in test_triangulation_and_depth_bookkeeping: this says where
    the truth lies between them)"""
    mp, Mx, Vx = _mp()
    c = gfamd.default_estimator_cfg()
    st, est_o, est_p, k, tp = fill_window(3)
    _seed_truth(st, est_o, est_p)
    est_p.debug("triangulateWithDepth")
    est_o.f_manager.triangulateWithDepth(est_o.Ps, est_o.Rs, est_o.tic, est_o.ric)
    w = sweep_window(est_o, est_p)
    est_p.debug("triangulate")
    est_o.f_manager.triangulate(est_o.Ps, est_o.Rs, est_o.tic, est_o.ric)
    fp = est_p.features()
    dep_o = np.array([f.estimated_depth for f in est_o.f_manager.feature])
    ric, tic = Mx(np.asarray(w["ric"]).reshape(-1)[:9].reshape(3, 3)), Vx(np.asarray(w["tic"]).reshape(-1)[:3])
    Rs, Ps = [Mx(R) for R in w["Rs"]], [Vx(P) for P in w["Ps"]]
    worst_p = worst_o = 0.0
    n = 0
    for f, obs in enumerate(w["obs"]):
        if float(w["estimated_depth"][f]) > 0 or len(obs) < 4:
            continue
        s0 = int(w["start_frame"][f])
        t0, R0 = Ps[s0] + Rs[s0] * tic, Rs[s0] * ric
        A = mp.zeros(2 * len(obs), 4)
        for i in range(len(obs)):
            t1, R1 = Ps[s0 + i] + Rs[s0 + i] * tic, Rs[s0 + i] * ric
            t, R = R0.T * (t1 - t0), R0.T * R1
            P = mp.zeros(3, 4)
            Rt = R.T
            mRt_t = -(Rt * t)
            for r in range(3):
                for cc in range(3):
                    P[r, cc] = Rt[r, cc]
                P[r, 3] = mRt_t[r]
            pt = Vx(obs[i][:3])
            fn = pt / mp.sqrt(pt[0] ** 2 + pt[1] ** 2 + pt[2] ** 2)
            for cc in range(4):
                A[2 * i, cc] = fn[0] * P[2, cc] - fn[2] * P[0, cc]
                A[2 * i + 1, cc] = fn[1] * P[2, cc] - fn[2] * P[1, cc]
        E, Q = mp.eigsy(A.T * A)
        j = min(range(4), key=lambda q: E[q])
        exact = Q[2, j] / Q[3, j]
        if exact < mp.mpf("0.1"):
            assert int(fp["estimate_flag"][f]) == 0 and fp["estimated_depth"][f] == c.init_depth
            continue
        assert int(fp["estimate_flag"][f]) == 2
        worst_p = max(worst_p, abs(float(fp["estimated_depth"][f]) - float(exact)) / float(exact))
        worst_o = max(worst_o, abs(float(dep_o[f]) - float(exact)) / float(exact))
        n += 1
    assert n > 5 and worst_p < 1e-8 and worst_o < 1e-8, (n, worst_p, worst_o)
    print("triangulate: %d depths; library's Jacobi SVD within %.1e of the exact singular vector, the oracle's LAPACK SVD within %.1e" % (n, worst_p, worst_o))


def test_remove_back_initial_phase():
    """slideWindowOld in the INITIAL phase uses removeBack (FM:858-874), no depth shift."""
    st, est_o, est_p, k, tp = fill_window(5)
    est_o.f_manager.removeBack()
    est_p.debug("removeBack")
    compare_features(est_o, est_p)
    est_o.f_manager.removeFront(est_o.W)
    est_p.debug("removeFront", [est_o.W])
    compare_features(est_o, est_p)


def test_rejects_unbuilt_configurations():
    with pytest.raises(gfamd.GfError):
        gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(estimate_extrinsic=2))
    with pytest.raises(gfamd.GfError):
        gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(window_size=40))


def test_group_argument_checks_and_no_cpu_fallback():
    """gf_estimator_group_*: argument errors are reported before any device work; without a GPU creation fails loudly (the shared
    back-end handle needs the device), it never degrades to a host solver"""
    with pytest.raises(gfamd.GfError) as e:
        gfamd.EstimatorGroup(gfamd.default_estimator_cfg(), 0)
    assert "1 <= n" in str(e.value)
    with pytest.raises(gfamd.GfError) as e:
        gfamd.EstimatorGroup(gfamd.default_estimator_cfg(with_tracker=1), 2)
    assert "with_tracker" in str(e.value)
    if gfamd.device_count() == 0:
        with pytest.raises(gfamd.GfError) as e:
            gfamd.EstimatorGroup(gfamd.default_estimator_cfg(), 2)
        assert "no HIP device" in str(e.value) or "HIP" in str(e.value)


def test_empty_and_sparse_feature_frames():
    """ragged inputs: frames without a single observation (vision failure) and frames with a handful of features, mixed into the fill
    phase -- keyframe votes, track bookkeeping and propagated states stay equal to the oracle's"""
    st, est_o, est_p = make_pair(4)
    tp, k = -1.0, 0
    while est_o.frame_count < est_o.W:
        kk = k * STRIDE
        for e in (est_o, est_p):
            t1 = st.feed(e, kk, tp)
        tp = t1
        frame = st.feature_frame(kk)
        if k in (2, 5):
            frame = {}
        elif k in (3, 6):
            frame = {i: frame[i] for i in sorted(frame)[:3]}
        for e in (est_o, est_p):
            e.inputFeature(float(st.cam_t[kk]), frame)
        compare_state(est_o, est_p)
        compare_features(est_o, est_p)
        k += 1
    assert est_o.frame_count == est_o.W


def test_gnss_intake_during_the_fill_phase_matches_oracle():
    """Estimator::inputGNSS / getGNSSInterval / processGNSS (estimator.cpp:397-404, :476-510, :1455-1535) while the window fills (no optimisation, no
    GPU): epochs older than the frame by more than 0.1 s are thrown away, the front epoch is taken whatever its age, the last taken epoch is
    processed again for every frame that finds the queue empty (gnss_msg is a member), satellites are admitted after gnss_track_num_thres good
    epochs in a row and fall back to zero on one bad psr_std, other constellations are dropped."""
    kw = dict(gnss_enable=1, gnss_track_num_thres=2, gnss_local_time_diff=18.0)
    st, est_o, est_p = make_pair(4, **kw)
    st.gnss_setup(sats_per_sys=2, n_low=1)
    rng = np.random.default_rng(5)
    tp, stale = -1.0, 0
    for k in range(10):
        tk = float(st.cam_t[k * STRIDE])
        epochs = []
        if k == 0:                                      # two epochs from long before the first frame: thrown away (EST:489-497)
            epochs += [st.gnss_epoch(tk - 0.9), st.gnss_epoch(tk - 0.5)]
        if k not in (3, 4, 8):                          # frames 3, 4 and 8 find the queue empty and reuse the previous epoch (EST:656-660)
            epochs.append(st.gnss_epoch(tk + rng.uniform(-0.03, 0.03), flaky_sat=101 if k == 5 else None))
        if k == 6:
            epochs[-1][1].append(dict(epochs[-1][1][0], sat=900, sys=-1))   # a QZSS / SBAS satellite: filtered by system (EST:1463-1465)
        for tg, ep in epochs:
            est_o.inputGNSS(tg, ep)
            est_p.inputGNSS(tg, ep)
        if k == 7:
            est_o.inputGNSSTimeDiff(18.0)
            est_p.inputGNSSTimeDiff(18.0)
        tp = feed(st, (est_o, est_p), k, tp)
        buf = est_p.debug("gnss_meas_buf")
        got, q = [], 0
        for _ in range(est_o.W + 1):
            n = int(buf[q])
            got.append([int(x) for x in buf[q + 1:q + 1 + n]])
            q += 1 + n
        assert got == [[o["sat"] for o in b] for b in est_o.gnss_meas_buf], k
        ts = est_p.debug("sat_track_status")
        assert {int(a): int(b) for a, b in zip(ts[0::2], ts[1::2])} == est_o.sat_track_status, k
        assert est_p.gnss_state()["queued"] == len(est_o.GNSSBuf)
        stale += int(k in (3, 4, 8))
    sizes = [len(b) for b in est_o.gnss_meas_buf]
    assert sizes[0] == 0 and sizes[1] > 0 and sizes[3] == sizes[2] and max(sizes) == 9   # admitted from the second epoch on; reused epochs fill frames 3, 4, 8
    assert est_o.sat_track_status[101] < est_o.sat_track_status[102]                 # the flaky satellite started over
    assert 900 not in est_o.sat_track_status and not est_p.gnss_state()["gnss_ready"]
    # inputs the boundary refuses
    with pytest.raises(gfamd.GfError):
        est_p.inputGNSS(1.0, [])
    plain = gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg())
    with pytest.raises(gfamd.GfError):
        plain.inputGNSS(*st.gnss_epoch(0.5))
    with pytest.raises(gfamd.GfError):
        gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(gnss_enable=1, gnss_ddt_sigma=0.0))


def _compare_latest(est_o, est_p, tol):
    l = est_p.latest()
    for k, v in (("time", est_o.latest_time), ("P", est_o.latest_P), ("Q", est_o.latest_Q), ("V", est_o.latest_V), ("time_wheel", est_o.latest_time_wheel),
                 ("P_wheel", est_o.latest_P_wheel), ("Q_wheel", est_o.latest_Q_wheel), ("V_wheel", est_o.latest_V_wheel)):
        np.testing.assert_allclose(l[k], v, rtol=0, atol=tol * max(1.0, float(np.abs(v).max())), err_msg=k)


def test_latest_states_at_sensor_rate_match_oracle():
    """latest_P / latest_Q / latest_V and the wheel counterparts (estimator.h:239-242, :354-356): fastPredictIMU inside inputIMU (estimator.cpp:332, :4014-4028),
    fastPredictWheel inside inputWheel (:363, :4079-4093; it integrates the IMU's latest_gyr_0, as written there) and the re-anchoring of
    updateLatestStates (:4141-4198: newest window state, then every sample still queued), checked after EVERY sample the way pubLatestOdometry would read them."""
    st, est_o, est_p, k, tp = fill_window(5)
    _seed_truth(st, est_o, est_p)
    est_o.updateLatestStates()
    est_p.debug("updateLatestStates")
    _compare_latest(est_o, est_p, 1e-13)
    assert est_o.latest_time == est_o.accBuf[-1][0] and est_o.latest_time_wheel == est_o.wheelVelBuf[-1][0]   # propagated through the queued samples
    ev = [(float(t), 0, i) for i, t in enumerate(st.imu_t) if tp < t <= tp + 0.3] + [(float(t), 1, i) for i, t in enumerate(st.wheel_t) if tp < t <= tp + 0.3]
    n = 0
    for t, kind, i in sorted(ev):
        for e in (est_o, est_p):
            if kind == 0:
                e.inputIMU(t, st.imu_acc[i], st.imu_gyr[i])
            else:
                e.inputWheel(t, st.wheel_vel[i], st.wheel_gyr[i])
        _compare_latest(est_o, est_p, 1e-12)
        n += 1
    assert n > 60 and np.linalg.norm(est_o.latest_P - est_o.Ps[est_o.frame_count]) > 1e-3 and np.linalg.norm(est_o.latest_P_wheel) > 0
    # the samples received since are still queued: re-anchoring replays them and lands on the same state
    before = est_p.latest()
    est_o.updateLatestStates()
    est_p.debug("updateLatestStates")
    _compare_latest(est_o, est_p, 1e-12)
    np.testing.assert_allclose(est_p.latest()["P"], before["P"], atol=1e-9)
    est_p.close()
