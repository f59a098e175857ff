"""The per-feature sweeps of the measurement side on the device: gf_triangulate_with_depth_batch and gf_moving_consistency_batch against the
host loops of the library (FeatureManager::triangulateWithDepth, feature_manager.cpp:726-799; Estimator::movingConsistencyCheckW, estimator.cpp:3955-3995) --
depths bit for bit, the same ids -- and against the numpy oracle."""
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gfamd  # noqa: E402
import test_estimator_host as TH  # noqa: E402

pytestmark = pytest.mark.gpu


def window_of(est_o, est_p):
    """the estimator's window as the batched entry points take it: poses from the library's state, observations from the (identical) feature list of the oracle"""
    s, fp = est_p.state(), est_p.features()
    fo = est_o.f_manager.feature
    assert [f.feature_id for f in fo] == list(fp["id"])
    obs = [np.array([[fr.point[0], fr.point[1], fr.point[2], fr.depth] for fr in f.feature_per_frame]) for f in fo]
    return dict(Rs=s["Rs"].copy(), Ps=s["Ps"].copy(), tic=est_o.tic.copy(), ric=est_o.ric.copy(), start_frame=fp["start_frame"].copy(), obs=obs,
                estimated_depth=fp["estimated_depth"].copy(), estimate_flag=fp["estimate_flag"].copy(), ids=fp["id"].copy())


def test_device_sweeps_match_the_host_loops_bit_for_bit():
    fs = gfamd.FeatureSweeps()
    wins, after, removed = [], [], []
    for seed in (3, 4, 5):
        st, est_o, est_p, k, tp = TH.fill_window(seed)
        TH._seed_truth(st, est_o, est_p)
        w0 = window_of(est_o, est_p)
        est_p.debug("triangulateWithDepth")
        est_o.f_manager.triangulateWithDepth(est_o.Ps, est_o.Rs, est_o.tic, est_o.ric)
        w1 = window_of(est_o, est_p)
        est_p.debug("triangulate")                      # the remaining tracks, so that movingConsistencyCheckW sees depths everywhere
        est_o.f_manager.triangulate(est_o.Ps, est_o.Rs, est_o.tic, est_o.ric)
        w2 = window_of(est_o, est_p)
        ids_host = set(int(i) for i in est_p.debug("movingConsistencyCheckW"))
        rem_o = set()
        est_o.movingConsistencyCheckW(rem_o)
        assert ids_host == rem_o
        wins.append((w0, w1, w2)); removed.append(ids_host)
    c = gfamd.default_estimator_cfg()
    out = fs.triangulate_with_depth([w[0] for w in wins], c.depth_threshold, c.init_depth)
    for (w0, w1, w2), (dep, flag) in zip(wins, out):
        assert np.array_equal(dep, w1["estimated_depth"]) and np.array_equal(flag, w1["estimate_flag"])      # bit for bit, three windows in one launch
        assert (flag == 1).sum() > 20
    rem = fs.moving_consistency([w[2] for w in wins], c.focal_length)
    for (w0, w1, w2), r, ids in zip(wins, rem, removed):
        assert set(int(i) for i in w2["ids"][r != 0]) == ids
    # a window whose newest poses are off by 30-40 cm: the check must throw tracks out, the same ones on both sides
    st, est_o, est_p, k, tp = TH.fill_window(6)
    TH._seed_truth(st, est_o, est_p)
    Ps = np.array(est_o.Ps); Ps[-5:] += np.array([0.35, -0.2, 0.0])
    est_p.set_state(est_o.W, 1, Ps, np.array(est_o.Rs)); est_o.Ps = [p.copy() for p in Ps]
    for e in (est_p,):
        e.debug("triangulateWithDepth"); e.debug("triangulate")
    est_o.f_manager.triangulateWithDepth(est_o.Ps, est_o.Rs, est_o.tic, est_o.ric); est_o.f_manager.triangulate(est_o.Ps, est_o.Rs, est_o.tic, est_o.ric)
    w = window_of(est_o, est_p)
    ids_host = set(int(i) for i in est_p.debug("movingConsistencyCheckW"))
    r = fs.moving_consistency([w], c.focal_length)[0]
    assert len(ids_host) > 5 and set(int(i) for i in w["ids"][r != 0]) == ids_host
    fs.close()


def test_device_sweeps_meet_the_reference_formulas_at_60_digits():
    """the two kernels against FeatureManager::triangulateWithDepth / Estimator::movingConsistencyCheckW transcribed into mpmath (tests/test_estimator_host.py): neither the
    library's host loops nor the oracle take part in the comparison"""
    fs = gfamd.FeatureSweeps()
    c = gfamd.default_estimator_cfg()
    for seed in (3, 5):
        st, est_o, est_p, k, tp = TH.fill_window(seed)
        TH._seed_truth(st, est_o, est_p)
        w0 = TH.sweep_window(est_o, est_p)
        dep, flag = fs.triangulate_with_depth([w0], c.depth_threshold, c.init_depth)[0]
        worst, n1 = TH.check_depth_sweep(w0, dep, flag, c.depth_threshold, c.init_depth)
        assert n1 > 20
        for e in (est_p,):
            e.debug("triangulateWithDepth"); e.debug("triangulate")
        est_o.f_manager.triangulateWithDepth(est_o.Ps, est_o.Rs, est_o.tic, est_o.ric); est_o.f_manager.triangulate(est_o.Ps, est_o.Rs, est_o.tic, est_o.ric)
        w2 = TH.sweep_window(est_o, est_p)
        rem = fs.moving_consistency([w2], c.focal_length)[0]
        ex = TH.exact_moving_consistency(w2, c.focal_length)
        for f, (r, margin) in enumerate(ex):
            if margin > 1e-9:
                assert bool(rem[f]) == r, (f, r, margin)
        print("seed %d: %d device depths within %.1e of the 60-digit average; decisions equal on %d tracks" % (seed, n1, worst, len(ex)))
    fs.close()


def test_throughput_of_the_sweeps():
    """256 windows of one camera frame in one launch each: kernel time (hipEvents), call time, and the host loop on one core"""
    st, est_o, est_p, k, tp = TH.fill_window(3)
    TH._seed_truth(st, est_o, est_p)
    w0 = window_of(est_o, est_p)
    t0 = time.perf_counter()
    for _ in range(20):
        est_p.set_state(est_o.W, 1, np.array(est_o.Ps), np.array(est_o.Rs))
        est_p.debug("triangulateWithDepth")
    host_tri = (time.perf_counter() - t0) / 20
    est_p.debug("triangulate")
    t0 = time.perf_counter()
    for _ in range(20):
        est_p.debug("movingConsistencyCheckW")
    host_mcc = (time.perf_counter() - t0) / 20
    w2 = window_of(est_o, est_p)
    fs = gfamd.FeatureSweeps()
    c = gfamd.default_estimator_cfg()
    B = 256
    fs.triangulate_with_depth([w0] * B, c.depth_threshold, c.init_depth); fs.moving_consistency([w2] * B, c.focal_length)
    k0 = fs.stats()["kernel_ms"]
    fs.triangulate_with_depth([w0] * B, c.depth_threshold, c.init_depth)
    k1 = fs.stats()["kernel_ms"]
    fs.moving_consistency([w2] * B, c.focal_length)
    k2 = fs.stats()["kernel_ms"]
    print("feature sweeps of %d windows x %d features: triangulateWithDepth kernel %.3f ms (host loop, one window, one core, with the debug call around it: %.3f ms), "
          "movingConsistencyCheckW kernel %.3f ms (host: %.3f ms)" % (B, len(w0["ids"]), k1 - k0, host_tri * 1e3, k2 - k1, host_mcc * 1e3))
    assert k1 - k0 < 2.0 and k2 - k1 < 1.0
    fs.close()
