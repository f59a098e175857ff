"""Back-end part of __graft_entry__.smoke(): one small window through the HIP solver + marginalisation, checked against the oracle."""
import numpy as np


def run():
    import gfamd
    import oracle_py
    import synth_window as SW
    w0 = SW.make_window(21, oracle_py, max_features=60, n_landmarks=90)
    wo, wg = w0.copy(), w0.copy()
    so = oracle_py.ba_solve(wo, 4)
    est = gfamd.Estimator(max_features=60, max_visual=600)
    sg = est.solve([wg], 4)[0]
    assert sg["iterations"] == so["iterations"] and sg["successful_steps"] == so["successful_steps"]
    pa, pb = wo["para_Pose"].reshape(-1, 7), wg["para_Pose"].reshape(-1, 7)
    assert np.abs(pa[:, :3] - pb[:, :3]).max() < 1e-6, "positions differ from the oracle"
    assert min(np.abs(pa[:, 3:] - pb[:, 3:]).max(), np.abs(pa[:, 3:] + pb[:, 3:]).max()) < 5e-7, "rotations differ from the oracle"
    pg = est.marginalize([wg], 0)[0]
    po = oracle_py.ba_marginalize(wo, 0)
    assert pg is not None and np.array_equal(pg["block_id"], po["block_id"]) and pg["n"] == po["n"]
    est.close()
    print("backend smoke ok: %d iterations, cost %.3f -> %.3f, prior n=%d" % (sg["iterations"], sg["initial_cost"], sg["final_cost"], pg["n"]))


def run_estimator():
    """Estimator::processImage call surface: a short synthetic RGB-D + IMU + wheel feature stream (stationary start, initialisation, first
    NON_LINEAR windows) through gf_estimator_* on the device against the numpy/C++ oracle pipeline."""
    import gfamd
    import estimator_oracle as EO
    import synth_stream as SS
    st = SS.Stream(1, t_still=1.0, t_move=1.2, v_max=0.4, yaw0=0.0, yaw_turn=-0.3, split_x=1.8, turn_delay=0.5)
    st._lm = st._landmarks(1600)
    st._pn = np.random.default_rng(4001).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
    est_p = gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(tio=SS.TIO, rio=SS.RIO, multiple_thread=1))
    est_o = EO.Estimator(dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1))
    tp = -1.0
    for k in range(len(st.cam_t)):
        for e in (est_o, est_p):
            t1 = st.feed(e, k, tp)
        tp = t1
        if k % 2:
            continue
        frame = st.feature_frame(k)
        est_p.inputFeature(float(st.cam_t[k]), frame)
        est_o.inputFeature(float(st.cam_t[k]), frame)
    s = est_p.state()
    assert s["solver_flag"] == est_o.solver_flag == 1 and s["n_optimizations"] == est_o.n_optimizations > 5
    assert [f.feature_id for f in est_o.f_manager.feature] == list(est_p.features()["id"])
    dp = float(np.abs(s["Ps"] - np.array(est_o.Ps)).max())
    assert dp < 5e-6, dp   # sanity bound of the smoke run; the 1e-6 bar is held by tests/test_estimator_gpu.py
    est_p.close()
    print("estimator smoke ok: %d optimisations, %d tracks, |dP| %.1e, newest position %.3f m" % (s["n_optimizations"], s["n_features"], dp, float(np.linalg.norm(s["Ps"][-1]))))
