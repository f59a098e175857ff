"""End-to-end synthetic replay (SURVEY.md §8c fixture vi): FeatureTracker::trackImage + Estimator::processImage on the HIP path
(gf_estimator_* -> gf_tracker_* / gf_ba_*) against the CPU oracle pipeline (oracle tracker + estimator_oracle + oracle BA) fed with the same
seeded RGB-D + IMU + wheel stream.  Bars: bit-exact feature ids at every frame, identical keyframe / marginalisation / stationarity decisions
and iteration counts, window poses within 1e-6 m / 1e-6 rad at every frame of the closed loop.

The stream starts at rest in front of the near wall (every tracked point carries a depth-camera depth) and sees the far wall only once it
moves: free-depth features without parallax make the reference's own solve noise-driven (the Schur block of such a feature is ~1e-27, its
step is a ratio of two rounding errors that the dogleg then rescales the whole step by), so no two implementations -- nor two builds of
Ceres -- agree there.  DESIGN.md, section "What parity can and cannot mean"."""
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
import oracle_py as O  # noqa: E402

pytestmark = pytest.mark.gpu


def make_stream(seed, t_move=3.0):
    return SS.Stream(seed, t_still=1.5, t_move=t_move, v_max=0.4, yaw0=0.0, yaw_turn=-0.6, split_x=1.8, turn_delay=0.8)


def rot_angle(Ra, Rb):
    """largest rotation angle between corresponding rotations [rad]"""
    out = 0.0
    for a, b in zip(Ra, Rb):
        c = (np.trace(a.T @ b) - 1.0) / 2.0
        s = np.linalg.norm(a.T @ b - (a.T @ b).T) / (2.0 * np.sqrt(2.0))
        out = max(out, float(np.arctan2(s, c)))
    return out


def compare_frame(est_o, est_p, worst, tag, latest_tol=1e-5):
    s = est_p.state()
    assert s["frame_count"] == est_o.frame_count and s["solver_flag"] == est_o.solver_flag, tag
    assert s["marginalization_flag"] == est_o.marginalization_flag and bool(s["systemstationary"]) == bool(est_o.systemstationary), tag
    fp = est_p.features()
    fo = est_o.f_manager.feature
    assert [f.feature_id for f in fo] == list(fp["id"]), tag                           # bit-exact ids, same list order
    assert [f.start_frame for f in fo] == list(fp["start_frame"]) and [len(f.feature_per_frame) for f in fo] == list(fp["n_obs"]), tag
    assert [f.estimate_flag for f in fo] == list(fp["estimate_flag"]), tag
    if est_o.last_summary is not None:
        assert s["iterations"] == est_o.last_summary["iterations"] and s["successful_steps"] == est_o.last_summary["successful_steps"], tag
    worst["p"] = max(worst["p"], float(np.abs(s["Ps"] - np.array(est_o.Ps)).max()))
    worst["r"] = max(worst["r"], rot_angle(s["Rs"], est_o.Rs))
    worst["v"] = max(worst["v"], float(np.abs(s["Vs"] - np.array(est_o.Vs)).max()))
    # what pubLatestOdometry / pubWheelLatestOdometry read (updateLatestStates estimator.cpp:4141-4198 after the frame, fastPredictIMU / fastPredictWheel per sample)
    lt = est_p.latest()
    assert abs(lt["time"] - est_o.latest_time) < 1e-12 and abs(lt["time_wheel"] - est_o.latest_time_wheel) < 1e-12, tag
    dev = {kk: float(np.abs(lt[kk] - vv).max()) for kk, vv in (("P", est_o.latest_P), ("V", est_o.latest_V), ("Q", est_o.latest_Q), ("P_wheel", est_o.latest_P_wheel),
                                                               ("V_wheel", est_o.latest_V_wheel), ("Q_wheel", est_o.latest_Q_wheel))}
    for kk, vv in dev.items():
        worst["latest_" + kk] = max(worst.get("latest_" + kk, 0.0), vv)
    worst["bias"] = max(worst.get("bias", 0.0), float(np.abs(s["Bas"] - np.array(est_o.Bas)).max()), float(np.abs(s["Bgs"] - np.array(est_o.Bgs)).max()))
    d_tio = float(np.abs(s["tio"] - est_o.tio).max())
    worst["tio"] = max(worst.get("tio", 0.0), d_tio)
    # With identical observations the propagated states sit at the pose level (1e-8 ... 3e-6; bound latest_tol).  latest_P_wheel = Rs tio + Ps carries the
    # translation of the wheel extrinsic, the weakest direction of these windows (1e-5 apart where the poses agree to 5e-8): it is held to that deviation.
    # With the tracker feedback of multiple_thread: 0 the two front ends already differ below LK's own 0.01 px and the accelerometer bias shows in
    # latest_V = Vs + dt (R (acc - Ba) - g): that test passes its own bound.
    assert max(dev["P"], dev["V"], dev["Q"], dev["V_wheel"], dev["Q_wheel"]) < latest_tol and dev["P_wheel"] < 2.0 * d_tio + latest_tol, (tag, dev, d_tio)
    return s


def test_replay_feature_frames_matches_oracle():
    """processImage driven by projected landmarks (no images): stationary initialisation, NON_LINEAR hand-over, both marginalisation kinds."""
    st = make_stream(1)
    st._lm = st._landmarks(1600)
    st._pn = np.random.default_rng(4001).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
    est_p = gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(tio=SS.TIO, rio=SS.RIO, multiple_thread=1))
    est_o = EO.Estimator(dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1))
    tp, worst, seen = -1.0, dict(p=0.0, r=0.0, v=0.0), set()
    for k in range(len(st.cam_t)):
        for e in (est_o, est_p):
            t1 = st.feed(e, k, tp)
        tp = t1
        if k % 2:
            continue
        frame = st.feature_frame(k)
        est_p.inputFeature(float(st.cam_t[k]), frame)
        est_o.inputFeature(float(st.cam_t[k]), frame)
        s = compare_frame(est_o, est_p, worst, "frame %d" % k)
        seen.add((s["solver_flag"], s["marginalization_flag"], s["systemstationary"]))
    assert est_o.solver_flag == EO.NON_LINEAR and est_o.n_optimizations > 40
    assert {(1, 0, 0), (1, 1, 0), (1, 1, 1)} <= seen                       # keyframes, non-keyframes and stationary frames all occurred
    assert np.linalg.norm(est_o.Ps[-1]) > 0.5                              # it really drove away
    assert any(f.estimate_flag == 2 for f in est_o.f_manager.feature)      # far-wall points were triangulated (free inverse depths)
    print("feature-frame replay worst deviation", worst)
    assert worst["p"] < 1e-6 and worst["r"] < 1e-6, worst


@pytest.mark.parametrize("use_wheel", [1, 0])
def test_replay_from_a_moving_start_initialises_through_sfm(use_wheel):
    """the recording begins in motion at constant speed, so neither the stationary nor the wheel-activated shortcut of initialStructure fires
    (estimator.cpp:1604-1682) and the window is initialised by the SfM branch (:1684-1847: solveRelativeRT_PNP, GlobalSFM::constructWithDepth, solvePnP per
    frame, visualInitialAlign; tests/test_init_sfm_host.py covers the pieces).  From there the replay is the usual closed loop: decisions identical, poses
    within 1e-6 of the oracle pipeline at every frame -- and, since the reference's alignment leaves the positions collapsed (`s' of estimator.cpp:1871) and,
    with the wheel, the velocities near zero, the optimisation has to pull the window back to the driven track, which it does.
    (Over 17 seeded recordings of this kind, scripts/sfm_init_sweep.py, the two pipelines stay within 6e-7 on 16 and end 5e-6 apart on one: the first
    marginalisation after such an initialisation is ill-determined in double precision, two eigen-solvers on the reference's own route differ by more there --
    DESIGN.md section 2.)"""
    st = SS.Stream(5, t_still=0.0, t_move=4.0, v_max=0.5, v_start=0.5, yaw_turn=0.4)
    st._lm = st._landmarks(1600)
    st._pn = np.random.default_rng(4005).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
    kw = dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, use_wheel=use_wheel, wdetect=use_wheel)
    est_p = gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(**kw))
    est_o = EO.Estimator(dict(kw))
    tp, worst, chord = -1.0, dict(p=0.0, r=0.0, v=0.0), []
    for k in range(len(st.cam_t)):
        for e in (est_o, est_p):
            t1 = st.feed(e, k, tp)
        tp = t1
        if k % 3:
            continue
        frame = st.feature_frame(k)
        est_p.inputFeature(float(st.cam_t[k]), frame)
        est_o.inputFeature(float(st.cam_t[k]), frame)
        compare_frame(est_o, est_p, worst, "frame %d" % k)
        if est_o.solver_flag == EO.NON_LINEAR:
            W = est_o.W
            chord.append(abs(np.linalg.norm(est_o.Ps[W - 1] - est_o.Ps[0]) - np.linalg.norm(st.p_wb(est_o.Headers[W - 1]) - st.p_wb(est_o.Headers[0]))))
    info = est_p.debug("init_info")
    assert est_o.solver_flag == EO.NON_LINEAR and not est_o.is_imu_excited and est_o.init_debug["n_tracked"] == int(info[1]) > 60 and int(info[6]) == 0
    assert est_o.n_optimizations > 25
    print("moving-start replay (use_wheel %d) worst deviation" % use_wheel, worst, "window chord error first / last %.3f / %.3f m" % (chord[0], chord[-1]))
    assert worst["p"] < 1e-6 and worst["r"] < 1e-6, worst
    assert chord[-1] < 0.05 and chord[-1] < chord[0]          # 0.5 m of chord: the window has found the driven track again
    est_p.close()


@pytest.mark.parametrize("variant", ["use_mcc", "estimate_td", "wheel_slip", "subset_cam", "subset_wheel"])
def test_replay_configurations_of_the_other_shipped_yaml_files(variant):
    """what the m2dgrp.yaml replays leave untouched: use_mcc: 1 (groundchallenge.yaml:10, idc_rs.yaml:13 -- the consistency check's outliers now reach
    removeOutlier and the tracker feedback, estimator.cpp:1104-1134), estimate_td: 1 (td becomes a free block once the vehicle moves, :3097-3100), and
    a wheel-slip segment (`wdetect`: |dP_wheel - dP_imu| > 0.02 raises wheelanomaly, the wheel factors of that frame are skipped in the solve and in
    the marginalisation, :633, :3132-3136, :3370), and the PoseSubsetParameterization masks every shipped file selects (`extrinsic_type: 3` = ADJUST_CAM_NO_Z with
    `estimate_extrinsic: 1`, :2969-2985; the same type on the wheel extrinsic, :3010-3026): the block becomes free once the vehicle moves, its z never does.
    Same bars as the other replays."""
    st = make_stream(2)
    st._lm = st._landmarks(1600)
    st._pn = np.random.default_rng(4002).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
    kw = dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1)
    if variant == "use_mcc":
        kw["use_mcc"] = 1
    if variant == "estimate_td":
        kw["estimate_td"] = 1
    if variant == "subset_cam":
        kw.update(estimate_extrinsic=1, extrinsic_type=3)
    if variant == "subset_wheel":
        kw.update(estimate_wheel_extrinsic=1, extrinsic_type_wheel=3, wdetect=0)   # (with wdetect this stream's wheel factors are dropped as anomalies once it moves)
    if variant == "wheel_slip":      # the odometer over-reports for 0.5 s in the middle of the drive (wheel spin)
        sel = (st.wheel_t > 3.0) & (st.wheel_t < 3.5)
        st.wheel_vel[sel] *= 1.6
    est_p = gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(**kw))
    est_o = EO.Estimator(dict(kw))
    tp, worst, td_free = -1.0, dict(p=0.0, r=0.0, v=0.0), 0
    for k in range(len(st.cam_t)):
        for e in (est_o, est_p):
            t1 = st.feed(e, k, tp)
        tp = t1
        if k % 2:
            continue
        frame = st.feature_frame(k)
        est_p.inputFeature(float(st.cam_t[k]), frame)
        est_o.inputFeature(float(st.cam_t[k]), frame)
        compare_frame(est_o, est_p, worst, "%s frame %d" % (variant, k))
        s = est_p.state()
        assert abs(s["td"] - est_o.td) < 1e-6      # seconds; observed 7e-9 on a td of 12 ms
        td_free += int(abs(est_o.td) > 0)
        if variant == "subset_cam":
            # z never moves; y is the window's weakest direction -- on this stream the solver lets tic wander by 2.3 m in three seconds (no prior worth the name on a
            # freshly freed block), and the two pipelines end 2e-5 apart in it (relative; the poses of the same frames agree to 1e-7): bar 1e-4 relative.
            # Round 5: one step of such a window against the same step at 60 digits (scripts/adjudicate_free_extrinsic_step.py, tests/golden `first_step`): the scaled,
            # damped system has condition 3e8 (its smallest eigenvalue IS the mu = 1e-8 damping: the direction is not observed at all), a single step of either
            # implementation is within 1e-9 m of the exact one -- 2.5e-8 of the step -- and a closed loop of ~40 such solves on a block that moves by metres compounds it.
            assert s["tic"][2] == 0.0 == est_o.tic[2] and np.abs(s["tic"] - est_o.tic).max() < 1e-4 * max(1.0, np.abs(est_o.tic).max())
        if variant == "subset_wheel":
            assert s["tio"][2] == SS.TIO[2] == est_o.tio[2]
    assert est_o.solver_flag == EO.NON_LINEAR and est_o.n_optimizations > 30
    if variant == "subset_cam":
        assert est_o.openExEstimation and np.abs(est_o.tic[:2]).max() > 1e-3      # the block was free and moved where the mask lets it
    if variant == "subset_wheel":
        assert est_o.openExWheelEstimation and np.abs(est_o.tio[:2] - np.asarray(SS.TIO)[:2]).max() > 1e-3
    if variant == "estimate_td":
        assert td_free > 5          # td really was estimated
    if variant == "wheel_slip":
        assert getattr(est_o, "n_wheel_anomaly", 0) > 0   # the slip was detected and wheel factors were dropped
    print("replay %s worst deviation" % variant, worst)
    assert worst["p"] < 1e-6 and worst["r"] < 1e-6, worst
    est_p.close()


def _image_replay(multiple_thread, t_move):
    st = make_stream(1, t_move)
    cfg = gfamd.default_estimator_cfg(tio=SS.TIO, rio=SS.RIO, multiple_thread=multiple_thread, with_tracker=1)
    cfg.tracker = gfamd.default_cfg()
    est_p = gfamd.SlidingWindowEstimator(cfg)
    est_o = EO.Estimator(dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=multiple_thread), tracker=O.Tracker())
    tp, worst = -1.0, dict(p=0.0, r=0.0, v=0.0)
    for k in range(len(st.cam_t)):
        for e in (est_o, est_p):
            t1 = st.feed(e, k, tp)
        tp = t1
        img, dep = st.image(k)
        fp = est_p.inputImage(float(st.cam_t[k]), img, dep)
        ids_o, obs_o = est_o.inputImage(float(st.cam_t[k]), img, dep)
        assert sorted(fp) == sorted(int(i) for i in ids_o), "tracker ids differ at image %d" % k
        for j, i in enumerate(ids_o):
            if multiple_thread:   # no feedback: the front end never sees the back end's floating-point result -> bit-exact
                assert np.array_equal(fp[int(i)], obs_o[j]), "tracker observation of id %d differs at image %d" % (i, k)
            else:                 # predicted start pixels are float32 roundings of poses that agree to ~1e-8: a rounding may flip by one ulp and
                #                   LK then stops (its 0.01 px criterion) at a slightly different sub-pixel position
                np.testing.assert_allclose(fp[int(i)][3:5], obs_o[j][3:5], rtol=0, atol=0.05, err_msg="id %d image %d" % (i, k))
        if multiple_thread and (k + 1) % 2 != 0:
            continue
        compare_frame(est_o, est_p, worst, "image %d" % k, latest_tol=1e-5 if multiple_thread else 1e-4)
    assert est_o.solver_flag == EO.NON_LINEAR and np.linalg.norm(est_o.Ps[-1]) > 0.3
    print("image replay (multiple_thread=%d) worst deviation" % multiple_thread, worst)
    return worst


def test_replay_images_closed_loop():
    """trackImage + processImage, MULTIPLE_THREAD data flow (every second image reaches the back end, no tracker feedback; m2dgrp.yaml:118)."""
    worst = _image_replay(1, 3.0)
    assert worst["p"] < 1e-6 and worst["r"] < 1e-6, worst


def test_replay_images_with_tracker_feedback():
    """multiple_thread: 0 -- setPrediction / removeOutliers feed the back end's result into the next trackImage (estimator.cpp:1132-1136):
    ids stay bit-exact; tracked pixels may differ below LK's own 0.01 px termination criterion because the predicted start pixels are float32
    roundings of poses that agree to ~1e-8 only."""
    worst = _image_replay(0, 2.0)
    assert worst["p"] < 1e-6 and worst["r"] < 1e-6, worst


@pytest.mark.parametrize("device_preint,group_threads,device_sweeps,use_mcc", [(False, None, False, 0), (True, None, False, 0), (False, 1, False, 0), (False, 3, False, 0),
                                                                               (True, 2, False, 0), (False, None, True, 0), (False, 2, True, 1), (True, None, True, 1)])
def test_group_of_sequences_on_one_batched_solver(device_preint, group_threads, device_sweeps, use_mcc, monkeypatch):
    """gf_estimator_group_*: four sequences keep the reference's per-sequence control flow on host threads while their solves and
    marginalisations reach the device as one batch; every member must end up where a stand-alone Estimator on the same inputs does.
    device_preint: the IMU intervals of a step are integrated by one device launch instead of on the members' threads -- bit-identical
    intervals (tests/test_preint_gpu.py), so the members must land on exactly the same bits as without it.
    device_sweeps (8(f)4 as well): triangulateWithDepth and movingConsistencyCheckW of all members as one launch each per step (before the solve with use_mcc, after
    it without) -- depths, flags and removed ids are bit-identical to the host loops (tests/test_featsweep_gpu.py), so again the same bits.
    group_threads: the members are user-level contexts on a pool of worker threads (GF_GROUP_THREADS); with 1 thread all four share one worker and hand it to
    each other at every rendezvous, with 3 the split is uneven -- the bits may not depend on it."""
    if group_threads is not None:
        monkeypatch.setenv("GF_GROUP_THREADS", str(group_threads))
    n = 4
    streams = []
    for s in range(n):
        st = SS.Stream(1 + s, t_still=1.5, t_move=1.6, v_max=0.4, yaw0=0.0, yaw_turn=-0.4 + 0.2 * s, split_x=1.8, turn_delay=0.6)
        st._lm = st._landmarks(900)
        st._pn = np.random.default_rng(4100 + s).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
        streams.append(st)
    cfg = gfamd.default_estimator_cfg(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, use_mcc=use_mcc)
    grp = gfamd.EstimatorGroup(cfg, n, device_preint=device_preint, device_sweeps=device_sweeps)
    solo = [gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, use_mcc=use_mcc)) for _ in range(n)]
    tp = [-1.0] * n
    nk = min(len(st.cam_t) for st in streams)
    worst = 0.0
    for k in range(nk):
        for s, st in enumerate(streams):
            st.feed(grp.members[s], k, tp[s])
            tp[s] = st.feed(solo[s], k, tp[s])
        if k % 2:
            continue
        frames = [st.feature_frame(k) for st in streams]
        grp.inputFeatures(list(range(n)), [float(st.cam_t[k]) for st in streams], frames)
        for s in range(n):
            solo[s].inputFeature(float(streams[s].cam_t[k]), frames[s])
            a, b = grp.members[s].state(), solo[s].state()
            assert a["frame_count"] == b["frame_count"] and a["solver_flag"] == b["solver_flag"] and a["marginalization_flag"] == b["marginalization_flag"], (k, s)
            assert a["iterations"] == b["iterations"] and a["successful_steps"] == b["successful_steps"], (k, s)
            assert list(grp.members[s].features()["id"]) == list(solo[s].features()["id"]), (k, s)
            worst = max(worst, float(np.abs(a["Ps"] - b["Ps"]).max()), rot_angle(a["Rs"], b["Rs"]))
    st = grp.stats()
    print("group of %d: worst deviation from stand-alone estimators %.2e; %s" % (n, worst, st))
    assert all(m.state()["solver_flag"] == 1 for m in grp.members)
    assert worst == 0.0                                                        # same kernels, same order of operations, same bits
    assert st["largest_batch"] == n and st["windows"] > 2 * st["batches"]      # the solves really went out together
    grp.close()
    for e in solo:
        e.close()


def test_group_submit_and_wait_on_a_padded_table():
    """gf_estimator_group_submit_features / _wait: inputFeature's two halves (estimator.cpp:447-459 queues the frame, processMeasurements works on it) with the
    observations taken from a padded table as a batched tracker leaves it (stride = its capacity).  Same bits as gf_estimator_group_input_features on the frames
    back to back; a second submit before the wait, and a frame longer than the stride, are refused without touching the step in flight."""
    n, cap = 3, 160
    streams = []
    for s in range(n):
        st = SS.Stream(11 + s, t_still=1.5, t_move=1.2, v_max=0.4, yaw0=0.0, yaw_turn=0.3 - 0.3 * s, split_x=1.8, turn_delay=0.6)
        st._lm = st._landmarks(900)
        st._pn = np.random.default_rng(5200 + s).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
        streams.append(st)
    cfg = gfamd.default_estimator_cfg(tio=SS.TIO, rio=SS.RIO, multiple_thread=1)
    ga, gb = gfamd.EstimatorGroup(cfg, n), gfamd.EstimatorGroup(cfg, n)
    tp = [-1.0] * n
    nk = min(len(st.cam_t) for st in streams)
    refused = 0
    for k in range(nk):
        for s, st in enumerate(streams):
            st.feed(ga.members[s], k, tp[s])
            tp[s] = st.feed(gb.members[s], k, tp[s])
        if k % 2:
            continue
        frames = [st.feature_frame(k) for st in streams]
        ts = [float(st.cam_t[k]) for st in streams]
        ga.inputFeatures(list(range(n)), ts, frames)
        table = np.zeros((n, cap), gfamd.OBS_DTYPE)
        table["id"] = -7                      # what lies beyond a frame's count is never looked at
        cnt = np.zeros(n, np.int32)
        for s, im in enumerate(frames):
            ids = sorted(im)[:cap]
            cnt[s] = len(ids)
            table["id"][s, :len(ids)] = ids
            table["v"][s, :len(ids)] = np.array([np.asarray(im[i], np.float64).reshape(-1)[:8] for i in ids]).reshape(len(ids), 8)
            assert len(ids) == len(im)
        gb.submitFeatures(np.arange(n), ts, table, cnt, stride=cap)
        if k == 4:
            with pytest.raises(gfamd.GfError, match="in flight"):
                gb.submitFeatures(np.arange(n), ts, table, cnt, stride=cap)
            refused += 1
        gb.wait()
        gb.wait()                             # nothing in flight: returns at once
        for s in range(n):
            a, b = ga.members[s].state(), gb.members[s].state()
            assert (a["frame_count"], a["solver_flag"], a["marginalization_flag"], a["iterations"]) == (b["frame_count"], b["solver_flag"], b["marginalization_flag"], b["iterations"]), (k, s)
            assert np.array_equal(a["Ps"], b["Ps"]) and np.array_equal(a["Rs"], b["Rs"]) and np.array_equal(a["Vs"], b["Vs"]), (k, s)
            assert gb.members[s].flags() == (b["frame_count"], b["solver_flag"], b["marginalization_flag"])
    assert refused == 1 and all(m.flags()[1] == 1 for m in gb.members)
    with pytest.raises(gfamd.GfError, match="stride"):
        gb.submitFeatures(np.arange(n), ts, table, np.full(n, cap + 1, np.int32), stride=cap)
    gb.wait()
    ga.close(); gb.close()


def test_group_at_the_largest_configuration_w20_500_features(monkeypatch):
    """BASELINE.json configs[4]'s size as group members (round-3 advisor): 20-frame windows with up to 500 tracked features each run whole frames -- window build,
    pack_slot of ~9 000 visual factors, initialStructure, the batched launches of whoever closes a rendezvous -- on the members' fiber stacks (GF_GROUP_STACK_KB,
    8 MB by default, pages touched on use), two members sharing one worker thread.  Same bits as stand-alone estimators."""
    monkeypatch.setenv("GF_GROUP_THREADS", "1")
    n = 2
    streams = []
    for s in range(n):
        st = SS.Stream(21 + s, t_still=1.5, t_move=2.2, v_max=0.4, yaw0=0.0, yaw_turn=-0.4 + 0.5 * s, split_x=1.8, turn_delay=0.6)
        st._lm = st._landmarks(2600)
        st._pn = np.random.default_rng(4500 + s).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
        streams.append(st)
    kw = dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, window_size=20, max_features=1100, max_visual=12000)
    grp = gfamd.EstimatorGroup(gfamd.default_estimator_cfg(**kw), n)
    solo = [gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(**kw)) for _ in range(n)]
    tp, worst, most = [-1.0] * n, 0.0, 0
    for k in range(min(len(st.cam_t) for st in streams)):
        for s, st in enumerate(streams):
            st.feed(grp.members[s], k, tp[s])
            tp[s] = st.feed(solo[s], k, tp[s])
        if k % 2:
            continue
        frames = [st.feature_frame(k) for st in streams]
        grp.inputFeatures(list(range(n)), [float(st.cam_t[k]) for st in streams], frames)
        for s in range(n):
            solo[s].inputFeature(float(streams[s].cam_t[k]), frames[s])
            a, b = grp.members[s].state(), solo[s].state()
            assert a["frame_count"] == b["frame_count"] and a["solver_flag"] == b["solver_flag"] and a["iterations"] == b["iterations"], (k, s)
            worst = max(worst, float(np.abs(a["Ps"] - b["Ps"]).max()), rot_angle(a["Rs"], b["Rs"]))
            most = max(most, len(grp.members[s].features()["id"]))
    print("W = 20 group: up to %d features per window, worst deviation from stand-alone %.2e, %s" % (most, worst, grp.stats()))
    assert all(m.state()["solver_flag"] == 1 and m.state()["frame_count"] == 20 for m in grp.members) and most >= 400
    assert worst == 0.0
    grp.close()
    for e in solo:
        e.close()


def test_group_members_initialise_through_sfm():
    """three recordings that begin in motion as members of one group: each member's SfM initialisation (host code on its own thread) and the batched solves
    behind it must land on the bits a stand-alone estimator produces"""
    n = 3
    streams = []
    for s in range(n):
        st = SS.Stream(5 + s, t_still=0.0, t_move=2.4, v_max=0.5, v_start=0.5, yaw_turn=0.4 - 0.3 * s)
        st._lm = st._landmarks(1200)
        st._pn = np.random.default_rng(4300 + s).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
        streams.append(st)
    kw = dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1)
    grp = gfamd.EstimatorGroup(gfamd.default_estimator_cfg(**kw), n)
    solo = [gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(**kw)) for _ in range(n)]
    tp = [-1.0] * n
    for k in range(0, min(len(st.cam_t) for st in streams), 3):
        for s, st in enumerate(streams):
            st.feed(grp.members[s], k, tp[s])
            tp[s] = st.feed(solo[s], k, tp[s])
        frames = [st.feature_frame(k) for st in streams]
        grp.inputFeatures(list(range(n)), [float(st.cam_t[k]) for st in streams], frames)
        for s in range(n):
            solo[s].inputFeature(float(streams[s].cam_t[k]), frames[s])
            a, b = grp.members[s].state(), solo[s].state()
            assert a["solver_flag"] == b["solver_flag"] and a["frame_count"] == b["frame_count"], (k, s)
            assert np.array_equal(a["Ps"], b["Ps"]) and np.array_equal(a["Rs"], b["Rs"]) and np.array_equal(a["Vs"], b["Vs"]), (k, s)
    for s in range(n):
        info = grp.members[s].debug("init_info")
        assert grp.members[s].state()["solver_flag"] == 1 and int(info[1]) > 60 and int(info[6]) == 0     # reached through the SfM branch
    grp.close()
    for e in solo:
        e.close()


@pytest.mark.parametrize("window_size,own_initialiser,raw", [(10, False, False), (20, False, False), (10, True, False), (10, True, True)])
def test_replay_with_gnss_matches_oracle(window_size, own_initialiser, raw):
    """GNSS raw measurements through the estimator: inputGNSS -> getGNSSInterval -> processGNSS gating
    (estimator.cpp:476-510, :1455-1535), the PoseAnchorFactor of the first optimisation (:2943-2951), GNSS-VI alignment under the reference's
    preconditions (:1928-1962; the initialiser's result is handed in), receiver-clock / anchor / yaw blocks and GnssPsrDoppFactor, DtDdtFactor,
    DdtSmoothFactor in the solve (:2904-2941, :3178-3230) and in the MARGIN_OLD marginalisation (:3398-3434), the `lowspeed` switch while the
    vehicle crawls at the end, the clock shifts of both slideWindow branches (:3674-3681, :3761-3768), updateGNSSStatistics (:2045-2058).
    Bars: identical decisions (gnss_ready, lowspeed, admitted satellites per frame, keyframes, iteration counts) at every frame; window poses within
    1e-6 m / 1e-6 rad; anchor, receiver clocks, ECEF position and the modelled range + clock of every admitted satellite within 5e-3 m
    (observed 1e-7 ... 1e-3; the floor of these states measured on the oracle itself is 5e-4 ... 4e-3 m, see the assertion).  The GNSS states carry ECEF-sized numbers (6.4e6 m) and hang on weak directions of the marginalisation prior
    (eigenvalues 1e-8 ... 1e-5 against 1e9 at the top): what the prior says about them is what is left of entries of 1e9 after the strong pivots
    have been eliminated, so every relative error of 1e-16 in the factorisation shows at the 1e-7 level there.  Measured: with 1 / L_kk from a
    plain Newton iteration (2-3 ulp) the same replays sat 1.2e-3 m (W = 10, raw) and 1.5e-3 m (W = 20, `lowspeed` anchor) from the oracle, with
    the correctly rounded reciprocal root 1e-4 and 2e-7 m -- local poses identical (5e-8 m) either way; the pseudoranges carry 0.5 m of noise.
    DESIGN.md, "GNSS chains".  While `lowspeed` keeps the GNSS factors out of the solve the anchor hangs on the prior alone (anc_low / ecef_low,
    same bar).
    own_initialiser: nobody hands an alignment in; the library runs GNSSVIInitializer itself (initial/gnss_vi_initializer.cpp: SPP fix of the window's
    measurements, yaw alignment on the Doppler residuals, anchor refinement) and must arrive where the numpy restatement does.
    raw: the satellites fly broadcast orbits; the estimators receive ephemerides (inputEphem) and raw observations and derive the satellite states
    themselves (nearest-toe ephemeris, transmission time, Kepler / GLONASS propagation: tests/test_gnss_ephem_host.py pins that code against physics)."""
    W = window_size
    st = SS.Stream(3, t_still=1.5, t_move=4.5 if W == 10 else 5.7, v_max=0.4 if W == 10 else 0.35, yaw0=0.0, yaw_turn=-0.6, split_x=1.8, turn_delay=0.8, slow_tail=1.5)
    st._lm = st._landmarks(1600)
    st._pn = np.random.default_rng(4003).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
    G = st.gnss_setup(orbits=EO if raw else None)
    kw = dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, gnss_enable=1, gnss_track_num_thres=3, gnss_local_time_diff=G["time_diff"], window_size=W, max_visual=8192)
    est_p = gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(**kw))
    est_o = EO.Estimator(dict(kw))
    for eph in G.get("ephems", []):
        est_p.inputEphem(eph)
        est_o.inputEphem(eph)
    tp, worst, orng = -1.0, dict(p=0.0, r=0.0, v=0.0, clk=0.0, anc=0.0, ecef=0.0, anc_low=0.0, ecef_low=0.0, rho=0.0), np.random.default_rng(99)
    seen, ready_frames, admitted, ate = set(), 0, set(), []
    R0w = st.R_wb(st.cam_t[0])
    theta0 = float(np.arctan2(R0w[1, 0], R0w[0, 0]))
    for k in range(len(st.cam_t)):
        for e in (est_o, est_p):
            t1 = st.feed(e, k, tp)
        tp = t1
        if k % 2:
            continue
        tk = float(st.cam_t[k])
        tg, epoch = st.gnss_epoch(tk + orng.uniform(-0.02, 0.02), flaky_sat=2 if (k // 2) % 6 == 5 else None)
        al = st.gnss_alignment(tk - W / 15.0)
        frame = st.feature_frame(k)
        for e in (est_o, est_p):
            e.inputGNSS(tg, epoch)
            if not own_initialiser:
                e.setGNSSAlignment(*al)      # a receiver-side SPP / alignment result is on offer at every frame; the estimator takes it when GNSSVIAlign's conditions hold
            e.inputFeature(tk, frame)
        compare_frame(est_o, est_p, worst, "gnss frame %d" % k)
        g = est_p.gnss_state()
        assert (g["gnss_ready"], g["lowspeed"], g["first_optimization"]) == (int(est_o.gnss_ready), int(est_o.lowspeed), int(est_o.first_optimization)), k
        buf = est_p.debug("gnss_meas_buf")
        got, q = [], 0
        for _ in range(W + 1):
            n = int(buf[q])
            got.append([int(x) for x in buf[q + 1:q + 1 + n]])
            q += 1 + n
        assert got == [[o["sat"] for o in b] for b in est_o.gnss_meas_buf], k
        seen.add((int(est_o.gnss_ready), int(est_o.lowspeed), est_o.marginalization_flag))
        if est_o.solver_flag == EO.NON_LINEAR:   # absolute trajectory error of the product's newest pose against the stream's ground truth, in the estimator's local frame
            s_ = est_p.state()
            ate.append(np.linalg.norm(s_["Ps"][W] - SS.rot_z(-theta0) @ st.p_wb(s_["Headers"][W])))
        admitted |= {s for b in got for s in b}
        if est_o.gnss_ready:
            ready_frames += 1
            worst["clk"] = max(worst["clk"], float(np.abs(g["rcv_dt"] - est_o.para_rcv_dt).max()), float(np.abs(g["rcv_ddt"] - est_o.para_rcv_ddt).max()))
            key = "_low" if est_o.lowspeed else ""
            worst["anc" + key] = max(worst["anc" + key], float(np.abs(g["anc_ecef"] - est_o.anc_ecef).max()))
            worst["ecef" + key] = max(worst["ecef" + key], float(np.abs(g["ecef_pos"] - est_o.ecef_pos).max()), float(np.abs(g["enu_pos"] - est_o.enu_pos).max()))
            assert abs(g["yaw_enu_local"] - est_o.yaw_enu_local) <= (1e-6 if own_initialiser else 0.0)   # held constant (estimator.cpp:2930); own fit: the window velocities it uses agree to 1e-7 m/s
            if own_initialiser:
                assert abs(est_o.yaw_enu_local - G["yaw_enu_local"]) < 0.05     # the Doppler alignment found the true ENU <- local yaw (0.05 rad: 0.4 m/s of speed, 5 cm/s of Doppler noise)
            # what the measurements see: modelled range + receiver clock of every admitted satellite of the newest frame, from either pipeline's states
            # (frame W - 1: the slide has already emptied the newest slot)
            s_ = est_p.state()
            Rz = SS.rot_z(est_o.yaw_enu_local)
            pe_o = est_o.anc_ecef + EO.ecef2rotation(est_o.anc_ecef) @ Rz @ est_o.Ps[W - 1]
            pe_p = g["anc_ecef"] + EO.ecef2rotation(g["anc_ecef"]) @ Rz @ s_["Ps"][W - 1]
            for o in est_o.gnss_meas_buf[W - 1]:
                rho_o = np.linalg.norm(o["sv_pos"] - pe_o) + est_o.para_rcv_dt[W - 1, o["sys"]]
                rho_p = np.linalg.norm(o["sv_pos"] - pe_p) + g["rcv_dt"][W - 1, o["sys"]]
                worst["rho"] = max(worst["rho"], abs(rho_p - rho_o))
    assert ready_frames > 25 and {(1, 0, 0), (1, 0, 1)} <= seen and seen & {(1, 1, 0), (1, 1, 1)}   # aligned; both marginalisation kinds; `lowspeed` solves
    if W == 10:
        assert {(1, 1, 0), (1, 1, 1)} <= seen                                                       # ... of both kinds
    low = {sv["sat"] for sv in G["sats"][3:5]}
    assert low <= admitted and not (low & {o["sat"] for o in est_o.gnss_meas_buf[W - 1]})   # admitted before the alignment, dropped by the elevation gate afterwards
    truth = G["anc"] + G["R_ew"] @ st.p_wb(est_o.Headers[W])
    assert np.linalg.norm(est_o.ecef_pos - truth) < 15.0                     # metres: a sane fix (the synthetic ranges carry no iono / tropo delay, the factor removes a modelled one)
    rmse = float(np.sqrt(np.mean(np.square(ate))))
    print("gnss replay W=%d worst deviation" % W, worst, "ATE rmse %.4f m over %d frames" % (rmse, len(ate)))
    assert rmse < 0.05                                                       # 1.4 m of driving; observed 0.01
    assert worst["p"] < 1e-6 and worst["r"] < 1e-6, worst
    # GNSS states: the floor measured on the oracle itself (scripts/gnss_replay_sensitivity.py --every --scaled 1e-9: every marginalisation prior of the ORACLE
    # pipeline replaced by another correct factorisation of a kept system that agrees entry by entry to 1e-9 sqrt(A_ii A_jj) -- less than the 4e-8 ... 2e-7
    # its own eigen pseudo-inverse carries against a 60-digit Schur complement, scripts/marg_precision.py): W = 10 anchor 5.3e-4 m, under lowspeed 9.3e-4 m,
    # clocks 7.3e-4 m; W = 20 anchor 1.4e-3 m, under lowspeed 4.2e-3 m, clocks 4.1e-4 m -- while the local positions move by 5e-6 / 2e-5 m.  Two
    # double-precision marginalisations cannot agree better than that on these states; the bar is the floor, 5e-3 m (observed here: 1e-7 ... 1e-3).
    print("gnss replay full worst:", {k: "%.2e" % float(v) for k, v in worst.items()})
    # raw: the two pipelines propagate the ephemerides themselves (C++ / numpy), so their factor tables already differ in the last bits, and the own
    # initialiser's first anchor hangs on few epochs: the same script with --own --raw moves anchor / clocks by 6e-3 ... 1e-2 m at 1e-9 and by 2e-2 m at 1e-7
    # (observed here: 2e-2 ... 6e-2 m, next to local poses at 4e-7 m)
    # Round 4: those floors were measured by swapping in factorisations whose right-hand side `r` treats the directions below the rank differently from the
    # reference's eigen projection (marginalization_factor.cpp:294-302) -- which is what the library did as well.  With the least-squares r (csrc/gf_ba_marg.hpp:
    # J^T r = orthogonal projection of b onto the factor's range, scripts/marg_rhs_projection.py) the same replays sit at 3e-8 ... 9e-6 m in every variant, raw
    # ephemerides and own initialiser included (before: 1e-4 handed-in, 4e-4 own initialiser, 1e-2 raw): the bar is 2e-5 m, twice the largest observed value
    # (the modelled range under `lowspeed` at W = 20).
    # Round 5: the same mechanism adjudicated at 60 digits on the two-window chain (scripts/adjudicate_gnss_chain.py): the oracle's own chain is 3.5e-5 m from the chain
    # with the exact prior in the absolute position / anchor and 4.6e-6 m in the receiver clocks, the library's 1.5e-5 / 1.5e-6 m -- a closed loop of priors between two
    # double-precision implementations cannot be held tighter than that in these states; the local poses are (1e-6, compare_frame).
    bar = 2e-5
    assert worst["clk"] < bar and worst["anc"] < bar and worst["ecef"] < bar, worst
    assert worst["anc_low"] < bar and worst["ecef_low"] < bar and worst["rho"] < bar, worst      # rho runs over the `lowspeed` frames too
    est_p.close()


def test_group_with_gnss_members():
    """gf_estimator_group_* with gnss_enable: the shared back-end handle carries the GNSS blocks (max_gnss), members take their GNSS epochs through their own
    handles and align themselves (GNSSVIInitializer inside the library); each member must end up where a stand-alone Estimator on the same inputs does --
    local poses, receiver clocks, anchor -- while solves, marginalisations on the resident windows and the GNSS kernels run as one batch."""
    n = 3
    streams, G = [], []
    for s in range(n):
        st = SS.Stream(3 + s, t_still=1.5, t_move=3.4, v_max=0.4, yaw0=0.0, yaw_turn=-0.5 + 0.3 * s, split_x=1.8, turn_delay=0.8)
        st._lm = st._landmarks(1200)
        st._pn = np.random.default_rng(4200 + s).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
        G.append(st.gnss_setup(alpha=0.3 + 0.4 * s))
        streams.append(st)
    kw = dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, gnss_enable=1, gnss_track_num_thres=3, gnss_local_time_diff=G[0]["time_diff"])
    grp = gfamd.EstimatorGroup(gfamd.default_estimator_cfg(**kw), n)
    solo = [gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(**kw)) for _ in range(n)]
    tp, orng = [-1.0] * n, [np.random.default_rng(50 + s) for s in range(n)]
    nk = min(len(st.cam_t) for st in streams)
    worst, ready = dict(p=0.0, clk=0.0, anc=0.0), 0
    for k in range(nk):
        for s, st in enumerate(streams):
            st.feed(grp.members[s], k, tp[s])
            tp[s] = st.feed(solo[s], k, tp[s])
        if k % 2:
            continue
        frames = [st.feature_frame(k) for st in streams]
        for s, st in enumerate(streams):
            tg, epoch = st.gnss_epoch(float(st.cam_t[k]) + orng[s].uniform(-0.02, 0.02))
            grp.members[s].inputGNSS(tg, epoch)
            solo[s].inputGNSS(tg, epoch)
        grp.inputFeatures(list(range(n)), [float(st.cam_t[k]) for st in streams], frames)
        for s in range(n):
            solo[s].inputFeature(float(streams[s].cam_t[k]), frames[s])
            a, b = grp.members[s].state(), solo[s].state()
            ga, gb = grp.members[s].gnss_state(), solo[s].gnss_state()
            assert a["frame_count"] == b["frame_count"] and a["solver_flag"] == b["solver_flag"] and a["iterations"] == b["iterations"], (k, s)
            assert (ga["gnss_ready"], ga["lowspeed"], ga["n_newest"]) == (gb["gnss_ready"], gb["lowspeed"], gb["n_newest"]), (k, s)
            worst["p"] = max(worst["p"], float(np.abs(a["Ps"] - b["Ps"]).max()))
            worst["clk"] = max(worst["clk"], float(np.abs(ga["rcv_dt"] - gb["rcv_dt"]).max()))
            worst["anc"] = max(worst["anc"], float(np.abs(ga["anc_ecef"] - gb["anc_ecef"]).max()))
            ready += ga["gnss_ready"]
    assert ready > 3 * 10
    assert abs(solo[1].gnss_state()["yaw_enu_local"] - G[1]["yaw_enu_local"]) < 0.08          # each member found its own ENU <- local yaw
    st_ = grp.stats()
    assert st_["largest_batch"] == n
    print("GNSS group vs stand-alone worst deviation", worst)
    # the batch and the single-window launches run the same arithmetic (bit-identical states in tests/test_backend_gpu.py); the members' host threads change nothing
    assert worst["p"] < 1e-9 and worst["clk"] < 1e-9 and worst["anc"] < 1e-9, worst
    grp.close()


def test_config4_replay_images_w20_gnss():
    """BASELINE.json configs[4] as one run: 640x480 RGB-D images through the HIP tracker at max_cnt 500 / min_dist 12 into a 20-frame
    window with IMU + wheel + GNSS factors (raw measurements + broadcast ephemerides, the library's own GNSSVIInitializer), against the oracle pipeline on
    the same stream (scripts/config4_replay.py).  Up to ~900 tracks live in the FeatureManager and ~9 500 visual factors in a window, which puts the reduced
    system (440 columns) and the kept system of the marginalisation (155) in global memory (ba_step<true>, ba_marg_finish<true>).  Bars: tracker ids and
    observations bit-exact at every image, identical decisions / iteration counts at every back-end frame, window poses within 1e-6 m / 1e-6 rad, GNSS
    states within the bar of test_replay_with_gnss_matches_oracle; the absolute trajectory error against the stream's ground truth is printed."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import config4_replay as C4
    r = C4.run(product=True, t_move=4.5)
    w = r["worst"]
    print("configs[4] replay: %d images, %d back-end frames (%d with gnss_ready), <= %d features / %d visual factors per window, worst deviation %s, "
          "ATE rmse %.4f m over %.2f m of driving; oracle pipeline %.1f s, HIP pipeline %.1f s (one sequence, synchronous calls)"
          % (r["images"], r["frames"], r["ready_frames"], r["n_feat_max"], r["n_visual_max"], w, r["ate_rmse"], r["moved_m"], r["t_oracle"], r["t_product"]))
    assert r["solver_flag"] == EO.NON_LINEAR and r["ready_frames"] >= 15 and r["n_visual_max"] > 5000
    assert {s[:3] for s in r["seen"]} >= {(1, 0, 0), (1, 0, 1)}, r["seen"]      # aligned windows of both marginalisation kinds were solved
    assert r["ate_rmse"] < 0.05
    assert w["p"] < 1e-6 and w["r"] < 1e-6, w
    assert w["clk"] < 1e-5 and w["anc"] < 1e-5 and w["ecef"] < 1e-5, w     # observed 1e-6 (bar and history: test_replay_with_gnss_matches_oracle)


def test_moving_start_sweep_of_seeded_recordings():
    """scripts/sfm_init_sweep.py as a test (round-3 review): 17 seeded recordings that begin in motion at a constant speed drawn per seed (0.3 ... 0.8 m/s, with
    and without the wheel, turning either way) go through the SfM branch of initialStructure and 2.4 s of closed loop on both pipelines.  Decisions identical at
    every frame, poses within 1e-6 on all 17 (observed 1e-9 ... 3e-7).  Round 3 had one recording at 5e-6 and blamed the conditioning of the first MARGIN_OLD
    marginalisation behind such an initialisation (cond(A_mm) ~ 3e12); it was the prior's right-hand side below the rank of its factor, which the reference projects
    and the library used to predict -- gone with the least-squares r of round 4 (csrc/gf_ba_marg.hpp)."""
    worst_all = {}
    for seed in range(1, 18):
        rng = np.random.default_rng(seed)
        use_wheel = int(seed % 2)
        v = float(rng.uniform(0.3, 0.8))
        st = SS.Stream(seed, t_still=0.0, t_move=2.4, v_max=v, v_start=v, yaw_turn=float(rng.uniform(-0.7, 0.7)))
        st._lm = st._landmarks(900)
        st._pn = np.random.default_rng(7000 + seed).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
        kw = dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, use_wheel=use_wheel, wdetect=use_wheel)
        eo, ep = EO.Estimator(dict(kw)), gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(**kw))
        tp, worst = -1.0, 0.0
        for k in range(0, len(st.cam_t), 3):
            for e in (eo, ep):
                t1 = st.feed(e, k, tp)
            tp = t1
            fr = st.feature_frame(k)
            eo.inputFeature(float(st.cam_t[k]), fr); ep.inputFeature(float(st.cam_t[k]), fr)
            s = ep.state()
            assert s["solver_flag"] == eo.solver_flag and s["frame_count"] == eo.frame_count and s["marginalization_flag"] == eo.marginalization_flag, (seed, k)
            worst = max(worst, float(np.abs(s["Ps"] - np.array(eo.Ps)).max()), rot_angle(s["Rs"], eo.Rs))
        assert eo.solver_flag == EO.NON_LINEAR and not eo.is_imu_excited, seed        # initialised through the SfM branch
        ep.close()
        worst_all[seed] = worst
    print("moving-start sweep, worst |dP| / |dR| per seed:", {k: "%.1e" % v for k, v in worst_all.items()})
    assert max(worst_all.values()) < 1e-6, worst_all
