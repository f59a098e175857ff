"""BASELINE.json configs[4] as one run: 640x480 RGB-D images -> FeatureTracker::trackImage at max_cnt 500 / min_dist 12 ->
Estimator::processImage with a 20-frame window, IMU + wheel + GNSS (raw measurements and broadcast ephemerides through inputGNSS / inputEphem, own
GNSSVIInitializer), MULTIPLE_THREAD data flow of m2dgrp.yaml -- the HIP pipeline (gf_estimator_* with its tracker) next to the CPU oracle pipeline
(oracle tracker + estimator_oracle + oracle BA) on the same seeded stream.  TEST / MEASUREMENT INFRASTRUCTURE (imports oracle/): used by
tests/test_estimator_gpu.py::test_config4_replay_images_w20_gnss and run stand-alone for the numbers kept in profiles/.

    python scripts/config4_replay.py [--oracle-only] [--t-move 5.7]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ground-fusion_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
import synth_stream as SS  # noqa: E402
import estimator_oracle as EO  # noqa: E402
import oracle_py as O  # noqa: E402

W, MAX_CNT, MIN_DIST = 20, 500, 12


def rot_angle(Ra, Rb):
    out = 0.0
    for a, b in zip(Ra, Rb):
        c = (np.trace(a.T @ b) - 1.0) / 2.0
        s = np.linalg.norm(a.T @ b - (a.T @ b).T) / (2.0 * np.sqrt(2.0))
        out = max(out, float(np.arctan2(s, c)))
    return out


def run(product=True, t_move=5.7, seed=3, verbose=False):
    """returns a dict of worst deviations / counters; raises AssertionError where a bit-exact or decision-level bar is broken"""
    st = SS.Stream(seed, t_still=1.5, t_move=t_move, v_max=0.35, yaw0=0.0, yaw_turn=-0.6, split_x=1.8, turn_delay=0.8, slow_tail=1.5)
    G = st.gnss_setup(orbits=EO)
    kw = dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, gnss_enable=1, gnss_track_num_thres=3, gnss_local_time_diff=G["time_diff"], window_size=W,
              max_features=1536, max_visual=16384)
    est_o = EO.Estimator(dict(kw), tracker=O.Tracker(O.default_cfg(max_cnt=MAX_CNT, min_dist=MIN_DIST)))
    est_p = None
    if product:
        import gfamd
        cfg = gfamd.default_estimator_cfg(with_tracker=1, **kw)
        cfg.tracker = gfamd.default_cfg(max_cnt=MAX_CNT, min_dist=MIN_DIST)
        est_p = gfamd.SlidingWindowEstimator(cfg)
    ests = [e for e in (est_o, est_p) if e is not None]
    for eph in G.get("ephems", []):
        for e in ests:
            e.inputEphem(eph)
    worst = dict(p=0.0, r=0.0, v=0.0, clk=0.0, anc=0.0, ecef=0.0, px=0.0)
    out = dict(frames=0, ready_frames=0, seen=set(), n_feat_max=0, n_visual_max=0, ate=[], t_oracle=0.0, t_product=0.0, tracked=0)
    R0w = st.R_wb(st.cam_t[0])
    theta0 = float(np.arctan2(R0w[1, 0], R0w[0, 0]))
    orng = np.random.default_rng(99)
    tp = -1.0
    for k in range(len(st.cam_t)):
        for e in ests:
            t1 = st.feed(e, k, tp)
        tp = t1
        tk = float(st.cam_t[k])
        img, dep = st.image(k)
        if k % 2 == 0:   # one GnssMeasMsg per back-end frame, ahead of the image that closes its interval
            tg, epoch = st.gnss_epoch(tk + orng.uniform(-0.02, 0.02), flaky_sat=2 if (k // 2) % 6 == 5 else None)
            for e in ests:
                e.inputGNSS(tg, epoch)
        t0 = time.perf_counter()
        ids_o, obs_o = est_o.inputImage(tk, img, dep)
        out["t_oracle"] += time.perf_counter() - t0
        out["tracked"] += len(ids_o)
        if est_p is not None:
            t0 = time.perf_counter()
            fp = est_p.inputImage(tk, img, dep)
            out["t_product"] += time.perf_counter() - t0
            assert sorted(fp) == sorted(int(i) for i in ids_o), "tracker ids differ at image %d" % k
            for j, i in enumerate(ids_o):   # MULTIPLE_THREAD: no feedback into the tracker -> observations bit-exact
                assert np.array_equal(fp[int(i)], obs_o[j]), "tracker observation of id %d differs at image %d" % (i, k)
        if (k + 1) % 2 != 0:
            continue
        out["frames"] += 1
        fo = est_o.f_manager.feature
        out["n_feat_max"] = max(out["n_feat_max"], len(fo))
        if est_o.last_summary is not None:
            out["n_visual_max"] = max(out["n_visual_max"], sum(len(f.feature_per_frame) - 1 for f in fo if len(f.feature_per_frame) >= 4))
        out["seen"].add((int(est_o.gnss_ready), int(est_o.lowspeed), est_o.marginalization_flag, est_o.solver_flag))
        if est_o.solver_flag == EO.NON_LINEAR:
            out["ate"].append(float(np.linalg.norm(np.array(est_o.Ps[W]) - SS.rot_z(-theta0) @ st.p_wb(est_o.Headers[W]))))
        if est_o.gnss_ready:
            out["ready_frames"] += 1
        if est_p is None:
            if verbose:
                print("image %d: %d tracks, %d features, flag %d, gnss_ready %d" % (k, len(ids_o), len(fo), est_o.solver_flag, est_o.gnss_ready), flush=True)
            continue
        s = est_p.state()
        tag = "image %d" % k
        assert s["frame_count"] == est_o.frame_count and s["solver_flag"] == est_o.solver_flag, tag
        assert s["marginalization_flag"] == est_o.marginalization_flag and bool(s["systemstationary"]) == bool(est_o.systemstationary), tag
        f_p = est_p.features()
        assert [f.feature_id for f in fo] == list(f_p["id"]), tag
        assert [f.start_frame for f in fo] == list(f_p["start_frame"]) and [len(f.feature_per_frame) for f in fo] == list(f_p["n_obs"]), tag
        assert [f.estimate_flag for f in fo] == list(f_p["estimate_flag"]), tag
        if est_o.last_summary is not None:
            assert s["iterations"] == est_o.last_summary["iterations"] and s["successful_steps"] == est_o.last_summary["successful_steps"], tag
        g = est_p.gnss_state()
        assert (g["gnss_ready"], g["lowspeed"], g["first_optimization"]) == (int(est_o.gnss_ready), int(est_o.lowspeed), int(est_o.first_optimization)), tag
        worst["p"] = max(worst["p"], float(np.abs(s["Ps"] - np.array(est_o.Ps)).max()))
        worst["r"] = max(worst["r"], rot_angle(s["Rs"], est_o.Rs))
        worst["v"] = max(worst["v"], float(np.abs(s["Vs"] - np.array(est_o.Vs)).max()))
        if est_o.gnss_ready:
            worst["clk"] = max(worst["clk"], float(np.abs(g["rcv_dt"] - est_o.para_rcv_dt).max()), float(np.abs(g["rcv_ddt"] - est_o.para_rcv_ddt).max()))
            worst["anc"] = max(worst["anc"], float(np.abs(g["anc_ecef"] - est_o.anc_ecef).max()))
            worst["ecef"] = max(worst["ecef"], float(np.abs(g["ecef_pos"] - est_o.ecef_pos).max()))
        if verbose:
            print(tag, "tracks %d features %d" % (len(ids_o), len(fo)), {k_: "%.2e" % v for k_, v in worst.items()}, flush=True)
    out["worst"] = worst
    out["ate_rmse"] = float(np.sqrt(np.mean(np.square(out["ate"])))) if out["ate"] else None
    out["solver_flag"] = est_o.solver_flag
    out["moved_m"] = float(np.linalg.norm(est_o.Ps[W]))
    out["images"] = len(st.cam_t)
    if est_p is not None:
        est_p.close()
    return out


if __name__ == "__main__":
    tm = float(sys.argv[sys.argv.index("--t-move") + 1]) if "--t-move" in sys.argv else 5.7
    r = run(product="--oracle-only" not in sys.argv, t_move=tm, verbose=True)
    r["seen"] = sorted(r["seen"])
    r.pop("ate")
    print(r)
