"""initialisation while moving over many seeded recordings: product and oracle pipelines side by side from a constant-speed start through the
SfM branch of initialStructure and 2.4 s of closed loop.  Prints, per recording, frame l / structure points / RANSAC-free facts and the worst pose deviation."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gfamd, synth_stream as SS, estimator_oracle as EO

worst_all = 0.0
for seed in range(1, int(sys.argv[1]) + 1 if len(sys.argv) > 1 else 11):
    rng = np.random.default_rng(seed)
    rng.uniform(size=int(os.environ.get("GF_SWEEP_SKIP_DRAWS", "0")))
    use_wheel = int(seed % 2)
    st = SS.Stream(seed, t_still=0.0, t_move=2.4, v_max=(v := float(rng.uniform(0.3, 0.8))), v_start=v, yaw_turn=float(rng.uniform(-0.7, 0.7)))
    st._lm = st._landmarks(int(os.environ.get("GF_SWEEP_LANDMARKS", "900")))
    st._pn = np.random.default_rng(7000 + seed).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
    kw = dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, use_wheel=use_wheel, wdetect=use_wheel)
    eo, ep = EO.Estimator(dict(kw)), gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(**kw))
    tp, worst, ok = -1.0, 0.0, True
    for k in range(0, len(st.cam_t), 3):
        for e in (eo, ep):
            t1 = st.feed(e, k, tp)
        tp = t1
        fr = st.feature_frame(k)
        eo.inputFeature(float(st.cam_t[k]), fr); ep.inputFeature(float(st.cam_t[k]), fr)
        s = ep.state()
        ok = ok and s["solver_flag"] == eo.solver_flag and s["frame_count"] == eo.frame_count
        dev = max(float(np.abs(s["Ps"] - np.array(eo.Ps)).max()), float(np.abs(s["Rs"] - np.array(eo.Rs)).max()))
        worst = max(worst, dev)
        if os.environ.get("GF_SWEEP_VERBOSE") and eo.solver_flag == 1:
            ls = eo.last_summary or {}
            fo, fp = eo.f_manager.feature, ep.features()
            dd = np.abs(fp["estimated_depth"] - np.array([f.estimated_depth for f in fo])) if len(fo) == len(fp["id"]) else np.array([np.nan])
            dep = np.array([f.estimated_depth for f in fo])
            j = int(np.nanargmax(dd))
            print("      dV %.2e dBa %.2e dBg %.2e  depth dev max %.2e (depth %.3f, %d obs, flag %d) rel %.2e" % (
                np.abs(s["Vs"] - np.array(eo.Vs)).max(), np.abs(s["Bas"] - np.array(eo.Bas)).max(), np.abs(s["Bgs"] - np.array(eo.Bgs)).max(), dd[j], dep[j],
                len(fo[j].feature_per_frame), fo[j].estimate_flag, np.nanmax(dd / np.maximum(np.abs(dep), 1e-9))))
            print("   k %3d dev %.2e  iterations %s/%s steps %s/%s cost %.3e -> %.3e" % (k, dev, s["iterations"], ls.get("iterations"), s["successful_steps"], ls.get("successful_steps"),
                                                                                   ls.get("initial_cost", float("nan")), ls.get("final_cost", float("nan"))), flush=True)
    d = getattr(eo, "init_debug", None)
    print("seed %2d wheel %d v %.2f: flag %d, sfm %s, l %s, points %s, decisions equal %s, worst |dP|,|dR| %.2e" % (
        seed, use_wheel, v, eo.solver_flag, d is not None and not eo.is_imu_excited, d and d["l"], d and d["n_tracked"], ok, worst), flush=True)
    worst_all = max(worst_all, worst)
    ep.close()
print("worst over all recordings %.2e" % worst_all)
