"""initialisation while moving on a synthetic recording that begins in motion at constant speed: the oracle pipeline on the CPU,
optionally next to the product (--product, needs a GPU).  Prints what the SfM branch of initialStructure did and the pose error afterwards."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import synth_stream as SS
import estimator_oracle as EO

def stream(seed=5, **kw):
    st = SS.Stream(seed, t_still=0.0, t_move=kw.pop("t_move", 3.0), v_max=0.5, v_start=0.5, yaw_turn=kw.pop("yaw_turn", 0.4))
    return st

if __name__ == "__main__":
    product = "--product" in sys.argv
    st = stream()
    kw = dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1)
    ests = [EO.Estimator(dict(kw))]
    if product:
        import gfamd
        ests.append(gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(**kw)))
    tp = -1.0
    STRIDE = 3
    for k in range(0, len(st.cam_t), STRIDE):
        t1 = tp
        for e in ests:
            t1 = st.feed(e, k, tp)
        tp = t1
        frame = st.feature_frame(k)
        t0 = time.time()
        for e in ests:
            e.inputFeature(float(st.cam_t[k]), frame)
        eo = ests[0]
        W = eo.W
        line = "k %3d fc %2d flag %d stationary %d excited %d  P[W] %s" % (k, eo.frame_count, eo.solver_flag, eo.systemstationary, eo.is_imu_excited, np.round(eo.Ps[min(eo.frame_count, W)], 3))
        if eo.solver_flag == 1:
            Wn = eo.W
            err = [abs(np.linalg.norm(eo.Ps[i] - eo.Ps[0]) - np.linalg.norm(st.p_wb(eo.Headers[i]) - st.p_wb(eo.Headers[0]))) for i in range(Wn)]
            line += "  chord err %.4f |V| %.3f (true %.3f)" % (max(err), np.linalg.norm(eo.Vs[Wn - 1]), np.linalg.norm(st._at(st._vw, eo.Headers[Wn - 1])))
        if product:
            s = ests[1].state()
            line += "  dP %.2e dR %.2e dV %.2e" % (np.abs(s["Ps"] - np.array(eo.Ps)).max(), np.abs(s["Rs"] - np.array(eo.Rs)).max(), np.abs(s["Vs"] - np.array(eo.Vs)).max())
        print(line, " %.2fs" % (time.time() - t0), flush=True)
        if getattr(eo, "init_debug", None) and not getattr(eo, "_shown", False):
            eo._shown = True
            d = eo.init_debug
            print("  SfM: l %d, %d points, relative_T %s, |g| %.4f g %s, s = x[-1] = %.3e" % (d["l"], d["n_tracked"], np.round(d["relative_T"], 4), np.linalg.norm(d["g_c0"]), np.round(d["g_c0"], 3), d["x"][-1]))
            print("  T (camera l frame):", np.round(d["T"], 3).tolist())
            Wn = eo.W
            err = [np.linalg.norm((eo.Ps[i] - eo.Ps[0]) - (st.p_wb(eo.Headers[i]) - st.p_wb(eo.Headers[0]))) for i in range(Wn)]
            print("  window position error against truth after the first optimisation (relative to frame 0):", np.round(err, 4).tolist())
