"""Replay one synthetic sequence through the HIP estimator and the CPU oracle side by side; prints the per-frame deviation.
usage: python scripts/replay_compare.py [seed] [images:0|1] [multiple_thread:0|1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gfamd  # noqa: E402
import synth_stream as SS  # noqa: E402
import estimator_oracle as EO  # noqa: E402
import oracle_py as O  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
images = int(sys.argv[2]) if len(sys.argv) > 2 else 0
mt = int(sys.argv[3]) if len(sys.argv) > 3 else 1
t_still = float(sys.argv[4]) if len(sys.argv) > 4 else 1.5
st = SS.Stream(seed, t_still=t_still, t_move=3.0, v_max=0.8 if t_still < 0.5 else 0.4, yaw0=0.0, yaw_turn=-0.6, split_x=1.8, turn_delay=0.8)
cfg = gfamd.default_estimator_cfg(tio=SS.TIO, rio=SS.RIO, multiple_thread=mt, with_tracker=images)
if images:
    cfg.tracker = gfamd.default_cfg()
est_p = gfamd.SlidingWindowEstimator(cfg)
est_o = EO.Estimator(dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=mt, svd=os.environ.get("GF_SVD", "jacobi")), tracker=O.Tracker() if images else None)
tp = -1.0
worst = 0.0
tg = tc = 0.0
for k in range(len(st.cam_t)):
    for e in (est_o, est_p):
        t1 = st.feed(e, k, tp)
    tp = t1
    t = float(st.cam_t[k])
    if images:
        img, dep = st.image(k)
        a = time.time(); fp = est_p.inputImage(t, img, dep); tg += time.time() - a
        a = time.time(); ids_o, obs_o = est_o.inputImage(t, img, dep); tc += time.time() - a
        same_ids = sorted(fp) == sorted(int(i) for i in ids_o)
        if (k + 1) % 2 != 0 and mt:
            if not same_ids:
                print(k, "TRACKER IDS DIFFER")
            continue
    else:
        if k % 2:
            continue
        frame = st.feature_frame(k)
        a = time.time(); est_p.inputFeature(t, frame); tg += time.time() - a
        a = time.time(); est_o.inputFeature(t, frame); tc += time.time() - a
        same_ids = True
    s = est_p.state()
    fo = est_o.f_manager.feature
    fpp = est_p.features()
    ids_eq = [f.feature_id for f in fo] == list(fpp["id"])
    dP = np.abs(s["Ps"] - np.array(est_o.Ps)).max()
    dR = np.abs(s["Rs"] - np.array(est_o.Rs)).max()
    dV = np.abs(s["Vs"] - np.array(est_o.Vs)).max()
    dB = max(np.abs(s["Bas"] - np.array(est_o.Bas)).max(), np.abs(s["Bgs"] - np.array(est_o.Bgs)).max())
    dd = np.abs(fpp["estimated_depth"] - np.array([f.estimated_depth for f in fo])).max() if ids_eq and len(fo) else -1
    it_o = est_o.last_summary["iterations"] if est_o.last_summary else -1
    worst = max(worst, dP, dR)
    print("%3d fc %2d flag %d/%d marg %d/%d stat %d/%d ids %s trk %s nfeat %3d it %d/%d cost %.4f/%.4f dP %.2e dR %.2e dV %.2e dB %.2e ddep %.2e |P| %.3f" % (
        k, s["frame_count"], s["solver_flag"], est_o.solver_flag, s["marginalization_flag"], est_o.marginalization_flag, s["systemstationary"], est_o.systemstationary,
        ids_eq, same_ids, len(fo), s["iterations"], it_o, s["final_cost"], est_o.last_summary["final_cost"] if est_o.last_summary else -1, dP, dR, dV, dB, dd,
        np.linalg.norm(s["Ps"][-1])))
print("worst pose deviation %.3e   time product %.2fs oracle %.2fs" % (worst, tg, tc))
