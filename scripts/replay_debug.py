import os, sys, pickle
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import synth_stream as SS, estimator_oracle as EO
seed, kdump = 1, int(sys.argv[1])
st = SS.Stream(seed, t_still=1.5, t_move=3.0, v_max=0.4, yaw0=0.38, yaw_turn=-0.55)
est_o = EO.Estimator(dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1))
tp = -1.0
for k in range(0, kdump + 1):
    tp = st.feed(est_o, k, tp)
    if k % 2: continue
    est_o.inputFeature(float(st.cam_t[k]), st.feature_frame(k))
pickle.dump(dict(est_o.last_window), open("/tmp/win_dump.pkl", "wb"))
print(est_o.last_summary)
