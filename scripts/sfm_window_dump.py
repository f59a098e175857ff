"""dumps the oracle pipeline's windows (the input of Estimator::optimization) of the first frames after the initialisation while moving, for the recording of
scripts/sfm_init_sweep.py on which product and oracle pipelines end 5e-6 m apart (GF_SWEEP_SKIP_DRAWS=2, 1400 landmarks, seed 1), into tmp_dump/win_k<frame>.pkl;
scripts/win_debug.py (solver, iteration by iteration), scripts/marg_window_compare.py (the two priors) and scripts/marg_conditioning.py (CPU) take them from there."""
import os, sys, pickle
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(ROOT, 'tmp_dump'), exist_ok=True)
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import synth_stream as SS, estimator_oracle as EO
seed = 1
rng = np.random.default_rng(seed); rng.uniform(size=2)
v = float(rng.uniform(0.3, 0.8)); yt = float(rng.uniform(-0.7, 0.7))
st = SS.Stream(seed, t_still=0.0, t_move=2.4, v_max=v, v_start=v, yaw_turn=yt)
st._lm = st._landmarks(1400)
st._pn = np.random.default_rng(7000 + seed).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
kw = dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, use_wheel=1, wdetect=1)
a = EO.Estimator(dict(kw))
tp = -1.0
for k in range(0, 34, 3):
    tp = st.feed(a, k, tp)
    a.inputFeature(float(st.cam_t[k]), st.feature_frame(k))
    if a.solver_flag == 1:
        pickle.dump(dict(a.last_window), open(os.path.join(ROOT, "tmp_dump", "win_k%d.pkl" % k), "wb"))
        print(k, a.last_summary)
