"""How much of a product-vs-oracle deviation after the initialisation while moving is the problem's own conditioning?  The recording of
scripts/sfm_init_sweep.py (GF_SWEEP_SKIP_DRAWS=2, 1400 landmarks, seed 1: 5e-6 m between the pipelines) through the ORACLE twice, the second time with the window
positions perturbed by 1e-11 m before the second solve after the initialisation (CPU only)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import synth_stream as SS, estimator_oracle as EO

seed = 1
rng = np.random.default_rng(seed); rng.uniform(size=2)
v = float(rng.uniform(0.3, 0.8)); yt = float(rng.uniform(-0.7, 0.7))
st = SS.Stream(seed, t_still=0.0, t_move=2.4, v_max=v, v_start=v, yaw_turn=yt)
st._lm = st._landmarks(1400)
st._pn = np.random.default_rng(7000 + seed).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
kw = dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, use_wheel=1, wdetect=1)
a, b = EO.Estimator(dict(kw)), EO.Estimator(dict(kw))
tp = -1.0
prng = np.random.default_rng(99)
for k in range(0, len(st.cam_t), 3):
    for e in (a, b):
        t1 = st.feed(e, k, tp)
    tp = t1
    fr = st.feature_frame(k)
    if k == 33:
        for i in range(b.W + 1):
            b.Ps[i] = b.Ps[i] + prng.normal(0, 1e-11, 3)
    a.inputFeature(float(st.cam_t[k]), fr); b.inputFeature(float(st.cam_t[k]), fr)
    if a.solver_flag == 1:
        dev = max(float(np.abs(np.array(a.Ps) - np.array(b.Ps)).max()), float(np.abs(np.array(a.Rs) - np.array(b.Rs)).max()))
        ls = a.last_summary or {}
        print("k %3d  oracle vs perturbed oracle %.2e   cost %.3e -> %.3e, %s successful steps of %s" % (k, dev, ls.get("initial_cost", 0), ls.get("final_cost", 0), ls.get("successful_steps"), ls.get("iterations")), flush=True)
