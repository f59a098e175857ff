"""How well is the ECEF anchor determined while `lowspeed` holds the GNSS factors out of the solve?  The oracle pipeline against itself with every
marginalisation prior (linearized_jacobians, linearized_residuals) perturbed by 1e-10 relative -- a tenth of the arithmetic difference between two
correct factorisations of the same prior (tests/test_backend_gpu.py pins them to 1e-9).  usage: gnss_replay_sensitivity.py [--w20] [rel]"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import synth_stream as SS  # noqa: E402
import estimator_oracle as EO  # noqa: E402

W = 20 if "--w20" in sys.argv else 10


REL = float([a for a in sys.argv[1:] if not a.startswith("--")][0]) if [a for a in sys.argv[1:] if not a.startswith("--")] else 1e-10


def run(nudge):
    prng = np.random.default_rng(123)
    st = SS.Stream(3, t_still=1.5, t_move=4.5 if W == 10 else 5.7, v_max=0.4 if W == 10 else 0.35, yaw0=0.0, yaw_turn=-0.6, split_x=1.8, turn_delay=0.8, slow_tail=1.5)
    st._lm = st._landmarks(1600)
    st._pn = np.random.default_rng(4003).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
    G = st.gnss_setup()
    e = EO.Estimator(dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, gnss_enable=1, gnss_track_num_thres=3, gnss_local_time_diff=G["time_diff"], window_size=W))
    tp, orng, out = -1.0, np.random.default_rng(99), []
    for k in range(len(st.cam_t)):
        tp = st.feed(e, k, tp)
        if k % 2:
            continue
        tk = float(st.cam_t[k])
        tg, epoch = st.gnss_epoch(tk + orng.uniform(-0.02, 0.02), flaky_sat=2 if (k // 2) % 6 == 5 else None)
        e.inputGNSS(tg, epoch)
        e.setGNSSAlignment(*st.gnss_alignment(tk - W / 15.0))
        e.inputFeature(tk, st.feature_frame(k))
        if nudge and e.prior is not None and not e.prior.get("nudged"):
            e.prior["J"] = e.prior["J"] * (1.0 + nudge * prng.normal(0, 1, e.prior["J"].shape))
            e.prior["r"] = e.prior["r"] * (1.0 + nudge * prng.normal(0, 1, e.prior["r"].shape))
            e.prior["nudged"] = True
        out.append((k, int(e.gnss_ready), int(e.lowspeed), np.array(e.Ps), e.anc_ecef.copy(), e.para_rcv_dt.copy()))
    return out


a, b = run(0.0), run(REL)
for (k, rdy, low, Pa, anca, dta), (_, _, _, Pb, ancb, dtb) in zip(a, b):
    if rdy:
        print("k %3d low %d  dP %.2e  danc %.2e  dclk %.2e" % (k, low, np.abs(Pa - Pb).max(), np.abs(anca - ancb).max(), np.abs(dta - dtb).max()))
