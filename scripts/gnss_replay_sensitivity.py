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


EVERY = "--every" in sys.argv   # perturb every marginalisation prior (what two implementations do to each other), not only the first one
NUDGED = [False]
OWN, RAW = "--own" in sys.argv, "--raw" in sys.argv   # the estimator's own GNSSVIInitializer instead of a handed-in alignment; broadcast ephemerides + raw observations
SCALED = "--scaled" in sys.argv   # perturb the kept system A = J^T J entry by entry (scaled), refactor it as the reference does (eigenvalues > 1e-8), keep b = J^T r


def run(nudge):
    NUDGED[0] = False
    prng = np.random.default_rng(123)
    st = SS.Stream(3, t_still=1.5, t_move=4.5 if W == 10 else 5.7, v_max=0.4 if W == 10 else 0.35, yaw0=0.0, yaw_turn=-0.6, split_x=1.8, turn_delay=0.8, slow_tail=1.5)
    st._lm = st._landmarks(1600)
    st._pn = np.random.default_rng(4003).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
    G = st.gnss_setup(orbits=EO if RAW else None)
    e = EO.Estimator(dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, gnss_enable=1, gnss_track_num_thres=3, gnss_local_time_diff=G["time_diff"], window_size=W))
    for eph in G.get("ephems", []):
        e.inputEphem(eph)
    tp, orng, out = -1.0, np.random.default_rng(99), []
    for k in range(len(st.cam_t)):
        tp = st.feed(e, k, tp)
        if k % 2:
            continue
        tk = float(st.cam_t[k])
        tg, epoch = st.gnss_epoch(tk + orng.uniform(-0.02, 0.02), flaky_sat=2 if (k // 2) % 6 == 5 else None)
        e.inputGNSS(tg, epoch)
        if not OWN:
            e.setGNSSAlignment(*st.gnss_alignment(tk - W / 15.0))
        e.inputFeature(tk, st.feature_frame(k))
        if nudge and e.prior is not None and (EVERY or not NUDGED[0]) and not e.prior.get("nudged"):
            NUDGED[0] = True
            if SCALED:   # another correct factorisation of a kept system that agrees entry by entry to `nudge` * sqrt(A_ii A_jj) (what tests/test_backend_gpu.py pins)
                n_ = len(e.prior["r"])
                J_ = np.asarray(e.prior["J"], float).reshape(n_, n_)
                A_, b_ = J_.T @ J_, J_.T @ np.asarray(e.prior["r"], float)
                sc = np.sqrt(np.maximum(np.diag(A_), 1e-300))
                N_ = prng.normal(0, 1, A_.shape); N_ = 0.5 * (N_ + N_.T)
                A2 = A_ + nudge * np.outer(sc, sc) * N_
                ev, V = np.linalg.eigh(A2)
                keep = ev > 1e-8
                S = np.where(keep, ev, 0.0); Sinv = np.where(keep, 1.0 / np.where(keep, ev, 1.0), 0.0)
                J2 = (np.sqrt(S)[:, None] * V.T)
                r2 = (np.sqrt(Sinv)[:, None] * V.T) @ b_
                e.prior["J"] = J2.reshape(np.asarray(e.prior["J"]).shape); e.prior["r"] = r2
            else:
                e.prior["J"] = e.prior["J"] * (1.0 + nudge * prng.normal(0, 1, e.prior["J"].shape))
                e.prior["r"] = e.prior["r"] * (1.0 + nudge * prng.normal(0, 1, e.prior["r"].shape))
            e.prior["nudged"] = True
        out.append((k, int(e.gnss_ready), int(e.lowspeed), np.array(e.Ps), e.anc_ecef.copy(), e.para_rcv_dt.copy()))
    return out


a, b = run(0.0), run(REL)
mx = {"dP": 0.0, "danc": 0.0, "danc_low": 0.0, "dclk": 0.0}
for (k, rdy, low, Pa, anca, dta), (_, _, _, Pb, ancb, dtb) in zip(a, b):
    if rdy:
        mx["dP"] = max(mx["dP"], float(np.abs(Pa - Pb).max())); mx["dclk"] = max(mx["dclk"], float(np.abs(dta - dtb).max()))
        mx["danc_low" if low else "danc"] = max(mx["danc_low" if low else "danc"], float(np.abs(anca - ancb).max()))
print("W = %d, relative perturbation %.0e of %s prior: worst local position %.2e m, anchor %.2e m, anchor under lowspeed %.2e m, clocks %.2e m"
      % (W, REL, "every" if EVERY else "the first", mx["dP"], mx["danc"], mx["danc_low"], mx["dclk"]))
for (k, rdy, low, Pa, anca, dta), (_, _, _, Pb, ancb, dtb) in zip(a, b):
    if rdy:
        print("k %3d low %d  dP %.2e  danc %.2e  dclk %.2e" % (k, low, np.abs(Pa - Pb).max(), np.abs(anca - ancb).max(), np.abs(dta - dtb).max()))
