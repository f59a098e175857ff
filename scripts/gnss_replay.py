"""GNSS through the estimator, closed loop: the numpy oracle alone (CPU) or oracle vs product (GPU).  usage: gnss_replay.py [--product]"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import synth_stream as SS  # noqa: E402
import estimator_oracle as EO  # noqa: E402

product = "--product" in sys.argv
W = 20 if "--w20" in sys.argv else 10
st = SS.Stream(3, t_still=1.5, t_move=4.5 if W == 10 else 5.7, v_max=0.4 if W == 10 else 0.35, yaw0=0.0, yaw_turn=-0.6, split_x=1.8, turn_delay=0.8, slow_tail=1.5)
st._lm = st._landmarks(1600)
st._pn = np.random.default_rng(4003).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
G = st.gnss_setup()
kw = dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1, gnss_enable=1, gnss_track_num_thres=3, gnss_local_time_diff=G["time_diff"], window_size=W, max_visual=8192)
ests = [EO.Estimator(dict(kw))]
if product:
    import gfamd
    ests.append(gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(**kw)))
tp = -1.0
orng = np.random.default_rng(99)
for k in range(len(st.cam_t)):
    for e in ests:
        t1 = st.feed(e, k, tp)
    tp = t1
    if k % 2:
        continue
    tk = float(st.cam_t[k])
    tg, epoch = st.gnss_epoch(tk + orng.uniform(-0.02, 0.02), flaky_sat=2 if (k // 2) % 6 == 5 else None)
    al = st.gnss_alignment(tk - W / 15.0)
    frame = st.feature_frame(k)
    for e in ests:
        e.inputGNSS(tg, epoch)
        e.setGNSSAlignment(*al)
        e.inputFeature(tk, frame)
    o = ests[0]
    line = "k %3d fc %2d flag %d marg %d ready %d low %d nmeas %s it %s cost %.3f->%.3f dt0 %.3f anc %s" % (
        k, o.frame_count, o.solver_flag, o.marginalization_flag, o.gnss_ready, o.lowspeed, [len(b) for b in o.gnss_meas_buf][-3:],
        o.last_summary["iterations"] if o.last_summary else None, o.last_summary["initial_cost"] if o.last_summary else 0,
        o.last_summary["final_cost"] if o.last_summary else 0, o.para_rcv_dt[W, 0], np.round(o.anc_ecef - G["anc"], 3))
    if product:
        s, g = ests[1].state(), ests[1].gnss_state()
        line += " | dP %.2e ddt %.2e danc %.2e ready %d low %d danc_enu %s" % (np.abs(s["Ps"] - np.array(o.Ps)).max(), np.abs(g["rcv_dt"] - o.para_rcv_dt).max(),
                                                                 np.abs(g["anc_ecef"] - o.anc_ecef).max(), g["gnss_ready"], g["lowspeed"], (G["Re"].T @ (g["anc_ecef"] - o.anc_ecef)).round(7))
    if product and o.prior is not None and o.gnss_ready:
        pp = ests[1].prior(cap_n=512)
        Jo = o.prior["J"].reshape(o.prior["n"], o.prior["n"])
        if pp["n"] == o.prior["n"] and list(pp["block_id"]) == list(o.prior["block_id"]):
            Ao, Ap, bo, bp = Jo.T @ Jo, pp["J"].T @ pp["J"], Jo.T @ o.prior["r"], pp["J"].T @ pp["r"]
            d = np.sqrt(np.maximum(np.diag(Ao), 1e-300))
            ev = np.linalg.eigvalsh(Ao)
            line += " | prior n %d dA %.1e dAs %.1e db %.1e eig %.1e..%.1e n<1e-6: %d" % (pp["n"], np.abs(Ap - Ao).max(), np.abs((Ap - Ao) / np.outer(d, d)).max(), np.abs(bp - bo).max(), ev.min(), ev.max(), int((ev < 1e-6).sum()))
            ia = [i for i, b in enumerate(pp["block_id"]) if b // 4096 == 13]
            if ia:
                off = sum((6 if (b // 4096) in (0, 2, 3) else 9 if b // 4096 == 1 else 1) for b in pp["block_id"][:ia[0]])
                line += " Aanc %s" % np.diag(Ao)[off:off + 3].round(3)
        else:
            line += " | prior shapes differ: n %d vs %d" % (pp["n"], o.prior["n"])
    if product and "--single" in sys.argv and o.gnss_ready:
        # the oracle's own window of this frame, solved by the product's back end alone: separates the solve from the closed loop
        import oracle_py as O
        if "ba1" not in globals():
            ba1 = gfamd.Estimator(window_size=W, max_features=512, max_visual=8192, batch=1, max_gnss=32 * (W + 1))
        w1, w2, w3 = o.last_window.copy(), o.last_window.copy(), o.last_window.copy()
        w3["para_Pose"] = w3["para_Pose"] + 1e-12 * np.random.default_rng(k).normal(0, 1, np.shape(w3["para_Pose"]))   # the oracle against itself, poses nudged by 1e-12 m
        ba1.solve([w1], 8)
        O.ba_solve(w2, 8)
        O.ba_solve(w3, 8)
        line += " | single: danc %.2e dP %.2e dclk %.2e  | self: danc %.2e dP %.2e" % (
            np.abs(w1["para_anc_ecef"] - w2["para_anc_ecef"]).max(), np.abs(w1["para_Pose"].reshape(-1, 7)[:, :3] - w2["para_Pose"].reshape(-1, 7)[:, :3]).max(),
            np.abs(w1["para_rcv_dt"] - w2["para_rcv_dt"]).max(), np.abs(w3["para_anc_ecef"] - w2["para_anc_ecef"]).max(),
            np.abs(w3["para_Pose"].reshape(-1, 7)[:, :3] - w2["para_Pose"].reshape(-1, 7)[:, :3]).max())
    if product and "--single" in sys.argv and o.gnss_ready and o.prior is not None and getattr(o, "last_marg_window", None) is not None:
        pm = ba1.marginalize([o.last_marg_window.copy()], o.marginalization_flag, cap_n=512)[0]
        po = O.ba_marginalize(o.last_marg_window.copy(), o.marginalization_flag)
        if pm is not None and po is not None and pm["n"] == po["n"]:
            n = po["n"]
            Jo, Jp = po["J"].reshape(n, n), pm["J"][:n * n].reshape(n, n)
            Ao, Ap, bo, bp = Jo.T @ Jo, Jp.T @ Jp, Jo.T @ po["r"], Jp.T @ pm["r"][:n]
            d = np.sqrt(np.maximum(np.diag(Ao), 1e-300))
            line += " | marg alone: dA %.1e dAs %.1e db %.1e" % (np.abs(Ap - Ao).max(), np.abs((Ap - Ao) / np.outer(d, d)).max(), np.abs(bp - bo).max())
        else:
            line += " | marg alone: shapes differ %s %s" % (pm and pm["n"], po and po["n"])
        o.last_marg_window = None
    print(line)
