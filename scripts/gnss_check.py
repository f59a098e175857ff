import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gfamd, oracle_py as O, synth_window as SW
W = int(sys.argv[1]) if len(sys.argv) > 1 else 10
F = int(sys.argv[2]) if len(sys.argv) > 2 else 150
ba = gfamd.Estimator(W, F, F * W, 1, max_gnss=12 * (W + 1))
for seed in (1, 2):
    for kw in (dict(), dict(anchor=True), dict(gnss_lowspeed=1)):
        w = SW.make_window(seed, O, W=W, n_landmarks=int(F * 1.5), max_features=F, gnss=True, **kw)
        lo = O.ba_linearize(w.copy(), cap=1024); lp = ba.linearize(w.copy(), cap=1024)
        same_ids = list(lo["ids"]) == list(lp["ids"])
        print("seed %d %s linearize: cost %.6f/%.6f n_f %d/%d n_e %d/%d ids %s dH %.3e (|H| %.3e) dg %.3e (|g| %.3e)" % (seed, kw, lp["cost"], lo["cost"], lp["n_f"], lo["n_f"], lp["n_e"], lo["n_e"], same_ids,
              np.abs(lp["H"] - lo["H"]).max() if lp["H"].shape == lo["H"].shape else -1, np.abs(lo["H"]).max(), np.abs(lp["g"] - lo["g"]).max() if lp["g"].shape == lo["g"].shape else -1, np.abs(lo["g"]).max()))
        a, b = w.copy(), w.copy()
        so = O.ba_solve(a, 8); sp = ba.solve([b], 8)[0]
        print("   solve: it %d/%d steps %d/%d cost %.6f/%.6f dPose %.2e dSB %.2e dClk %.2e dDdt %.2e dAnc %.2e" % (sp["iterations"], so["iterations"], sp["successful_steps"], so["successful_steps"], sp["final_cost"], so["final_cost"],
              np.abs(a["para_Pose"] - b["para_Pose"]).max(), np.abs(a["para_SpeedBias"] - b["para_SpeedBias"]).max(), np.abs(a["para_rcv_dt"] - b["para_rcv_dt"]).max(), np.abs(a["para_rcv_ddt"] - b["para_rcv_ddt"]).max(),
              np.abs(a["para_anc_ecef"] - b["para_anc_ecef"]).max()))
        for mode in (0,):
            po = O.ba_marginalize(a, mode, cap_n=512); pp = ba.marginalize([a], mode, cap_n=512)[0]
            n = po["n"]; Jo = po["J"].reshape(n, n)
            if pp["n"] != n: print("   marg n differs", pp["n"], n, list(pp["block_id"]), list(po["block_id"])); continue
            Jp = pp["J"].reshape(n, n)
            print("   marg mode %d: n %d ids %s dA %.2e (|A| %.2e) db %.2e (|b| %.2e) dx0 %.2e" % (mode, n, list(pp["block_id"]) == list(po["block_id"]), np.abs(Jp.T @ Jp - Jo.T @ Jo).max(), np.abs(Jo.T @ Jo).max(),
                  np.abs(Jp.T @ pp["r"] - Jo.T @ po["r"]).max(), np.abs(Jo.T @ po["r"]).max(), np.abs(pp["x0"] - po["x0"]).max()))
            w2 = SW.make_window(seed, O, W=W, n_landmarks=int(F * 1.5), max_features=F, gnss=True, frame0=1, prior=pp, **{k: v for k, v in kw.items() if k != "anchor"})
            c, e = w2.copy(), w2.copy()
            s2o = O.ba_solve(c, 8); s2p = ba.solve([e], 8)[0]
            print("   chained: it %d/%d steps %d/%d cost %.6f/%.6f dPose %.2e dClk %.2e" % (s2p["iterations"], s2o["iterations"], s2p["successful_steps"], s2o["successful_steps"], s2p["final_cost"], s2o["final_cost"],
                  np.abs(c["para_Pose"] - e["para_Pose"]).max(), np.abs(c["para_rcv_dt"] - e["para_rcv_dt"]).max()))
            p1o = O.ba_marginalize(c, 1, cap_n=512); p1p = ba.marginalize([c], 1, cap_n=512)[0]
            if p1o is not None and p1p is not None:
                n1 = p1o["n"]; J1o = p1o["J"].reshape(n1, n1); J1p = p1p["J"].reshape(n1, n1)
                print("   second-new marg: n %d/%d ids %s dA %.2e (|A| %.2e)" % (p1p["n"], n1, list(p1p["block_id"]) == list(p1o["block_id"]), np.abs(J1p.T @ J1p - J1o.T @ J1o).max(), np.abs(J1o.T @ J1o).max()))
            else: print("   second-new marg:", p1o is not None, p1p is not None)
