"""W=20 / 500-feature windows (BASELINE.json config 5 without GNSS): HIP vs oracle solve + both marginalisation kinds."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gfamd, oracle_py as O, synth_window as SW
W = int(sys.argv[1]) if len(sys.argv) > 1 else 20
F = int(sys.argv[2]) if len(sys.argv) > 2 else 500
ba = gfamd.Estimator(W, F, F * W)
for seed in (1, 2):
    w = SW.make_window(seed, O, W=W, n_landmarks=int(F * 1.5), max_features=F)
    lo = O.ba_linearize(w.copy(), cap=1024); lp = ba.linearize(w.copy(), cap=1024)
    print('linearize: cost %.6f/%.6f n_f %d/%d n_e %d/%d dH %.3e (|H| %.3e) dg %.3e (|g| %.3e)' % (lp['cost'], lo['cost'], lp['n_f'], lo['n_f'], lp['n_e'], lo['n_e'], np.abs(lp['H'] - lo['H']).max(), np.abs(lo['H']).max(), np.abs(lp['g'] - lo['g']).max(), np.abs(lo['g']).max()))
    for it in (1, 2):
        a, b2 = w.copy(), w.copy(); so = O.ba_solve(a, it); sp = ba.solve([b2], it)[0]
        print('   iters %d: steps %d/%d cost %.6f/%.6f dPose %.2e dFeat %.2e' % (it, sp['successful_steps'], so['successful_steps'], sp['final_cost'], so['final_cost'], np.abs(a['para_Pose'] - b2['para_Pose']).max(), np.abs(a['para_Feature'] - b2['para_Feature']).max()))
    w1, w2 = w.copy(), w.copy()
    t0 = time.time(); so = O.ba_solve(w1, 8); t1 = time.time(); sp = ba.solve([w2], 8)[0]; t2 = time.time()
    print("W %d seed %d n_feat %d n_vis %d: it %d/%d steps %d/%d cost %.6f/%.6f dPose %.2e dSB %.2e dFeat %.2e  oracle %.3fs hip %.3fs" % (
        W, seed, w["n_feature"], w["n_visual"], sp["iterations"], so["iterations"], sp["successful_steps"], so["successful_steps"], sp["final_cost"], so["final_cost"],
        np.abs(w1["para_Pose"] - w2["para_Pose"]).max(), np.abs(w1["para_SpeedBias"] - w2["para_SpeedBias"]).max(), np.abs(w1["para_Feature"] - w2["para_Feature"]).max(), t1 - t0, t2 - t1))
    for mode in (0, 1):
        po = O.ba_marginalize(w1, mode, cap_n=512)
        pp = ba.marginalize([w1], mode, cap_n=512)[0]
        if po is None or pp is None:
            print("   mode", mode, "oracle", po is not None, "hip", pp is not None); continue
        n = po["n"]; Jo = po["J"].reshape(n, n); Jp = pp["J"].reshape(pp["n"], pp["n"])
        Ao, Ap = Jo.T @ Jo, Jp.T @ Jp
        print("   marg mode %d: n %d/%d ids %s dA %.2e (|A| %.2e) db %.2e (|b| %.2e)" % (mode, pp["n"], n, list(pp["block_id"]) == list(po["block_id"]), np.abs(Ap - Ao).max(), np.abs(Ao).max(),
              np.abs(Jp.T @ pp["r"] - Jo.T @ po["r"]).max(), np.abs(Jo.T @ po["r"]).max()))
        if mode == 0:
            # chain: next window with the new prior
            wn = SW.make_window(seed, O, W=W, n_landmarks=int(F * 1.5), max_features=F, frame0=1, prior=None)
            a, b2 = wn.copy(), wn.copy()
            a.set_prior(po); b2.set_prior(pp)
            so2 = O.ba_solve(a, 8); sp2 = ba.solve([b2], 8)[0]
            print("   chained solve: it %d/%d cost %.6f/%.6f dPose %.2e" % (sp2["iterations"], so2["iterations"], sp2["final_cost"], so2["final_cost"], np.abs(a["para_Pose"] - b2["para_Pose"]).max()))
