import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gfamd as gf, oracle_py as oracle, synth_window as SW
seed = 12
for rep in range(8):
    est = gf.Estimator()
    w = SW.make_window(seed, oracle)
    wg = w.copy()
    so = oracle.ba_solve(w, 8); sg = est.solve([wg], 8)[0]
    d1 = np.abs(w["para_Pose"] - wg["para_Pose"]).max()
    po = oracle.ba_marginalize(w, 0)
    pg = est.marginalize([wg], 0)[0]
    n = po["n"]; Jo = po["J"].reshape(n, n); Jp = pg["J"].reshape(n, n)
    dA = np.abs(Jp.T @ Jp - Jo.T @ Jo).max(); db = np.abs(Jp.T @ pg["r"] - Jo.T @ po["r"]).max()
    w2o = SW.make_window(seed, oracle, frame0=1, prior=po)
    w2g = SW.make_window(seed, oracle, frame0=1, prior=pg)
    w2x = SW.make_window(seed, oracle, frame0=1, prior=pg)
    s2o = oracle.ba_solve(w2o, 8); s2g = est.solve([w2g], 8)[0]; s2x = oracle.ba_solve(w2x, 8)
    print("rep %d: solve1 dPose %.2e it %d/%d | prior dA %.2e db %.2e | chained dPose(gpu prior+gpu solve vs oracle) %.2e  (gpu prior + oracle solve vs oracle) %.2e  it %d/%d/%d steps %d/%d/%d" % (
        rep, d1, sg["iterations"], so["iterations"], dA, db, np.abs(w2o["para_Pose"] - w2g["para_Pose"]).max(), np.abs(w2o["para_Pose"] - w2x["para_Pose"]).max(),
        s2g["iterations"], s2o["iterations"], s2x["iterations"], s2g["successful_steps"], s2o["successful_steps"], s2x["successful_steps"]))
    est.close()
