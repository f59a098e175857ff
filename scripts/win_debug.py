import os, sys, pickle
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gfamd, oracle_py as O, gfwindow as gw
d = pickle.load(open(sys.argv[1], "rb"))
w = gw.Window(); w.update(d); w.finalize()
ba = gfamd.Estimator(10, 512, 4096)
lo = O.ba_linearize(w.copy(), cap=1024); lp = ba.linearize(w.copy(), cap=1024)
print("linearize: cost %.6f/%.6f dH %.3e (|H| %.3e) dg %.3e (|g| %.3e)" % (lp["cost"], lo["cost"], np.abs(lp["H"] - lo["H"]).max(), np.abs(lo["H"]).max(), np.abs(lp["g"] - lo["g"]).max(), np.abs(lo["g"]).max()))
print("para_Feature range", w["para_Feature"].min(), w["para_Feature"].max())
for it in range(1, 9):
    w1, w2 = w.copy(), w.copy()
    so = O.ba_solve(w1, it); sp = ba.solve([w2], it)[0]
    print("iters %d: it %d/%d steps %d/%d cost %.6f/%.6f radius %.4e/%.4e dPose %.2e dSB %.2e dFeat %.2e" % (it, sp["iterations"], so["iterations"], sp["successful_steps"], so["successful_steps"],
          sp["final_cost"], so["final_cost"], sp["radius"], so["radius"], np.abs(w1["para_Pose"] - w2["para_Pose"]).max(), np.abs(w1["para_SpeedBias"] - w2["para_SpeedBias"]).max(),
          np.abs(w1["para_Feature"] - w2["para_Feature"]).max()))
