"""which device buffers of a gf_ba handle are read before this handle wrote them?  GF_BA_POISON_RANGE=lo:hi fills the allocations lo..hi-1 of the process with 0xFF
instead of zeros; a run whose results change names a buffer that is read unwritten.  python scripts/poison_bisect.py  (GPU)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = r'''
import sys, numpy as np
sys.path.insert(0, "ground-fusion_amd"); sys.path.insert(0, "oracle")
import gfamd, oracle_py as O, synth_window as SW
w = SW.make_window(1, O, gnss=%s)
est = gfamd.Estimator(10, 150, 1500, 2, max_gnss=%s)
ws = [w.copy(), w.copy()]
s = est.solve(ws, 8)
p = est.marginalize(ws, 0)
print(repr((s[1]["final_cost"], s[1]["iterations"], float(np.abs(p[1]["r"]).sum()), float(np.abs(ws[1]["para_Pose"]).sum()))))
'''
def run(lo, hi, gnss):
    env = dict(os.environ, GF_BA_POISON_RANGE="%d:%d" % (lo, hi))
    out = subprocess.run([sys.executable, "-c", PROBE % (("True", "132") if gnss else ("False", "0"))], capture_output=True, text=True, cwd=ROOT, env=env)
    return out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "ERR " + out.stderr[-200:]
for gnss in (False, True):
    ref = run(10000, 10001, gnss)
    bad = []
    def rec(lo, hi):
        if run(lo, hi, gnss) == ref:
            return
        if hi - lo == 1:
            bad.append(lo); return
        mid = (lo + hi) // 2
        rec(lo, mid); rec(mid, hi)
    rec(0, 96)
    print("gnss", gnss, "reference", ref, "-> allocations read before written:", bad)
