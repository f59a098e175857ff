import sys, ctypes as C
sys.path.insert(0, 'ground-fusion_amd')
import numpy as np, gfamd, synth_window as SW
est = gfamd.Estimator(batch=256)
base = [SW.make_window(1000 + b, gfamd) for b in range(8)]
wins = [base[b % 8] for b in range(256)]
est.upload(wins)
for it in (1, 8):
    est.solve_resident(it, -1, True)
    st = np.zeros(64, np.int64)
    gfamd._chk(gfamd.lib().gf_ba_debug_stamps(est.h, st.ctypes.data_as(C.POINTER(C.c_longlong)), 64))
    t0 = st[32]
    for wv in range(4):
        v = st[32 + 8 * wv: 32 + 8 * wv + 6] - t0
        print(it, "wave", wv, "start %d | phase1 done %d | after sync %d | task1 done %d | task2 done %d | all tasks %d" % tuple(v.tolist()))
