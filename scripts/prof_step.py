import sys, ctypes as C
sys.path.insert(0,'ground-fusion_amd')
import numpy as np, gfamd, synth_window as SW
est=gfamd.Estimator(batch=256)
base=[SW.make_window(1000+b, gfamd) for b in range(8)]; wins=[base[b%8] for b in range(256)]
est.upload(wins)
for it in (1,2,3):
    est.solve_resident(it, -1, True)
    st=np.zeros(96, np.int64)
    gfamd._chk(gfamd.lib().gf_ba_debug_stamps(est.h, st.ctypes.data_as(C.POINTER(C.c_longlong)), 96))
    d=np.diff(st[:15]); print(it, 'total', (st[14]-st[0]), 'phases', d.tolist(), 'chol diag/panel/trail', st[20:23].tolist(), 'diag sub-phases (accumulated over the blocks after the first)', st[24:29].tolist(), 'backsubst: wave0 work / wave0 barrier wait / wave1 barrier wait', st[29:32].tolist(), 'per block column: wave 0 (tile + diag)', st[40:52].tolist(), 'wave 1 trailing', st[56:68].tolist(), 'wave 4 trailing (shares SIMD 0)', st[72:84].tolist())
