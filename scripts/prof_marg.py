import sys, ctypes as C
sys.path.insert(0,'ground-fusion_amd')
import numpy as np, gfamd, synth_window as SW
est=gfamd.Estimator(batch=64)
wins=[SW.make_window(1000+b, gfamd) for b in range(64)]
est.upload(wins)
est.solve_resident(8, 0, True)
_, pri = est.download(wins, True)
w1=[SW.make_window(1000+b, gfamd, frame0=1, prior=pri[b]) for b in range(64)]
est.upload(w1)
for it in range(2):
    est.solve_resident(8, 0, True)
    st=np.zeros(128, np.int64)
    gfamd._chk(gfamd.lib().gf_ba_debug_stamps(est.h, st.ctypes.data_as(C.POINTER(C.c_longlong)), 128))
    print('marg phases', np.diff(st[32:38]).tolist(), 'least-squares rhs: total', int(st[38] - st[37]), 'staging', int(st[40] - st[37]), 'backward subst', int(st[41] - st[40]),
          'G', int(st[42] - st[41]), 'solve', int(st[43] - st[42]), 'T w + forward subst', int(st[38] - st[43]), 'rank*1000+n', int(st[39]), 'pivot loop: search / swap / column / hand-over / update', st[100:105].tolist(), 'entry -> first stamp / last stamp -> exit / block 0 whole / longest block since the handle was made', st[106:110].tolist(), est.stats()['ms_marginalize'])
