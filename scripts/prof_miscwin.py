"""stage clocks of ba_linearize_misc_win (profiling build, scripts/build_profile.py): stamps 64..75 of block 0"""
import sys, ctypes as C
sys.path.insert(0, 'ground-fusion_amd')
import numpy as np, gfamd, synth_window as SW
est = gfamd.Estimator(batch=256)
base = [SW.make_window(1000 + b, gfamd) for b in range(8)]
est.upload([base[b % 8] for b in range(256)])
est.solve_resident(2, -1, True)
st = np.zeros(128, np.int64)
gfamd._chk(gfamd.lib().gf_ba_debug_stamps(est.h, st.ctypes.data_as(C.POINTER(C.c_longlong)), 128))
t0 = st[64]
names = {65: "maps + prior dx", 66: "wave 0: IMU factors", 67: "wave 1: wheel factors", 68: "waves 2..: prior A dx -> g", 69: "waves 2..: prior A -> H", 70: "after phase 1 barrier", 71: "phase 2", 75: "end"}
for i in sorted(names):
    print("%3d %-28s +%d cycles" % (i, names[i], st[i] - t0))
