"""stage clocks of one IMU and one wheel wavefront of ba_linearize_misc<false> (profiling build: python scripts/build_profile.py)"""
import sys, ctypes as C
sys.path.insert(0, 'ground-fusion_amd')
import numpy as np, gfamd, synth_window as SW
est = gfamd.Estimator(batch=256)
base = [SW.make_window(1000 + b, gfamd) for b in range(8)]
wins = [base[b % 8] for b in range(256)]
est.upload(wins)
for it in (1, 2, 8):
    est.solve_resident(it, -1, True)
    st = np.zeros(64, np.int64)
    gfamd._chk(gfamd.lib().gf_ba_debug_stamps(est.h, st.ctypes.data_as(C.POINTER(C.c_longlong)), 64))
    for o, name in ((32, "imu"), (40, "wheel")):
        v = st[o:o + 6]
        print(it, name, "raw eval %d, whiten r + cost %d, S J %d, J^T J + atomics %d, drain %d  (cycles; total %d)" % (*np.diff(v).tolist(), v[5] - v[0]))
