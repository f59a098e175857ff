"""phase clocks of the two linearisation kernels and of ba_step (profiling build: scripts/build_profile.py, run with GF_LIB_PATH=.../libgroundfusion_hip_prof.so)"""
import sys, ctypes as C
sys.path.insert(0, 'ground-fusion_amd')
import numpy as np, gfamd, synth_window as SW
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
est = gfamd.Estimator(batch=B)
base = [SW.make_window(1000 + b, gfamd) for b in range(8)]
wins = [base[b % 8] for b in range(B)]
est.upload(wins)
for it in (1, 2, 3):
    est.solve_resident(it, -1, True)
    st = np.zeros(128, np.int64)
    gfamd._chk(gfamd.lib().gf_ba_debug_stamps(est.h, st.ctypes.data_as(C.POINTER(C.c_longlong)), 128))
    print("iters", it)
    print("  ba_step   total %d phases %s | chol diag/panel/trail %s | diag block copy-in/factor/inverse/write-back (sum over blocks) %s" % (st[14] - st[0], np.diff(st[:15]).tolist(), st[20:23].tolist(), st[24:28].tolist()))
    m = st[64:76]
    print("  misc_win  total %d | zero+tables %d | imu eval end +%d wheel eval end +%d prior g/cost end +%d H gather end +%d | barrier %d | tiles %d | band gather %d" % (
        m[11] - m[0], m[1] - m[0], m[2] - m[1], m[3] - m[1], m[4] - m[1], m[5] - m[1], m[6] - m[1], m[7] - m[6], m[11] - m[7]))
    v = st[80:87]
    print("  visual    total %d | zero+ranges %d | main loop (wave 0) %d | barrier %d | fold %d | Vc reduce %d | Et rows %d" % (v[6] - v[0], v[1] - v[0], v[2] - v[1], v[3] - v[2], v[4] - v[3], v[5] - v[4], v[6] - v[5]))
