"""cycle split of ba_step in its chain form (profiling build: scripts/build_profile.py, GF_LIB_PATH=.../libgroundfusion_hip_prof.so GF_BA_CHAIN=1)"""
import sys, ctypes as C
sys.path.insert(0, 'ground-fusion_amd')
import numpy as np, gfamd, synth_window as SW
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
est = gfamd.Estimator(batch=B)
base = [SW.make_window(1000 + b, gfamd) for b in range(8)]; wins = [base[b % 8] for b in range(B)]
est.upload(wins)
for it in (1, 2, 3):
    est.solve_resident(it, -1, True)
    st = np.zeros(96, np.int64)
    gfamd._chk(gfamd.lib().gf_ba_debug_stamps(est.h, st.ctypes.data_as(C.POINTER(C.c_longlong)), 96))
    d = np.diff(st[:15])
    print(it, 'total', st[14] - st[0], 'phases 0..14', d.tolist())
    print('   chain loop: (1) update + dense update %d, (2) factor | next panel cleared %d, (3) panel + next front %d; dense Cholesky + backward %d; backward through the chain %d'
          % (st[87], st[88], st[89], st[91] - st[90], st[10] - st[91]))
