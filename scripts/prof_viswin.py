"""stage clocks of ba_linearize_visual_win (profiling build, scripts/build_profile.py): stamps 80..86 of block 0"""
import sys, ctypes as C
sys.path.insert(0, 'ground-fusion_amd')
import numpy as np, gfamd, synth_window as SW
est = gfamd.Estimator(batch=256)
base = [SW.make_window(1000 + b, gfamd) for b in range(8)]
est.upload([base[b % 8] for b in range(256)])
est.solve_resident(2, -1, True)   # run with GF_BA_COST_ONLY=0: else the last sweep of the solve is the cost-only one and leaves no stamps
if "marg" in sys.argv:   # the MARGIN_OLD sweep (extrinsic columns kept, 6 wavefronts, only the factors of features that start at frame 0) is the last visual launch then
    est.solve_resident(0, 0, False)
st = np.zeros(128, np.int64)
gfamd._chk(gfamd.lib().gf_ba_debug_stamps(est.h, st.ctypes.data_as(C.POINTER(C.c_longlong)), 128))
print("stamps 80..86 relative to 80:", (st[80:87] - st[80]).tolist(), "first chunk of wavefront 0: evaluation", int(st[93] - st[92]), "rest of the chunk loop (staging + MFMA, later chunks)", int(st[94] - st[93]), "last flush", int(st[82] - st[94]), "first chunk: staging", int(st[95]), "MFMA loops + flushes", int(st[96]))
