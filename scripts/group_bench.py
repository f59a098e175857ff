"""end-to-end throughput of gf_estimator_group: n sequences (the same seeded stream replicated) through inputFeature -> processImage ->
batched solve + marginalisation; reports window-solves/s once the windows are live"""
import sys, time
sys.path.insert(0, "ground-fusion_amd")
import numpy as np, gfamd, synth_stream as SS

st = SS.Stream(1, t_still=1.5, t_move=2.0, v_max=0.4, yaw0=0.0, yaw_turn=-0.6, split_x=1.8, turn_delay=0.8)
st._lm = st._landmarks(1600)
st._pn = np.random.default_rng(4001).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
import ctypes as C
OBS = np.dtype([("id", "i4"), ("camera_id", "i4"), ("v", "f8", 8)])
frames = {}
for k in range(0, len(st.cam_t), 2):     # marshalled once, outside the timed region: the C-ABI is what is measured, not ctypes loops
    f = st.feature_frame(k)
    a = np.zeros(len(f), OBS)
    for j, i in enumerate(sorted(f)):
        a[j]["id"] = i; a[j]["v"] = f[i]
    frames[k] = a
for n in [int(a) for a in sys.argv[1:]] or [64]:
    cfg = gfamd.default_estimator_cfg(tio=SS.TIO, rio=SS.RIO, multiple_thread=1)
    grp = gfamd.EstimatorGroup(cfg, n)
    tp, live, t_live, steps = -1.0, False, 0.0, 0
    for k in range(len(st.cam_t)):
        for m in grp.members:
            t1 = st.feed(m, k, tp)
        tp = t1
        if k % 2:
            continue
        obs = np.tile(frames[k], n)
        sq = np.arange(n, dtype=np.int32); tt = np.full(n, float(st.cam_t[k])); no = np.full(n, len(frames[k]), np.int32)
        t0 = time.perf_counter()
        gfamd._chk(gfamd.lib().gf_estimator_group_input_features(grp.g, n, sq.ctypes.data_as(C.POINTER(C.c_int)), tt.ctypes.data_as(C.POINTER(C.c_double)),
                                                                 obs.ctypes.data_as(C.c_void_p), no.ctypes.data_as(C.POINTER(C.c_int))))
        dt = time.perf_counter() - t0
        if live:
            t_live += dt; steps += 1
        live = grp.members[0].state()["solver_flag"] == 1
    s = grp.stats()
    print("n=%d: %d live steps, %.2f ms per group step -> %.0f window-solves/s end to end (host bookkeeping + upload + solve + marginalise + download); %s"
          % (n, steps, 1e3 * t_live / max(steps, 1), n * steps / max(t_live, 1e-9), s))
    grp.close()
