"""where the wall time of the drop-in loop goes (bench.end_to_end_sample instrumented): python scripts/e2e_timeline.py [nseq]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd"))
import ctypes as C
import numpy as np, torch, gfamd, synth_stream as SS
nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
for rep in range(2):
    st = SS.Stream(1, t_still=1.5, t_move=1.5, v_max=0.4, yaw0=0.0, yaw_turn=-0.6, split_x=1.8, turn_delay=0.8)
    cfg = gfamd.default_estimator_cfg(tio=SS.TIO, rio=SS.RIO, multiple_thread=1)
    grp = gfamd.EstimatorGroup(cfg, nseq)
    trk = gfamd.FeatureTracker(gfamd.default_cfg(batch=nseq))
    frames, cache = [], {}
    for k in range(len(st.cam_t)):
        key = (tuple(np.round(st.p_wb(st.cam_t[k]), 9)), round(float(st._at(st._psi, st.cam_t[k])), 9))
        if key not in cache:
            img, dep = st.image(k)
            cache[key] = (torch.from_numpy(img).to(dev), torch.from_numpy(dep.view(np.int16)).to(dev))
        frames.append(cache[key])
    sq = np.arange(nseq, dtype=np.int32)
    state = {"thread": None, "t_group": 0.0, "n_group": 0}
    def group_step(tk, obs, no):
        t0 = time.perf_counter()
        tt = np.full(nseq, tk)
        gfamd._chk(gfamd.lib().gf_estimator_group_input_features(grp.g, nseq, sq.ctypes.data_as(C.POINTER(C.c_int)), tt.ctypes.data_as(C.POINTER(C.c_double)), obs.ctypes.data_as(C.c_void_p), no.ctypes.data_as(C.POINTER(C.c_int))))
        state["t_group"] += time.perf_counter() - t0; state["n_group"] += 1
    T = dict(expand=0.0, track=0.0, pack=0.0, join=0.0, feed=0.0)
    tp, live, t_start = -1.0, False, None
    for k in range(len(st.cam_t)):
        t0 = time.perf_counter()
        g = frames[k][0].unsqueeze(0).expand(nseq, -1, -1).contiguous(); d = frames[k][1].unsqueeze(0).expand(nseq, -1, -1).contiguous(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        n = trk.trackImageBatchDevice([float(st.cam_t[k])] * nseq, g.data_ptr(), d.data_ptr(), unpack=False)
        t2 = time.perf_counter()
        if live: T["expand"] += t1 - t0; T["track"] += t2 - t1
        if k % 2 == 0:
            out = trk._out
            obs = np.ascontiguousarray(out[np.arange(out.shape[1])[None, :] < n[:, None]]); no = np.ascontiguousarray(n, np.int32).copy()
            t3 = time.perf_counter()
            if state["thread"] is not None: state["thread"].join(); state["thread"] = None
            t4 = time.perf_counter()
            now_live = grp.members[0].state()["solver_flag"] == 1
            if now_live and not live:
                live, t_start = True, time.perf_counter(); state["t_group"] = 0.0; state["n_group"] = 0
            for kk in (k - 1, k):
                if kk >= 0:
                    for m in grp.members: t1_ = st.feed(m, kk, tp)
                    tp = t1_
            t5 = time.perf_counter()
            if live: T["pack"] += t3 - t2; T["join"] += t4 - t3; T["feed"] += t5 - t4
            state["thread"] = threading.Thread(target=group_step, args=(float(st.cam_t[k]), obs, no)); state["thread"].start()
    state["thread"].join()
    tot = time.perf_counter() - t_start
    print("pass %d: live wall %.1f ms (feed %.1f excluded in the bench): expand frames %.1f, tracker calls %.1f, obs packing %.1f, waiting for the group %.1f; group steps %d x %.2f ms"
          % (rep, 1e3 * tot, 1e3 * T["feed"], 1e3 * T["expand"], 1e3 * T["track"], 1e3 * T["pack"], 1e3 * T["join"], state["n_group"], 1e3 * state["t_group"] / max(state["n_group"], 1)))
    grp.close(); trk.close()
