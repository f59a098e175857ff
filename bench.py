#!/usr/bin/env python3
"""bench.py — Ground-Fusion sliding-window hot path on MI355X (contract: task brief; details in DESIGN.md "Measurement").

A step = one pass of the hot path over one batch of synthetic input already resident in HBM: every one of the `--batch`
independent 640x480 RGBD+IMU+wheel sequences owned by this GPU advances by one frame:
  front end  FeatureTracker::trackImage   (HIP pyramid + Scharr + LK forward/reverse + Shi-Tomasi top-up), and
  back end   Estimator::optimization()    (8 dogleg iterations of the 10-frame / 150-feature window + MARGIN_OLD prior).
One process per GPU.  Sequences are independent, so they are sharded across ranks with no data-path collective (weak
scaling); the one exchange north_star names -- the gather of the newest pose of every sequence over RCCL -- is part of every
timed step.  After the timed region the three roofline kernels are timed again on an otherwise idle GPU (tracker alone, back
end alone), and rank 0 at N = 1 runs two bounded side measurements: the end-to-end drop-in path (trackImage -> inputFeature ->
processImage -> solve -> marginalise through gf_estimator_group_*) and the CPU oracle.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ground-fusion_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

W, H = 640, 480
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_MFMA_PEAK_TF = 78.6  # MI355X FP64 matrix peak (v_mfma_f64_16x16x4_f64)


def make_frames(n_frames, batch, seed0, device):
    """Synthetic sequences generated on the GPU (plumbing): band-limited texture, similarity warp per frame.
    Returns uint8 [n_frames, batch, H, W] and uint16-as-int16 depth [batch, H, W]."""
    import synth
    n_tex = min(batch, 8)
    tex = torch.stack([torch.from_numpy(synth.make_texture(seed0 + i)) for i in range(n_tex)]).to(device)
    ys, xs = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32), torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
    frames = torch.empty((n_frames, batch, H, W), dtype=torch.uint8, device=device)
    rng = np.random.default_rng(seed0)
    for b in range(batch):
        vx, vy = rng.uniform(-6, 6), rng.uniform(-4, 4)
        wz = rng.uniform(-0.004, 0.004)
        ox, oy = rng.uniform(150, 300), rng.uniform(150, 300)
        t = tex[b % n_tex]
        for k in range(n_frames):
            ang, sc = wz * k, 1.0 + 0.002 * k
            c, s = np.cos(ang), np.sin(ang)
            xc, yc = xs - W / 2, ys - H / 2
            u = sc * (c * xc - s * yc) + W / 2 + vx * k + 0.37 * np.sin(0.9 * k) + ox
            v = sc * (s * xc + c * yc) + H / 2 + vy * k + oy
            u = torch.remainder(u, 1022.0); v = torch.remainder(v, 1022.0)
            x0 = u.floor().long(); y0 = v.floor().long()
            a = u - x0; bb = v - y0
            img = (1 - a) * (1 - bb) * t[y0, x0] + a * (1 - bb) * t[y0, x0 + 1] + (1 - a) * bb * t[y0 + 1, x0] + a * bb * t[y0 + 1, x0 + 1]
            frames[k, b] = img.round().clamp(0, 255).to(torch.uint8)
    depth = torch.full((batch, H, W), 1800, dtype=torch.int16, device=device)
    return frames, depth


def make_windows(gfamd, batch, seed0, features, window=10, gnss=False, distinct=None):
    """Steady-state windows (with a marginalisation prior) for `batch` sequences: window 0 is solved and marginalised on the
    GPU with the product path itself, the resulting prior feeds window 1 which is what the bench solves.  `distinct` < batch: only that
    many seeded windows are generated (host synthesis of a 500-feature window takes seconds) and dealt round-robin to the sequences."""
    import synth_window as SW
    nd = batch if not distinct else min(batch, distinct)
    est = gfamd.Estimator(window_size=window, max_features=features, max_visual=features * window, batch=batch, max_gnss=12 * (window + 1) if gnss else 0)
    kw = dict(W=window, max_features=features, n_landmarks=int(features * 1.5), gnss=gnss)
    w0 = [SW.make_window(seed0 + b, gfamd, **kw) for b in range(nd)]
    w0 = [w0[b % nd] if b < nd else w0[b % nd].copy() for b in range(batch)]
    est.upload(w0)
    est.solve_resident(8, 0, True)
    _, priors = est.download(w0, True)
    w1 = [SW.make_window(seed0 + b, gfamd, frame0=1, prior=priors[b], **kw) for b in range(nd)]
    w1 = [w1[b % nd] if b < nd else w1[b % nd].copy() for b in range(batch)]
    return est, w1


def cpu_baseline(frames_host, dt, wins, threads, ba_iters, max_cnt, min_dist, repeat=4):
    """CPU oracle (kind 'port', built -O3 -march=native on this box: BASELINE.md section 2) on a bounded sample of the same workload.
    Variant (a): every sequence single-threaded like the reference's solver (Ceres num_threads = 1, estimator.cpp:3306); `threads` sequences run side by
    side, one per core, which is the most favourable way to use the cores for THROUGHPUT.  Variant (b): one sequence at a time with the reference's own
    intra-sequence parallelism (OpenCV parallel_for_ over the LK points on all cores, 4 marginalisation threads, marginalization_factor.h:22)."""
    import threading
    import oracle_py
    native = oracle_py.use_native()
    oracle_py.lib()
    n_frames, nseq = frames_host.shape[0], frames_host.shape[1]
    depth = np.full((H, W), 1800, np.uint16)
    counts = [0] * nseq
    t_track = [0.0] * nseq
    t_ba = [0.0] * nseq
    n_ba = [0] * nseq

    def run(b, rep=repeat):
        t0 = time.perf_counter()
        for _ in range(rep):          # the same frames again through a fresh tracker: ~10 s of CPU work over all threads
            tr = oracle_py.Tracker(oracle_py.default_cfg(max_cnt=max_cnt, min_dist=min_dist))
            prev = set()
            for k in range(n_frames):
                ids, _ = tr.track(dt * k, frames_host[k, b], depth)
                if k > 0:
                    counts[b] += len(prev & set(ids.tolist()))
                prev = set(ids.tolist())
        t_track[b] = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(rep * max(1, n_frames // 4)):
            w = wins[b].copy()
            oracle_py.ba_solve(w, ba_iters)
            oracle_py.ba_marginalize(w, 0)
            n_ba[b] += 1
        t_ba[b] = time.perf_counter() - t0

    t0 = time.perf_counter()
    ths = [threading.Thread(target=run, args=(b,)) for b in range(nseq)]
    for i in range(0, nseq, threads):
        for th in ths[i:i + threads]:
            th.start()
        for th in ths[i:i + threads]:
            th.join()
    el = time.perf_counter() - t0
    # time for one full step of one sequence on one core = tracker frame + solve; `threads` cores run in parallel
    per_frame = sum(t_track) / (nseq * repeat * (n_frames - 1)) + 0.0
    per_solve = sum(t_ba) / sum(n_ba)
    steps_per_s = threads / (per_frame + per_solve)
    tracked_per_frame = sum(counts) / (nseq * repeat * (n_frames - 1))
    # variant (b): sequence 0 alone, LK over all cores + 4 marginalisation threads
    ncpu = min(os.cpu_count() or 1, 8)   # "per-point parallel LK over all 8 cores"; 150-500 points do not feed more threads than that
    oracle_py.set_threads(ncpu)
    t_track[0] = t_ba[0] = 0.0; n_ba[0] = 0; counts[0] = 0
    tb0 = time.perf_counter()
    run(0, 1)
    tb = time.perf_counter() - tb0
    oracle_py.set_threads(1)
    b_frame, b_solve = t_track[0] / (n_frames - 1), t_ba[0] / max(n_ba[0], 1)
    return {"steps_per_s": steps_per_s, "tracked_features_per_s": threads * tracked_per_frame / per_frame, "repeat": repeat,
            "solves_only_per_s": threads / per_solve, "ms_track_frame_1core": 1e3 * per_frame, "ms_solve_marg_1core": 1e3 * per_solve, "wall_s": el + tb,
            "build": "g++ -O3 -march=native -ffp-contract=off (built on this box)" if native else "g++ -O3 -ffp-contract=off (portable build: the native build failed on this box)",
            "variant_a_one_core_steps_per_s": 1.0 / (per_frame + per_solve),
            "variant_b": {"threads_lk": ncpu, "threads_marginalisation": 4, "ms_track_frame": 1e3 * b_frame, "ms_solve_marg": 1e3 * b_solve,
                          "steps_per_s": 1.0 / (b_frame + b_solve), "tracked_features_per_s": (counts[0] / (n_frames - 1)) / b_frame,
                          "note": "one sequence at a time, the reference's own intra-sequence parallelism; Ceres itself stays single-threaded (estimator.cpp:3306)"}}


def usable_cpus():
    """hardware threads this process can really use: its affinity mask AND its container's CPU quota (cgroup v2 cpu.max / v1 cfs quota) -- the boxes of round 6 show 256
    hardware threads and grant 16 cores' worth of bandwidth; the library sizes its pools the same way (csrc/gf_host_cpus.hpp)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def pmc_summary():
    """counter-derived figures of the roofline kernels, collected by separate rocprofv3 --pmc passes (scripts/pmc_collect.sh) and kept
    under profiles/ -- never hard-coded here.  Missing file: the fields that need it are null."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_summary.json")) as f:
            S = json.load(f)
    except (OSError, ValueError):
        return {}
    # counters describe the kernels they were collected on: a summary taken on other device code is not quoted (round-4 review: the LK kernel had changed under it)
    now = kernel_source_sha16()
    if S.get("kernel_source_sha16") != now:
        return {"_stale": "profiles/pmc_summary.json was collected on kernel sources %s, this tree has %s: counter-derived fields are null (scripts/pmc_collect.sh + pmc_summarize.py refresh it)"
                          % (S.get("kernel_source_sha16", "<unrecorded>"), now)}
    return S


def kernel_source_sha16():
    """first 16 hex digits of the SHA-256 over the device / host sources of the library (ground-fusion_amd/csrc/*.hip, *.hpp, sorted by name)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ground-fusion_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".hpp")):
            h.update(fn.encode()); h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


_E2E_INPUTS = {}
_E2E_STAGED = {}


def e2e_inputs(n_streams, dev):
    """`n_streams` distinct seeded RGB-D + IMU + wheel recordings of the same length, rendered once (input synthesis, outside every timed part).  Stream 0 is the
    recording the homogeneous sample replicates; the others differ in seed (texture, landmarks, noise), speed, turn and -- the point -- in when they start to move
    (0.1 s apart), so that at any frame some sequences vote "keyframe" (MARGIN_OLD) while others still stand (MARGIN_SECOND_NEW) and the batches the group forms
    are mixed.  Returns (streams, gray [frames][n_streams, H, W] u8, depth [frames][n_streams, H, W] i16)."""
    if n_streams in _E2E_INPUTS:
        return _E2E_INPUTS[n_streams]
    import synth_stream as SS
    T = 3.0
    streams = []
    for q in range(n_streams):
        t_still = 1.5 + 0.1 * q
        streams.append(SS.Stream(1 + q, t_still=t_still, t_move=T - t_still, v_max=0.4 - 0.02 * (q % 4), yaw0=0.0, yaw_turn=-0.6 if q % 2 == 0 else 0.5, split_x=1.8,
                                 turn_delay=0.8 - 0.05 * q))
    n = len(streams[0].cam_t)
    assert all(len(st.cam_t) == n and np.array_equal(st.cam_t, streams[0].cam_t) for st in streams)
    gray, depth = [], []
    for k in range(n):
        gs, ds = [], []
        for st in streams:
            cache = st.__dict__.setdefault("_bench_cache", {})
            key = (tuple(np.round(st.p_wb(st.cam_t[k]), 9)), round(float(st._at(st._psi, st.cam_t[k])), 9))   # identical poses (the stationary lead-in) share a frame
            if key not in cache:
                img, dep = st.image(k)
                cache[key] = (torch.from_numpy(img), torch.from_numpy(dep.view(np.int16)))
            gs.append(cache[key][0]); ds.append(cache[key][1])
        gray.append(torch.stack(gs).to(dev)); depth.append(torch.stack(ds).to(dev))
    _E2E_INPUTS[n_streams] = (streams, gray, depth)
    return _E2E_INPUTS[n_streams]


def end_to_end_sample(gfamd, nseq, dev, max_cnt, min_dist, device_preint=None, device_sweeps=False, n_streams=1, n_groups=1, stagger=True):
    """The drop-in path on a bounded sample: `nseq` sequences -- `n_streams` distinct seeded recordings dealt round-robin (1: one recording replicated, every
    member takes the same decisions; 8: staggered starts, mixed batches) -- through the batched tracker (trackImage on every camera frame) and gf_estimator_group_*
    (inputFeature -> processImage -> batched solve + marginalisation on every second frame, the reference's MULTIPLE_THREAD flow): the back end is fed by the
    tracker's own output; as in the reference the tracker (sync_process thread) and the estimators (processThread) run concurrently, one frame apart.  Returns
    window-solves/s over the frames on which all windows are live (NON_LINEAR): wall-clock including the tracker, host bookkeeping, uploads and downloads; the IMU /
    wheel samples are queued before the first frame (round 5: nothing is excluded from the clock).
    n_groups > 1: the sequences are split over that many estimator groups (gf_estimator_group_*: own back-end handle, own stream, own worker threads).  With
    `stagger` the groups alternate in which of every two camera frames is "the second one" (inputImageCnt % 2, estimator.cpp:420-428: a sequence that started one
    frame later) -- every sequence still hands every second frame of its own recording to its back end, but one group's host phase falls on the other group's batch."""
    import synth_stream as SS
    streams, gray, depth = e2e_inputs(n_streams, dev)
    st0 = streams[0]
    cfg = gfamd.default_estimator_cfg(tio=SS.TIO, rio=SS.RIO, multiple_thread=1)
    bounds = [nseq * q // n_groups for q in range(n_groups + 1)]
    # the library gives ONE group up to 32 workers (half of this rank's hardware threads when that is less); several groups of one process share that many.  Round 6, honest
    # clock, 256 hardware threads, gate spin off: 2 groups x 16 workers 34.6-45.4 k window-solves/s, 2 x 32: 40.6-44.6 k, 2 x 64: 28.3 k, 4 x 8: 40.2 k, 4 x 16: 43.7 k,
    # 8 x 4: 34.0 k (profiles/r06_e2e_pools.txt; the figures quoted here in round 5 -- "64 workers each 70.6 k" -- came from the clock that subtracted 0.56 s of 0.65)
    own_env = n_groups > 1 and "GF_GROUP_THREADS" not in os.environ
    if own_env:
        os.environ["GF_GROUP_THREADS"] = str(max(1, min(32, 2 * usable_cpus() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))) // n_groups))
    workers = os.environ.get("GF_GROUP_THREADS", "library default")
    grps = [gfamd.EstimatorGroup(cfg, bounds[q + 1] - bounds[q], device_preint=device_preint, device_sweeps=device_sweeps) for q in range(n_groups)]
    if own_env:
        del os.environ["GF_GROUP_THREADS"]
    members = [m for g_ in grps for m in g_.members]
    trk = gfamd.FeatureTracker(gfamd.default_cfg(batch=nseq, max_cnt=max_cnt, min_dist=min_dist))
    # every camera frame of every sequence resident in HBM before the clock starts (the contract of `value`: inputs already on the device), dealt to the sequences
    key = (n_streams, nseq)
    if key not in _E2E_STAGED:
        assign = torch.arange(nseq, device=dev) % n_streams
        _E2E_STAGED.clear()            # one staged set at a time (~0.9 MB per sequence and frame)
        torch.cuda.empty_cache()
        _E2E_STAGED[key] = ([gray[k].index_select(0, assign) for k in range(len(gray))], [depth[k].index_select(0, assign) for k in range(len(depth))])
    sgray, sdepth = _E2E_STAGED[key]
    torch.cuda.synchronize()
    G = range(n_groups)
    phase = [(q % 2) if stagger else 0 for q in G]          # the camera frames with k % 2 == phase[q] reach group q's back ends
    seqs = [np.arange(bounds[q + 1] - bounds[q], dtype=np.int32) for q in G]
    # the tracker writes every frame's padded output table into one of three, the estimators read their rows of it where it lies (no copy): a group holds the table
    # of frame k until it is waited for behind the tracker call of frame k + 2, which writes the table frame k - 1 used
    ring = [(np.zeros((nseq, trk.cap), gfamd.OBS_DTYPE), np.zeros(nseq, np.int32)) for _ in range(3)]
    ts_all = [np.full(nseq, float(t_)) for t_ in st0.cam_t]
    reps = [[next(b for b in range(bounds[q], bounds[q + 1]) if b % n_streams == s_) for s_ in range(n_streams)] for q in G]   # one member per recording and group
    tp = [[-1.0] * n_streams for _ in G]
    glive = [False] * n_groups
    solves, frames_live = 0, 0
    live, t_start = False, None
    # IMU / wheel samples: ALL of them are queued before the first camera frame (inputIMU / inputWheel only push into the members' buffers, estimator.cpp:318-372; a
    # member takes the interval of a frame out of its queue when the frame arrives).  Rounds 3-4 fed them frame by frame through ~4 k ctypes calls per back-end frame
    # and took that time out of the clock -- but with two alternating groups the OTHER group's step kept running during the excluded time (round-4 advisor): the rate
    # was flattered.  Round 5 queued the samples here but LEFT the per-frame feed loop and its subtraction in the live span (0.56 s of a 0.65 s span taken out: every
    # end_to_end figure of BENCH_r05.json is void).  Round 6: the loop below makes no feed() call and the live span is pc() - t_start
    # (tests/test_bench_clock.py runs this function against stubs and fails on an excluded time, on a late feed() and on rate x span != solves).
    for q in G:
        lo, hi = bounds[q], bounds[q + 1]
        for kk in range(len(st0.cam_t)):
            t1 = list(tp[q])
            for b in range(lo, hi):
                t1[b % n_streams] = streams[b % n_streams].feed(members[b], kk, tp[q][b % n_streams])
            tp[q] = t1
    steps_live = mixed = keyframe_votes = votes = 0
    clk = {"tracker": 0.0, "observations": 0.0, "wait_for_estimators": 0.0, "bookkeeping": 0.0}   # where the main thread's wall time goes while live [s]
    pc = time.perf_counter
    for k in range(len(st0.cam_t)):
        c1 = pc()
        # the tracker of this frame runs while the estimators still work on earlier back-end frames (separate threads in the reference too: rosNodeTest.cpp:713)
        tab, n = ring[k % 3]
        trk.trackImageBatchDevice(ts_all[k], sgray[k].data_ptr(), sdepth[k].data_ptr(), unpack=False, out=tab, n_out=n)
        c2 = pc()
        if live:
            clk["tracker"] += c2 - c1
            frames_live += 1
        for q in G:
            if k % 2 != phase[q]:
                continue
            lo, hi = bounds[q], bounds[q + 1]
            c2 = pc()
            buf, no = tab[lo:hi], n[lo:hi]
            c3 = pc()
            grps[q].wait()
            c4 = pc()
            if live:
                clk["observations"] += c3 - c2; clk["wait_for_estimators"] += c4 - c3
            fl = [members[b].flags() for b in reps[q]]        # (frame_count, solver_flag, marginalization_flag) of one member per recording
            if live:          # the decisions of the step that just finished
                flags = [r[2] for r in fl]
                steps_live += 1; mixed += int(len(set(flags)) > 1); keyframe_votes += sum(1 for f in flags if f == 0); votes += len(flags)
            glive[q] = all(r[1] == 1 for r in fl)
            if all(glive) and not live:
                live, t_start = True, pc()
            if live:
                solves += hi - lo
            grps[q].submitFeatures(seqs[q], ts_all[k][lo:hi], buf, no, stride=buf.shape[1])   # inputFeature of every sequence of the group (returns at once)
            if live:
                clk["bookkeeping"] += pc() - c4
    for g_ in grps:
        g_.wait()
    ts_ = trk.stats()
    calls_ = max(len(st0.cam_t), 1)
    trk_anatomy = {k_: round(ts_[k_] / calls_, 3) for k_ in ("ms_host_pre", "ms_wait_lk", "ms_host_mid", "ms_wait_detect", "ms_host_post") if k_ in ts_}
    t_live = (pc() - t_start) if t_start is not None else 0.0     # the live span is the wall clock, full stop
    stts = [g_.stats() for g_ in grps]
    stt = {"batches": sum(s_["batches"] for s_ in stts), "largest_batch": max(s_["largest_batch"] for s_ in stts)}
    pos = float(np.linalg.norm(members[0].state()["Ps"][-1]))
    for g_ in grps:
        g_.close()
    trk.close()
    bf = max(frames_live / 2.0, 1e-9)       # back-end frames of a sequence inside the live span
    return {"window_solves_per_s": solves / max(t_live, 1e-9), "sequences": nseq, "distinct_recordings": n_streams, "estimator_groups": n_groups,
            "groups_alternate_frames": bool(stagger and n_groups > 1), "live_camera_frames": frames_live, "window_solves": solves,
            "wall_s": t_live, "ms_per_backend_frame": 1e3 * t_live / bf, "group_batches": stt["batches"], "largest_batch": stt["largest_batch"],
            "backend_frames_with_mixed_decisions": mixed, "group_steps_live": steps_live, "keyframe_vote_share": keyframe_votes / max(votes, 1),
            "newest_position_norm_m": pos,
            "device_preint": bool(device_preint) if device_preint is not None else "library default (on when a worker thread carries >= 16 members, i.e. on small hosts)",
            "host_hardware_threads": os.cpu_count(), "host_usable_cpus": usable_cpus(), "group_worker_threads": workers, "tracker_ms_per_call": trk_anatomy,
            "main_thread_ms_per_backend_frame": {k_: round(1e3 * v_ / bf, 3) for k_, v_ in clk.items()},
            "imu_wheel_feed": "every sample queued before the first camera frame; the loop below it makes no feed() call and nothing is subtracted from the wall clock",
            "path": "gf_tracker_track_batch_device -> gf_estimator_group_submit_features / _wait (inputFeature -> processImage -> gf_ba solve + marginalise, "
                    "windows packed / uploaded / downloaded every frame)"}


def pcie_sample(gfamd, trk, est, args, B, frames, depth, dt, step, frame_index, n_host=4, K=24):
    """The boundary as the reference states it -- trackImage(const cv::Mat&) on HOST images (feature_tracker.h:47) -- next to the device-resident loop of `value`:
    the same step with every frame (gray u8 + depth u16) coming from page-locked host memory through gf_tracker_prefetch_batch / gf_tracker_track_prefetched, i.e.
    the copy of frame k + 1's gray image (307 200 B per sequence) on a copy stream under frame k's kernels; the depth image stays on the host, where the tracker samples it.  Reports the step both ways, the tracker alone both ways,
    and the bus rate the copies reached: where the copy is longer than the kernels the path is bound by the bus and that bound is the number."""
    hg = [frames[frame_index(i)].cpu().pin_memory() for i in range(n_host)]     # a ring of n_host frames x B sequences
    hd = depth.cpu().pin_memory()
    bytes_per_step = B * H * W      # round 5: only the gray image travels; the depth image's <= max_cnt samples per sequence are taken on the host (feature_tracker.cpp:360)

    def run(host, backend):
        torch.cuda.synchronize()
        if host:
            trk.prefetchHost(hg[0].data_ptr(), hd.data_ptr())
        t0 = time.perf_counter()
        for i in range(K):
            if backend:
                est.solve_resident_async(args.ba_iters, 0, True)
            ts = [dt * step[0]] * B
            if host:
                trk.prefetchHost(hg[(i + 1) % n_host].data_ptr(), hd.data_ptr())     # frame i + 1 starts to travel ...
                trk.trackPrefetched(ts, unpack=False)                                 # ... while frame i is tracked
            else:
                trk.trackImageBatchDevice(ts, frames.data_ptr() + frame_index(i % n_host) * B * H * W, depth.data_ptr(), unpack=False)
            if backend:
                est.wait()
            step[0] += 1
        if host:
            trk.trackPrefetched([dt * step[0]] * B, unpack=False)     # drain the frame that is still staged (outside the timed part)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K

    # a plain copy of one step's images, alone on the bus: what the host delivers
    torch.cuda.synchronize()
    dg = torch.empty_like(frames[0])
    t0 = time.perf_counter()
    for i in range(8):
        dg.copy_(hg[i % n_host], non_blocking=True)
    torch.cuda.synchronize()
    t_copy = (time.perf_counter() - t0) / 8
    run(True, False); run(False, False)       # warm: second pair of frame buffers, copy stream
    t_trk_dev, t_trk_host = run(False, False), run(True, False)
    t_dev, t_host = run(False, True), run(True, True)
    return {"bytes_per_step": bytes_per_step, "sequences": B, "steps": K, "host_frames_in_ring": n_host,
            "h2d_copy_alone_ms": 1e3 * t_copy, "h2d_GBps": bytes_per_step / t_copy / 1e9,
            "tracker_only_ms_per_step": {"device_resident": 1e3 * t_trk_dev, "host_images": 1e3 * t_trk_host},
            "step_ms": {"device_resident": 1e3 * t_dev, "host_images": 1e3 * t_host},
            "window_solves_per_s": {"device_resident": B / t_dev, "host_images": B / t_host},
            "host_over_device": t_dev / t_host,
            "bound": "bus" if t_copy > 0.9 * t_host else "kernels",
            "note": "host_images: the gray image of every sequence crosses PCIe every frame (gf_tracker_prefetch_batch: the copy of frame k + 1 runs under the kernels of frame k), the depth image is sampled on the host; "
                    "when h2d_copy_alone_ms exceeds the device-resident step the path is bound by the bus, not by the kernels"}


def small_batch_sample(gfamd, dev, args, WIN, GNSS, sizes=(1, 8, 64), K=40, distinct=None):
    """The same step at the batch sizes BASELINE.json's other configurations name -- one sequence (configs[0] / [1] as written), 8 per GPU (configs[3]: 64 sequences
    over 8 GPUs), 64 -- on handles of their own: ms per step (tracker frame + solve + MARGIN_OLD, device-resident inputs) and the back end alone.  The default run's
    256 sequences are what fills the chip; these are the latency end of the same path.  `large_batch` (round 6) is the same function at 512 / 1024 sequences per GPU
    (64 distinct seeded windows dealt round-robin: host synthesis of a thousand windows is not what is measured): where more than one window per CU is resident."""
    out = {}
    dt = 1.0 / 15.0
    for Bs in sizes:
        frames, depth = make_frames(8, Bs, 5000 + Bs, dev)
        trk = gfamd.FeatureTracker(gfamd.default_cfg(batch=Bs, max_cnt=args.max_cnt, min_dist=args.min_dist))
        est, wins = make_windows(gfamd, Bs, 5000 + Bs, args.max_cnt, WIN, GNSS, distinct if distinct else args.distinct)
        est.upload(wins)
        fb = Bs * H * W

        def fi(k):
            m = k % 14
            return m if m < 8 else 14 - m

        def run(n, tracker=True, backend=True, k0=0):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                if backend:
                    est.solve_resident_async(args.ba_iters, 0, True)
                if tracker:
                    trk.trackImageBatchDevice([dt * (k0 + i)] * Bs, frames.data_ptr() + fi(k0 + i) * fb, depth.data_ptr(), unpack=False)
                if backend:
                    est.wait()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n
        run(4)
        t_step = run(K, k0=4)
        t_be = run(K, tracker=False)
        t_tr = run(K, backend=False, k0=4 + K)
        out[str(Bs)] = {"ms_per_step": 1e3 * t_step, "window_solves_per_s": Bs / t_step, "backend_alone_ms": 1e3 * t_be, "tracker_alone_ms": 1e3 * t_tr}
        trk.close(); est.close()
        del frames, depth
    return out


CONFIGS = {1: dict(max_cnt=150, min_dist=30, window=10, gnss=False, batch=256, distinct=None),
           2: dict(max_cnt=300, min_dist=20, window=10, gnss=False, batch=256, distinct=None),
           4: dict(max_cnt=500, min_dist=12, window=20, gnss=True, batch=64, distinct=16)}


def config_sample(gfamd, dev, cfg_index, K=20, Wm=3, ba_iters=8, distinct=None):
    """BASELINE.json configs[cfg_index] as a short kernel-rate sample inside the default run (round-5 review: configs[2] and configs[4] only existed as builder-run files
    under profiles/): the same step as `value` -- tracker frame + 8 dogleg iterations + MARGIN_OLD for every sequence, inputs resident in HBM -- on handles of its own,
    K timed steps, then the three roofline kernels on an otherwise idle GPU for their fractions.  `python bench.py --config N` is the long form of the same."""
    C = CONFIGS[cfg_index]
    B, dt = C["batch"], 1.0 / 15.0
    nf = 8
    frames, depth = make_frames(nf, B, 7000 + cfg_index, dev)
    trk = gfamd.FeatureTracker(gfamd.default_cfg(batch=B, max_cnt=C["max_cnt"], min_dist=C["min_dist"]))
    trk.set_profiling(True)
    est, wins = make_windows(gfamd, B, 7000 + cfg_index, C["max_cnt"], C["window"], C["gnss"], distinct if distinct is not None else (C["distinct"] or 32))
    est.upload(wins)
    fb = B * H * W
    step = [0]

    def fi(k):
        m = k % (2 * nf - 2)
        return m if m < nf else 2 * nf - 2 - m

    def do_step():
        est.solve_resident_async(ba_iters, 0, True)
        trk.trackImageBatchDevice([dt * step[0]] * B, frames.data_ptr() + fi(step[0]) * fb, depth.data_ptr(), unpack=False)
        est.wait()
        step[0] += 1
    for _ in range(Wm + 1):
        do_step()
    trk.reset_stats(); est.reset_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        do_step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st, bs = trk.stats(), est.stats()
    trk.reset_stats(); est.reset_stats()
    for _ in range(3):
        trk.trackImageBatchDevice([dt * step[0]] * B, frames.data_ptr() + fi(step[0]) * fb, depth.data_ptr(), unpack=False)
        step[0] += 1
    torch.cuda.synchronize()
    for _ in range(2):
        est.solve_resident(ba_iters, 0, True)
    ti, bi = trk.stats(), est.stats()
    tf = lambda fl, ms: fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    lk_ms = ti["ms_lk"] / max(ti["lk_launches"], 1)
    lk_bytes = (484.0 * 5.0 * ti["lk_level_passes"] + 484.0 * ti["lk_iterations"]) / max(ti["lk_launches"], 1)
    jtj_ms, step_ms = bi["ms_jtj"] / max(bi["jtj_launches"], 1), bi["ms_step"] / max(bi["step_launches"], 1)
    out = {"workload": "configs[%d]: %d features / min_dist %d, %d-frame window, visual+IMU+wheel%s+prior, %d sequences on one GPU" % (cfg_index, C["max_cnt"], C["min_dist"], C["window"], "+GNSS" if C["gnss"] else "", B),
           "sequences_per_gpu": B, "steps": K, "warmup": Wm, "value": bs["solves"] / el, "unit": "window-solves/s (each with its tracker frame)", "ms_per_step": 1e3 * el / K,
           "tracked_features_per_s": st["tracked_features"] / el,
           "roofline_frac_isolated": lk_bytes / (lk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if lk_ms > 0 else None,
           "roofline_jtj_frac": tf(bi["jtj_alg_flops"] / max(bi["jtj_launches"], 1), jtj_ms) / FP64_MFMA_PEAK_TF,
           "roofline_step_frac": tf(bi["step_flops"] / max(bi["step_launches"], 1), step_ms) / FP64_MFMA_PEAK_TF,
           "launch_ms_isolated": {"lk_track_kernel": lk_ms, "ba_linearize_visual_win": jtj_ms, "ba_step": step_ms},
           "ba_solve_ms_isolated": bi["ms_solve"] / 2, "ba_marginalize_ms_isolated": bi["ms_marginalize"] / 2,
           "reduced_system_in": "global memory (ba_step<true>)" if C["window"] > 10 or C["gnss"] else "LDS"}
    trk.close(); est.close()
    del frames, depth
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=1, choices=(1, 2, 4),
                    help="index into BASELINE.json configs: 1 = 150 features, W = 10 (the configuration the metric is quoted on; default); 2 = 300 features / min_dist 20, "
                         "W = 10, full HIP path; 4 = 500 features / min_dist 12, W = 20, wheel + GNSS factors (reduced system and kept system in global memory)")
    ap.add_argument("--batch", type=int, default=None, help="independent sequences per GPU (default 256; 64 for --config 4)")
    ap.add_argument("--strong", type=int, default=0, help="strong scaling: this many sequences IN TOTAL, split over the ranks (BASELINE.json configs[3]: 64)")
    ap.add_argument("--distinct", type=int, default=None, help="number of distinct seeded windows (dealt round-robin to the sequences); default all (16 for --config 4)")
    ap.add_argument("--max-cnt", type=int, default=None)
    ap.add_argument("--min-dist", type=int, default=None)
    ap.add_argument("--ba-iters", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frontend", action="store_true")
    ap.add_argument("--no-backend", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the bounded end-to-end (drop-in path) sample")
    ap.add_argument("--e2e-seqs", type=int, default=256)
    ap.add_argument("--e2e-streams", type=int, default=8, help="distinct seeded recordings of the end-to-end sample (staggered starts: mixed keyframe / non-keyframe batches)")
    ap.add_argument("--e2e-groups", type=int, default=2, help="estimator groups the end-to-end sample splits its sequences over (own handle, stream and workers each; they alternate in which camera frames reach their back ends, so one group's host phase falls on the other's batch)")
    ap.add_argument("--e2e-same-frames", action="store_true", help="with several estimator groups: all groups take the same camera frames (default: they alternate, see end_to_end_sample)")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-image (PCIe-inclusive) sample")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the step times at 1 / 8 / 64 sequences")
    ap.add_argument("--no-large-batch", action="store_true", help="skip the step times at 512 / 1024 sequences per GPU")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the 20-step kernel-rate samples of configs[2] and configs[4]")
    ap.add_argument("--no-long-run", action="store_true", help="skip the extra 200-step run next to a short timed loop")
    ap.add_argument("--e2e-device-preint", action="store_true", help="also run the end-to-end sample with the batched device pre-integration of the estimator group")
    ap.add_argument("--e2e-device-sweeps", action="store_true", help="also run the end-to-end sample with the batched device feature sweeps (triangulateWithDepth, movingConsistencyCheckW) of the estimator group")
    ap.add_argument("--e2e-only", action="store_true", help="only the end-to-end (drop-in path) sample, as one JSON line (used by the default run for its small-host sample)")
    ap.add_argument("--host-threads", type=int, default=0, help="with --e2e-only: confine this process to the first N hardware threads (sched_setaffinity) and size the worker pools for them")
    ap.add_argument("--no-small-host", action="store_true", help="skip the end-to-end sample on 8 hardware threads")
    ap.add_argument("--small-host-all", action="store_true", help="the 8-thread end-to-end sample with one estimator group as well (default: two alternating groups only)")
    args = ap.parse_args()

    if args.e2e_only:
        if args.host_threads > 0:   # the host BASELINE.md plans for: everything this process runs (tracker bookkeeping pool, group workers, Python) shares N hardware threads
            os.sched_setaffinity(0, set(range(args.host_threads)))   # the library sizes its pools from the affinity mask (half of it for the group's workers / the tracker's pool)
            os.environ["GF_HOST_THREADS"] = str(max(1, args.host_threads // 2))
            os.environ["GF_GROUP_THREADS"] = str(max(1, args.host_threads // (2 * max(1, args.e2e_groups))))   # several groups of one process share that half
        else:
            os.environ.setdefault("GF_HOST_THREADS", str(max(1, min(16, usable_cpus() // 2))))
        import gfamd
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(0)
        gfamd._chk(gfamd.lib().gf_set_device(0))
        S = max(1, args.e2e_streams)
        kw = dict(n_streams=S, n_groups=args.e2e_groups, stagger=not args.e2e_same_frames, device_preint=True if args.e2e_device_preint else None, device_sweeps=args.e2e_device_sweeps)
        cold = end_to_end_sample(gfamd, args.e2e_seqs, dev, 150, 30, **kw)
        warm = end_to_end_sample(gfamd, args.e2e_seqs, dev, 150, 30, **kw)
        warm["passes_window_solves_per_s"] = [cold["window_solves_per_s"], warm["window_solves_per_s"]]
        warm["affinity_hardware_threads"] = len(os.sched_getaffinity(0))
        warm["tracker_host_threads"] = int(os.environ["GF_HOST_THREADS"])
        print(json.dumps(warm))
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # host bookkeeping threads of the tracker (up to 16 per process): keep all ranks of the node within its cores (the estimator group takes the same share)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    os.environ.setdefault("GF_HOST_THREADS", str(max(1, min(16, usable_cpus() // (2 * max(local_world, 1))))))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the HIP path has no CPU fallback)")
    # GF_BENCH_SINGLE_DEVICE=1 (tests only): every rank on device 0 with the gloo backend, to run the N > 1 path on a one-GPU box
    single = os.environ.get("GF_BENCH_SINGLE_DEVICE") == "1"
    device_index = 0 if single else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if single:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    import gfamd
    import shard
    gfamd._chk(gfamd.lib().gf_set_device(device_index))
    CFG = CONFIGS[args.config]
    args.max_cnt = args.max_cnt or CFG["max_cnt"]
    args.min_dist = args.min_dist or CFG["min_dist"]
    args.batch = args.batch or CFG["batch"]
    args.distinct = args.distinct if args.distinct is not None else CFG["distinct"]
    WIN, GNSS = CFG["window"], CFG["gnss"]
    if args.strong:
        if args.strong % world:
            raise SystemExit("--strong %d does not divide over %d ranks" % (args.strong, world))
        args.batch = args.strong // world
    B, K, Wm = args.batch, args.steps, args.warmup
    dt = 1.0 / 15.0
    # distinct frames per sequence; longer runs walk the clip forwards and backwards (continuous motion, tracks persist) instead of synthesising hundreds of frames
    n_frames = min(Wm + K + 1 + 4, 32)

    def frame_index(k):
        m = k % max(2 * n_frames - 2, 1)
        return m if m < n_frames else 2 * n_frames - 2 - m
    seq0 = shard.first_sequence(rank, B)  # global sequence ids [seq0, seq0 + B)
    frames, depth = make_frames(n_frames, B, 1000 + seq0, dev)
    torch.cuda.synchronize()
    frame_bytes = B * H * W

    trk = gfamd.FeatureTracker(gfamd.default_cfg(batch=B, max_cnt=args.max_cnt, min_dist=args.min_dist))
    trk.set_profiling(True)
    est, wins = make_windows(gfamd, B, 1000 + seq0, args.max_cnt, WIN, GNSS, args.distinct)
    est.upload(wins)
    step = [0]

    newest = torch.zeros((B, 7), dtype=torch.float64, device=dev)   # payload of the pose gather
    gathered = [None]
    pose_gather = shard.PoseGather(dist, world, [B] * world, dev)     # persistent send / receive buffers, all_gather_into_tensor over RCCL

    def do_step(exchange=True):
        k = frame_index(step[0])
        # the back end of this step is enqueued first and runs on its own stream while the tracker (GPU kernels + host bookkeeping)
        # proceeds -- the reference runs processImage and trackImage on separate threads as well (estimator.cpp:209, rosNodeTest.cpp:713)
        if not args.no_backend:
            est.solve_resident_async(args.ba_iters, 0, True)
        if not args.no_frontend:
            trk.trackImageBatchDevice([dt * step[0]] * B, frames.data_ptr() + k * frame_bytes, depth.data_ptr(), unpack=False)
        if not args.no_backend:
            est.wait()
            if exchange:   # north_star's only exchange: the newest pose of every sequence, all_gather over RCCL (56 B per sequence, latency-bound)
                est.export_newest_poses(newest.data_ptr(), B)
                gathered[0] = pose_gather(newest)
        step[0] += 1

    t_phase = [time.perf_counter()]

    def phase(name):   # where the wall clock of the whole run goes (stderr; the JSON line stays the only thing on stdout)
        now = time.perf_counter()
        if rank == 0:
            print("bench.py: %-28s %6.1f s" % (name, now - t_phase[0]), file=sys.stderr, flush=True)
        t_phase[0] = now
    phase("setup (frames, windows)")
    for _ in range(Wm + 1):  # frame 0 only detects; it is part of the warm-up
        do_step()
    trk.reset_stats()
    est.reset_stats()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        do_step()
    barrier()
    el = time.perf_counter() - t0
    phase("warm-up + timed loop")
    st = trk.stats()
    bs = est.stats()
    sums = est.download(wins) if not args.no_backend else [None]
    # the driver's command times 20 steps (57 ms): box-to-box and host jitter are larger than that sample resolves (round-5 review), so the same loop once more over 200
    # steps, reported next to `value` (N = 1 only: the driver's own multi-GPU lines stay what its command measures)
    long_run = None
    if world == 1 and K < 200 and not args.no_long_run and not (args.no_frontend or args.no_backend):
        s0 = est.stats()["solves"]
        barrier()
        t1 = time.perf_counter()
        for _ in range(200):
            do_step()
        barrier()
        el2 = time.perf_counter() - t1
        long_run = {"steps": 200, "value": (est.stats()["solves"] - s0) / el2, "ms_per_step": 1e3 * el2 / 200}
        phase("200-step run")

    # the roofline kernels once more on an otherwise idle GPU: their hipEvent times without the other half's kernels in between
    iso = {}
    if not args.no_frontend and not args.no_backend:
        trk.reset_stats(); est.reset_stats()
        for _ in range(4):
            trk.trackImageBatchDevice([dt * step[0]] * B, frames.data_ptr() + frame_index(step[0]) * frame_bytes, depth.data_ptr(), unpack=False)
            step[0] += 1
        torch.cuda.synchronize()
        for _ in range(3):
            est.solve_resident(args.ba_iters, 0, True)
        ti, bi = trk.stats(), est.stats()
        iso = {"lk_ms": ti["ms_lk"] / max(ti["lk_launches"], 1), "lk_alg_bytes": (484.0 * 5.0 * ti["lk_level_passes"] + 484.0 * ti["lk_iterations"]) / max(ti["lk_launches"], 1),
               "pyramid_ms": ti["ms_pyramid"] / 4, "detect_ms": ti["ms_detect"] / 4,
               "jtj_ms": bi["ms_jtj"] / max(bi["jtj_launches"], 1), "step_ms": bi["ms_step"] / max(bi["step_launches"], 1),
               "ba_solve_ms": bi["ms_solve"] / 3, "ba_marginalize_ms": bi["ms_marginalize"] / 3}
        if not GNSS and WIN <= 10:   # north_star's split formulation next to the fused kernel (fixed extrinsic, LDS-resident tiles): same solves, the visual sweep as two kernels
            est.set_split_jtj(True)
            est.solve_resident(args.ba_iters, 0, True)     # warm (allocation of the block-row buffer)
            est.reset_stats()
            for _ in range(3):
                est.solve_resident(args.ba_iters, 0, True)
            bsplit = est.stats()
            est.set_split_jtj(False)
            iso["split"] = {"sweep_plus_contraction_ms": bsplit["ms_jtj"] / max(bsplit["jtj_launches"], 1),
                            "contraction_ms": bsplit["ms_jtj_contract"] / max(bsplit["jtj_contract_launches"], 1), "ba_solve_ms": bsplit["ms_solve"] / 3}

    # the gather keeps the global sequence order: this rank's block of the gathered poses is what it exported
    own_block_ok = True
    if not args.no_backend:
        own_block_ok = bool(torch.equal(gathered[0][rank * B:(rank + 1) * B], newest))
        assert own_block_ok, "rank %d: its block of the gathered poses differs from what it exported" % rank
    tot = torch.tensor([el, float(st["tracked_features"]), float(st["output_features"]), float(bs["solves"])], dtype=torch.float64, device=dev)
    if dist is not None:
        mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        el_max, tracked, outf, solves = mx[0].item(), sm[1].item(), sm[2].item(), sm[3].item()
    else:
        el_max, tracked, outf, solves = el, st["tracked_features"], st["output_features"], bs["solves"]

    if rank == 0:
        if not args.no_backend:
            assert gathered[0].shape == (B * world, 7) and bool(torch.isfinite(gathered[0]).all())
        pmc = pmc_summary()
        # roofline of the dominant front-end kernel (lk_track_kernel): algorithmic bytes per SURVEY.md 8(d):
        #   484*(1+4) B per (point, level pass) [u8 window + s16x2 derivative window] + 484 B per iteration [moving window].
        # The formula is the survey's and is kept as the numerator; since round 2 the kernel does not read derivative windows any more (it evaluates the
        # Scharr derivative from four u8 rows of 16 B per lane): `bytes_loaded_by_design` below is what its loads request.
        launches = max(st["lk_launches"], 1)
        alg_bytes = 484.0 * 5.0 * st["lk_level_passes"] + 484.0 * st["lk_iterations"]
        lk_ms = st["ms_lk"] / launches
        achieved = alg_bytes / launches / (lk_ms * 1e-3) / 1e9 if lk_ms > 0 else 0.0
        jtj_ms = bs["ms_jtj"] / max(bs["jtj_launches"], 1)
        step_ms = bs["ms_step"] / max(bs["step_launches"], 1)
        step_flops = bs["step_flops"] / max(bs["step_launches"], 1)
        jtj_alg = bs["jtj_alg_flops"] / max(bs["jtj_launches"], 1)
        jtj_issued = bs["jtj_flops"] / max(bs["jtj_launches"], 1)
        jtj_t = iso.get("jtj_ms", jtj_ms)      # kernel time on an idle GPU when available (the in-region span includes waiting for CUs held by tracker kernels)
        step_t = iso.get("step_ms", step_ms)
        tf = lambda fl, ms: fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        lk_pmc = pmc.get("lk_track_kernel", {})
        traffic = lk_pmc.get("hbm_bytes_per_point")   # per point: the same kernel at any batch
        unit = "window-solves/s (each with its tracker frame)" if not (args.no_frontend or args.no_backend) else ("tracked-features/s" if args.no_backend else "window-solves/s")
        value = (tracked / el_max) if args.no_backend else (solves / el_max)
        res = {
            "metric": "sliding-window solves/sec + tracked-features/sec, 640x480x10-frame window",
            "value": value, "unit": unit, "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": 1e3 * el_max / K, "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "front end u8/s16 fixed point + f32; back end f64", "data": "synthetic",
            "config": {"workload": "configs[%d] (%d features / min_dist %d, %d-frame window, visual+IMU+wheel%s+prior factors, %d dogleg iterations + MARGIN_OLD marginalisation; "
                                   "LK 21x21 3 levels + flow-back, Shi-Tomasi top-up) run as %s; every step ends with the all_gather of the newest poses"
                                   % (args.config, args.max_cnt, args.min_dist, WIN, "+GNSS" if GNSS else "", args.ba_iters,
                                      ("%d independent 640x480 RGBD sequences in total, %d per GPU (strong scaling)" % (args.strong, B)) if args.strong
                                      else ("%d independent 640x480 RGBD sequences per GPU (weak scaling)" % B)),
                       "baseline_config_index": args.config, "sequences_per_gpu": B, "sequences_total": B * world, "window": WIN, "features": args.max_cnt, "min_dist": args.min_dist,
                       "gnss": GNSS, "distinct_windows": args.distinct or B, "ba_iterations": args.ba_iters,
                       "reduced_system_in": "global memory (ba_step<true>)" if WIN > 10 or GNSS else "LDS (ba_step<false>)"},
            "pose_gather": {"rows": int(gathered[0].shape[0]) if gathered[0] is not None else 0, "bytes_per_step": 56 * B * world, "own_block_matches_export": own_block_ok,
                            "backend": "none (1 rank)" if dist is None else ("gloo (GF_BENCH_SINGLE_DEVICE test mode)" if single else "nccl (RCCL)")},
            "host_threads_per_rank": int(os.environ["GF_HOST_THREADS"]),
            "kernel_rate": value,
            "value_200_steps": long_run["value"] if long_run else (value if K >= 200 else None), "ms_per_step_200_steps": long_run["ms_per_step"] if long_run else (1e3 * el_max / K if K >= 200 else None),
            "value_is": "kernel_rate: tracker frame + solve + marginalisation per step with the windows resident on the device (inputs in HBM, the contract of `value`); "
                        "the rate through the reference's own call surface (trackImage -> inputFeature -> processImage, windows packed / uploaded / downloaded every frame) is "
                        "end_to_end.window_solves_per_s, reported at 1 GPU",
            "solves_per_s": solves / el_max, "tracked_features_per_s": tracked / el_max, "frames_per_s": B * world * K / el_max,
            "output_features_per_s": outf / el_max,
            "gpu_ms_per_step": {"pyramid": st["ms_pyramid"] / K, "lk": st["ms_lk"] / K, "detect": st["ms_detect"] / K, "tracker_total": st["ms_total_gpu"] / K,
                                "ba_solve": bs["ms_solve"] / K, "ba_marginalize": bs["ms_marginalize"] / K},
            "gpu_ms_isolated": iso,
            "host_ms_per_step": {k: st[k] / K for k in ("ms_host_pre", "ms_wait_lk", "ms_host_mid", "ms_wait_detect", "ms_host_post")},
            "roofline": {"kernel": "lk_track_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": (traffic * st["lk_points"] / launches) if traffic else None, "launch_ms": lk_ms,
                         "launch_ms_isolated": iso.get("lk_ms"),
                         "frac_isolated": (iso["lk_alg_bytes"] / (iso["lk_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if iso.get("lk_ms") else None,
                         "traffic_note": lk_pmc.get("note", "no PMC summary under profiles/"),
                         "algorithmic_bytes_per_launch": alg_bytes / launches,
                         "bytes_loaded_by_design_per_launch": (63.0 * 64.0 * st["lk_level_passes"] + 1024.0 * st["lk_tile_refills"]) / launches if "lk_tile_refills" in st else (63.0 * 64.0 * st["lk_level_passes"]) / launches,
                         "bytes_loaded_by_design_note": "template phase: 63 lanes x 4 rows x 16 B per (point, level pass); moving image: 32 x 32 B tile refills on top (not counted by the kernel)",
                         "points_per_launch": st["lk_points"] / launches, "iterations_per_launch": st["lk_iterations"] / launches},
            "roofline_jtj": {"kernel": "ba_linearize_visual_win", "bound": "mfma", "achieved": tf(jtj_alg, jtj_t), "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                             "frac": tf(jtj_alg, jtj_t) / FP64_MFMA_PEAK_TF, "launch_ms": jtj_t, "launch_ms_in_timed_region": jtj_ms,
                             "algorithmic_flops_per_launch": jtj_alg, "issued_mfma_flops_per_launch": jtj_issued, "frac_issued": tf(jtj_issued, jtj_t) / FP64_MFMA_PEAK_TF,
                             "mfma_utilisation_pmc": pmc.get("ba_linearize_visual_win", {}).get("mfma_utilisation") if args.config == 1 and B == 256 else None,   # the PMC passes ran configs[1] at 256 windows
                             "note": "algorithmic flops = Nv*2*2*91 per window (SURVEY.md 8d); issued = 2048 per v_mfma_f64_16x16x4_f64 (16-column tiles, 13 used); the kernel also "
                                     "evaluates every factor's residual / Jacobian (FP64 VALU), reduces the pair tiles and builds the E^T F rows; " + pmc.get("ba_linearize_visual_win", {}).get("note", "")},
            "roofline_jtj_split": None if "split" not in iso else {
                "kernel": "ba_linearize_visual_win<MODE 2> (contraction only) behind <MODE 1> (sweep -> block rows in HBM)", "bound": "mfma",
                "achieved": tf(jtj_alg, iso["split"]["contraction_ms"]), "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf(jtj_alg, iso["split"]["contraction_ms"]) / FP64_MFMA_PEAK_TF,
                "frac_issued": tf(jtj_issued, iso["split"]["contraction_ms"]) / FP64_MFMA_PEAK_TF, "launch_ms": iso["split"]["contraction_ms"],
                "sweep_plus_contraction_ms": iso["split"]["sweep_plus_contraction_ms"], "fused_ms": iso.get("jtj_ms"),
                "ba_solve_ms": {"split": iso["split"]["ba_solve_ms"], "fused": iso.get("ba_solve_ms")},
                "block_row_bytes_per_launch": 2 * 256.0 * jtj_alg / (4 * 91) if jtj_alg else None,
                "mfma_utilisation_pmc": pmc.get("ba_linearize_visual_win_contract", {}).get("mfma_utilisation") if args.config == 1 and B == 256 else None,
                "note": "north_star's formulation measured: the sweep writes 256 B of block rows per factor to HBM and a second kernel only contracts them (then reduces the tiles and builds "
                        "Vc / E^T F as the fused kernel does); same bits as the fused kernel (tests/test_backend_gpu.py::test_split_jtj_formulation_gives_the_same_bits); off by default "
                        "because the iteration gets slower, not faster"},
            "roofline_step": {"kernel": "ba_step", "bound": "mfma", "achieved": tf(step_flops, step_t), "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf(step_flops, step_t) / FP64_MFMA_PEAK_TF,
                              "launch_ms": step_t, "launch_ms_in_timed_region": step_ms, "flops_per_launch": step_flops,
                              "mfma_utilisation_pmc": pmc.get("ba_step", {}).get("mfma_utilisation") if args.config == 1 and B == 256 else None,
                              "note": "largest kernel by time; algorithmic flops = Schur SYRK NE*n_c^2 + Cholesky R^3/3 + substitutions 2R^2 per window; one block per window, "
                                      "bound by the latency of R/16 sequential 16x16 factor+inverse blocks, not by MFMA issue"},
            "ba_summary_seq0": sums[0],
            "kernel_source_sha16": kernel_source_sha16(), "pmc_summary": pmc.get("_stale", "profiles/pmc_summary.json matches this tree's kernel sources"),
        }
        phase("isolated passes")
        if world == 1 and not args.no_pcie and not args.strong and not (args.no_frontend or args.no_backend):
            res["pcie_inclusive"] = pcie_sample(gfamd, trk, est, args, B, frames, depth, dt, step, frame_index)
            phase("pcie_inclusive")
        if world == 1 and not args.no_small_batch and not args.strong and not (args.no_frontend or args.no_backend):
            res["small_batch"] = small_batch_sample(gfamd, dev, args, WIN, GNSS)
            phase("small_batch")
        if world == 1 and not args.no_large_batch and not args.strong and args.config == 1 and B == 256 and not (args.no_frontend or args.no_backend):
            # round-5 review: "256 sequences per GPU is the builder's number" -- the same step with two and four windows per CU resident
            lb = small_batch_sample(gfamd, dev, args, WIN, GNSS, sizes=(512, 1024), K=20, distinct=64)
            for k_, v_ in lb.items():
                v_["sequences_per_gpu"] = int(k_); v_["vs_256_sequences"] = v_["window_solves_per_s"] / value
            res["large_batch"] = lb
            # the same with ba_step in its chain form (round 6: speed-bias blocks eliminated first, dense part only in LDS, 256 threads: two windows per CU; the switch is
            # read when a handle is created, results agree with the dense form to 1e-9: tests/test_backend_gpu.py::test_chain_form_...)
            had = os.environ.get("GF_BA_CHAIN")
            os.environ["GF_BA_CHAIN"] = "1"
            try:
                lc = small_batch_sample(gfamd, dev, args, WIN, GNSS, sizes=(256, 512, 1024), K=20, distinct=64)
                for k_, v_ in lc.items():
                    v_["sequences_per_gpu"] = int(k_); v_["vs_256_sequences_dense_form"] = v_["window_solves_per_s"] / value
                res["large_batch_chain_form"] = lc
            except Exception as ex:
                res["large_batch_chain_form"] = {"error": repr(ex)[:300]}
            finally:
                if had is None:
                    os.environ.pop("GF_BA_CHAIN", None)
                else:
                    os.environ["GF_BA_CHAIN"] = had
            phase("large_batch")
        if world == 1 and not args.no_other_configs and not args.strong and args.config == 1 and not (args.no_frontend or args.no_backend):
            res["other_configs"] = {}
            for ci in (2, 4):
                try:
                    res["other_configs"]["configs[%d]" % ci] = config_sample(gfamd, dev, ci, ba_iters=args.ba_iters)
                except Exception as ex:   # a side measurement: its failure must not cost the line
                    res["other_configs"]["configs[%d]" % ci] = {"error": repr(ex)[:300]}
            phase("other_configs")
        if world == 1 and not args.no_e2e and not args.strong and args.config == 1 and not (args.no_frontend or args.no_backend):
            trk.close(); trk = None
            S = max(1, args.e2e_streams)
            cold = end_to_end_sample(gfamd, args.e2e_seqs, dev, args.max_cnt, args.min_dist, n_streams=S, n_groups=args.e2e_groups, stagger=not args.e2e_same_frames)
            # warm passes (host allocations and worker threads up, as in a running service): the path alternates host and device phases and a pass moves by +-10 % with
            # whatever else the host runs, so two are taken, the better one is the sample, and all three are listed
            warm = [end_to_end_sample(gfamd, args.e2e_seqs, dev, args.max_cnt, args.min_dist, n_streams=S, n_groups=args.e2e_groups, stagger=not args.e2e_same_frames) for _ in range(2)]
            res["end_to_end"] = max(warm, key=lambda r: r["window_solves_per_s"])
            res["end_to_end"]["passes_window_solves_per_s"] = [cold["window_solves_per_s"]] + [r["window_solves_per_s"] for r in warm]
            if args.e2e_groups > 1:   # next to it: all sequences in ONE group, every batch 256 windows (the arrangement of rounds 2-3 and of the first half of round 4)
                one = end_to_end_sample(gfamd, args.e2e_seqs, dev, args.max_cnt, args.min_dist, n_streams=S, n_groups=1)
                res["end_to_end_one_group"] = {k: one[k] for k in ("window_solves_per_s", "sequences", "distinct_recordings", "estimator_groups", "ms_per_backend_frame", "group_batches", "largest_batch",
                                                                   "main_thread_ms_per_backend_frame")}
            if S > 1:   # the best case for the batching next to it: ONE recording replicated, every member takes the same keyframe decision, every rendezvous is one homogeneous batch
                homo = end_to_end_sample(gfamd, args.e2e_seqs, dev, args.max_cnt, args.min_dist, n_streams=1, n_groups=args.e2e_groups, stagger=not args.e2e_same_frames)
                res["end_to_end_homogeneous"] = {k: homo[k] for k in ("window_solves_per_s", "sequences", "distinct_recordings", "estimator_groups", "ms_per_backend_frame", "group_batches", "largest_batch",
                                                                      "backend_frames_with_mixed_decisions", "group_steps_live", "groups_alternate_frames", "keyframe_vote_share")}
            if args.e2e_device_preint:   # the steps' IMU intervals as one device launch instead of on the members' threads (same bits; slower on a many-core host)
                alt = end_to_end_sample(gfamd, args.e2e_seqs, dev, args.max_cnt, args.min_dist, device_preint=True)
                res["end_to_end"]["with_device_preintegration_window_solves_per_s"] = alt["window_solves_per_s"]
                res["end_to_end"]["with_device_preintegration_newest_position_norm_m"] = alt["newest_position_norm_m"]
            if args.e2e_device_sweeps:   # the members' per-feature loops as one device launch each per step (same bits; two more rendezvous per frame)
                alt = end_to_end_sample(gfamd, args.e2e_seqs, dev, args.max_cnt, args.min_dist, device_sweeps=True)
                res["end_to_end"]["with_device_feature_sweeps_window_solves_per_s"] = alt["window_solves_per_s"]
                res["end_to_end"]["with_device_feature_sweeps_newest_position_norm_m"] = alt["newest_position_norm_m"]
            res["end_to_end"]["first_pass_window_solves_per_s"] = cold["window_solves_per_s"]
            if not args.no_small_host:
                # The same sample on the host BASELINE.md plans for: this thread and every thread the library creates from here on (the tracker's bookkeeping pool, the
                # groups' workers: both pools are sized from the affinity mask) confined to 8 hardware threads.  In this process -- the rendered recordings are reused --;
                # the HIP runtime's own helper threads, created earlier, keep their mask (`python bench.py --e2e-only --host-threads 8` confines a whole process).
                all_cpus = os.sched_getaffinity(0)
                saved = {k_: os.environ.get(k_) for k_ in ("GF_HOST_THREADS", "GF_GROUP_THREADS")}
                for ng in ((1, 2) if args.small_host_all else (2,)):
                    key_ = "end_to_end_8_host_threads" + ("" if ng == 1 else "_two_groups")
                    try:
                        os.sched_setaffinity(0, set(sorted(all_cpus)[:8]))
                        os.environ["GF_HOST_THREADS"] = "4"
                        os.environ["GF_GROUP_THREADS"] = str(max(1, 4 // ng))
                        p8 = [end_to_end_sample(gfamd, args.e2e_seqs, dev, args.max_cnt, args.min_dist, n_streams=S, n_groups=ng) for _ in range(2)]
                        r8 = p8[1]
                        res[key_] = {k_: r8[k_] for k_ in ("window_solves_per_s", "sequences", "distinct_recordings", "estimator_groups", "ms_per_backend_frame", "group_worker_threads",
                                                            "main_thread_ms_per_backend_frame", "tracker_ms_per_call", "device_preint")}
                        res[key_].update({"passes_window_solves_per_s": [r_["window_solves_per_s"] for r_ in p8], "affinity_hardware_threads": len(os.sched_getaffinity(0)), "tracker_host_threads": 4})
                    except Exception as ex:   # the sample is a side measurement: its failure must not cost the line
                        res[key_] = {"error": repr(ex)[:300]}
                    finally:
                        os.sched_setaffinity(0, all_cpus)
                        for k_, v_ in saved.items():
                            if v_ is None:
                                os.environ.pop(k_, None)
                            else:
                                os.environ[k_] = v_
        phase("end_to_end samples")
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is an N = 1 measurement (the other ranks would sit in the closing barrier meanwhile)
            nseq = min(8, B)
            cores = min(os.cpu_count() or 1, nseq)
            fh = frames[: min(n_frames, 13), :nseq].cpu().numpy()
            if args.config != 1:   # heavier windows: bound the sample
                fh = fh[:5]
            cb = cpu_baseline(fh, dt, wins[:nseq], cores, args.ba_iters, args.max_cnt, args.min_dist, repeat=4 if args.config == 1 else 1)
            res["cpu_baseline"] = {"value": cb["steps_per_s"] if not args.no_backend else cb["tracked_features_per_s"],
                                   "unit": unit, "cores": cores, "kind": "port",
                                   "sample": "%d sequences x %d x %d frames (tracker) and %d solve+marginalise per sequence through the CPU oracle (%s), %d threads, one sequence per thread "
                                             "(variant a x %d cores); then sequence 0 once more as variant b; %.1f s wall, ~%.0f s of CPU work; box shows %d hardware threads, %d usable under its CPU quota"
                                             % (nseq, cb["repeat"], fh.shape[0], cb["repeat"] * max(1, fh.shape[0] // 4), cb["build"], cores, cores, cb["wall_s"], cb["wall_s"] * cores, os.cpu_count() or 1, usable_cpus()),
                                   "detail": cb}
        phase("cpu_baseline")
        print(json.dumps(res))
    if trk is not None:
        trk.close()
    est.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
