"""Results of a few back-end scenarios as digests, one line each: `name sha1`.  scripts/stale_bisect.py and tests/test_stale_memory_gpu.py run this in sub-processes under
different GF_BA_POISON settings (device buffers that start as zeros / as plausible stale data) and behind different predecessors in the same handle: a result that
depends on anything but the window handed in shows as a digest that moves.  python scripts/stale_probe.py [scenario ...]  (GPU)

Scenarios
  plain, gnss             fresh handle: solve + MARGIN_OLD + next-window solve (+ MARGIN_SECOND_NEW) of one seeded window
  plain_after, gnss_after the same window and calls in a handle that solved and marginalised OTHER windows first (different sizes, factor families, priors):
                          must give the digest of plain / gnss
  slots, slots_fresh      gf_ba_pack_slot / gf_ba_solve_packed with slots that sit batches out, behind other windows in the same slots / in a fresh handle:
                          must give the same digest
  group, group_gnss       gf_estimator_group_* on staggered streams (members initialise at different times: batches with inactive and never-packed slots)"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ground-fusion_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
import gfamd  # noqa: E402
import gfwindow as gw  # noqa: E402
import oracle_py as O  # noqa: E402  (only the pre-integration helpers that synth_window needs to BUILD a window; nothing is compared with the oracle here)
import synth_window as SW  # noqa: E402


class Digest:
    def __init__(self):
        self.h = hashlib.sha1()

    def add(self, *xs):
        for x in xs:
            if isinstance(x, dict):
                for k in sorted(x):
                    self.add(k, x[k])
            elif isinstance(x, (list, tuple)):
                for y in x:
                    self.add(y)
            elif isinstance(x, np.ndarray):
                self.h.update(np.ascontiguousarray(x).tobytes())
            elif isinstance(x, float):
                self.h.update(np.float64(x).tobytes())
            else:
                self.h.update(repr(x).encode())
        return self

    def hex(self):
        return self.h.hexdigest()


def gnss_est(batch=1, W=10, F=150):
    return gfamd.Estimator(W, F, F * W, batch, max_gnss=12 * (W + 1))


def chain(est, seed, kw, d):
    """solve + MARGIN_OLD of the seeded window, then the next window with that prior: solve (+ MARGIN_SECOND_NEW without GNSS)"""
    w = SW.make_window(seed, O, **kw)
    s = est.solve([w], 8)[0]
    p0 = est.marginalize([w], 0)[0]
    d.add({k: w[k] for k in gw.STATE_KEYS}, s, p0["J"], p0["r"], list(p0["block_id"]), p0["x0"])
    w2 = SW.make_window(seed, O, frame0=1, prior=p0, **kw)
    s2 = est.solve([w2], 8)[0]
    d.add({k: w2[k] for k in gw.STATE_KEYS}, s2)
    if not kw.get("gnss"):
        p1 = est.marginalize([w2], 1)[0]
        d.add(p1["J"], p1["r"])


def predecessors(est, gnss):
    """other windows through the same handle first: fewer features, no wheel factors, a free camera extrinsic, a prior chain of their own"""
    sink = Digest()
    base = dict(gnss=True) if gnss else {}
    for seed, kw in ((31, dict(max_features=60, n_landmarks=100)), (32, dict(use_wheel=False)), (33, dict(fix_ex_pose=0)), (34, dict())):
        kw = dict(kw, **base)
        if gnss and seed == 33:
            kw = dict(base, gnss_lowspeed=1)
        chain(est, seed, kw, sink)


def sc_plain(after):
    est = gfamd.Estimator()
    if after:
        predecessors(est, False)
    d = Digest()
    chain(est, 12, {}, d)
    est.close()
    return d.hex()


def sc_gnss(after):
    est = gnss_est()
    if after:
        predecessors(est, True)
    d = Digest()
    chain(est, 12, dict(gnss=True), d)
    est.close()
    return d.hex()


def sc_slots(behind=True):
    """the estimator group's route into the solver, driven by hand: slots packed one by one, batches in which some slots sit out, marginalisation of listed slots;
    the windows of the second round must not see what the first round left in their slots"""
    import ctypes as C
    L = gfamd.lib()
    est = gfamd.Estimator(batch=4)
    d = Digest()

    def run(round_seed, record):
        wins = {b: SW.make_window(round_seed + b, O, **({} if b != 2 else dict(use_wheel=False))) for b in (0, 1, 2)}   # slot 3 is never packed
        for act in ((0, 2), (1,), (0, 1, 2)):
            cw = {b: wins[b].to_c() for b in act}
            for b in act:
                gfamd._chk(L.gf_ba_pack_slot(est.h, b, C.byref(cw[b])))
            arr = (C.c_int * len(act))(*act)
            gfamd._chk(L.gf_ba_solve_packed(est.h, arr, len(act), 8))
            for b in act:
                summ = gw.SummaryC()
                gfamd._chk(L.gf_ba_unpack_slot(est.h, b, C.byref(cw[b]), C.byref(summ)))
                if record:
                    d.add(b, {k: wins[b][k] for k in gw.STATE_KEYS}, {k: getattr(summ, k) for k, _ in gw.SummaryC._fields_})
    if behind:
        run(40, False)
    run(12, True)
    est.close()
    return d.hex()


def sc_group(gnss):
    import synth_stream as SS
    n = 3
    streams, G = [], []
    for s in range(n):
        # staggered: member s starts to move 0.6 s later than member s - 1, so that solves begin at different frames (inactive / never-packed slots in the batches)
        st = SS.Stream(3 + s, t_still=1.5 + 0.6 * s, t_move=3.0 - 0.6 * s, v_max=0.4, yaw0=0.0, yaw_turn=-0.5 + 0.3 * s, split_x=1.8, turn_delay=0.8)
        st._lm = st._landmarks(1000)
        st._pn = np.random.default_rng(4200 + s).normal(0, 1.0, (len(st.cam_t), len(st._lm), 2))
        if gnss:
            G.append(st.gnss_setup(alpha=0.3 + 0.4 * s))
        streams.append(st)
    kw = dict(tio=SS.TIO, rio=SS.RIO, multiple_thread=1)
    if gnss:
        kw.update(gnss_enable=1, gnss_track_num_thres=3, gnss_local_time_diff=G[0]["time_diff"])
    grp = gfamd.EstimatorGroup(gfamd.default_estimator_cfg(**kw), n)
    tp, orng = [-1.0] * n, [np.random.default_rng(50 + s) for s in range(n)]
    nk = min(len(st.cam_t) for st in streams)
    d = Digest()
    for k in range(nk):
        for s, st in enumerate(streams):
            tp[s] = st.feed(grp.members[s], k, tp[s])
        if k % 2:
            continue
        frames = [st.feature_frame(k) for st in streams]
        if gnss:
            for s, st in enumerate(streams):
                tg, epoch = st.gnss_epoch(float(st.cam_t[k]) + orng[s].uniform(-0.02, 0.02))
                grp.members[s].inputGNSS(tg, epoch)
        grp.inputFeatures(list(range(n)), [float(st.cam_t[k]) for st in streams], frames)
        for s in range(n):
            a = grp.members[s].state()
            d.add(a["frame_count"], a["solver_flag"], a["iterations"], a["Ps"], a["Vs"], a["Bas"], a["Bgs"])
            if gnss:
                ga = grp.members[s].gnss_state()
                d.add(ga["rcv_dt"], ga["anc_ecef"])
    grp.close()
    return d.hex()


SCENARIOS = {
    "plain": lambda: sc_plain(False), "plain_after": lambda: sc_plain(True),
    "gnss": lambda: sc_gnss(False), "gnss_after": lambda: sc_gnss(True),
    "slots": lambda: sc_slots(True), "slots_fresh": lambda: sc_slots(False),
    "group": lambda: sc_group(False), "group_gnss": lambda: sc_group(True),
}

if __name__ == "__main__":
    names = sys.argv[1:] or list(SCENARIOS)
    for nm in names:
        print(nm, SCENARIOS[nm](), flush=True)
