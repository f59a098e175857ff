"""What the documented int64 accumulation of the LK sums can change against the float accumulation of an x86 OpenCV build (oracle mode 1):
the three tracker configurations of the parity tests, 12 frames each; counts of differing status flags, feature ids and coordinates."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O, synth

def run(cfgkw, seed, nframes, mode):
    O.set_lk_accum(mode)
    tr = O.Tracker(O.default_cfg(**cfgkw))
    frames = synth.tracker_sequence(seed, nframes)
    depth = np.full(frames[0].shape, 1800, np.uint16)
    out = []
    for k, f in enumerate(frames):
        ids, obs = tr.track(0.0666 * k, f, depth)
        out.append((ids.copy(), obs.copy()))
    O.set_lk_accum(0)
    return out

def compare(cfgkw, seed=1000, nframes=12):
    a, b = run(cfgkw, seed, nframes, 0), run(cfgkw, seed, nframes, 1)
    tot = same_ids = 0
    max_d, n_moved, frames_diff = 0.0, 0, 0
    for (ia, oa), (ib, ob) in zip(a, b):
        tot += len(ia)
        common = np.intersect1d(ia, ib)
        same_ids += len(common)
        if not np.array_equal(ia, ib):
            frames_diff += 1
        pa = {int(i): o[3:5] for i, o in zip(ia, oa)}; pb = {int(i): o[3:5] for i, o in zip(ib, ob)}
        for i in common:
            d = float(np.abs(pa[int(i)] - pb[int(i)]).max())
            if d > 0:
                n_moved += 1; max_d = max(max_d, d)
    return dict(cfg=cfgkw, observations=tot, ids_in_both=same_ids, frames_with_different_id_lists=frames_diff, coordinates_changed=n_moved, largest_pixel_change=max_d)

def survey(cfgkw, n_seq, nframes, seed0=1000):
    """Round-5 review, item 8: the margin the unpinned OpenCV accumulation order leaves, as a number.  `n_seq` seeded sequences of `nframes` frames through both modes;
    counts over every (frame, feature id) that either mode reports: ids reported by one mode only (a status flag or a later corner differs), and -- for ids in both --
    pixel coordinates whose cvRound differs (what setMask / inBorder / the depth sample see) and whose float value differs at all."""
    tot = dict(config=cfgkw, sequences=n_seq, frames=n_seq * nframes, observations=0, ids_in_one_mode_only=0, rounded_pixel_differs=0, float_coordinate_differs=0,
               largest_pixel_change=0.0, frames_with_different_id_lists=0, sequences_with_any_id_difference=0)
    for q in range(n_seq):
        a, b = run(cfgkw, seed0 + q, nframes, 0), run(cfgkw, seed0 + q, nframes, 1)
        any_diff = False
        for (ia, oa), (ib, ob) in zip(a, b):
            sa, sb = set(ia.tolist()), set(ib.tolist())
            tot["observations"] += len(sa | sb)
            tot["ids_in_one_mode_only"] += len(sa ^ sb)
            if not np.array_equal(ia, ib):
                tot["frames_with_different_id_lists"] += 1; any_diff = True
            pa = {int(i): o[3:5] for i, o in zip(ia, oa)}; pb = {int(i): o[3:5] for i, o in zip(ib, ob)}
            for i in sa & sb:
                d = np.abs(pa[i] - pb[i])
                if d.max() > 0:
                    tot["float_coordinate_differs"] += 1; tot["largest_pixel_change"] = max(tot["largest_pixel_change"], float(d.max()))
                    if not np.array_equal(np.rint(pa[i]), np.rint(pb[i])):   # cvRound = round half to even
                        tot["rounded_pixel_differs"] += 1
        tot["sequences_with_any_id_difference"] += int(any_diff)
    n = max(tot["observations"], 1)
    tot["fraction_ids_in_one_mode_only"] = tot["ids_in_one_mode_only"] / n
    tot["fraction_rounded_pixel_differs"] = tot["rounded_pixel_differs"] / n
    return tot


if __name__ == "__main__":
    import argparse, json
    ap = argparse.ArgumentParser()
    ap.add_argument("--survey", type=int, default=0, help="number of seeded sequences per configuration (x --frames frames each); 0: the 12-frame comparison of round 2")
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    cfgs = (dict(max_cnt=150, min_dist=30), dict(max_cnt=300, min_dist=20), dict(max_cnt=500, min_dist=12))
    if not args.survey:
        for kw in cfgs:
            print(compare(kw), flush=True)
    else:
        O.set_threads(args.threads)
        res = []
        for kw in cfgs:
            res.append(survey(kw, args.survey, args.frames)); print(json.dumps(res[-1]), flush=True)
        if args.out:
            json.dump({"what": "int64 (parity mode) against float-lane accumulation (oracle mode 1: OpenCV 4.2 x86 order as restated in oracle/tracker_oracle.cpp) of the LK sums; "
                               "every (frame, feature id) either mode reports", "results": res}, open(args.out, "w"), indent=1)
