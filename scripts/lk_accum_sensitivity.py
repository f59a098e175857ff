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

if __name__ == "__main__":
    for kw in (dict(max_cnt=150, min_dist=30), dict(max_cnt=300, min_dist=20), dict(max_cnt=500, min_dist=12)):
        print(compare(kw), flush=True)
