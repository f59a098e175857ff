"""bench.py's end-to-end sample on the CPU, against stand-ins for the library: the clock of the live span must be the wall clock.

Round 5's record carried `end_to_end` figures from which 0.56 s of a 0.65 s span had been subtracted: a feed loop inside the live span (re-enabled by `fed[q] = k`) whose time was
taken out of the clock while the other estimator group's workers kept running.  This test runs `end_to_end_sample` itself -- the tracker, the estimator groups and the rendered
recordings replaced by stubs that sleep -- and fails if (a) the result carries an excluded time, (b) rate x span != solves, (c) any IMU / wheel sample is pushed after the first
camera frame, or (d) the reported span is shorter than the time the stubs provably slept inside it."""
import time
import types

import numpy as np
import pytest
import torch

import bench


class _Stream:
    def __init__(self, log, n_frames):
        self.cam_t = np.arange(n_frames) / 15.0
        self.log = log

    def feed(self, est, k, t_prev):
        self.log.append(("feed", time.perf_counter()))
        return float(self.cam_t[k]) + 0.03


class _Member:
    def __init__(self, grp):
        self.grp = grp

    def flags(self):
        return (10, 1 if self.grp.submitted >= 2 else 0, 0)     # NON_LINEAR from the third back-end frame on

    def state(self):
        return {"Ps": np.ones((11, 3))}


class _Group:
    SLEEP = 0.004

    def __init__(self, cfg, n, device_preint=None, device_sweeps=False):
        self.members = [_Member(self) for _ in range(n)]
        self.submitted = 0
        self.slept = []

    def wait(self):
        time.sleep(self.SLEEP)
        self.slept.append((time.perf_counter(), self.SLEEP))

    def submitFeatures(self, seqs, ts, buf, no, stride=None):
        self.submitted += 1

    def stats(self):
        return {"batches": self.submitted, "largest_batch": len(self.members)}

    def close(self):
        pass


class _Tracker:
    cap = 8

    def __init__(self, cfg, log):
        self.log = log

    def trackImageBatchDevice(self, ts, gray, depth, unpack=False, out=None, n_out=None):
        self.log.append(("track", time.perf_counter()))

    def stats(self):
        return {}

    def close(self):
        pass


class _Tensor:
    def data_ptr(self):
        return 0

    def index_select(self, dim, idx):
        return self


@pytest.mark.parametrize("n_groups,n_streams", [(1, 1), (2, 8)])
def test_end_to_end_sample_subtracts_nothing_from_its_clock(monkeypatch, n_groups, n_streams):
    log, groups = [], []
    n_frames = 24
    streams = [_Stream(log, n_frames) for _ in range(n_streams)]

    def make_group(*a, **kw):
        groups.append(_Group(*a, **kw))
        return groups[-1]
    fake = types.SimpleNamespace(default_estimator_cfg=lambda **kw: kw, default_cfg=lambda **kw: kw, EstimatorGroup=make_group, FeatureTracker=lambda cfg: _Tracker(cfg, log),
                                 OBS_DTYPE=np.dtype([("id", np.int32), ("v", np.float64, 8)]))
    monkeypatch.setattr(bench, "e2e_inputs", lambda n, dev: (streams, [_Tensor()] * n_frames, [_Tensor()] * n_frames))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda *a, **k: None)
    bench._E2E_STAGED.clear()
    t_in = time.perf_counter()
    r = bench.end_to_end_sample(fake, 16, "cpu", 150, 30, n_streams=n_streams, n_groups=n_groups)
    t_out = time.perf_counter()
    bench._E2E_STAGED.clear()

    assert "t_feed_excluded_s" not in r and not any("excluded" in k for k in r), "an excluded time is back in the record"
    assert r["window_solves"] > 0 and r["wall_s"] > 0
    assert r["window_solves_per_s"] == pytest.approx(r["window_solves"] / r["wall_s"], rel=1e-12)
    first_track = min(t for what, t in log if what == "track")
    late = [t for what, t in log if what == "feed" and t > first_track]
    assert not late, "%d feed() calls after the first camera frame: the live span marshals inputs again" % len(late)
    assert sum(1 for what, _ in log if what == "feed") == 16 * n_frames          # every sample queued once, before the clock
    # the span cannot be shorter than what the stubs slept inside it: every wait() of a live back-end frame sleeps SLEEP on the main thread
    live_waits = r["group_steps_live"] * _Group.SLEEP
    assert r["wall_s"] >= 0.98 * live_waits, "reported span %.4f s is shorter than the %.4f s the estimators were waited for inside it" % (r["wall_s"], live_waits)
    assert r["wall_s"] <= t_out - t_in
    # and the anatomy of the main thread adds up to no more than the span
    bf = r["live_camera_frames"] / 2.0
    assert sum(r["main_thread_ms_per_backend_frame"].values()) * bf / 1e3 <= r["wall_s"] * 1.001
