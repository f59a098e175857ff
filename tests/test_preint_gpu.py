"""Batched IMU pre-integration on the device (SURVEY.md §8(f)4, row B2 on the GPU): gf_imu_preintegrate_batch against the host loop gf_imu_preintegrate
(IntegrationBase::push_back, factor/integration_base.h:39-167) -- bit for bit -- and, through it, against the C oracle."""
import os
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gfamd  # noqa: E402
import oracle_py as O  # noqa: E402

pytestmark = pytest.mark.gpu
NOISE = np.array([1.2374091609523514e-02, 3.0032654435730201e-03, 1.9218003442176448e-04, 5.4692100664858005e-05])


def intervals(seed, n, max_len=24):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len + 1, n)
    lens[:3] = [0, 1, max_len][:min(n, 3)]          # an empty interval (two frames on one IMU stamp), a single sample, the longest
    first = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    m = int(first[-1])
    dt = rng.uniform(0.003, 0.007, m)
    acc = rng.normal(0, 1.5, (m, 3)) + np.array([0, 0, 9.8])
    gyr = rng.normal(0, 0.4, (m, 3))
    acc0, gyr0 = rng.normal(0, 1.5, (n, 3)) + np.array([0, 0, 9.8]), rng.normal(0, 0.4, (n, 3))
    ba, bg = rng.normal(0, 0.05, (n, 3)), rng.normal(0, 0.01, (n, 3))
    return first, dt, acc, gyr, acc0, gyr0, ba, bg


def test_device_intervals_are_bit_identical_to_the_host_loop():
    first, dt, acc, gyr, acc0, gyr0, ba, bg = intervals(1, 300)
    pb = gfamd.PreintBatch()
    out = pb.run(first, dt, acc, gyr, acc0, gyr0, ba, bg, NOISE)
    for i in range(len(first) - 1):
        a, b = first[i], first[i + 1]
        h = gfamd.imu_preintegrate(dt[a:b], acc[a:b], gyr[a:b], acc0[i], gyr0[i], ba[i], bg[i], NOISE)
        for k in ("delta_p", "delta_q", "delta_v", "jacobian", "covariance"):
            assert np.array_equal(out[k][i], h[k]), (i, k, np.abs(out[k][i] - h[k]).max())
        assert out["sum_dt"][i] == h["sum_dt"]
    # ... and the host loop is the C oracle's to rounding (tests/test_backend_* hold that bar for the factors built on it)
    i = 2
    a, b = first[i], first[i + 1]
    o = O.imu_preintegrate(dt[a:b], acc[a:b], gyr[a:b], acc0[i], gyr0[i], ba[i], bg[i], NOISE)
    np.testing.assert_allclose(out["jacobian"][i], np.asarray(o["jacobian"]).reshape(-1), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(out["covariance"][i], np.asarray(o["covariance"]).reshape(-1), rtol=1e-11, atol=1e-22)
    pb.close()


def test_device_intervals_meet_the_reference_formulas_at_60_digits():
    """the device kernel against integration_base.h evaluated with 60 digits (tests/golden/ref_preint.json.gz): all intervals of the fixture in ONE launch"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_golden import check_preint_against_ref
    pb = gfamd.PreintBatch()

    def one(dt, acc, gyr, acc0, gyr0, ba, bg, noise):
        out = pb.run([0, len(dt)], dt, acc, gyr, [acc0], [gyr0], [ba], [bg], noise)
        return {k: v[0] for k, v in out.items()}
    print("device pre-integration vs 60 digits: %.1e" % check_preint_against_ref(one, None))
    pb.close()


def test_batch_sizes_and_repeated_calls():
    pb = gfamd.PreintBatch()
    ref = None
    for n in (1, 7, 512, 2048, 512):                 # growing and shrinking batches reuse one context
        first, dt, acc, gyr, acc0, gyr0, ba, bg = intervals(5, n, max_len=12)
        out = pb.run(first, dt, acc, gyr, acc0, gyr0, ba, bg, NOISE)
        i = n - 1
        a, b = first[i], first[i + 1]
        h = gfamd.imu_preintegrate(dt[a:b], acc[a:b], gyr[a:b], acc0[i], gyr0[i], ba[i], bg[i], NOISE)
        assert np.array_equal(out["covariance"][i], h["covariance"]) and np.array_equal(out["jacobian"][i], h["jacobian"])
        if n == 512:
            assert ref is None or all(np.array_equal(ref[k], out[k]) for k in out)   # the same call twice: the same bits
            ref = out
    st = pb.stats()
    assert st["launches"] == 5 and st["intervals"] == 1 + 7 + 512 + 2048 + 512
    pb.close()


def test_throughput_against_the_host_loop():
    """512 intervals of 7 samples -- one camera frame of 256 sequences, two interval objects each (the frame's own and the window's, estimator.cpp:760-768,
    :866-869): kernel time from hipEvents, the call's wall time (upload, launch, download into pinned memory), and the host loop on one core."""
    n = 512
    rng = np.random.default_rng(9)
    first = np.arange(0, 7 * n + 1, 7, dtype=np.int32)
    m = 7 * n
    dt, acc, gyr = np.full(m, 0.005), rng.normal(0, 1.5, (m, 3)) + np.array([0, 0, 9.8]), rng.normal(0, 0.4, (m, 3))
    acc0, gyr0, ba, bg = acc[:n].copy(), gyr[:n].copy(), np.zeros((n, 3)), np.zeros((n, 3))
    pb = gfamd.PreintBatch()
    pb.run(first, dt, acc, gyr, acc0, gyr0, ba, bg, NOISE)
    k0 = pb.stats()["kernel_ms"]
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        pb.run(first, dt, acc, gyr, acc0, gyr0, ba, bg, NOISE)
    wall = (time.perf_counter() - t0) / reps
    kern = (pb.stats()["kernel_ms"] - k0) / reps
    t0 = time.perf_counter()
    for i in range(64):
        gfamd.imu_preintegrate(dt[7 * i:7 * i + 7], acc[7 * i:7 * i + 7], gyr[7 * i:7 * i + 7], acc0[i], gyr0[i], ba[i], bg[i], NOISE)
    host = (time.perf_counter() - t0) / 64 * n
    print("pre-integration of %d intervals x 7 samples: kernel %.3f ms, call %.3f ms, host loop on one core %.1f ms" % (n, kern, wall * 1e3, host * 1e3))
    assert kern < 0.5 and wall < host                # one launch beats one core; the estimator group spreads the host loop over its threads
    pb.close()
