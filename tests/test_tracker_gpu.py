"""Parity tests proper: HIP path (through the C-ABI) vs the CPU oracle, bit-exact. Run with -m gpu."""
import numpy as np
import pytest
import synth

pytestmark = pytest.mark.gpu


def _frames(seed, n, w=640, h=480):
    return synth.tracker_sequence(seed, n, w, h)


@pytest.mark.parametrize("shape", [(480, 640), (120, 160), (97, 132)])
def test_pyramid_and_scharr_bit_exact(gf, oracle, shape):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    ref = img
    level = 0
    while True:
        out, der = gf.pyramid_level(img, level)
        assert np.array_equal(out, ref), "level %d image" % level
        assert np.array_equal(der, oracle.scharr(ref)), "level %d derivative" % level
        nxt = oracle.pyr_down(ref)
        if level == 3 or nxt.shape[0] <= 21 or nxt.shape[1] <= 21:
            break
        ref = nxt
        level += 1


@pytest.mark.parametrize("max_level,use_init", [(3, False), (1, True), (0, False), (1, False)])
def test_lk_bit_exact(gf, oracle, max_level, use_init):
    tex = synth.make_texture(21)
    f0 = synth.warp_frame(tex, 0, 0)
    f1 = synth.warp_frame(tex, -3.7, 2.2, 0.004, 1.003)
    pts = oracle.good_features(f0, 300, min_dist=15.0)
    rng = np.random.default_rng(4)
    # add hard cases: image border, outside the image, flat-ish areas
    extra = np.array([[0.2, 0.3], [639.5, 479.2], [-30.0, 10.0], [700.0, 100.0], [320.25, 1.5], [5.5, 470.1]], np.float32)
    pts = np.concatenate([pts, extra, rng.uniform(0, 1, (40, 2)).astype(np.float32) * [640, 480]]).astype(np.float32)
    init = (pts + rng.normal(0, 1.5, pts.shape)).astype(np.float32) if use_init else None
    r_pts, r_st, r_it = oracle.lk(f0, f1, pts, init, max_level=max_level)
    g_pts, g_st, g_it = gf.lk_track(f0, f1, pts, init, max_level=max_level)
    assert np.array_equal(r_st, g_st)
    ok = r_st > 0
    assert ok.sum() > 250
    assert np.array_equal(r_pts[ok].view(np.uint32), g_pts[ok].view(np.uint32)), "tracked coordinates differ"
    assert r_it == g_it


def test_lk_small_image_fewer_levels(gf, oracle):
    tex = synth.make_texture(8)
    f0 = synth.warp_frame(tex, 0, 0, w=160, h=120)
    f1 = synth.warp_frame(tex, 1.2, -0.8, w=160, h=120)
    pts = oracle.good_features(f0, 60, min_dist=8.0)
    r_pts, r_st, r_it = oracle.lk(f0, f1, pts, None, max_level=3)
    g_pts, g_st, g_it = gf.lk_track(f0, f1, pts, None, max_level=3)
    assert np.array_equal(r_st, g_st) and r_it == g_it
    assert np.array_equal(r_pts[r_st > 0].view(np.uint32), g_pts[g_st > 0].view(np.uint32))


@pytest.mark.parametrize("shape", [(480, 640), (96, 128)])
def test_min_eigen_val_bit_exact(gf, oracle, shape):
    tex = synth.make_texture(13)
    img = synth.warp_frame(tex, 3, 4, w=shape[1], h=shape[0])
    a = oracle.min_eigen_val(img)
    b = gf.min_eigen_val(img)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("max_corners,min_dist,masked", [(150, 30, False), (300, 20, False), (500, 12, True), (40, 30, True), (2000, 5, False)])
def test_good_features_bit_exact(gf, oracle, max_corners, min_dist, masked):
    tex = synth.make_texture(17)
    img = synth.warp_frame(tex, 0, 0)
    mask = None
    if masked:
        mask = np.full(img.shape, 255, np.uint8)
        for (cx, cy) in [(100, 100), (320, 240), (600, 20), (10, 470)]:
            oracle.fill_circle(mask, cx, cy, min_dist)
        mask[200:260, :] = 0
    a = oracle.good_features(img, max_corners, min_dist=float(min_dist), mask=mask)
    b = gf.good_features(img, max_corners, min_dist=min_dist, mask=mask)
    assert len(a) == len(b) and np.array_equal(a, b)


@pytest.mark.parametrize("shape,min_dist", [((120, 160), 8), ((150, 200), 10), ((122, 188), 6), ((62, 36), 4), ((33, 64), 3), ((480, 640), 30),
                                            ((61, 64), 3), ((91, 132), 3), ((481, 640), 6)])   # heights = 1 mod 30: the last strip's halo row is image row h - 1
def test_good_features_on_odd_image_sizes(gf, oracle, shape, min_dist):
    """the Shi-Tomasi pass walks strips of 60 columns x 30 rows: widths / heights that are no multiples of either, images narrower than one strip, and a mask
    that leaves only pieces of the border strips -- corners and their order must still be OpenCV's"""
    tex = synth.make_texture(5)
    img = synth.warp_frame(tex, 2, -3, w=shape[1], h=shape[0])
    for masked in (False, True):
        mask = None
        if masked:
            mask = np.zeros(img.shape, np.uint8)
            mask[:, -7:] = 255; mask[-5:, :] = 255; mask[:3, :] = 255; mask[:, :2] = 255   # only the image's rim may hold corners
            oracle.fill_circle(mask, shape[1] // 2, shape[0] - 1, min_dist)
        a = oracle.good_features(img, 400, min_dist=float(min_dist), mask=mask)
        b = gf.good_features(img, 400, min_dist=min_dist, mask=mask)
        assert len(a) == len(b) and np.array_equal(a, b), (shape, masked)
        assert masked or len(a) > 20


def test_good_features_degenerate_images(gf, oracle):
    flat = np.full((480, 640), 128, np.uint8)
    assert len(gf.good_features(flat, 50)) == 0 == len(oracle.good_features(flat, 50))
    allmasked = np.zeros((480, 640), np.uint8)
    img = synth.warp_frame(synth.make_texture(2), 0, 0)
    assert len(gf.good_features(img, 50, mask=allmasked)) == 0 == len(oracle.good_features(img, 50, mask=allmasked))


def test_track_image_through_featureless_and_saturated_frames(gf, oracle):
    """empty inputs: frames without any corner (no track exists, LK runs on zero points), then texture, then a saturated frame whose tracks
    the brightness test (> 250, feature_tracker.cpp:160-163) drops, then texture again -- ids and observations stay bit-exact"""
    tex = _frames(1003, 3)
    flat = np.full(tex[0].shape, 128, np.uint8)
    white = np.full(tex[0].shape, 255, np.uint8)
    seq = [flat, flat, tex[0], tex[1], white, tex[2], flat]
    depth = np.full(flat.shape, 1500, np.uint16)
    otr, gtr = oracle.Tracker(oracle.default_cfg()), gf.FeatureTracker(gf.default_cfg())
    counts = []
    for k, f in enumerate(seq):
        oi, oo = otr.track(0.0666 * k, f, depth)
        gi, go = gtr.trackImage(0.0666 * k, f, depth)
        assert np.array_equal(oi, gi), "frame %d: feature id lists differ" % k
        assert np.array_equal(oo.view(np.uint64), go.view(np.uint64)), "frame %d: observations differ" % k
        counts.append(len(gi))
    assert counts[0] == 0 and counts[1] == 0 and counts[2] > 100 and counts[3] > 100
    gtr.close()


@pytest.mark.parametrize("max_cnt,min_dist", [(150, 30), (300, 20), (500, 12)])
def test_track_image_sequence_bit_exact_ids(gf, oracle, max_cnt, min_dist):
    frames = _frames(1000, 8)
    depth = [np.full(f.shape, 1000 + 37 * k, np.uint16) for k, f in enumerate(frames)]
    ocfg = oracle.default_cfg(max_cnt=max_cnt, min_dist=min_dist)
    otr = oracle.Tracker(ocfg)
    gtr = gf.FeatureTracker(gf.default_cfg(max_cnt=max_cnt, min_dist=min_dist))
    for k, f in enumerate(frames):
        t = 0.0666 * k
        oi, oo = otr.track(t, f, depth[k])
        gi, go = gtr.trackImage(t, f, depth[k])
        assert np.array_equal(oi, gi), "frame %d: feature id lists differ" % k
        assert np.array_equal(oo.view(np.uint64), go.view(np.uint64)), "frame %d: observations differ" % k
        os_, gs_ = otr.state(), gtr.state()
        assert np.array_equal(os_[0], gs_[0]) and np.array_equal(os_[1], gs_[1]) and np.array_equal(os_[2], gs_[2])
    # the HIP kernel skips the reverse pass of points whose forward pass already failed (result unchanged)
    assert 0.9 * otr.lk_iters() <= gtr.stats()["lk_iterations"] <= otr.lk_iters()
    gtr.close()


def test_track_image_prediction_and_outlier_feedback(gf, oracle):
    """setPrediction / removeOutliers path (feature_tracker.cpp:118-133, :1006-1045), incl. the <10 fallback."""
    frames = _frames(1003, 6)
    otr = oracle.Tracker(oracle.default_cfg(depth_cam=0))
    gtr = gf.FeatureTracker(gf.default_cfg(depth_cam=0))
    rng = np.random.default_rng(9)
    for k, f in enumerate(frames):
        t = 0.0666 * k
        oi, oo = otr.track(t, f, None)
        gi, go = gtr.trackImage(t, f, None)
        assert np.array_equal(oi, gi) and np.array_equal(oo.view(np.uint64), go.view(np.uint64)), "frame %d" % k
        rm = oi[rng.random(len(oi)) < 0.05]
        otr.remove_outliers(rm); gtr.removeOutliers(rm)
        ids, _, pts = otr.state()
        sel = rng.random(len(ids)) < 0.7
        # predictions: normalised rays of the current pixel + noise; frame 3 gets garbage to force the fallback
        noise = 200.0 if k == 3 else 1.0
        uv = pts[sel] + rng.normal(0, noise, (sel.sum(), 2))
        xyz = np.stack([(uv[:, 0] - 324.0858154296875) / 603.95556640625 * 2.0, (uv[:, 1] - 232.72303771972656) / 603.1257934570312 * 2.0,
                        np.full(len(uv), 2.0)], 1)
        otr.set_prediction(ids[sel], xyz); gtr.setPrediction(ids[sel], xyz)
    gtr.close()


def test_depth_camera_without_depth_image_returns_an_empty_frame(gf, oracle):
    """depth_cam = 1 and no depth image: neither packing loop of trackImage runs (feature_tracker.cpp:320, :344) -- the featureFrame is empty
    while ids / track_cnt / prev_pts advance as usual; the next frame with a depth image reports the aged tracks"""
    frames = _frames(1004, 4)
    depth = np.full(frames[0].shape, 1500, np.uint16)
    otr, gtr = oracle.Tracker(oracle.default_cfg()), gf.FeatureTracker(gf.default_cfg())
    for k, f in enumerate(frames):
        d = None if k in (1, 2) else depth
        oi, oo = otr.track(0.05 * k, f, d)
        gi, go = gtr.trackImage(0.05 * k, f, d)
        assert np.array_equal(oi, gi) and np.array_equal(oo.view(np.uint64), go.view(np.uint64))
        assert (len(gi) == 0) == (d is None)
        os_, gs_ = otr.state(), gtr.state()
        assert len(gs_[0]) > 100 and all(np.array_equal(a, b) for a, b in zip(os_, gs_))
    assert gs_[1].max() == 4     # tracked through the frames without output
    gtr.close()


def test_track_image_with_lens_distortion(gf, oracle):
    """SURVEY.md row T8 with non-zero distortion (config/realsense/idc_cam.yaml): undistortedPts -> PinholeCamera::liftProjective's 8 fixed-point
    iterations (PinholeCamera.cc:450-510), ptsVelocity on the undistorted plane, setPrediction -> spaceToPlane with distortion (:520-542)"""
    K = dict(fx=6.2097277909374247e+02, fy=6.2212293397677581e+02, cx=3.1175896455154810e+02, cy=2.4718077836114819e+02,
             k1=1.4865749308203452e-01, k2=-4.6815685578576460e-01, p1=1.6205585303208318e-03, p2=-8.9101576735577930e-03)
    ocfg, gcfg = oracle.default_cfg(), gf.default_cfg()
    for k, v in K.items():
        setattr(ocfg, k, v); setattr(gcfg, k, v)
    frames = _frames(1005, 6)
    depth = np.full(frames[0].shape, 2100, np.uint16)
    otr, gtr = oracle.Tracker(ocfg), gf.FeatureTracker(gcfg)
    rng = np.random.default_rng(3)
    outs = []
    for k, f in enumerate(frames):
        oi, oo = otr.track(0.0666 * k, f, depth)
        gi, go = gtr.trackImage(0.0666 * k, f, depth)
        outs.append((gi.copy(), go.copy()))
        assert np.array_equal(oi, gi), "frame %d: ids differ" % k
        assert np.array_equal(oo.view(np.uint64), go.view(np.uint64)), "frame %d: observations differ" % k
        # the undistorted coordinates really differ from the pinhole ones
        pin = np.stack([(go[:, 3] - K["cx"]) / K["fx"], (go[:, 4] - K["cy"]) / K["fy"]], 1)
        assert np.abs(pin - go[:, :2]).max() > 1e-3
        ids, _, pts = otr.state()
        sel = rng.random(len(ids)) < 0.6
        xyz = np.stack([go[sel, 0] * 2.0, go[sel, 1] * 2.0, np.full(sel.sum(), 2.0)], 1) + rng.normal(0, 0.002, (sel.sum(), 3))
        otr.set_prediction(ids[sel], xyz); gtr.setPrediction(ids[sel], xyz)
    gtr.close()
    # ... and not only against the oracle: the library's x, y, vx, vy against the reference's formulas in other arithmetic (mpmath at 60 digits / numpy IEEE floats)
    from test_golden import check_t8_against_the_reference_formulas
    exact, total = check_t8_against_the_reference_formulas(outs, [0.0666 * k for k in range(len(frames))], K)
    print("undistorted coordinates: %d of %d floats equal the 60-digit value rounded to float" % (exact, total))
    assert exact >= 0.999 * total


def test_batched_sequences_match_individual_runs(gf, oracle):
    B = 3
    seqs = [_frames(1000 + b, 4) for b in range(B)]
    gtr = gf.FeatureTracker(gf.default_cfg(batch=B, depth_cam=0))
    otrs = [oracle.Tracker(oracle.default_cfg(depth_cam=0)) for _ in range(B)]
    for k in range(4):
        res = gtr.trackImageBatch([0.05 * k] * B, [seqs[b][k] for b in range(B)], None)
        for b in range(B):
            oi, oo = otrs[b].track(0.05 * k, seqs[b][k], None)
            assert np.array_equal(oi, res[b][0]) and np.array_equal(oo.view(np.uint64), res[b][1].view(np.uint64))
    gtr.close()


def test_device_resident_entry_point(gf, oracle):
    import torch
    frames = _frames(1001, 3)
    depth = [np.full(f.shape, 2000, np.uint16) for f in frames]
    gtr = gf.FeatureTracker(gf.default_cfg())
    otr = oracle.Tracker(oracle.default_cfg())
    for k, f in enumerate(frames):
        dg = torch.from_numpy(f).cuda()
        dd = torch.from_numpy(depth[k].view(np.int16)).cuda()
        torch.cuda.synchronize()
        gi, go = gtr.trackImageBatchDevice([0.1 * k], dg.data_ptr(), dd.data_ptr())[0]
        oi, oo = otr.track(0.1 * k, f, depth[k])
        assert np.array_equal(oi, gi) and np.array_equal(oo.view(np.uint64), go.view(np.uint64))
    gtr.close()


def test_prefetched_host_frames_match_the_device_route(gf):
    """gf_tracker_prefetch_batch / gf_tracker_track_prefetched (the host-image boundary with the copy of frame k + 1 under frame k's kernels): ids and
    observations identical to the same frames handed over on the device, over a sequence that alternates the two frame-buffer pairs"""
    import torch
    B = 3
    seqs = [synth.tracker_sequence(40 + b, 6) for b in range(B)]
    depth = np.full(seqs[0][0].shape, 1800, np.uint16)
    a = gf.FeatureTracker(gf.default_cfg(batch=B))
    c = gf.FeatureTracker(gf.default_cfg(batch=B))
    host_g = [torch.from_numpy(np.stack([seqs[b][k] for b in range(B)])).pin_memory() for k in range(6)]
    host_d = torch.from_numpy(np.stack([depth] * B).view(np.int16)).pin_memory()
    c.prefetchHost(host_g[0].data_ptr(), host_d.data_ptr())
    for k in range(6):
        dg = host_g[k].cuda()
        dd = host_d.cuda()
        # the caller's own output table (one of a ring, as a caller does whose estimators still read the previous frames' tables): same content, nothing else touched
        tab, cnt = np.zeros((B, a.cap), gf.OBS_DTYPE), np.zeros(B, np.int32)
        tab["id"] = -9
        ra = a.trackImageBatchDevice([0.0666 * k] * B, dg.data_ptr(), dd.data_ptr(), out=tab, n_out=cnt)
        assert all(cnt[b] == len(ra[b][0]) and np.all(tab["id"][b, cnt[b]:] == -9) for b in range(B))
        if k + 1 < 6:
            c.prefetchHost(host_g[k + 1].data_ptr(), host_d.data_ptr())     # two frames staged: k (oldest) and k + 1, whose copy runs under the kernels of k
        if k == 0:
            with pytest.raises(gf.GfError, match="two frames are staged"):
                c.prefetchHost(host_g[2].data_ptr(), host_d.data_ptr())
        rc_ = c.trackPrefetched([0.0666 * k] * B)
        for (ia, oa), (ic, oc) in zip(ra, rc_):
            assert np.array_equal(ia, ic) and np.array_equal(oa.view(np.uint64), oc.view(np.uint64)), k
        assert len(ra[0][0]) > 50
    with pytest.raises(gf.GfError, match="without a staged frame"):
        c.trackPrefetched([1.0] * B)
    a.close(); c.close()


@pytest.mark.parametrize("points", [2, 4])
def test_lk_with_several_points_per_wavefront_is_bit_identical(gf, oracle, monkeypatch, points):
    """Round 6: lk_track_mp_kernel<P> (GF_LK_POINTS = 2 | 4, read when a tracker is created): every lane keeps its seven template pixels of P points, the wave-uniform work and
    the exact sums are shared between them.  Same arithmetic per point: the building block (hard cases: border, outside, flat areas; plain and predicted-start passes; a point
    count that is no multiple of P) and a tracked sequence must come out bit for bit as the oracle's -- which is what the one-point kernel is held to."""
    monkeypatch.setenv("GF_LK_POINTS", str(points))
    tex = synth.make_texture(21)
    f0 = synth.warp_frame(tex, 0, 0)
    f1 = synth.warp_frame(tex, -3.7, 2.2, 0.004, 1.003)
    pts = oracle.good_features(f0, 301, min_dist=15.0)
    rng = np.random.default_rng(4)
    extra = np.array([[0.2, 0.3], [639.5, 479.2], [-30.0, 10.0], [700.0, 100.0], [320.25, 1.5], [5.5, 470.1]], np.float32)
    pts = np.concatenate([pts, extra, rng.uniform(0, 1, (41, 2)).astype(np.float32) * [640, 480]]).astype(np.float32)
    for max_level, use_init in ((3, False), (1, True)):
        init = (pts + rng.normal(0, 1.5, pts.shape)).astype(np.float32) if use_init else None
        r_pts, r_st, r_it = oracle.lk(f0, f1, pts, init, max_level=max_level)
        g_pts, g_st, g_it = gf.lk_track(f0, f1, pts, init, max_level=max_level)
        assert np.array_equal(r_st, g_st) and r_it == g_it
        assert np.array_equal(r_pts[r_st > 0].view(np.uint32), g_pts[r_st > 0].view(np.uint32))
    frames = _frames(1001, 6)
    depth = np.full(frames[0].shape, 1500, np.uint16)
    otr = oracle.Tracker(oracle.default_cfg(max_cnt=300, min_dist=20))
    gtr = gf.FeatureTracker(gf.default_cfg(max_cnt=300, min_dist=20))
    for k, f in enumerate(frames):
        oi, oo = otr.track(0.0666 * k, f, depth)
        gi, go = gtr.trackImage(0.0666 * k, f, depth)
        assert np.array_equal(oi, gi) and np.array_equal(oo.view(np.uint64), go.view(np.uint64)), k
    gtr.close()


@pytest.mark.parametrize("switch", ["GF_PYR_HEAD", "GF_SELECT_TOPK"])
def test_the_older_forms_of_the_pyramid_head_and_the_corner_selection_stay_bit_identical(gf, oracle, monkeypatch, switch):
    """Round 6: levels 0 + 1 of the pyramid come from one kernel (pyr_head_kernel) and the handful of corners a frame with its tracks alive asks for from repeated block-wide
    maxima (select_topk_kernel); the forms they replace stay behind GF_PYR_HEAD=0 / GF_SELECT_TOPK=0 (read when a tracker is created).  Both forms are held to the oracle's
    bits: a tracked sequence in which the first frame wants every corner (the sort) and the later ones a few (the maxima), and a batch of sequences with different needs."""
    monkeypatch.setenv(switch, "0")
    frames = _frames(1003, 7)
    depth = np.full(frames[0].shape, 1500, np.uint16)
    otr = oracle.Tracker(oracle.default_cfg(max_cnt=150, min_dist=30))
    gtr = gf.FeatureTracker(gf.default_cfg(max_cnt=150, min_dist=30))
    for k, f in enumerate(frames):
        oi, oo = otr.track(0.0666 * k, f, depth)
        gi, go = gtr.trackImage(0.0666 * k, f, depth)
        assert np.array_equal(oi, gi) and np.array_equal(oo.view(np.uint64), go.view(np.uint64)), k
    gtr.close()
    monkeypatch.delenv(switch)
    gtr = gf.FeatureTracker(gf.default_cfg(max_cnt=150, min_dist=30))      # and the default forms on the same frames
    otr = oracle.Tracker(oracle.default_cfg(max_cnt=150, min_dist=30))
    for k, f in enumerate(frames):
        oi, oo = otr.track(0.0666 * k, f, depth)
        gi, go = gtr.trackImage(0.0666 * k, f, depth)
        assert np.array_equal(oi, gi) and np.array_equal(oo.view(np.uint64), go.view(np.uint64)), k
    gtr.close()
