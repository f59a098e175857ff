"""SURVEY.md §8(f)2, first item: ROS bag (format 2.0) -> the messages the reference's node subscribes to -> raw frames, without ROS.
CPU tests of ground-fusion_amd/host/rosbag_reader.h through the C-ABI (gf_bag_*, gf_ros_decode_*): bags written by tests/bagwriter.py in every
container variant rosbag produces (uncompressed / bz2 / lz4 chunks, with and without index records, with and without the trailing connection
records), read back message for message; cv_bridge's MONO8 / MONO16 conversions (rosNodeTest.cpp:238-288)."""
import os
import struct
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gfamd  # noqa: E402
import bagwriter as BW  # noqa: E402

TOPICS = dict(imu="/camera/imu", wheel="/odom", img0="/camera/color/image_raw", img1="/camera/aligned_depth_to_color/image_raw")


def _messages(n_cam=6, seed=0, w=64, h=48):
    """one second of a synthetic recording: 200 Hz IMU, 50 Hz odometry, colour + depth frames; (time ns, topic key, datatype, payload, decoded truth)"""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(40):
        t = 1_600_000_000_000_000_000 + k * 5_000_000
        acc, gyr = rng.normal(0, 1, 3), rng.normal(0, 0.1, 3)
        out.append((t, "imu", "sensor_msgs/Imu", BW.imu(k, t, acc, gyr), (acc, gyr)))
        if k % 4 == 0:
            lin, ang, pos = rng.normal(0, 1, 3), rng.normal(0, 0.1, 3), rng.normal(0, 5, 3)
            out.append((t + 1_000, "wheel", "nav_msgs/Odometry", BW.odometry(k, t + 1_000, lin, ang, pos), (lin, ang, pos)))
    for k in range(n_cam):
        t = 1_600_000_000_000_000_000 + k * 33_333_333
        rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        dep = rng.integers(0, 6000, (h, w), dtype=np.uint16)
        out.append((t, "img0", "sensor_msgs/Image", BW.image(k, t, rgb, "rgb8", step_pad=5), rgb))
        out.append((t + 2_000_000, "img1", "sensor_msgs/Image", BW.image(k, t + 2_000_000, dep, "16UC1"), dep))
    out.sort(key=lambda m: m[0])
    return out


def _gray_of_rgb(rgb):
    """OpenCV 4.2 cvtColor RGB2GRAY on 8-bit data (color_rgb.cpp RGB2Gray<uchar>)"""
    a = rgb.astype(np.int64)
    return ((a[..., 0] * 4899 + a[..., 1] * 9617 + a[..., 2] * 1868 + (1 << 13)) >> 14).astype(np.uint8)


@pytest.mark.parametrize("compression,index,trailer", [("none", True, True), ("bz2", True, True), ("lz4", True, True), ("none", False, True), ("lz4", False, False),
                                                       ("bz2", True, False)])
def test_bag_round_trip(tmp_path, compression, index, trailer):
    msgs = _messages()
    path = tmp_path / "a.bag"
    wr = BW.BagWriter(str(path), compression=compression, chunk_bytes=40_000, index=index, trailer=trailer)
    for t, key, dtype, payload, _ in msgs:
        wr.write(TOPICS[key], dtype, t, payload)
    wr.close()
    bag = gfamd.Bag(path)
    conns = bag.connections()
    assert sorted((t, d) for _, t, d in conns) == sorted({(TOPICS[k], d) for _, k, d, _, _ in msgs})
    by_id = {cid: topic for cid, topic, _ in conns}
    assert bag.select() == len(msgs)
    for i, (t, key, dtype, payload, truth) in enumerate(msgs):
        cid, t_rec, data = bag.message(i)
        assert by_id[cid] == TOPICS[key] and data == payload and abs(t_rec - t * 1e-9) < 1e-6
        stamp = float(t // 1_000_000_000) + 1e-9 * float(t % 1_000_000_000)       # ros::Time::toSec
        if key == "imu":
            ts, acc, gyr = gfamd.ros_decode_imu(data)
            assert ts == stamp and np.array_equal(acc, truth[0]) and np.array_equal(gyr, truth[1])
        elif key == "wheel":
            ts, lin, ang, pos = gfamd.ros_decode_odometry(data)
            assert ts == stamp and np.array_equal(lin, truth[0]) and np.array_equal(ang, truth[1]) and np.array_equal(pos, truth[2])
        elif key == "img0":
            ts, g = gfamd.ros_decode_image(data, depth=False)
            assert ts == stamp and np.array_equal(g, _gray_of_rgb(truth))
        else:
            ts, d = gfamd.ros_decode_image(data, depth=True)
            assert ts == stamp and np.array_equal(d, truth)
    # topic selection keeps the play order
    n = bag.select([TOPICS["img0"], TOPICS["img1"]])
    assert n == sum(1 for m in msgs if m[1] in ("img0", "img1"))
    assert [by_id[bag.message(i)[0]] for i in range(n)] == [TOPICS[m[1]] for m in msgs if m[1] in ("img0", "img1")]
    bag.close()


def test_play_order_is_record_time_not_file_order(tmp_path):
    """a recorder writes messages as they arrive per connection queue: chunks are not globally sorted, `rosbag play` sorts by record time"""
    path = tmp_path / "b.bag"
    wr = BW.BagWriter(str(path), chunk_bytes=600)
    times = [50, 10, 40, 20, 30, 20]
    for k, t in enumerate(times):
        wr.write("/imu" if k % 2 else "/imu2", "sensor_msgs/Imu", 1_000_000_000 + t, BW.imu(k, 7, (k, 0, 0), (0, 0, 0)))
    wr.close()
    bag = gfamd.Bag(path)
    assert bag.select() == 6
    seqs = [int(gfamd.ros_decode_imu(bag.message(i)[2])[1][0]) for i in range(6)]
    assert seqs == [1, 3, 5, 4, 2, 0]        # by time; the two messages at t = 20 in file order (3 before 5)


def test_image_conversions_follow_cv_bridge():
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (9, 7, 3), dtype=np.uint8)
    gray = _gray_of_rgb(rgb)
    for enc, arr in (("rgb8", rgb), ("bgr8", rgb[..., ::-1]), ("rgba8", np.dstack([rgb, np.full((9, 7), 9, np.uint8)])),
                     ("bgra8", np.dstack([rgb[..., ::-1], np.full((9, 7), 9, np.uint8)]))):
        t, g = gfamd.ros_decode_image(BW.image(1, 2_500_000_000, arr, enc, step_pad=3))
        assert t == 2.5 and np.array_equal(g, gray), enc
    m8 = rng.integers(0, 256, (9, 7), dtype=np.uint8)
    for enc in ("mono8", "8UC1"):               # 8UC1 is relabelled mono8 (rosNodeTest.cpp:241-252)
        assert np.array_equal(gfamd.ros_decode_image(BW.image(1, 0, m8, enc, step_pad=1))[1], m8)
    d = rng.integers(0, 65536, (9, 7), dtype=np.uint16)
    for enc, be in (("16UC1", False), ("mono16", False), ("16UC1", True)):     # the depth topic is relabelled MONO16 whatever it says (:265-286)
        assert np.array_equal(gfamd.ros_decode_image(BW.image(1, 0, d, enc, big_endian=be), depth=True)[1], d)
    # pure colours: the fixed-point weights (4899, 9617, 1868) / 16384
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255]]], np.uint8)
    assert list(gfamd.ros_decode_image(BW.image(1, 0, px, "rgb8"))[1][0]) == [76, 150, 29, 255]


def test_reader_refuses_what_it_cannot_read(tmp_path):
    p = tmp_path / "x.bag"
    p.write_bytes(b"#ROSBAG V1.2\n" + b"\0" * 64)
    with pytest.raises(gfamd.GfError, match="format 2.0"):
        gfamd.Bag(p)
    with pytest.raises(gfamd.GfError, match="cannot open"):
        gfamd.Bag(tmp_path / "absent.bag")
    with pytest.raises(gfamd.GfError, match="32FC1"):
        gfamd.ros_decode_image(BW.image(1, 0, np.zeros((4, 4), np.uint8), "32FC1"))
    with pytest.raises(gfamd.GfError, match="two bytes"):
        gfamd.ros_decode_image(BW.image(1, 0, np.zeros((4, 4), np.uint8), "mono8"), depth=True)
    with pytest.raises(gfamd.GfError, match="truncated"):
        gfamd.ros_decode_imu(BW.imu(1, 0, (0, 0, 0), (0, 0, 0))[:-3])
    # a chunk cut off in the middle of the file
    wr = BW.BagWriter(str(p), chunk_bytes=1 << 20)
    for k in range(20):
        wr.write("/imu", "sensor_msgs/Imu", k, BW.imu(k, k, (0, 0, 0), (0, 0, 0)))
    wr.close()
    data = p.read_bytes()
    p.write_bytes(data[:4096 + 13 + 2000])
    with pytest.raises(gfamd.GfError, match="truncated"):
        gfamd.Bag(p)


def test_lz4_block_decoder_against_hand_made_blocks(tmp_path):
    """the block format itself: a run (overlapping match, offset 1), a long literal run and a long match (length bytes of 255), stored blocks"""
    raw = b"A" * 1000 + bytes(range(256)) * 3 + b"tail-of-the-chunk"
    for stored_every in (0, 1, 2):
        path = tmp_path / ("l%d.bag" % stored_every)
        wr = BW.BagWriter(str(path), compression="lz4", chunk_bytes=1 << 30)
        wr.write("/blob", "sensor_msgs/Image", 5, BW.image(0, 5, np.frombuffer(raw, np.uint8).reshape(1, -1), "mono8"))
        # force small blocks so that several blocks (compressed and stored) make up the chunk
        orig = BW.lz4_frame
        BW.lz4_frame = lambda d, stored_every=stored_every, **kw: orig(d, block=700, stored_every=stored_every)
        try:
            wr.close()
        finally:
            BW.lz4_frame = orig
        bag = gfamd.Bag(path)
        assert bag.select() == 1
        assert gfamd.ros_decode_image(bag.message(0)[2])[1].tobytes() == raw
    assert len(BW.lz4_block(b"A" * 1000)) < 30            # the writer really emits matches
