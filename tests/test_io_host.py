"""ROS-free I/O around the path, no GPU needed: the YAML configuration reader against the reference's key names and
cv::FileStorage conventions, the trajectory line, raw PGM frames, and the replay node's callback logic (rosNodeTest.cpp) with a recorder in
place of the estimator."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd"))
import gfamd  # noqa: E402
import synth_stream as SS  # noqa: E402

CAM = """%YAML:1.0
---
model_type: PINHOLE
camera_name: camera
image_width: 640
image_height: 480
distortion_parameters:
   k1: -0.01
   k2: 0.002
   p1: 1.0e-4
   p2: -2.0e-4
projection_parameters:
   fx: 611.5
   fy: 610.25
   cx: 320.5
   cy: 241.0
"""

CFG = """%YAML:1.0

#common parameters
imu: 1         # comments after values
wheel: 1
depth: 1
gnss_enable: 0
w_replace: 1
wdetect: 0
stationary_detect: 1
use_mcc: 1  #wheel-aided mcc.  do not use when wheel slips
num_of_cam: 1
depth_threshold: 7  #  7 for parking1
imu_topic: "/camera/imu"   # a string with a # inside quotes stays: "/a#b"
output_path: "/tmp/out#1/"
cam0_calib: "cam_x.yaml"
image_width: 640
image_height: 480
estimate_extrinsic: 1
extrinsic_type: 0
body_T_cam0: !!opencv-matrix
   rows: 4
   cols: 4
   dt: d
   data: [ 0.0, 0.04, 1.0, 0.05,
           -1.0  ,0.0, -0.00 ,-0.01,
             -0.0 , -1.0  ,0.04 , 0.2,
            0.     ,     0.     ,     0.  ,        1.     ]
estimate_wheel_extrinsic: 0
body_T_wheel: !!opencv-matrix
  rows: 4
  cols: 4
  dt: d
  data: [1, 0, 0, 0.1, 0, 1, 0, 0.2, 0, 0, 1, 0.3, 0, 0, 0, 1]
multiple_thread: 0
max_cnt: 200
min_dist: 25
flow_back: 1
max_solver_time: 0.04
max_num_iterations: 6
keyframe_parallax: 12.5
acc_n: 1.0e-02
gyr_n: 2.0E-03
acc_w: 3e-4
gyr_w: 4.0e-05
g_norm: 9.81
wheel_gyro_noise_sigma: 0.004
wheel_velocity_noise_sigma: 0.02
estimate_wheel_intrinsic: 0
sx: 1.01
sy: 0.99
sw: 1.02
estimate_td: 1
td: -0.005
estimate_td_wheel: 0
td_wheel: 0.01
"""


def write_cfg(tmp_path, text=CFG, cam=CAM):
    (tmp_path / "cam_x.yaml").write_text(cam)
    p = tmp_path / "cfg.yaml"
    p.write_text(text)
    return str(p)


def test_yaml_reader_follows_read_parameters(tmp_path):
    c = gfamd.estimator_cfg_from_yaml(write_cfg(tmp_path))
    assert (c.use_imu, c.use_wheel, c.depth, c.use_mcc, c.wdetect, c.stationary_detect, c.multiple_thread) == (1, 1, 1, 1, 0, 1, 0)
    assert (c.estimate_extrinsic, c.estimate_wheel_extrinsic, c.estimate_wheel_intrinsic, c.estimate_td, c.estimate_td_wheel) == (1, 0, 0, 1, 0)
    assert c.num_iterations == 6 and c.min_parallax_px == 12.5 and c.depth_threshold == 7.0
    assert (c.acc_n, c.gyr_n, c.acc_w, c.gyr_w, c.g_norm) == (1.0e-2, 2.0e-3, 3e-4, 4.0e-5, 9.81)
    assert (c.wheel_vel_n, c.wheel_gyr_n, c.sx, c.sy, c.sw, c.td, c.td_wheel) == (0.02, 0.004, 1.01, 0.99, 1.02, -0.005, 0.01)
    assert (c.window_size, c.focal_length, c.init_depth, c.with_tracker) == (10, 600.0, 5.0, 1)    # compile-time constants of parameters.h / .cpp:478
    # body_T_cam0's rotation block is written to two digits: the reference orthonormalises it through a quaternion (parameters.cpp:386-392)
    ric = np.array(c.ric).reshape(3, 3)
    assert np.allclose(ric @ ric.T, np.eye(3), atol=1e-15) and np.allclose(ric, [[0, 0.04, 1], [-1, 0, 0], [0, -1, 0.04]], atol=0.03) and ric[0, 1] != 0.04
    assert list(c.tic) == [0.05, -0.01, 0.2] and list(c.tio) == [0.1, 0.2, 0.3] and np.array_equal(np.array(c.rio).reshape(3, 3), np.eye(3))
    t = c.tracker
    assert (t.width, t.height, t.batch, t.max_cnt, t.min_dist, t.flow_back, t.depth_cam) == (640, 480, 1, 200, 25, 1, 1)
    assert (t.fx, t.fy, t.cx, t.cy, t.k1, t.k2, t.p1, t.p2) == (611.5, 610.25, 320.5, 241.0, -0.01, 0.002, 1.0e-4, -2.0e-4)


def test_yaml_reader_takes_the_subset_parameterisation_types(tmp_path):
    """extrinsic_type / extrinsic_type_wheel (parameters.cpp:394-420, :280-306) select PoseSubsetParameterization masks (EST:2969-2985, :3010-3026); every shipped
    config/realsense/*.yaml says extrinsic_type: 3.  The wheel key is only read when the wheel extrinsic is estimated."""
    c = gfamd.estimator_cfg_from_yaml(write_cfg(tmp_path, CFG.replace("extrinsic_type: 0", "extrinsic_type: 3")))
    assert (c.estimate_extrinsic, c.extrinsic_type, c.extrinsic_type_wheel) == (1, 3, 0)
    c = gfamd.estimator_cfg_from_yaml(write_cfg(tmp_path, CFG.replace("estimate_wheel_extrinsic: 0", "estimate_wheel_extrinsic: 1\nextrinsic_type_wheel: 4")))
    assert (c.estimate_wheel_extrinsic, c.extrinsic_type_wheel) == (1, 4)
    c = gfamd.estimator_cfg_from_yaml(write_cfg(tmp_path, CFG.replace("estimate_extrinsic: 1", "estimate_extrinsic: 0").replace("extrinsic_type: 0", "extrinsic_type: 3")))
    assert (c.estimate_extrinsic, c.extrinsic_type) == (0, 0)      # not read when the extrinsic is fixed (parameters.cpp:392)
    lib = gfamd.lib()
    assert [lib.gf_pose_subset_mask(t) for t in (0, 1, 2, 3, 4)] == [0x00, 0x38, 0x07, 0x04, 0x3c]
    assert lib.gf_pose_subset_mask(7) == 0x38                      # out of range: the reference warns and keeps its zero-initialised enum = *_TRANSLATION
    import estimator_oracle
    assert [estimator_oracle.subset_mask(t) for t in (0, 1, 2, 3, 4, 7)] == [0x00, 0x38, 0x07, 0x04, 0x3c, 0x38]


GNSS_KEYS = """gnss_local_online_sync: 0
gnss_local_time_diff: 18.0
gnss_elevation_thres: 30
gnss_psr_std_thres: 2.0
gnss_dopp_std_thres: 2.5
gnss_track_num_thres: 20
gnss_ddt_sigma: 0.1
gnss_iono_default_parameters: !!opencv-matrix
  rows: 1
  cols: 8
  dt: d
  data: [0.1118E-07,  0.2235E-07, -0.4172E-06,  0.6557E-06,
         0.1249E+06, -0.4424E+06,  0.1507E+07, -0.2621E+06]
"""


def test_yaml_reader_takes_the_gnss_keys(tmp_path):
    # parameters.cpp:519-552
    c = gfamd.estimator_cfg_from_yaml(write_cfg(tmp_path, CFG.replace("gnss_enable: 0", "gnss_enable: 1\n" + GNSS_KEYS)))
    assert (c.gnss_enable, c.gnss_track_num_thres, c.gnss_local_time_diff, c.gnss_elevation_thres) == (1, 20, 18.0, 30.0)
    assert (c.gnss_psr_std_thres, c.gnss_dopp_std_thres, c.gnss_ddt_sigma) == (2.0, 2.5, 0.1)
    assert list(c.gnss_iono) == [0.1118e-07, 0.2235e-07, -0.4172e-06, 0.6557e-06, 0.1249e+06, -0.4424e+06, 0.1507e+07, -0.2621e+06]
    assert gfamd.estimator_cfg_from_yaml(write_cfg(tmp_path)).gnss_enable == 0


@pytest.mark.parametrize("edit,needle", [
    (("gnss_enable: 0", "gnss_enable: 1"), "gnss_iono_default_parameters"),
    (("gnss_enable: 0", "gnss_enable: 1\n" + GNSS_KEYS.replace("gnss_local_online_sync: 0", "gnss_local_online_sync: 1")), "gnss_local_online_sync"),
    (("num_of_cam: 1", "num_of_cam: 2"), "num_of_cam"),
    (("estimate_extrinsic: 1", "estimate_extrinsic: 2"), "estimate_extrinsic"),
    (("w_replace: 1", "w_replace: 1\nuse_line: 1"), "use_line"),
    (("cam0_calib: \"cam_x.yaml\"", "cam0_calib: \"nope.yaml\""), "cannot open"),
    (("0.     ,     0.     ,     0.  ,        1.     ]", "0., 0., 0. ]"), "rows*cols"),
    (("  data: [1, 0, 0, 0.1,", "  data: [1, zero, 0, 0.1,"), "bad number"),
])
def test_yaml_reader_refuses_what_the_build_does_not_carry(tmp_path, edit, needle):
    assert edit[0] in CFG
    with pytest.raises(gfamd.GfError) as e:
        gfamd.estimator_cfg_from_yaml(write_cfg(tmp_path, CFG.replace(*edit)))
    assert needle in str(e.value)


def test_yaml_reader_missing_file_and_missing_keys(tmp_path):
    with pytest.raises(gfamd.GfError):
        gfamd.estimator_cfg_from_yaml(str(tmp_path / "absent.yaml"))
    # cv::FileNode: a missing numeric key reads as 0
    c = gfamd.estimator_cfg_from_yaml(write_cfg(tmp_path, CFG.replace("keyframe_parallax: 12.5\n", "").replace("use_mcc: 1", "")))
    assert c.min_parallax_px == 0.0 and c.use_mcc == 0


REF_CFG = "/root/reference/config/realsense/m2dgrp.yaml"


@pytest.mark.skipif(not os.path.exists(REF_CFG), reason="the reference tree is only present in the build container")
def test_shipped_configs_parse_and_match_the_defaults():
    c, d = gfamd.estimator_cfg_from_yaml(REF_CFG), gfamd.default_estimator_cfg()
    for k, _ in gfamd.EstimatorCfg._fields_:
        if k in ("tracker", "with_tracker"):
            continue
        a, b = getattr(c, k), getattr(d, k)
        if k == "max_solver_time":
            assert (a, b) == (0.04, 0.0), k                             # the file's cap is read; a default handle runs without one (parity mode)
        elif k == "gnss_iono":
            assert list(a) == list(b), k
        elif k in ("tic", "ric", "tio", "rio"):
            assert np.allclose(list(a), list(b), atol=1e-6), k      # the yaml's body_T_wheel is orthonormal to 6 digits only; both are normalised
        else:
            assert a == b, k
    t, u = c.tracker, gfamd.default_cfg()
    for k, _ in gfamd.TrackerCfg._fields_:
        assert getattr(t, k) == getattr(u, k), k
    for name in ("gc_test.yaml", "groundchallenge.yaml", "idc_rs.yaml"):
        gfamd.estimator_cfg_from_yaml(os.path.join(os.path.dirname(REF_CFG), name))


def test_exported_dataset_config_round_trips(tmp_path):
    st = SS.Stream(3, t_still=0.2, t_move=0.2)
    assert st.export(str(tmp_path), 2, multiple_thread=0, max_cnt=120) == 2
    c = gfamd.estimator_cfg_from_yaml(str(tmp_path / "config.yaml"))
    d = gfamd.default_estimator_cfg(tio=SS.TIO, rio=SS.RIO, multiple_thread=0, with_tracker=1)
    d.tracker = gfamd.default_cfg(max_cnt=120)
    for k, _ in gfamd.EstimatorCfg._fields_:
        a, b = getattr(c, k), getattr(d, k)
        if k == "tracker":
            assert all(getattr(a, f) == getattr(b, f) for f, _ in gfamd.TrackerCfg._fields_)
        elif k in ("tic", "ric", "tio", "rio", "gnss_iono"):
            assert list(a) == list(b), k
        elif k == "max_solver_time":
            assert (a, b) == (0.04, 0.0), k
        else:
            assert a == b, k
    img, dep = st.image(1)
    assert np.array_equal(gfamd.read_pgm(str(tmp_path / "frames" / "000001_gray.pgm")), img)
    assert np.array_equal(gfamd.read_pgm(str(tmp_path / "frames" / "000001_depth.pgm")), dep)
    assert len(open(tmp_path / "image0.csv").read().split()) == 2


def test_pgm_header_with_absurd_numbers_is_rejected(tmp_path):
    """a garbage header must fail cleanly, not overflow the parser's accumulator"""
    for hdr in (b"P5\n99999999999999999999 480\n255\n", b"P5\n640 480\n70000\n", b"P5\n640 -3\n255\n"):
        q = str(tmp_path / "bad.pgm")
        with open(q, "wb") as f:
            f.write(hdr + b"\0" * 64)
        with pytest.raises(gfamd.GfError):
            gfamd.read_pgm(q)


def test_tum_line_and_pgm(tmp_path):
    p = str(tmp_path / "vio.txt")
    a = 0.3
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    gfamd.tum_append(p, 1690000000.123456789, [1.5, -2.25, 0.125], R)
    gfamd.tum_append(p, 2.0, [0, 0, 0], np.diag([-1.0, -1.0, 1.0]))       # trace <= 0 branch of Eigen's conversion: q = (0, 0, 1; 0)
    l0, l1 = open(p).read().splitlines()
    f = l0.split(" ")
    assert len(f) == 8 and all(len(x.split(".")[1]) == 9 for x in f)        # ios::fixed, setprecision(9)
    assert f[:4] == ["1690000000.123456717", "1.500000000", "-2.250000000", "0.125000000"]
    assert np.allclose([float(x) for x in f[4:]], [0, 0, np.sin(a / 2), np.cos(a / 2)], atol=1e-9)
    assert [float(x) for x in l1.split(" ")[4:]] == [0.0, 0.0, 1.0, 0.0]
    rng = np.random.default_rng(0)
    for dt in (np.uint8, np.uint16):
        img = rng.integers(0, np.iinfo(dt).max, (7, 13)).astype(dt)
        q = str(tmp_path / "a.pgm")
        gfamd.write_pgm(q, img)
        assert np.array_equal(gfamd.read_pgm(q), img) and gfamd.read_pgm(q).dtype == dt
    with open(q, "r+b") as fh:
        fh.truncate(40)
    with pytest.raises(gfamd.GfError):
        gfamd.read_pgm(q)
    (tmp_path / "b.pgm").write_bytes(b"P2\n1 1\n255\n0\n")
    with pytest.raises(gfamd.GfError):
        gfamd.read_pgm(str(tmp_path / "b.pgm"))


NODE_TEST = r"""
#include <cmath>
#include <cstdio>
#include "ground-fusion_amd/host/replay_node.h"
struct Rec {   // stands in for gf::Estimator
    std::vector<double> imu_t, wheel_t, wheel_gz, img_t; std::vector<int> img_px;
    void inputIMU(double t, const gf::Vec3&, const gf::Vec3&) { imu_t.push_back(t); }
    void inputWheel(double t, const gf::Vec3&, const gf::Vec3& g) { wheel_t.push_back(t); wheel_gz.push_back(g[2]); }
    void inputImage(double t, const gf::GrayImage& g, const gf::DepthImage& d) { img_t.push_back(t); img_px.push_back(g.data[1] + d.data[1]); }
    std::vector<double> gnss_t; std::vector<int> gnss_n; int n_align = 0;
    void inputGNSS(double t, const std::vector<gf_gnss_obs>& m) { gnss_t.push_back(t); gnss_n.push_back((int)m.size()); }
    void setGNSSAlignment(const gf::Vec3&, double, const double*, double) { n_align++; }
};
int main(int argc, char** argv) {
    Rec r; gf::ReplayNode<Rec> n(r); n.w_replace = atoi(argv[2]);
    const std::string src = argv[1];
    if (src.size() > 4 && src.substr(src.size() - 4) == ".bag") n.run_bag(src, "/imu", "/odom", "/img0", "/img1");
    else n.run(src);
    printf("imu %zu wheel %zu pairs %ld thrown %ld %ld\n", r.imu_t.size(), r.wheel_t.size(), n.n_pairs, n.n_thrown0, n.n_thrown1);
    for (size_t i = 0; i < r.gnss_t.size(); i++) printf("g %.9f %d\n", r.gnss_t[i], r.gnss_n[i]);
    for (size_t i = 0; i < r.wheel_t.size(); i++) printf("w %.9f %.12f\n", r.wheel_t[i], r.wheel_gz[i]);
    for (size_t i = 0; i < r.img_t.size(); i++) printf("i %.9f %d\n", r.img_t[i], r.img_px[i]);
    return 0;
}
"""


def test_replay_node_callbacks(tmp_path):
    """wheel_callback's yaw-rate substitution (rosNodeTest.cpp:81-147) and sync_process's 3 ms pairing (:395-428) on a hand-made message log."""
    d = tmp_path / "ds"
    (d / "frames").mkdir(parents=True)
    imu_t = np.arange(0, 0.1001, 0.005)
    gy = 0.1 + 2.0 * imu_t                                           # IMU gyro y, linear in t: interpolation is exact
    with open(d / "imu.csv", "w") as f:
        for t, g in zip(imu_t, gy):
            f.write("%r,0,0,9.8,0.01,%r,0.02\n" % (float(t), float(g)))
    # messages arrive in stamp order, so only IMU samples older than the odometry stamp are queued: the substitution happens when two of them
    # lie within 6 ms of it (0.0258, 0.0459; extrapolated from both), not when one does (0.0811) or when an earlier odometry message has
    # already consumed the queue (0.0487)
    wheel_t = [0.0258, 0.0459, 0.0487, 0.0811]
    with open(d / "wheel.csv", "w") as f:
        for t in wheel_t:
            f.write("%r,1,0,0,0.5,0.6,0.7\n" % t)
    for k in range(4):
        gfamd.write_pgm(str(d / "frames" / ("g%d.pgm" % k)), np.full((4, 6), 10 + k, np.uint8))
        gfamd.write_pgm(str(d / "frames" / ("d%d.pgm" % k)), np.full((4, 6), 1000 + k, np.uint16))
    # gray 0 has no depth partner (thrown), depth 1 lags by 2 ms (paired), depth 2 lags by 5 ms (gray 2 thrown, depth 2 thrown once gray 3 leads)
    (d / "image0.csv").write_text("0.010,frames/g0.pgm\n0.040,frames/g1.pgm\n0.070,frames/g2.pgm\n0.090,frames/g3.pgm\n")
    (d / "image1.csv").write_text("0.042,frames/d1.pgm\n0.075,frames/d2.pgm\n0.0905,frames/d3.pgm\n")
    src = tmp_path / "n.cpp"
    src.write_text(NODE_TEST)
    exe = tmp_path / "n"
    lib = os.path.join(ROOT, "ground-fusion_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", ROOT, str(src), "-L", lib, "-lgroundfusion_hip", "-Wl,-rpath," + lib, "-o", str(exe)])

    def run(w_replace):
        out = subprocess.check_output([str(exe), str(d), str(w_replace)]).decode().splitlines()
        return out[0], [tuple(float(x) for x in l.split()[1:]) for l in out if l.startswith("w ")], [l.split()[1:] for l in out if l.startswith("i ")]

    head, wheel, imgs = run(1)
    assert head == "imu 21 wheel 4 pairs 2 thrown 2 1"
    expect = [-(0.1 + 2.0 * 0.0258), -(0.1 + 2.0 * 0.0459), 0.7, 0.7]      # -(gyro y) carried linearly to the odometry stamp
    assert [w[0] for w in wheel] == wheel_t and np.allclose([w[1] for w in wheel], expect, atol=1e-12)
    assert imgs == [["0.040000000", str(11 + 1001)], ["0.090000000", str(13 + 1003)]]
    _, wheel0, _ = run(0)
    assert [w[1] for w in wheel0] == [0.7] * 4
    # optional GNSS topic: rows of one message share t_msg; the message reaches inputGNSS with its own stamp and all its observations
    row = "%r,%d,0,%r,2.2e7,100.0,0.5,0.3,0.19,1e7,1e7,1e7,100,200,300,1e-5,1e-12,1e-9,2,2,345600.5"
    (d / "gnss.csv").write_text("# header\n" + "\n".join([row % (18.02, 1, 18.02), row % (18.02, 2, 18.02), row % (18.09, 1, 18.09)]) + "\n")
    (d / "gnss_align.csv").write_text("0.015,1,2,3,0.3,10,20,30,40,2.0\n")
    out = subprocess.check_output([str(exe), str(d), "1"]).decode().splitlines()
    assert [l for l in out if l.startswith("g ")] == ["g 18.020000000 2", "g 18.090000000 1"]


def test_replay_node_from_a_bag_matches_the_csv_route(tmp_path):
    """ReplayNode::run_bag: the hand-made log of test_replay_node_callbacks as a ROS bag (bz2 chunks, an unrelated topic in between, colour as bgr8) drives the
    same callbacks to the same result -- yaw-rate substitution, pairing, thrown frames and all."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import bagwriter as BW
    d = tmp_path / "ds"
    (d / "frames").mkdir(parents=True)
    imu_t = np.arange(0, 0.1001, 0.005)
    gy = 0.1 + 2.0 * imu_t
    wheel_t = [0.0258, 0.0459, 0.0487, 0.0811]
    img0 = [(0.010, 0), (0.040, 1), (0.070, 2), (0.090, 3)]
    img1 = [(0.042, 1), (0.075, 2), (0.0905, 3)]
    with open(d / "imu.csv", "w") as f:
        for t, g in zip(imu_t, gy):
            f.write("%r,0,0,9.8,0.01,%r,0.02\n" % (float(t), float(g)))
    with open(d / "wheel.csv", "w") as f:
        for t in wheel_t:
            f.write("%r,1,0,0,0.5,0.6,0.7\n" % t)
    for k in range(4):
        gfamd.write_pgm(str(d / "frames" / ("g%d.pgm" % k)), np.full((4, 6), 10 + k, np.uint8))
        gfamd.write_pgm(str(d / "frames" / ("d%d.pgm" % k)), np.full((4, 6), 1000 + k, np.uint16))
    (d / "image0.csv").write_text("".join("%r,frames/g%d.pgm\n" % tk for tk in img0))
    (d / "image1.csv").write_text("".join("%r,frames/d%d.pgm\n" % tk for tk in img1))
    ns = lambda t: int(round(t * 1e9))
    ev = [(ns(t), 0, (t, g)) for t, g in zip(imu_t, gy)] + [(ns(t), 1, t) for t in wheel_t] + [(ns(t), 2, k) for t, k in img0] + [(ns(t), 3, k) for t, k in img1]
    ev.sort(key=lambda e: (e[0], e[1]))
    bag = tmp_path / "log.bag"
    wr = BW.BagWriter(str(bag), compression="bz2", chunk_bytes=700)
    for seq, (t, kind, x) in enumerate(ev):
        if kind == 0:
            wr.write("/imu", "sensor_msgs/Imu", t, BW.imu(seq, t, (0, 0, 9.8), (0.01, float(x[1]), 0.02)))
            wr.write("/tf", "tf2_msgs/TFMessage", t, b"\0\0\0\0")       # a topic nobody subscribes
        elif kind == 1:
            wr.write("/odom", "nav_msgs/Odometry", t, BW.odometry(seq, t, (1, 0, 0), (0.5, 0.6, 0.7)))
        elif kind == 2:
            wr.write("/img0", "sensor_msgs/Image", t, BW.image(seq, t, np.full((4, 6, 3), 10 + x, np.uint8), "bgr8", step_pad=2))
        else:
            wr.write("/img1", "sensor_msgs/Image", t, BW.image(seq, t, np.full((4, 6), 1000 + x, np.uint16), "16UC1"))
    wr.close()
    src = tmp_path / "n.cpp"
    src.write_text(NODE_TEST)
    exe = tmp_path / "n"
    lib = os.path.join(ROOT, "ground-fusion_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", ROOT, str(src), "-L", lib, "-lgroundfusion_hip", "-Wl,-rpath," + lib, "-ldl", "-o", str(exe)])
    for w_replace in (1, 0):
        a = subprocess.check_output([str(exe), str(d), str(w_replace)]).decode()
        b = subprocess.check_output([str(exe), str(bag), str(w_replace)]).decode()
        assert a == b and a.splitlines()[0] == "imu 21 wheel 4 pairs 2 thrown 2 1"
    # a topic with the wrong datatype is refused, not misread
    wr = BW.BagWriter(str(bag))
    wr.write("/imu", "nav_msgs/Odometry", 5, BW.odometry(0, 5, (1, 0, 0), (0, 0, 0)))
    wr.close()
    r = subprocess.run([str(exe), str(bag), "0"], capture_output=True, text=True)
    assert r.returncode != 0 and "subscribes it as sensor_msgs/Imu" in r.stderr
