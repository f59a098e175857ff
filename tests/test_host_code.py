"""Host-side product code that needs no GPU (pre-integration, double2vector gauge fix) against the oracle."""
import numpy as np
import synth_window as SW


def test_host_preintegration_matches_oracle(oracle):
    import gfamd
    rng = np.random.default_rng(3)
    n = 14
    dt = np.full(n, 0.005)
    acc = rng.normal(0, 1, (n, 3)) + [0, 0, 9.8]
    gyr = rng.normal(0, 0.3, (n, 3))
    a = oracle.imu_preintegrate(dt, acc, gyr, acc[0], gyr[0], [0.01, -0.02, 0.03], [0.001, 0.002, -0.001], [0.1, 0.01, 0.001, 0.0001])
    b = gfamd.imu_preintegrate(dt, acc, gyr, acc[0], gyr[0], [0.01, -0.02, 0.03], [0.001, 0.002, -0.001], [0.1, 0.01, 0.001, 0.0001])
    for k in a:
        assert np.allclose(a[k], b[k], rtol=1e-12, atol=1e-15), k
    vel = rng.normal(1, 0.1, (n, 3))
    a = oracle.wheel_preintegrate(dt, vel, gyr, vel[0], gyr[0], [1.01, 0.99, 1.02], [0.1, 0.01])
    b = gfamd.wheel_preintegrate(dt, vel, gyr, vel[0], gyr[0], [1.01, 0.99, 1.02], [0.1, 0.01])
    for k in a:
        assert np.allclose(a[k], b[k], rtol=1e-12, atol=1e-15), k


def test_double2vector_keeps_yaw_and_position_of_pose0(oracle):
    import gfamd
    w = SW.make_window(3, oracle)
    W = w["W"]
    pp = w["para_Pose"].reshape(-1, 7).copy()
    q = pp[0, 3:]
    x, y, z, qw = q
    R0 = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * qw), 2 * (x * z + y * qw)], [2 * (x * y + z * qw), 1 - 2 * (x * x + z * z), 2 * (y * z - x * qw)],
                   [2 * (x * z - y * qw), 2 * (y * z + x * qw), 1 - 2 * (x * x + y * y)]])
    P0 = pp[0, :3].copy()
    oracle.ba_solve(w, 4)  # moves pose 0 (gauge drift)
    a = oracle.double2vector(W, R0, P0, w["para_Pose"], w["para_SpeedBias"])
    b = gfamd.double2vector(W, R0, P0, w["para_Pose"], w["para_SpeedBias"])
    for u, v in zip(a, b):
        assert np.allclose(u, v, rtol=0, atol=1e-13)
    Rs, Ps = b[0].reshape(-1, 3, 3), b[1].reshape(-1, 3)
    assert np.allclose(Ps[0], P0, atol=1e-12)
    yaw = lambda R: np.arctan2(R[1, 0], R[0, 0])
    assert abs(yaw(Rs[0]) - yaw(R0)) < 1e-12
    assert np.allclose(Rs[5] @ Rs[5].T, np.eye(3), atol=1e-12)


def test_cpp_host_mirror_compiles_and_links(tmp_path):
    """The C++ drop-in classes (host/feature_tracker.h, host/estimator.h, host/estimator_backend.h) compile against the C-ABI with plain g++."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.cpp"
    src.write_text('#include <cmath>\n#include "ground-fusion_amd/host/feature_tracker.h"\n#include "ground-fusion_amd/host/estimator_backend.h"\n#include "ground-fusion_amd/host/estimator.h"\n'
                   'int main() { gf::FeatureTracker t; t.setIntrinsics(640, 480, 600, 600, 320, 240); (void)sizeof(gf::EstimatorBackend);\n'
                   '  gf::Estimator e; e.setParameter(); gf::Vec3 a{0, -9.8, 0}, w{0, 0, 0}; e.inputIMU(0.0, a, w); e.inputWheel(0.0, w, w);\n'
                   '  gf::FeatureFrame f; e.inputFeature(1.0, f);   /* waits: IMU data do not cover t yet */\n'
                   '  e.inputrawodom(0.5, w);                          /* estimator.h:116: queued, never consumed */\n'
                   '  bool threw = false; try { gf::Estimator fresh; fresh.setParameter(); fresh.processImage(f, 0.1); } catch (const std::runtime_error&) { threw = true; }   /* no IMU sample yet */\n'
                   '  gf::Estimator g; g.cfg.gnss_enable = 1; g.setParameter();   /* the GNSS surface: estimator.h:97-100, members read by visualization.cpp:454-545 */\n'
                   '  gf_gnss_obs o{}; o.sat = 3; o.sys = 0; o.time = 18.1; o.psr = 2.2e7; o.psr_std = 0.5; o.dopp_std = 0.3; o.wavelength = 0.19; o.sv_pos[0] = 2.6e7;\n'
                   '  g.inputGNSS(18.1, std::vector<gf_gnss_obs>{o}); g.inputGNSSTimeDiff(18.0); g.inputIonoParams(0.0, std::vector<double>(8, 1e-8));\n'
                   '  gf_gnss_ephem eph{}; eph.sat = 3; eph.sys = 0; eph.A = 2.656e7; eph.toe = 0.0; g.inputEphem(eph);\n'
                   '  gf_gnss_raw_obs r{}; r.sat = 3; r.sys = 0; r.time = 18.2; r.psr = 2.2e7; r.freq = 1575.42e6; g.inputGNSSRaw(18.2, std::vector<gf_gnss_raw_obs>{r});\n'
                   '  g.inputIMU(0.0, a, w);                          /* refresh() pulls the GNSS members */\n'
                   '  const bool gn = !g.gnss_ready && (int)g.para_rcv_dt.size() == 44 && g.key_poses.empty() && g.R_enu_local[0] == 1.0 && g.anc_ecef[0] == 0.0;\n'
                   '  /* the callbacks of rosNodeTest.cpp read these right after inputIMU / inputWheel (estimator.cpp:332-335, :363-366) and visualization.cpp reads rpw / zpw */\n'
                   '  gf::Vec3 a2{0.1, -9.8, 0.2}; e.inputIMU(0.005, a2, w); const double lp = e.latest_P[0] + e.latest_V[1] + e.latest_Q[0] + e.latest_P_wheel[2] + e.latest_Q_wheel[4];\n'
                   '  g.inputGNSSTimeDiff(18.0);\n'
                   '  const bool lat = e.latest_time == 0.005 && e.latest_time_wheel == 0.0 && std::isfinite(lp) && e.rpw[0] == 1.0 && e.rpw[1] == 0.0 && e.zpw == 0.0 && g.diff_t_gnss_local == 18.0;\n'
                   '  return (t.MAX_CNT == 150 && e.frame_count == 0 && (int)e.Ps.size() == 11 && e.wheelxyztBuf.size() == 1 && threw && gn && lat) ? 0 : 1; }\n')
    exe = tmp_path / "t"
    lib = os.path.join(root, "ground-fusion_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-I", root, str(src), "-L", lib, "-lgroundfusion_hip", "-Wl,-rpath," + lib, "-o", str(exe)])
    assert subprocess.call([str(exe)]) == 0


def test_machine_without_rccl_gets_an_error_not_a_crash(tmp_path):
    """gf_comm_* resolve RCCL at run time; where no librccl can be loaded every entry point must return GF_ERR_NO_DEVICE with a message (round-4 advisor: the
    message was built from a second dlerror() call, which returns NULL -- std::string + nullptr -- and the advertised graceful path crashed).  Own process: the
    resolution happens once per process, and GF_RCCL_LIBRARY names the only library it may try."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, ctypes as C\n"
            "sys.path.insert(0, %r)\n"
            "import gfamd\n"
            "L = gfamd.lib()\n"
            "buf = (C.c_ubyte * 128)()\n"
            "rc = L.gf_comm_unique_id(buf)\n"
            "msg = L.gf_last_error().decode()\n"
            "h = C.c_void_p()\n"
            "rc2 = L.gf_comm_create(buf, 1, 0, 0, C.byref(h))\n"
            "print(rc, rc2, msg)\n") % os.path.join(root, "ground-fusion_amd")
    env = dict(os.environ, GF_RCCL_LIBRARY=str(tmp_path / "no_such_librccl.so"), GF_NO_TORCH_PRELOAD="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stderr[-500:]
    rc, rc2, msg = out.stdout.strip().split(" ", 2)
    assert int(rc) != 0 and int(rc) == int(rc2)
    assert "RCCL is not available" in msg and "no_such_librccl.so" in msg
