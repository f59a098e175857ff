// estimator.h — C++ host mirror of the reference's Estimator call surface (vins_estimator/src/estimator/estimator.h:86-132, :262-356)
// on top of gf_estimator_* of libgroundfusion_hip.so.  The ROS callbacks of rosNodeTest.cpp keep calling
//   estimator.inputIMU(t, acc, gyr)      estimator.h:104        estimator.inputWheel(t, vel, gyr)   :106
//   estimator.inputImage(t, img, depth)  :105                   estimator.inputFeature(t, frame)    :108
// and the publishers of utility/visualization.cpp keep reading Ps, Rs, Vs, Bas, Bgs, tic, ric, tio, rio, sx, sy, sw, td, td_wheel,
// Headers, solver_flag, marginalization_flag, key_poses and (gnss_enable) gnss_ready, anc_ecef, ecef_pos, enu_pos, R_enu_local, para_rcv_dt, which are
// refreshed after every call.
// Build with -DGF_WITH_EIGEN / -DGF_WITH_OPENCV to get the Eigen / cv::Mat signatures (absent in this container: std::array / gf::ImageView).
// Errors: a failing C-ABI call throws std::runtime_error(gf_last_error()) instead of the reference's log-and-continue.
#pragma once
#include <array>
#include <cmath>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "feature_tracker.h"

namespace gf {

#ifdef GF_WITH_EIGEN
typedef Eigen::Matrix3d Mat3;
#else
typedef std::array<double, 9> Mat3;  // row-major
#endif

class Estimator {
  public:
    enum SolverFlag { INITIAL, NON_LINEAR };
    enum MarginalizationFlag { MARGIN_OLD = 0, MARGIN_SECOND_NEW = 1 };
    gf_estimator_cfg cfg;                         // what parameters.cpp reads from the YAML; defaults = config/realsense/m2dgrp.yaml
    SolverFlag solver_flag = INITIAL;
    MarginalizationFlag marginalization_flag = MARGIN_OLD;
    int frame_count = 0;
    std::vector<Vec3> Ps, Vs, Bas, Bgs;           // (WINDOW_SIZE + 1)
    std::vector<Mat3> Rs;
    std::vector<double> Headers;
    Vec3 tic[1], tio; Mat3 ric[1], rio;
    double td = 0, td_wheel = 0, sx = 1, sy = 1, sw = 1;
    bool systemstationary = false;
    std::vector<Vec3> key_poses;                  // estimator.h:290: the window's positions after a NON_LINEAR frame
    // GNSS members read by pubGnssResult (utility/visualization.cpp:454-545); refreshed when cfg.gnss_enable
    bool gnss_ready = false;
    Vec3 anc_ecef{}, ecef_pos{}, enu_pos{};
    Mat3 R_enu_local{};
    double yaw_enu_local = 0;
    std::vector<double> para_rcv_dt, para_rcv_ddt;   // (WINDOW_SIZE + 1) * 4, (WINDOW_SIZE + 1)
    std::vector<std::pair<double, Vec3>> wheelxyztBuf;   // estimator.h:204
    // IMU- / wheel-rate state for pubLatestOdometry / pubWheelLatestOdometry (estimator.h:239-242, :354-356; fastPredictIMU estimator.cpp:4014-4028,
    // fastPredictWheel :4079-4093, updateLatestStates :4141-4198): refreshed after every inputIMU / inputWheel and after every frame
    double latest_time = 0, latest_time_wheel = 0;
    Vec3 latest_P{}, latest_V{}, latest_P_wheel{}, latest_V_wheel{};
#ifdef GF_WITH_EIGEN
    Eigen::Quaterniond latest_Q{1, 0, 0, 0}, latest_Q_wheel{1, 0, 0, 0};
#else
    Mat3 latest_Q{{1, 0, 0, 0, 1, 0, 0, 0, 1}}, latest_Q_wheel{{1, 0, 0, 0, 1, 0, 0, 0, 1}};   // rotation matrices, row-major
#endif
    // plane parameters (estimator.h:216-217): only initPlane (USE_PLANE, estimator.cpp:1537-1554) and the plane factor write them; `plane: 1` is outside the
    // built path (gf_estimator_cfg_from_yaml refuses it), so they keep the values clearState gives them (estimator.cpp:123-124) for the publishers that read them
#ifdef GF_WITH_EIGEN
    Mat3 rpw = Mat3::Identity();
#else
    Mat3 rpw{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
#endif
    double zpw = 0;
    double diff_t_gnss_local = 0;                        // estimator.h:310: set by inputGNSSTimeDiff / cfg.gnss_local_time_diff

    Estimator() { gf_estimator_default_cfg(&cfg); }
    ~Estimator() { if (h_) gf_estimator_destroy(h_); }
    Estimator(const Estimator&) = delete;
    Estimator& operator=(const Estimator&) = delete;

    void setParameter() {                          // estimator.cpp:176-211: (re)create with the current cfg
        if (h_) { gf_estimator_destroy(h_); h_ = nullptr; }
        check(gf_estimator_create(&cfg, &h_));
        refresh();
    }
    void clearState() { setParameter(); }          // estimator.cpp:51-174
    // readParameters(config_file) (parameters.cpp:138-558) fills cfg, tracker part included; call setParameter() afterwards as main() does
    void readParameters(const std::string& config_file) { check(gf_estimator_cfg_from_yaml(config_file.c_str(), &cfg)); }
    // VINS_RESULT_PATH: pubOdometry's trajectory file (visualization.cpp:346-357), written by the library after every processed frame
    void setResultPath(const std::string& vio_txt) { need(); check(gf_estimator_set_result_path(h_, vio_txt.c_str())); }

    void inputIMU(double t, const Vec3& linearAcceleration, const Vec3& angularVelocity) {
        need();
        const double a[3] = {linearAcceleration[0], linearAcceleration[1], linearAcceleration[2]}, g[3] = {angularVelocity[0], angularVelocity[1], angularVelocity[2]};
        check(gf_estimator_input_imu(h_, t, a, g));
        refreshLatest();     // estimator.cpp:332-335: the caller publishes latest_P / latest_Q / latest_V right after
    }
    void inputWheel(double t, const Vec3& linearVelocity, const Vec3& angularVelocity) {
        need();
        const double v[3] = {linearVelocity[0], linearVelocity[1], linearVelocity[2]}, g[3] = {angularVelocity[0], angularVelocity[1], angularVelocity[2]};
        check(gf_estimator_input_wheel(h_, t, v, g));
        refreshLatest();     // estimator.cpp:363-366
    }
    // inputGNSS (estimator.h:99): one epoch of L1 observations with the satellite states their ephemerides give (gf_gnss_obs); inputGNSSTimeDiff (:100);
    // setGNSSAlignment: the result of GNSSVIInitializer, applied by the library where the reference runs GNSSVIAlign (estimator.cpp:1928-2043)
    gf_estimator* handle() { need(); return h_; }   // for the C entry points this class does not wrap (gf_estimator_get_gnss_state, _get_features, ...)
    void inputGNSS(double t, const std::vector<gf_gnss_obs>& meas) { need(); check(gf_estimator_input_gnss(h_, t, meas.data(), (int)meas.size())); }
    void inputGNSSRaw(double t, const std::vector<gf_gnss_raw_obs>& meas) { need(); check(gf_estimator_input_gnss_raw(h_, t, meas.data(), (int)meas.size())); }
    void inputEphem(const gf_gnss_ephem& eph) { need(); check(gf_estimator_input_ephem(h_, &eph)); }                 // estimator.h:97 (GPS / Galileo / BeiDou)
    void inputEphem(const gf_gnss_glo_ephem& geph) { need(); check(gf_estimator_input_glo_ephem(h_, &geph)); }      // (GLONASS)
    void inputGNSSTimeDiff(double t_diff) { need(); check(gf_estimator_input_gnss_time_diff(h_, t_diff)); diff_t_gnss_local = t_diff; }
    void inputIonoParams(double /*ts*/, const std::vector<double>& iono_params) { need(); if (iono_params.size() != 8) return; check(gf_estimator_input_iono_params(h_, iono_params.data())); }
    void setGNSSAlignment(const Vec3& anc_ecef, double yaw_enu_local, const double rcv_dt[4], double rcv_ddt) {
        need();
        const double a[3] = {anc_ecef[0], anc_ecef[1], anc_ecef[2]};
        check(gf_estimator_set_gnss_alignment(h_, a, yaw_enu_local, rcv_dt, rcv_ddt));
    }
    void inputrawodom(double t, const Vec3& wheel_xyz) { wheelxyztBuf.emplace_back(t, wheel_xyz); }   // estimator.h:116, estimator.cpp:388-395: queued, never consumed (as in the reference)
    void processImage(const FeatureFrame& image, double header) {   // estimator.h:110: the frame directly, without the measurement queues
        need();
        std::vector<gf_feature_obs> obs;
        for (auto& kv : image) {
            gf_feature_obs o;
            o.id = kv.first; o.camera_id = kv.second[0].first;
            for (int k = 0; k < 8; k++) o.v[k] = kv.second[0].second[k];
            obs.push_back(o);
        }
        check(gf_estimator_process_image(h_, header, obs.data(), (int)obs.size()));
        refresh();
    }
    void inputFeature(double t, const FeatureFrame& featureFrame) {   // + processMeasurements -> processImage
        need();
        std::vector<gf_feature_obs> obs;
        for (auto& kv : featureFrame) {
            gf_feature_obs o;
            o.id = kv.first; o.camera_id = kv.second[0].first;
            for (int k = 0; k < 8; k++) o.v[k] = kv.second[0].second[k];
            obs.push_back(o);
        }
        check(gf_estimator_input_feature(h_, t, obs.data(), (int)obs.size()));
        refresh();
    }
    // needs cfg.with_tracker = 1 and cfg.tracker filled before setParameter()
    void inputImage(double t, const GrayImage& _img, const DepthImage& _img1 = DepthImage()) {
        need();
        check(gf_estimator_input_image(h_, t, _img.data, _img.stride, _img1.empty() ? nullptr : _img1.data, _img1.stride, nullptr, 0, nullptr));
        refresh();
    }
#ifdef GF_WITH_OPENCV
    void inputImage(double t, const cv::Mat& _img, const cv::Mat& _img1 = cv::Mat()) {
        GrayImage g{_img.ptr<uint8_t>(), _img.rows, _img.cols, (int)_img.step};
        DepthImage d;
        if (!_img1.empty()) d = DepthImage{_img1.ptr<uint16_t>(), _img1.rows, _img1.cols, (int)(_img1.step / 2)};
        inputImage(t, g, d);
    }
#endif

  private:
    gf_estimator* h_ = nullptr;
    static void check(int rc) { if (rc != GF_OK) throw std::runtime_error(std::string("groundfusion_hip: ") + gf_last_error()); }
    void need() { if (!h_) setParameter(); }
    static Vec3 v3(const double* p) { Vec3 v; v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; return v; }
    static Mat3 m3(const double* p) {
        Mat3 m;
#ifdef GF_WITH_EIGEN
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) m(r, c) = p[3 * r + c];
#else
        for (int i = 0; i < 9; i++) m[i] = p[i];
#endif
        return m;
    }
    void refreshLatest() {
        double a[16], b[16];
        check(gf_estimator_get_latest(h_, a, b));
        latest_time = a[0]; latest_P = v3(a + 1); latest_V = v3(a + 13); latest_time_wheel = b[0]; latest_P_wheel = v3(b + 1); latest_V_wheel = v3(b + 13);
#ifdef GF_WITH_EIGEN
        latest_Q = Eigen::Quaterniond(m3(a + 4)); latest_Q_wheel = Eigen::Quaterniond(m3(b + 4));
#else
        latest_Q = m3(a + 4); latest_Q_wheel = m3(b + 4);
#endif
    }
    void refresh() {
        refreshLatest();
        if (cfg.gnss_enable && diff_t_gnss_local == 0) diff_t_gnss_local = cfg.gnss_local_time_diff;
        const int N = cfg.window_size + 1;
        std::vector<double> P(3 * N), R(9 * N), V(3 * N), Ba(3 * N), Bg(3 * N);
        Headers.assign(N, 0.0);
        int info[16]; double extr[32];
        check(gf_estimator_get_state(h_, P.data(), R.data(), V.data(), Ba.data(), Bg.data(), Headers.data(), info, extr));
        Ps.resize(N); Vs.resize(N); Bas.resize(N); Bgs.resize(N); Rs.resize(N);
        for (int i = 0; i < N; i++) { Ps[i] = v3(&P[3 * i]); Vs[i] = v3(&V[3 * i]); Bas[i] = v3(&Ba[3 * i]); Bgs[i] = v3(&Bg[3 * i]); Rs[i] = m3(&R[9 * i]); }
        frame_count = info[0]; solver_flag = info[1] ? NON_LINEAR : INITIAL; marginalization_flag = info[2] ? MARGIN_SECOND_NEW : MARGIN_OLD; systemstationary = info[6] != 0;
        tic[0] = v3(extr); ric[0] = m3(extr + 3); tio = v3(extr + 12); rio = m3(extr + 15);
        sx = extr[24]; sy = extr[25]; sw = extr[26]; td = extr[27]; td_wheel = extr[28];
        if (solver_flag == NON_LINEAR) key_poses = Ps;     // estimator.cpp:1153-1155 (what pubKeyPoses reads)
        if (cfg.gnss_enable) {                              // what pubGnssResult reads (visualization.cpp:454-545)
            int gi[8]; double anc[3], ecef[3], enu[3];
            para_rcv_dt.assign(4 * N, 0.0); para_rcv_ddt.assign(N, 0.0);
            check(gf_estimator_get_gnss_state(h_, gi, para_rcv_dt.data(), para_rcv_ddt.data(), &yaw_enu_local, anc, ecef, enu));
            gnss_ready = gi[0] != 0; anc_ecef = v3(anc); ecef_pos = v3(ecef); enu_pos = v3(enu);
            const double c = std::cos(yaw_enu_local), s_ = std::sin(yaw_enu_local);
            const double Rz[9] = {c, -s_, 0, s_, c, 0, 0, 0, 1};
            R_enu_local = m3(Rz);
        }
    }
};

}  // namespace gf
