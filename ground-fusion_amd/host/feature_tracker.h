// feature_tracker.h — C++ host mirror of the reference's FeatureTracker call surface
// (vins_estimator/src/featureTracker/feature_tracker.h:43-99) on top of the C-ABI of libgroundfusion_hip.so.
//
// Same member names, argument meaning and return type as the reference, so that Estimator::inputImage
// (estimator.cpp:213-240) and the ROS plumbing compile against it unchanged once cv::Mat / Eigen are present:
//   * build with -DGF_WITH_OPENCV to take cv::Mat arguments directly (CV_8UC1 image, CV_16UC1 depth);
//   * build with -DGF_WITH_EIGEN to return Eigen::Matrix<double,8,1> observations;
//   * without them (this container has neither) the light gf::Image view and std::array<double,8> stand in.
// Errors: the reference logs and carries on; here a failing C-ABI call throws std::runtime_error with gf_last_error().
#pragma once
#include <array>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/groundfusion_hip.h"
#ifdef GF_WITH_OPENCV
#include <opencv2/core.hpp>
#endif
#ifdef GF_WITH_EIGEN
#include <eigen3/Eigen/Dense>
#endif

namespace gf {

template <class T> struct ImageView {  // stands in for cv::Mat when OpenCV is absent
    const T* data = nullptr; int rows = 0, cols = 0, stride = 0;  // stride in elements
    bool empty() const { return data == nullptr; }
};
typedef ImageView<uint8_t> GrayImage;
typedef ImageView<uint16_t> DepthImage;

#ifdef GF_WITH_EIGEN
typedef Eigen::Matrix<double, 8, 1> Obs8;
typedef Eigen::Vector3d Vec3;
#else
typedef std::array<double, 8> Obs8;
typedef std::array<double, 3> Vec3;
#endif
typedef std::map<int, std::vector<std::pair<int, Obs8>>> FeatureFrame;

class FeatureTracker {
  public:
    // globals of parameters.h the tracker reads: MAX_CNT, MIN_DIST, FLOW_BACK (config/realsense/m2dgrp.yaml:131-136)
    int MAX_CNT = 150, MIN_DIST = 30, FLOW_BACK = 1;
    int row = 0, col = 0;
    bool stereo_cam = false, depth_cam = false;
    std::vector<int> ids, track_cnt;          // refreshed after every trackImage (feature_tracker.h:85-86)
    std::vector<std::pair<float, float>> prev_pts;
    int n_id = 0;

    FeatureTracker() {}
    ~FeatureTracker() { if (h_) gf_tracker_destroy(h_); }
    FeatureTracker(const FeatureTracker&) = delete;
    FeatureTracker& operator=(const FeatureTracker&) = delete;

    // camodocal pinhole YAML (camera_models/src/camera_models/PinholeCamera.cc Parameters::readFromYamlFile); `depth` = depth camera flag
    void readIntrinsicParameter(const std::vector<std::string>& calib_file, const int depth) {
        if (calib_file.empty()) throw std::runtime_error("readIntrinsicParameter: no calibration file");
        std::ifstream f(calib_file[0]);
        if (!f) throw std::runtime_error("cannot open " + calib_file[0]);
        std::string line;
        while (std::getline(f, line)) {
            std::istringstream ss(line);
            std::string key;
            ss >> key;
            double v;
            if (!(ss >> v)) continue;
            if (key == "image_width:") col = (int)v; else if (key == "image_height:") row = (int)v;
            else if (key == "k1:") k1_ = v; else if (key == "k2:") k2_ = v; else if (key == "p1:") p1_ = v; else if (key == "p2:") p2_ = v;
            else if (key == "fx:") fx_ = v; else if (key == "fy:") fy_ = v; else if (key == "cx:") cx_ = v; else if (key == "cy:") cy_ = v;
        }
        depth_cam = depth != 0;
        if (calib_file.size() == 2) stereo_cam = true;
    }
    void setIntrinsics(int width, int height, double fx, double fy, double cx, double cy, double k1 = 0, double k2 = 0, double p1 = 0, double p2 = 0) {
        col = width; row = height; fx_ = fx; fy_ = fy; cx_ = cx; cy_ = cy; k1_ = k1; k2_ = k2; p1_ = p1; p2_ = p2;
    }

    FeatureFrame trackImage(double _cur_time, const GrayImage& _img, const DepthImage& _img1 = DepthImage()) {
        if (!h_) create(_img.cols, _img.rows);
        std::vector<gf_feature_obs> out((size_t)((MAX_CNT + 3) & ~3));
        int n = 0;
        check(gf_tracker_track(h_, 0, _cur_time, _img.data, _img.stride, _img1.empty() ? nullptr : _img1.data, _img1.stride, out.data(), (int)out.size(), &n));
        FeatureFrame featureFrame;
        for (int i = 0; i < n; i++) {
            Obs8 o;
            for (int k = 0; k < 8; k++) o[k] = out[i].v[k];
            featureFrame[out[i].id].emplace_back(out[i].camera_id, o);
        }
        refresh();
        return featureFrame;
    }
#ifdef GF_WITH_OPENCV
    FeatureFrame trackImage(double _cur_time, const cv::Mat& _img, const cv::Mat& _img1 = cv::Mat()) {
        GrayImage g{_img.ptr<uint8_t>(), _img.rows, _img.cols, (int)_img.step};
        DepthImage d;
        if (!_img1.empty()) d = DepthImage{_img1.ptr<uint16_t>(), _img1.rows, _img1.cols, (int)(_img1.step / 2)};
        return trackImage(_cur_time, g, d);
    }
#endif
    void setPrediction(std::map<int, Vec3>& predictPts) {  // feature_tracker.cpp:1006-1027
        std::vector<int> pid; std::vector<double> xyz;
        for (auto& kv : predictPts) { pid.push_back(kv.first); xyz.push_back(kv.second[0]); xyz.push_back(kv.second[1]); xyz.push_back(kv.second[2]); }
        check(gf_tracker_set_prediction(h_, 0, pid.data(), xyz.data(), (int)pid.size()));
    }
    void removeOutliers(std::set<int>& removePtsIds) {  // feature_tracker.cpp:1029-1045
        std::vector<int> v(removePtsIds.begin(), removePtsIds.end());
        check(gf_tracker_remove_outliers(h_, 0, v.data(), (int)v.size()));
        refresh();
    }

  private:
    gf_tracker* h_ = nullptr;
    double fx_ = 1, fy_ = 1, cx_ = 0, cy_ = 0, k1_ = 0, k2_ = 0, p1_ = 0, p2_ = 0;
    static void check(int rc) { if (rc != GF_OK) throw std::runtime_error(std::string("groundfusion_hip: ") + gf_last_error()); }
    void create(int w, int h) {
        gf_tracker_cfg c{};
        c.width = col ? col : w; c.height = row ? row : h; c.batch = 1; c.max_cnt = MAX_CNT; c.min_dist = MIN_DIST; c.flow_back = FLOW_BACK; c.depth_cam = depth_cam ? 1 : 0;
        c.fx = fx_; c.fy = fy_; c.cx = cx_; c.cy = cy_; c.k1 = k1_; c.k2 = k2_; c.p1 = p1_; c.p2 = p2_;
        row = c.height; col = c.width;
        check(gf_tracker_create(&c, &h_));
    }
    void refresh() {
        const int cap = (MAX_CNT + 3) & ~3;
        ids.assign(cap, 0); track_cnt.assign(cap, 0);
        std::vector<float> p(2 * cap);
        int n = 0;
        check(gf_tracker_get_state(h_, 0, ids.data(), track_cnt.data(), p.data(), cap, &n));
        ids.resize(n); track_cnt.resize(n); prev_pts.resize(n);
        for (int i = 0; i < n; i++) prev_pts[i] = {p[2 * i], p[2 * i + 1]};
        for (int id : ids) n_id = id + 1 > n_id ? id + 1 : n_id;
    }
};

}  // namespace gf
