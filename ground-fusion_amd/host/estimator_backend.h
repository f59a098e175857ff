// estimator_backend.h — C++ host mirror of the optimisation part of Estimator (vins_estimator/src/estimator/estimator.h:262-356):
// the sliding-window state arrays, vector2double / double2vector (estimator.cpp:2276-2353, :2440-2569) and optimization()
// (:2890-3631) expressed on the C-ABI gf_ba_* of libgroundfusion_hip.so.  The factor tables (what the reference builds from
// f_manager / pre_integrations[] when it adds residual blocks) are handed over as a gf_ba_window by the caller; the feature
// manager, initialisation and slideWindow bookkeeping of Estimator::processImage stay with the caller.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/groundfusion_hip.h"

namespace gf {

class EstimatorBackend {
  public:
    enum MarginalizationFlag { MARGIN_OLD = 0, MARGIN_SECOND_NEW = 1 };
    const int WINDOW_SIZE;
    int NUM_ITERATIONS = 8;                       // max_num_iterations, m2dgrp.yaml:140
    // sliding-window state, row-major 3x3 rotations (estimator.h:267-275)
    std::vector<double> Ps, Vs, Rs, Bas, Bgs;
    // para_* arrays (estimator.h:335-341)
    std::vector<double> para_Pose, para_SpeedBias, para_Feature;
    double para_Ex_Pose[7] = {0, 0, 0, 0, 0, 0, 1}, para_Ex_Pose_wheel[7] = {0, 0, 0, 0, 0, 0, 1}, para_Ix[3] = {1, 1, 1}, para_Td[1] = {0}, para_Td_wheel[1] = {0};
    // last_marginalization_info (estimator.h:292-294) in C-ABI form
    bool prior_valid = false;
    std::vector<int> prior_block_id; std::vector<double> prior_J, prior_r, prior_x0; int prior_n = 0;

    EstimatorBackend(int window_size, int max_features, int max_visual) : WINDOW_SIZE(window_size) {
        const int NP = window_size + 1;
        Ps.assign(3 * NP, 0); Vs.assign(3 * NP, 0); Bas.assign(3 * NP, 0); Bgs.assign(3 * NP, 0); Rs.assign(9 * NP, 0);
        for (int i = 0; i < NP; i++) Rs[9 * i] = Rs[9 * i + 4] = Rs[9 * i + 8] = 1;
        para_Pose.assign(7 * NP, 0); para_SpeedBias.assign(9 * NP, 0); para_Feature.assign(max_features, 0);
        gf_ba_cfg c{window_size, max_features, max_visual, 1};
        check(gf_ba_create(&c, &h_));
        pJ_.resize(256 * 256); pr_.resize(256); px0_.resize(512); pid_.resize(64);
    }
    ~EstimatorBackend() { if (h_) gf_ba_destroy(h_); }
    EstimatorBackend(const EstimatorBackend&) = delete;

    void vector2double() {  // estimator.cpp:2276-2353
        for (int i = 0; i <= WINDOW_SIZE; i++) {
            double* p = &para_Pose[7 * i];
            p[0] = Ps[3 * i]; p[1] = Ps[3 * i + 1]; p[2] = Ps[3 * i + 2];
            rot_to_quat(&Rs[9 * i], p + 3);
            double* s = &para_SpeedBias[9 * i];
            for (int k = 0; k < 3; k++) { s[k] = Vs[3 * i + k]; s[3 + k] = Bas[3 * i + k]; s[6 + k] = Bgs[3 * i + k]; }
        }
    }
    void double2vector() {  // estimator.cpp:2440-2497 (gauge fix) — uses Rs[0], Ps[0] from before the solve
        const std::vector<double> R0(Rs.begin(), Rs.begin() + 9), P0(Ps.begin(), Ps.begin() + 3);
        check(gf_ba_double2vector(WINDOW_SIZE, R0.data(), P0.data(), para_Pose.data(), para_SpeedBias.data(), Rs.data(), Ps.data(), Vs.data(), Bas.data(), Bgs.data()));
    }
    // Estimator::optimization(): the caller fills every factor table / flag of `w`; the para_* pointers and the prior are wired here.
    gf_ba_summary optimization(gf_ba_window& w, MarginalizationFlag marginalization_flag) {
        vector2double();
        w.W = WINDOW_SIZE;
        w.para_Pose = para_Pose.data(); w.para_SpeedBias = para_SpeedBias.data(); w.para_Ex_Pose = para_Ex_Pose; w.para_Ex_Pose_wheel = para_Ex_Pose_wheel;
        w.para_Ix = para_Ix; w.para_Td = para_Td; w.para_Td_wheel = para_Td_wheel; w.para_Feature = para_Feature.data();
        if (prior_valid) { w.prior_n = prior_n; w.prior_nblocks = (int)prior_block_id.size(); w.prior_block_id = prior_block_id.data(); w.prior_J = prior_J.data(); w.prior_r = prior_r.data(); w.prior_x0 = prior_x0.data(); }
        else { w.prior_n = 0; w.prior_nblocks = 0; }
        gf_ba_summary s{};
        check(gf_ba_solve(h_, &w, 1, NUM_ITERATIONS, &s));           // estimator.cpp:3303-3318
        double2vector();                                               // :3327
        vector2double();                                               // :3337 / :3541
        gf_ba_prior p{};                                               // :3334-3631
        p.cap_n = 256; p.cap_blocks = 64; p.block_id = pid_.data(); p.J = pJ_.data(); p.r = pr_.data(); p.x0 = px0_.data();
        check(gf_ba_marginalize(h_, &w, 1, (int)marginalization_flag, &p));
        if (p.valid) {
            prior_valid = true; prior_n = p.n;
            prior_block_id.assign(pid_.begin(), pid_.begin() + p.nblocks);
            prior_J.assign(pJ_.begin(), pJ_.begin() + (size_t)p.n * p.n); prior_r.assign(pr_.begin(), pr_.begin() + p.n);
            int gs = 0;
            for (int id : prior_block_id) { const int k = id / 4096; gs += (k == GF_POSE || k == GF_EX_POSE || k == GF_EX_WHEEL) ? 7 : k == GF_SPEEDBIAS ? 9 : 1; }
            prior_x0.assign(px0_.begin(), px0_.begin() + gs);
        }
        return s;
    }

  private:
    gf_ba* h_ = nullptr;
    std::vector<double> pJ_, pr_, px0_; std::vector<int> pid_;
    static void check(int rc) { if (rc != GF_OK) throw std::runtime_error(std::string("groundfusion_hip: ") + gf_last_error()); }
    static void rot_to_quat(const double* R, double* q_xyzw) {  // Eigen::Quaterniond(Matrix3d)
        double t = R[0] + R[4] + R[8], w, x, y, z;
        if (t > 0) { t = std::sqrt(t + 1.0); w = 0.5 * t; t = 0.5 / t; x = (R[7] - R[5]) * t; y = (R[2] - R[6]) * t; z = (R[3] - R[1]) * t; }
        else {
            int i = 0; if (R[4] > R[0]) i = 1; if (R[8] > R[4 * i]) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            double v[3];
            t = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0); v[i] = 0.5 * t; t = 0.5 / t;
            w = (R[3 * k + j] - R[3 * j + k]) * t; v[j] = (R[3 * j + i] + R[3 * i + j]) * t; v[k] = (R[3 * k + i] + R[3 * i + k]) * t;
            x = v[0]; y = v[1]; z = v[2];
        }
        q_xyzw[0] = x; q_xyzw[1] = y; q_xyzw[2] = z; q_xyzw[3] = w;
    }
};

}  // namespace gf
