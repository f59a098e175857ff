// gf_estimator.hip — host side of the sliding-window back end: the parts of Estimator (vins_estimator/src/estimator/estimator.cpp) and
// FeatureManager (feature_manager.cpp) that decide WHAT the optimiser sees, written on plain arrays.
// All dense numerics go through the HIP back end (gf_ba_*); nothing in this file evaluates factors or solves on the CPU.
//
//   reference                                                   here
//   Estimator::inputIMU / inputWheel / inputImage  EST:213-360   gf_estimator_input_imu / _input_wheel / _input_image / _input_feature
//   Estimator::processMeasurements                 EST:526-709   Estimator::processMeasurements
//   Estimator::processIMU / processWheel           EST:743-842   Estimator::processIMU / processWheel
//   Estimator::processImage                        EST:843-1163  Estimator::processImage
//   Estimator::initialStructure (stationary and wheel-activated shortcuts EST:1557-1682; SfM branch EST:1684-1847 with visualInitialAlign, numerics in gf_init_sfm.hpp)
//   Estimator::vector2double / double2vector       EST:2276-2353, :2440-2569
//   Estimator::optimization                        EST:2890-3636 (problem construction -> one gf_ba_window)
//   Estimator::slideWindow / slideWindowNew / Old  EST:3638-3837
//   Estimator::predictPtsInNextFrame, movingConsistencyCheckW, reprojectionError(3D)  EST:3862-4010
//   FeatureManager::*                              FM:43-110, :198-302, :669-934, :978-1010
// Unsupported switches are rejected at create time: ESTIMATE_EXTRINSIC==2, USE_LINE, USE_PLANE, USE_MOTION, STEREO, !USE_IMU.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <ctime>
#include <deque>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <sched.h>
#include <vector>
#include <climits>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <ucontext.h>
#include <sys/mman.h>

#include "../../include/groundfusion_hip.h"
#include "gf_comm.hpp"
#include "gf_dmath.hpp"
#include "gf_preint.hpp"
#include "gf_host_cpus.hpp"
#include "gf_init_sfm.hpp"

namespace gf { int set_err(int code, const char* fmt, ...); }
using namespace gfd;

namespace {

V3 arr3(const double* p) { return v3(p[0], p[1], p[2]); }
M3 arr9(const double* p) { M3 m; for (int i = 0; i < 9; i++) m.m[i] = p[i]; return m; }
double norm(V3 a) { return sqrt(sqn(a)); }
V3 normalized(V3 a) { return a / norm(a); }

// Utility::R2ypr / ypr2R in DEGREES (utility/utility.h:78-118)
V3 R2ypr(const M3& R) {
    const V3 n = v3(R.m[0], R.m[3], R.m[6]), o = v3(R.m[1], R.m[4], R.m[7]), a = v3(R.m[2], R.m[5], R.m[8]);
    const double y = atan2(n.y, n.x);
    const double p = atan2(-n.z, n.x * cos(y) + n.y * sin(y));
    const double r = atan2(a.x * sin(y) - a.y * cos(y), -o.x * sin(y) + o.y * cos(y));
    return v3(y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0);
}
M3 ypr2R(V3 ypr) {
    const double y = ypr.x / 180.0 * M_PI, p = ypr.y / 180.0 * M_PI, r = ypr.z / 180.0 * M_PI;
    M3 Rz = m3_zero(), Ry = m3_zero(), Rx = m3_zero();
    Rz.m[0] = cos(y); Rz.m[1] = -sin(y); Rz.m[3] = sin(y); Rz.m[4] = cos(y); Rz.m[8] = 1;
    Ry.m[0] = cos(p); Ry.m[2] = sin(p); Ry.m[4] = 1; Ry.m[6] = -sin(p); Ry.m[8] = cos(p);
    Rx.m[0] = 1; Rx.m[4] = cos(r); Rx.m[5] = -sin(r); Rx.m[7] = sin(r); Rx.m[8] = cos(r);
    return Rz * Ry * Rx;
}
// Utility::g2R (utility/utility.cpp:12-22) with Eigen's Quaterniond::FromTwoVectors (regular branch; the antiparallel branch needs g ≈ -z)
M3 g2R(V3 g) {
    const V3 v0 = normalized(g), v1 = v3(0, 0, 1);
    const double c = dot(v1, v0);
    M3 R0;
    if (c < -1.0 + 1e-12) R0 = m3_diag(1, -1, -1);  // rotation by pi about x: any axis orthogonal to z serves
    else {
        const V3 axis = cross(v0, v1);
        const double s = sqrt((1.0 + c) * 2.0), invs = 1.0 / s;
        R0 = qmat(Q4{s * 0.5, axis.x * invs, axis.y * invs, axis.z * invs});
    }
    const double yaw = R2ypr(R0).x;
    return ypr2R(v3(-yaw, 0, 0)) * R0;
}

// right singular vector of the smallest singular value of A (rows x 4), one-sided Jacobi (stands in for Eigen::JacobiSVD, FM:710)
void smallest_right_singular_vector(std::vector<double>& A, int rows, double out[4]) {
    double V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                double a = 0, b = 0, g = 0;
                for (int r = 0; r < rows; r++) { const double x = A[4 * r + p], y = A[4 * r + q]; a += x * x; b += y * y; g += x * y; }
                if (g == 0.0 || fabs(g) <= 1e-300) continue;
                const double lim = 1e-15 * sqrt(a * b);
                if (fabs(g) <= lim) continue;
                off = std::max(off, fabs(g) / sqrt(a * b));
                const double zeta = (b - a) / (2.0 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int r = 0; r < rows; r++) { const double x = A[4 * r + p], y = A[4 * r + q]; A[4 * r + p] = c * x - s * y; A[4 * r + q] = s * x + c * y; }
                for (int r = 0; r < 4; r++) { const double x = V[4 * r + p], y = V[4 * r + q]; V[4 * r + p] = c * x - s * y; V[4 * r + q] = s * x + c * y; }
            }
        if (off == 0.0) break;
    }
    int best = 0; double bn = -1;
    for (int c = 0; c < 4; c++) { double n = 0; for (int r = 0; r < rows; r++) n += A[4 * r + c] * A[4 * r + c]; if (bn < 0 || n < bn) { bn = n; best = c; } }
    for (int r = 0; r < 4; r++) out[r] = V[4 * r + best];
}

// ---------------------------------------------------------------- FeatureManager (feature_manager.h:30-80, :139-215)
struct FeaturePerFrame { V3 point; double uv[2], velocity[2], depth, cur_td; };
struct FeaturePerId {
    int feature_id, start_frame;
    std::vector<FeaturePerFrame> feature_per_frame;
    int used_num = 0; double estimated_depth = -1.0; int estimate_flag = 0, solve_flag = 0;
    FeaturePerId(int id, int sf) : feature_id(id), start_frame(sf) {}
    int endFrame() const { return start_frame + (int)feature_per_frame.size() - 1; }
};

struct FeatureManager {
    std::list<FeaturePerId> feature;
    bool ascending = true;   // `feature` is ascending in feature_id (see addFeatureCheckParallax)
    int last_track_num = 0, new_feature_num = 0, long_track_num = 0;
    double last_average_parallax = 0;
    int WINDOW_SIZE = 10; double FOCAL_LENGTH = 600, MIN_PARALLAX = 10.0 / 600, INIT_DEPTH = 5, depth_threshold = 3;

    int getFeatureCount() {  // FM:43-55
        int cnt = 0;
        for (auto& it : feature) { it.used_num = (int)it.feature_per_frame.size(); if (it.used_num >= 4) cnt++; }
        return cnt;
    }
    static double compensatedParallax2(const FeaturePerId& it, int frame_count) {  // FM:978-1010 (the compensation is the identity)
        const FeaturePerFrame& fi = it.feature_per_frame[frame_count - 2 - it.start_frame];
        const FeaturePerFrame& fj = it.feature_per_frame[frame_count - 1 - it.start_frame];
        const double u_j = fj.point.x, v_j = fj.point.y;
        const double dep_i = fi.point.z, u_i = fi.point.x / dep_i, v_i = fi.point.y / dep_i;
        const double du = u_i - u_j, dv = v_i - v_j;
        return std::max(0.0, sqrt(std::min(du * du + dv * dv, du * du + dv * dv)));
    }
    bool addFeatureCheckParallax(int frame_count, const gf_feature_obs* obs, int n, double td) {  // FM:57-116; `obs` sorted by id (std::map order)
        double parallax_sum = 0; int parallax_num = 0;
        last_track_num = 0; last_average_parallax = 0; new_feature_num = 0; long_track_num = 0;
        // The reference looks every observation up with find_if over the list (FM:68-72): O(tracks) per observation.  The list is in insertion order, and a tracker hands
        // out ascending ids (feature_tracker.cpp:85-93), so the list is normally ascending in feature_id like the frame (std::map order): one merge walk finds the same
        // elements.  `ascending` is given up -- for good, the find_if route takes over -- the moment an id is appended behind a larger one (callers with ids of their own).
        auto hint = feature.begin();
        long long prev_id = -(1LL << 40);
        for (int k = 0; k < n; k++) {
            const double* p = obs[k].v;
            FeaturePerFrame f{v3(p[0], p[1], p[2]), {p[3], p[4]}, {p[5], p[6]}, p[7], td};
            const int feature_id = obs[k].id;
            std::list<FeaturePerId>::iterator it;
            if (ascending && feature_id > prev_id) {
                while (hint != feature.end() && hint->feature_id < feature_id) ++hint;
                it = (hint != feature.end() && hint->feature_id == feature_id) ? hint : feature.end();
            } else it = std::find_if(feature.begin(), feature.end(), [feature_id](const FeaturePerId& x) { return x.feature_id == feature_id; });
            prev_id = feature_id;
            if (it == feature.end()) {
                if (!feature.empty() && feature.back().feature_id > feature_id) ascending = false;
                feature.emplace_back(feature_id, frame_count); feature.back().feature_per_frame.push_back(f); new_feature_num++;
            }
            else { it->feature_per_frame.push_back(f); last_track_num++; if (it->feature_per_frame.size() >= 4) long_track_num++; }
        }
        if (frame_count < 2 || last_track_num < 20 || long_track_num < 40 || new_feature_num > 0.5 * last_track_num) return true;
        for (auto& it : feature)
            if (it.start_frame <= frame_count - 2 && it.start_frame + (int)it.feature_per_frame.size() - 1 >= frame_count - 1) { parallax_sum += compensatedParallax2(it, frame_count); parallax_num++; }
        if (parallax_num == 0) return true;
        last_average_parallax = parallax_sum / parallax_num * FOCAL_LENGTH;
        return parallax_sum / parallax_num >= MIN_PARALLAX;
    }
    // FM:219-247: pairs (a, b) of depth-scaled observations, flat as 6 doubles
    std::vector<double> getCorrespondingWithDepth(int l, int r) const {
        std::vector<double> c;
        for (auto& it : feature)
            if (it.start_frame <= l && it.endFrame() >= r) {
                const FeaturePerFrame &fa = it.feature_per_frame[l - it.start_frame], &fb = it.feature_per_frame[r - it.start_frame];
                if (fa.depth < 0.1 || fa.depth > 10) continue;
                if (fb.depth < 0.1 || fb.depth > 10) continue;
                const V3 a = fa.point * fa.depth, b = fb.point * fb.depth;
                c.insert(c.end(), {a.x, a.y, a.z, b.x, b.y, b.z});
            }
        return c;
    }
    void setDepth(const double* x) {  // FM:249-267
        int idx = -1;
        for (auto& it : feature) {
            it.used_num = (int)it.feature_per_frame.size();
            if (it.used_num < 4) continue;
            it.estimated_depth = 1.0 / x[++idx];
            it.solve_flag = it.estimated_depth < 0 ? 2 : 1;
        }
    }
    void clearDepth() { for (auto& it : feature) it.estimated_depth = -1; }  // FM:280-284
    void removeFailures() { for (auto it = feature.begin(); it != feature.end();) it = it->solve_flag == 2 ? feature.erase(it) : std::next(it); }  // FM:269-278
    void getDepthVector(std::vector<double>& dep) {  // FM:286-302
        dep.clear();
        for (auto& it : feature) { it.used_num = (int)it.feature_per_frame.size(); if (it.used_num < 4) continue; dep.push_back(1.0 / it.estimated_depth); }
    }
    void triangulate(const V3* Ps, const M3* Rs, V3 tic, const M3& ric) {  // FM:669-724
        for (auto& it : feature) {
            if (it.estimated_depth > 0) continue;
            it.used_num = (int)it.feature_per_frame.size();
            if (it.used_num < 4) continue;
            const int imu_i = it.start_frame; int imu_j = imu_i - 1;
            std::vector<double> A(2 * it.feature_per_frame.size() * 4);
            int row = 0;
            const V3 t0 = Ps[imu_i] + Rs[imu_i] * tic; const M3 R0 = Rs[imu_i] * ric;
            for (auto& fr : it.feature_per_frame) {
                imu_j++;
                const V3 t1 = Ps[imu_j] + Rs[imu_j] * tic; const M3 R1 = Rs[imu_j] * ric;
                const V3 t = transpose(R0) * (t1 - t0); const M3 R = transpose(R0) * R1;
                const M3 Rt = transpose(R); const V3 pt = -(Rt * t);
                const double P[3][4] = {{Rt.m[0], Rt.m[1], Rt.m[2], pt.x}, {Rt.m[3], Rt.m[4], Rt.m[5], pt.y}, {Rt.m[6], Rt.m[7], Rt.m[8], pt.z}};
                const V3 f = normalized(fr.point);
                for (int c = 0; c < 4; c++) A[4 * row + c] = f.x * P[2][c] - f.z * P[0][c];
                row++;
                for (int c = 0; c < 4; c++) A[4 * row + c] = f.y * P[2][c] - f.z * P[1][c];
                row++;
            }
            double sv[4];
            smallest_right_singular_vector(A, row, sv);
            it.estimated_depth = sv[2] / sv[3];
            it.estimate_flag = 2;
            if (it.estimated_depth < 0.1) { it.estimated_depth = INIT_DEPTH; it.estimate_flag = 0; }
        }
    }
    void triangulateWithDepth(const V3* Ps, const M3* Rs, V3 tic, const M3& ric) {  // FM:726-799
        for (auto& it : feature) {
            it.used_num = (int)it.feature_per_frame.size();
            if (it.used_num < 4) continue;
            if (it.estimated_depth > 0) continue;
            const int s = it.start_frame, n = (int)it.feature_per_frame.size();
            double depth_sum = 0.0; size_t cnt = 0;
            const V3 tr = Ps[s] + Rs[s] * tic; const M3 Rr = Rs[s] * ric;
            for (int i = 0; i < n; i++) {
                const V3 t0 = Ps[s + i] + Rs[s + i] * tic; const M3 R0 = Rs[s + i] * ric;
                const double d = it.feature_per_frame[i].depth;
                if (d < 0.1 || d > depth_threshold) continue;
                const V3 point0 = it.feature_per_frame[i].point * d;
                const V3 t2r = transpose(Rr) * (t0 - tr); const M3 R2r = transpose(Rr) * R0;
                for (int j = 0; j < n; j++) {
                    if (i == j) continue;
                    const V3 t1 = Ps[s + j] + Rs[s + j] * tic; const M3 R1 = Rs[s + j] * ric;
                    const V3 t20 = transpose(R0) * (t1 - t0); const M3 R20 = transpose(R0) * R1;
                    const V3 pp = transpose(R20) * point0 - transpose(R20) * t20;
                    const double rx = it.feature_per_frame[j].point.x - pp.x / pp.z, ry = it.feature_per_frame[j].point.y - pp.y / pp.z;
                    if (sqrt(rx * rx + ry * ry) < 10.0 / 460) { const V3 pr = R2r * point0 + t2r; depth_sum += pr.z; cnt++; }
                }
            }
            if (cnt == 0) continue;
            it.estimated_depth = depth_sum / cnt;
            it.estimate_flag = 1;
            if (it.estimated_depth < 0.1) { it.estimated_depth = INIT_DEPTH; it.estimate_flag = 0; }
        }
    }
    void removeOutlier(const std::set<int>& idx) { for (auto it = feature.begin(); it != feature.end();) it = idx.count(it->feature_id) ? feature.erase(it) : std::next(it); }  // FM:801-816
    void removeBackShiftDepth(const M3& marg_R, V3 marg_P, const M3& new_R, V3 new_P) {  // FM:818-856
        for (auto it = feature.begin(); it != feature.end();) {
            auto cur = it++;
            if (cur->start_frame != 0) { cur->start_frame--; continue; }
            const V3 uv_i = cur->feature_per_frame[0].point;
            cur->feature_per_frame.erase(cur->feature_per_frame.begin());
            if (cur->feature_per_frame.size() < 2) { feature.erase(cur); continue; }
            const V3 pts_i = uv_i * cur->estimated_depth, w_pts_i = marg_R * pts_i + marg_P, pts_j = transpose(new_R) * (w_pts_i - new_P);
            cur->estimated_depth = pts_j.z > 0 ? pts_j.z : INIT_DEPTH;
        }
    }
    void removeBack() {  // FM:858-874
        for (auto it = feature.begin(); it != feature.end();) {
            auto cur = it++;
            if (cur->start_frame != 0) cur->start_frame--;
            else { cur->feature_per_frame.erase(cur->feature_per_frame.begin()); if (cur->feature_per_frame.empty()) feature.erase(cur); }
        }
    }
    void removeFront(int frame_count) {  // FM:914-934
        for (auto it = feature.begin(); it != feature.end();) {
            auto cur = it++;
            if (cur->start_frame == frame_count) cur->start_frame--;
            else {
                const int j = WINDOW_SIZE - 1 - cur->start_frame;
                if (cur->endFrame() < frame_count - 1) continue;
                cur->feature_per_frame.erase(cur->feature_per_frame.begin() + j);
                if (cur->feature_per_frame.empty()) feature.erase(cur);
            }
        }
    }
};

// ---------------------------------------------------------------- pre-integration holders (IntegrationBase / WheelIntegrationBase as sample buffers)
struct ImuPre {  // integration_base.h:22-62: linearized_acc/gyr, linearized_ba/bg, the three sample buffers; results evaluated on demand
    V3 acc0, gyr0, lin_ba, lin_bg;
    std::vector<double> dt, acc, gyr;
    double sum_dt = 0, delta_p[3] = {0, 0, 0}, delta_q[4] = {1, 0, 0, 0}, delta_v[3] = {0, 0, 0};
    std::vector<double> jacobian, covariance;
    bool dirty = true, restart = true;
    gf::ImuPreState st;   // state after st.n_done samples: appended samples are integrated on top (same operations as starting over)
    // round 6: the 3-vector / quaternion part on a track of its own (st_s; its J / P are unused).  checkimu reads delta_v / sum_dt of every frame of all_image_frame on every
    // image (estimator.cpp:2173-2216) and nothing else; evaluating those intervals in full was a second 15 x 15 pre-integration of every IMU sample, ~100 us of a member's
    // ~220 us of host time per frame.  The full evaluation, when somebody asks for the Jacobian / covariance (the initialisation, an IMU factor), runs as before.
    gf::ImuPreState st_s; bool s_dirty = true, s_restart = true;
    ImuPre(V3 a0, V3 g0, V3 ba, V3 bg) : acc0(a0), gyr0(g0), lin_ba(ba), lin_bg(bg), jacobian(225), covariance(225) {}
    void push_back(double t, V3 a, V3 g) { dt.push_back(t); acc.insert(acc.end(), {a.x, a.y, a.z}); gyr.insert(gyr.end(), {g.x, g.y, g.z}); dirty = true; s_dirty = true; }
    void repropagate(V3 ba, V3 bg) { lin_ba = ba; lin_bg = bg; dirty = true; restart = true; s_dirty = true; s_restart = true; }  // integration_base.h:51-62
    void adopt() {   // st was filled by the batched device kernel (bit-identical to the host loop over all samples)
        memcpy(delta_p, st.dp, 24); memcpy(delta_q, st.dq, 32); memcpy(delta_v, st.dv, 24);
        memcpy(jacobian.data(), st.J, 225 * 8); memcpy(covariance.data(), st.P, 225 * 8);
        sum_dt = st.sum_dt;
        dirty = false; restart = false; s_dirty = false;
    }
    void eval_state() {   // delta_p / delta_q / delta_v / sum_dt up to date; jacobian / covariance possibly not (dirty stays as it is)
        if (!dirty || !s_dirty) return;   // a clean full evaluation carries the same values
        const double a0[3] = {acc0.x, acc0.y, acc0.z}, g0[3] = {gyr0.x, gyr0.y, gyr0.z}, ba[3] = {lin_ba.x, lin_ba.y, lin_ba.z}, bg[3] = {lin_bg.x, lin_bg.y, lin_bg.z};
        if (s_restart || st_s.n_done > (int)dt.size()) { gf::imu_preint_reset(st_s, a0, g0); s_restart = false; }
        gf::imu_preint_state_range(st_s, ba, bg, dt.data(), acc.data(), gyr.data(), st_s.n_done, (int)dt.size());
        memcpy(delta_p, st_s.dp, 24); memcpy(delta_q, st_s.dq, 32); memcpy(delta_v, st_s.dv, 24);
        sum_dt = st_s.sum_dt;
        s_dirty = false;
    }
    int eval(const double* noise) {
        if (!dirty) return GF_OK;
        const double a0[3] = {acc0.x, acc0.y, acc0.z}, g0[3] = {gyr0.x, gyr0.y, gyr0.z}, ba[3] = {lin_ba.x, lin_ba.y, lin_ba.z}, bg[3] = {lin_bg.x, lin_bg.y, lin_bg.z};
        if (restart || st.n_done > (int)dt.size()) { gf::imu_preint_reset(st, a0, g0); restart = false; }
        gf::imu_preint_range(st, ba, bg, noise, dt.data(), acc.data(), gyr.data(), st.n_done, (int)dt.size());
        memcpy(delta_p, st.dp, 24); memcpy(delta_q, st.dq, 32); memcpy(delta_v, st.dv, 24);
        memcpy(jacobian.data(), st.J, 225 * 8); memcpy(covariance.data(), st.P, 225 * 8);
        sum_dt = st.sum_dt;
        dirty = false; s_dirty = false;
        return GF_OK;
    }
};
struct WheelPre {  // wheel_integration_base.h:23-60
    V3 vel0, gyr0; double lin[4];  // linearized sx, sy, sw, td
    std::vector<double> dt, vel, gyr;
    double sum_dt = 0, delta_p[3] = {0, 0, 0}, delta_q[4] = {1, 0, 0, 0}, jacobian[18], covariance[36];
    bool dirty = true;
    WheelPre(V3 v0, V3 g0, double sx, double sy, double sw, double td) : vel0(v0), gyr0(g0), lin{sx, sy, sw, td} {}
    void push_back(double t, V3 v, V3 g) { dt.push_back(t); vel.insert(vel.end(), {v.x, v.y, v.z}); gyr.insert(gyr.end(), {g.x, g.y, g.z}); dirty = true; }
    V3 vel_1() const { return dt.empty() ? vel0 : arr3(&vel[vel.size() - 3]); }
    V3 gyr_1() const { return dt.empty() ? gyr0 : arr3(&gyr[gyr.size() - 3]); }
    int eval(const double* noise) {
        if (!dirty) return GF_OK;
        const double v0[3] = {vel0.x, vel0.y, vel0.z}, g0[3] = {gyr0.x, gyr0.y, gyr0.z};
        const int rc = gf_wheel_preintegrate((int)dt.size(), dt.data(), vel.data(), gyr.data(), v0, g0, lin, noise, delta_p, delta_q, jacobian, covariance, &sum_dt);
        dirty = rc != GF_OK;
        return rc;
    }
};
struct ImageFrame {  // initial/initial_alignment.h:25-40
    M3 R = m3_identity(); V3 T = v3(0, 0, 0);
    std::vector<gf_feature_obs> points;   // the frame's observations in id order (std::map order), for the per-frame solvePnP of initialStructure
    bool is_key_frame = false;
    std::shared_ptr<ImuPre> pre_integration;
    std::shared_ptr<WheelPre> pre_integration_wheel;
    bool pre_deleted = false;  // the reference deletes the pointer of the oldest frame but keeps the map entry (EST:3722-3726)
};

}  // namespace

// ---------------------------------------------------------------- many sequences, one batched solver (not in the reference)
// Estimators of a group run their processImage on one host thread each; when one reaches ceres::Solve or the marginalisation it hands
// its window to this rendezvous and sleeps.  As soon as every member that is still busy with its frame is asleep here, all waiting
// windows go to the device in one gf_ba_solve / gf_ba_marginalize call of the shared handle -- the batched kernels see B windows per
// launch although each Estimator keeps the reference's single-sequence control flow.  A member outside a group step (a frame that
// waited for IMU data and is taken by a later inputIMU) is simply a batch of one.
// ---------------------------------------------------------------- broadcast ephemerides -> satellite state (gnss_comm eph2pos / geph2pos / eph2svdt / eph2vel,
// not vendored by the reference: restated from the published broadcast-orbit algorithms, RTKLIB ephemeris.c lineage)
namespace gnss_eph {
constexpr double kC = 2.99792458e8;
constexpr double MU_GPS = 3.9860050e14, MU_GAL = 3.986004418e14, MU_CMP = 3.986004418e14, OMGE_GPS = 7.2921151467e-5, OMGE_GAL = 7.2921151467e-5, OMGE_CMP = 7.292115e-5;
constexpr double MU_GLO = 3.9860044e14, J2_GLO = 1.0826257e-3, OMGE_GLO = 7.292115e-5, RE_GLO = 6378136.0, TSTEP = 60.0;
constexpr double SIN_5 = -0.0871557427476582, COS_5 = 0.9961946980917456;   // sin(-5 deg), cos(-5 deg): BeiDou GEO frame
inline double eph2svdt(double t, const gf_gnss_ephem& e) {   // eph2clk
    double tk = t - e.toc;
    for (int i = 0; i < 2; i++) tk -= e.af0 + e.af1 * tk + e.af2 * tk * tk;
    return e.af0 + e.af1 * tk + e.af2 * tk * tk;
}
inline V3 eph2pos(double t, const gf_gnss_ephem& e, double* svdt) {   // IS-GPS-200 table 20-IV; BeiDou GEO satellites in their inclined frame
    const double mu = e.sys == 2 ? MU_GAL : e.sys == 3 ? MU_CMP : MU_GPS, omge = e.sys == 2 ? OMGE_GAL : e.sys == 3 ? OMGE_CMP : OMGE_GPS;
    double tk = t - e.toe;
    const double M = e.M0 + (sqrt(mu / (e.A * e.A * e.A)) + e.delta_n) * tk;
    double E = M, Ek = 0;
    for (int n = 0; fabs(E - Ek) > 1e-13 && n < 30; n++) { Ek = E; E -= (E - e.e * sin(E) - M) / (1.0 - e.e * cos(E)); }
    const double sinE = sin(E), cosE = cos(E);
    double u = atan2(sqrt(1.0 - e.e * e.e) * sinE, cosE - e.e) + e.omg, r = e.A * (1.0 - e.e * cosE), i = e.i0 + e.i_dot * tk;
    const double sin2u = sin(2.0 * u), cos2u = cos(2.0 * u);
    u += e.cus * sin2u + e.cuc * cos2u; r += e.crs * sin2u + e.crc * cos2u; i += e.cis * sin2u + e.cic * cos2u;
    const double x = r * cos(u), y = r * sin(u), cosi = cos(i);
    V3 rs;
    if (e.sys == 3 && (e.prn <= 5 || e.prn >= 59)) {
        const double O = e.OMG0 + e.OMG_dot * tk - omge * e.toe_tow, sinO = sin(O), cosO = cos(O);
        const double xg = x * cosO - y * cosi * sinO, yg = x * sinO + y * cosi * cosO, zg = y * sin(i), sino = sin(omge * tk), coso = cos(omge * tk);
        rs = v3(xg * coso + yg * sino * COS_5 + zg * sino * SIN_5, -xg * sino + yg * coso * COS_5 + zg * coso * SIN_5, -yg * SIN_5 + zg * COS_5);
    } else {
        const double O = e.OMG0 + (e.OMG_dot - omge) * tk - omge * e.toe_tow, sinO = sin(O), cosO = cos(O);
        rs = v3(x * cosO - y * cosi * sinO, x * sinO + y * cosi * cosO, y * sin(i));
    }
    if (svdt) { tk = t - e.toc; *svdt = e.af0 + e.af1 * tk + e.af2 * tk * tk - 2.0 * sqrt(mu * e.A) * e.e * sinE / (kC * kC); }
    return rs;
}
inline void glo_deq(const double* x, double* xdot, const double* acc) {
    const double r2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2], r3 = r2 * sqrt(r2), omg2 = OMGE_GLO * OMGE_GLO;
    const double a = 1.5 * J2_GLO * MU_GLO * RE_GLO * RE_GLO / r2 / r3, b = 5.0 * x[2] * x[2] / r2, c = -MU_GLO / r3 - a * (1.0 - b);
    xdot[0] = x[3]; xdot[1] = x[4]; xdot[2] = x[5];
    xdot[3] = (c + omg2) * x[0] + 2.0 * OMGE_GLO * x[4] + acc[0];
    xdot[4] = (c + omg2) * x[1] - 2.0 * OMGE_GLO * x[3] + acc[1];
    xdot[5] = (c - 2.0 * a) * x[2] + acc[2];
}
inline void glorbit(double t, double* x, const double* acc) {   // one Runge-Kutta step
    double k1[6], k2[6], k3[6], k4[6], w[6];
    glo_deq(x, k1, acc); for (int i = 0; i < 6; i++) w[i] = x[i] + k1[i] * t / 2.0;
    glo_deq(w, k2, acc); for (int i = 0; i < 6; i++) w[i] = x[i] + k2[i] * t / 2.0;
    glo_deq(w, k3, acc); for (int i = 0; i < 6; i++) w[i] = x[i] + k3[i] * t;
    glo_deq(w, k4, acc);
    for (int i = 0; i < 6; i++) x[i] += (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]) * t / 6.0;
}
inline double geph2svdt(double t, const gf_gnss_glo_ephem& g) {   // geph2clk
    double tk = t - g.toe;
    for (int i = 0; i < 2; i++) tk -= -g.tau_n + g.gamma * tk;
    return -g.tau_n + g.gamma * tk;
}
inline V3 geph2pos(double t, const gf_gnss_glo_ephem& g, double* svdt) {
    double tk = t - g.toe;
    if (svdt) *svdt = -g.tau_n + g.gamma * tk;
    double x[6] = {g.pos[0], g.pos[1], g.pos[2], g.vel[0], g.vel[1], g.vel[2]};
    for (double tt = tk < 0.0 ? -TSTEP : TSTEP; fabs(tk) > 1e-9; tk -= tt) { if (fabs(tk) < TSTEP) tt = tk; glorbit(tt, x, g.acc); }
    return v3(x[0], x[1], x[2]);
}
// GnssPsrDoppFactor's constructor (gnss_psr_dopp_factor.cpp:3-47)
inline void sat_state(const gf_gnss_raw_obs& r, const gf_gnss_ephem* e, const gf_gnss_glo_ephem* g, gf_gnss_obs* o) {
    memset(o, 0, sizeof(*o));
    o->sat = r.sat; o->sys = r.sys; o->time = r.time; o->psr = r.psr; o->dopp = r.dopp; o->psr_std = r.psr_std; o->dopp_std = r.dopp_std; o->wavelength = kC / r.freq; o->tow = r.tow;
    double sv_tx = r.time - r.psr / kC, svdt, svddt, d1, d2;
    V3 p, p2;
    const double tt = 1e-3;   // eph2vel / geph2vel: difference quotient over 1 ms, position and clock alike
    if (g) {
        svdt = geph2svdt(sv_tx, *g); sv_tx -= svdt;
        p = geph2pos(sv_tx, *g, &d1); p2 = geph2pos(sv_tx + tt, *g, &d2);
        o->tgd = 0.0; o->pr_uura = 2.0 * (r.psr_std / 0.16); o->dp_uura = 2.0 * (r.dopp_std / 0.256);
    } else {
        svdt = eph2svdt(sv_tx, *e); sv_tx -= svdt;
        p = eph2pos(sv_tx, *e, &d1); p2 = eph2pos(sv_tx + tt, *e, &d2);
        o->tgd = e->tgd0;
        const double k = e->sys == 2 ? e->ura - 2.0 : e->ura - 1.0;
        o->pr_uura = k * (r.psr_std / 0.16); o->dp_uura = k * (r.dopp_std / 0.256);
    }
    svdt = d1; svddt = (d2 - d1) / tt;
    const V3 v = (p2 - p) / tt;
    o->sv_pos[0] = p.x; o->sv_pos[1] = p.y; o->sv_pos[2] = p.z; o->sv_vel[0] = v.x; o->sv_vel[1] = v.y; o->sv_vel[2] = v.z; o->svdt = svdt; o->svddt = svddt;
}
}  // namespace gnss_eph

// A generation counter many threads sleep on (futex): bump() wakes them all at once and none of them has to take a lock to find out why it woke.
// (A condition variable makes 256 sleepers queue up on its mutex one after the other -- milliseconds per rendezvous at this group size.)
struct Gate {
    std::atomic<int> gen{0};
    int now() const { return gen.load(std::memory_order_acquire); }
    // GF_GROUP_SPIN_US (default 300): look at the counter for that long before going to sleep -- most waits of a group step are shorter than a sleep and a wake-up
    // of 128 threads.  Measured end to end, 256 members on 128 workers: 0 us 46.9 k window-solves/s (first..last request of the solve rendezvous 1.2 ms),
    // 300 us 48.8 k (0.77 ms), 3000 us 16 k (the spinning workers take the cores the tracker's pool and the batch's own thread need).
    // Round 6, with nothing taken out of the end-to-end clock any more (bench.py): spinning workers cost the caller's tracker thread and its bookkeeping pool more than they
    // save (two groups of 128 members, 256 hardware threads: 16 workers each 40.8 k window-solves/s at 300 us against 45.4 k at 0; 32: 32.3 k / 44.6 k; 64: 21.5 k / 28.3 k;
    // profiles/r06_e2e_pools.txt).  Default 0: straight to the futex.
    static int spin_us() { static const int v = [] { const char* e = getenv("GF_GROUP_SPIN_US"); return e ? atoi(e) : 0; }(); return v; }
    void wait_while(int seen) {   // returns at once if gen != seen
        if (const int us = spin_us()) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int k = 0;; k++) {
                if (gen.load(std::memory_order_acquire) != seen) return;
                __builtin_ia32_pause();
                if ((k & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(us)) break;
            }
        }
        syscall(SYS_futex, reinterpret_cast<int*>(&gen), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
    }
    void bump() { gen.fetch_add(1, std::memory_order_acq_rel); syscall(SYS_futex, reinterpret_cast<int*>(&gen), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0); }
};

// Members of a group run as user-level contexts (fibers) on a pool of worker threads: a member that has to wait for a batch hands its thread to the next
// member of the same thread instead of putting an operating-system thread to sleep.  The pool has as many threads as the caller allows (GF_GROUP_THREADS; default
// min(members, hardware threads)): 8 ranks x 256 members on one node are 8 x cores/8 threads, not 2 048.  A fiber always runs on the thread that owns it.
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;   // mmap'ed: pages are touched on use only, the lowest page is a guard (an overflow faults instead of running into a neighbour)
    int state = 0;           // 0 idle, 1 running / runnable, 2 waiting for a batch, 3 frame finished
    static constexpr size_t kGuard = 4096;
    // GF_GROUP_STACK_KB (default 8 MB, what a worker thread of its own would have): MAP_NORESERVE pages cost nothing until a frame touches them
    static size_t stack_bytes() {
        static const size_t n = [] { const char* e = getenv("GF_GROUP_STACK_KB"); long kb = e ? atol(e) : 8192; if (kb < 256) kb = 256; return ((size_t)kb << 10) & ~(size_t)4095; }();
        return n;
    }
    size_t kStack = 0;
    bool alloc() {
        if (stack) return true;
        kStack = stack_bytes();
        void* p = mmap(nullptr, kStack + kGuard, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) return false;
        (void)mprotect(p, kGuard, PROT_NONE);
        stack = static_cast<char*>(p);
        return true;
    }
    ~Fiber() { if (stack) munmap(stack, kStack + kGuard); }
    Fiber() = default;
    Fiber(const Fiber&) = delete;
    Fiber& operator=(const Fiber&) = delete;
};
static thread_local ucontext_t* tl_sched = nullptr;   // the worker's scheduler context while a fiber runs
static thread_local Fiber* tl_fiber = nullptr;
static inline void fiber_yield() { Fiber* f = tl_fiber; f->state = 2; swapcontext(&f->ctx, tl_sched); }

// One member's window as the batched feature sweeps take it (gf_triangulate_with_depth_batch / gf_moving_consistency_batch): the member fills it from its
// FeatureManager, the rendezvous concatenates the members' tables, launches once and scatters the results back.
struct SweepIn {
    int W = 0, nf = 0;
    std::vector<double> Rs, Ps;          // (W + 1) x 9, (W + 1) x 3
    double tic[3], ric[9];
    std::vector<int> start_frame, first_obs, flag, remove;   // per feature; first_obs has nf + 1 entries, relative to this member
    std::vector<double> obs, depth;      // 4 per observation (x, y, z, depth-camera depth); estimated depth per feature (in / out)
    double depth_threshold = 0, init_depth = 0, focal_length = 0;
};

struct BatchSolver {
    struct Req { int kind; gf_ba_window* w; int iters, mode; gf_ba_summary* sum; gf_ba_prior* prior; int rc; std::atomic<bool> done; std::string err;
                 std::vector<ImuPre*> pres; const double* noise; int slot = -1; SweepIn* sweep = nullptr; };
    // kind 0 solve, 1 marginalise, 2 IMU pre-integrations of this frame, 3 a per-feature sweep of this member's window (mode 0 triangulateWithDepth, 1 movingConsistencyCheckW;
    // SURVEY.md 8(f)4)
    // slot: the member's own slot of the shared handle.  A solve request arrives with its window already packed into that slot by the member's thread
    // (gf_ba_pack_slot), and the member unpacks its result itself: the batching thread only uploads, launches and downloads.
    gf_ba* ba = nullptr;
    gf::PreintBatch* pre = nullptr;   // GF_GROUP_DEVICE_PREINT=1: the members' pending pre-integrations run as one launch per camera frame
    gf_featsweep* sweeps = nullptr;   // GF_GROUP_DEVICE_SWEEPS=1: triangulateWithDepth / movingConsistencyCheckW of the members as one launch each per camera frame
    long long sweep_batches = 0, sweep_features = 0; double t_sweep = 0;
    double t_pre = 0; long long pre_batches = 0, pre_intervals = 0;
    std::mutex m;
    Gate finished;                  // bumped after every batch
    int active = 0;                 // members currently inside a frame of a group step
    std::vector<Req*> pending;
    long long batches = 0, windows = 0, largest = 0;
    std::vector<gf_ba_window*> resident;   // per slot: the member's window its last solve left on the device (the marginalisation reuses the device copy)
    size_t mem_count = 0;
    double t_solve = 0, t_marg = 0;        // wall time inside the batched calls [s] (GF_GROUP_TIMING=1 prints them when the group is destroyed)
    // wall-clock anatomy of a group step (GF_GROUP_TIMING=1): the rendezvous of a step in order -- from the step's start to the first request of a rendezvous
    // (the fastest member's host phase), from the first to the last request (the slowest member: what the batch waits for), inside the batch
    std::chrono::steady_clock::time_point t_mark, t_first;
    double w_to_first[4] = {0, 0, 0, 0}, w_spread[4] = {0, 0, 0, 0}, w_batch[4] = {0, 0, 0, 0}; long long w_n[4] = {0, 0, 0, 0};
    int rdv = 0;                           // index of the rendezvous inside the current step

    int submit(Req& r) {
        {
            std::unique_lock<std::mutex> lk(m);
            if (pending.empty()) t_first = std::chrono::steady_clock::now();
            pending.push_back(&r);
            maybe_run();
        }
        for (;;) {
            const int seen = finished.now();
            if (r.done.load(std::memory_order_acquire)) break;
            if (tl_fiber) fiber_yield();   // the worker's scheduler resumes this member when a batch has been closed (or just to look again)
            else finished.wait_while(seen);
        }
        if (r.rc != GF_OK) gf::set_err(r.rc, "%s", r.err.c_str());   // the batch may have run on another thread: carry its message over
        return r.rc;
    }
    void leave() { std::unique_lock<std::mutex> lk(m); active--; maybe_run(); }
    void leave_with_error() {}   // a member that fails before its request ends its frame with the error; the worker's leave() takes it out of the rendezvous
    // with the lock held: run everything that waits once nobody is left computing on the host
    void maybe_run() {
        if (pending.empty() || (int)pending.size() < active) return;
        std::vector<Req*> reqs;
        reqs.swap(pending);
        const auto t_close = std::chrono::steady_clock::now();
        // Partition by a snapshot of kind / mode BEFORE anything runs, and publish `done` only after the last pass: a submitter that sees done == true
        // returns and destroys its stack-allocated Req, so no request may be looked at again once any flag of this batch is up.
        std::vector<Req*> groups[4];   // pre-integrations, solves, MARGIN_OLD, MARGIN_SECOND_NEW
        std::vector<Req*> sweep_grp[2];   // triangulateWithDepth, movingConsistencyCheckW
        for (Req* r : reqs) {
            if (r->kind == 3) sweep_grp[r->mode != 0].push_back(r);
            else groups[r->kind == 2 ? 0 : r->kind == 0 ? 1 : 2 + (r->mode != 0)].push_back(r);
        }
        for (int mode = 0; mode < 2; mode++) {   // the feature sweeps: every member's table behind the other, one launch, results back into the members' tables
            const std::vector<Req*>& grp = sweep_grp[mode];
            if (grp.empty()) continue;
            const auto tc0 = std::chrono::steady_clock::now();
            const int B = (int)grp.size(), W = grp[0]->sweep->W;
            std::vector<double> Rs, Ps, tic, ric, obs, depth;
            std::vector<int> first_feature(1, 0), start_frame, first_obs(1, 0), flag;
            for (Req* r : grp) {
                const SweepIn& in = *r->sweep;
                Rs.insert(Rs.end(), in.Rs.begin(), in.Rs.end()); Ps.insert(Ps.end(), in.Ps.begin(), in.Ps.end());
                tic.insert(tic.end(), in.tic, in.tic + 3); ric.insert(ric.end(), in.ric, in.ric + 9);
                const int o0 = first_obs.back();
                for (int f = 0; f < in.nf; f++) first_obs.push_back(o0 + in.first_obs[f + 1]);
                start_frame.insert(start_frame.end(), in.start_frame.begin(), in.start_frame.end());
                obs.insert(obs.end(), in.obs.begin(), in.obs.end()); depth.insert(depth.end(), in.depth.begin(), in.depth.end());
                flag.insert(flag.end(), in.flag.begin(), in.flag.end());
                first_feature.push_back(first_feature.back() + in.nf);
            }
            const int F = first_feature.back();
            std::vector<int> remove(std::max(F, 1), 0);
            int rc = GF_OK;
            if (F > 0) {
                const SweepIn& c0 = *grp[0]->sweep;   // members of a group share their configuration
                if (mode == 0) rc = gf_triangulate_with_depth_batch(sweeps, B, W, Rs.data(), Ps.data(), tic.data(), ric.data(), first_feature.data(), start_frame.data(), first_obs.data(),
                                                                    obs.data(), c0.depth_threshold, c0.init_depth, depth.data(), flag.data());
                else rc = gf_moving_consistency_batch(sweeps, B, W, Rs.data(), Ps.data(), tic.data(), ric.data(), first_feature.data(), start_frame.data(), first_obs.data(), obs.data(),
                                                      depth.data(), c0.focal_length, remove.data());
            }
            const std::string err = rc == GF_OK ? std::string() : std::string(gf_last_error());
            for (int b = 0; b < B; b++) {
                Req* r = grp[b];
                SweepIn& in = *r->sweep;
                if (rc == GF_OK) {
                    const int f0 = first_feature[b];
                    if (mode == 0) { std::copy(depth.begin() + f0, depth.begin() + f0 + in.nf, in.depth.begin()); std::copy(flag.begin() + f0, flag.begin() + f0 + in.nf, in.flag.begin()); }
                    else std::copy(remove.begin() + f0, remove.begin() + f0 + in.nf, in.remove.begin());
                }
                r->rc = rc; r->err = err;
            }
            t_sweep += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count();
            sweep_batches++; sweep_features += F;
        }
        {   // pre-integrations first: nothing else of the same step can be pending next to them
            const std::vector<Req*>& grp = groups[0];
            if (!grp.empty()) {
                const auto tc0 = std::chrono::steady_clock::now();
                std::vector<gf::PreintJob> jobs;
                for (Req* r : grp)
                    for (ImuPre* p : r->pres) {
                        gf::PreintJob j{&p->st, &p->lin_ba.x, &p->lin_bg.x, p->dt.data(), p->acc.data(), p->gyr.data(), (int)p->dt.size(), {p->acc0.x, p->acc0.y, p->acc0.z}, {p->gyr0.x, p->gyr0.y, p->gyr0.z}};
                        jobs.push_back(j);
                    }
                const int rc = gf::preint_batch_run(pre, jobs, grp[0]->noise);
                if (rc == GF_OK) for (Req* r : grp) for (ImuPre* p : r->pres) p->adopt();
                const std::string err = rc == GF_OK ? std::string() : std::string(gf_last_error());
                for (Req* r : grp) { r->rc = rc; r->err = err; }
                t_pre += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count();
                pre_batches++; pre_intervals += (long long)jobs.size();
            }
        }
        for (int pass = 0; pass < 3; pass++) {   // solves, MARGIN_OLD, MARGIN_SECOND_NEW
            const std::vector<Req*>& grp = groups[1 + pass];
            if (grp.empty()) continue;
            // the shared handle takes its iteration count per call: group by it (members of one group share a configuration)
            int rc;
            const auto tc0 = std::chrono::steady_clock::now();
            std::vector<int> slots(grp.size());
            for (size_t i = 0; i < grp.size(); i++) slots[i] = grp[i]->slot;
            if (pass == 0) {
                rc = gf_ba_solve_packed(ba, slots.data(), (int)grp.size(), grp[0]->iters);
                if (resident.size() != mem_count) resident.assign(mem_count, nullptr);
                if (rc == GF_OK) for (Req* r : grp) resident[r->slot] = r->w;   // this member's window is what its slot holds now
            } else {
                std::vector<gf_ba_window> wins(grp.size());
                for (size_t i = 0; i < grp.size(); i++) wins[i] = *grp[i]->w;
                // the windows of this step's solve are still on the device: only their states (double2vector's gauge fix sits in between) go up again;
                // every member then copies its own prior out of the handle's host mirrors (gf_ba_unpack_prior_slot on its own thread)
                bool all = resident.size() == mem_count;
                for (size_t i = 0; i < grp.size() && all; i++) all = resident[grp[i]->slot] == grp[i]->w;
                if (all) rc = gf_ba_marginalize_resident(ba, slots.data(), wins.data(), (int)grp.size(), pass - 1, nullptr);
                else rc = gf::set_err(GF_ERR_INVALID, "marginalisation of a window that is not resident in its member's slot");
            }
            (pass == 0 ? t_solve : t_marg) += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc0).count();
            const std::string err = rc == GF_OK ? std::string() : std::string(gf_last_error());
            for (Req* r : grp) { r->rc = rc; r->err = err; }
            batches++; windows += (long long)grp.size(); largest = std::max(largest, (long long)grp.size());
        }
        {
            const auto t_end = std::chrono::steady_clock::now();
            const int q = std::min(rdv, 3);
            w_to_first[q] += std::chrono::duration<double>(t_first - t_mark).count(); w_spread[q] += std::chrono::duration<double>(t_close - t_first).count();
            w_batch[q] += std::chrono::duration<double>(t_end - t_close).count(); w_n[q]++;
            t_mark = t_end; rdv++;
        }
        for (Req* r : reqs) r->done.store(true, std::memory_order_release);   // last touch of every request
        finished.bump();
    }
};

// ---------------------------------------------------------------- Estimator
struct gf_estimator {
    gf_estimator_cfg cfg;
    int WINDOW_SIZE;
    enum { INITIAL = 0, NON_LINEAR = 1 };
    enum { MARGIN_OLD = 0, MARGIN_SECOND_NEW = 1 };
    FeatureManager f_manager;
    gf_ba* ba = nullptr;
    BatchSolver* group = nullptr;      // member of a gf_estimator_group: solves go through the group's shared handle
    int group_slot = -1;               // ... in this slot of it
    gf_tracker* tracker = nullptr;
    // measurement queues (estimator.h:182-190)
    std::deque<std::pair<double, V3>> accBuf, gyrBuf, wheelVelBuf, wheelGyrBuf;
    std::deque<std::pair<double, std::vector<gf_feature_obs>>> featureBuf;
    double prevTime = -1, curTime = 0, prevTime_wheel = -1, curTime_wheel = 0;
    int inputImageCnt = 0;
    // window state (estimator.h:262-300)
    std::vector<V3> Ps, Vs, Bas, Bgs; std::vector<M3> Rs; std::vector<double> Headers;
    V3 tic = v3(0, 0, 0), tio = v3(0, 0, 0), g = v3(0, 0, 9.805); M3 ric = m3_identity(), rio = m3_identity();
    double td = 0, td_wheel = 0, sx = 1, sy = 1, sw = 1;
    int frame_count = 0, solver_flag = INITIAL, marginalization_flag = MARGIN_OLD, sum_of_back = 0, sum_of_front = 0;
    bool first_imu = false, first_wheel = false, initFirstPoseFlag = false;
    V3 acc_0 = v3(0, 0, 0), gyr_0 = v3(0, 0, 0), vel_0_wheel = v3(0, 0, 0), gyr_0_wheel = v3(0, 0, 0), latest_vel_wheel_0 = v3(0, 0, 0);
    // IMU- / wheel-rate propagation of the newest state (estimator.h:239-242, :354-356): what pubLatestOdometry / pubWheelLatestOdometry publish.
    // Uninitialised in the reference until the first updateLatestStates (UB): zero / identity here.  Rotations are kept as matrices.
    double latest_time = 0, latest_time_wheel = 0, latest_sx = 1, latest_sy = 1, latest_sw = 1;
    V3 latest_P = v3(0, 0, 0), latest_V = v3(0, 0, 0), latest_Ba = v3(0, 0, 0), latest_Bg = v3(0, 0, 0), latest_acc_0 = v3(0, 0, 0), latest_gyr_0 = v3(0, 0, 0);
    V3 latest_P_wheel = v3(0, 0, 0), latest_V_wheel = v3(0, 0, 0), latest_gyr_wheel_0 = v3(0, 0, 0);
    M3 latest_Q = m3_identity(), latest_Q_wheel = m3_identity();
    std::vector<std::shared_ptr<ImuPre>> pre_integrations; std::vector<std::shared_ptr<WheelPre>> pre_integrations_wheel;
    std::shared_ptr<ImuPre> tmp_pre_integration; std::shared_ptr<WheelPre> tmp_wheel_pre_integration;
    std::map<double, ImageFrame> all_image_frame;
    double initial_timestamp = 0;
    // stationarity / anomaly votes (estimator.h, EST:26-35)
    bool wheelanomaly = false, visualstationary = false, wheelstationary = false, imustationary = false, systemstationary = false, varstationary = false,
         preintegrationstationary = false, is_imu_excited = false, Bas_calibok = false;
    std::string result_path;   // VINS_RESULT_PATH ("" = no trajectory file)
    V3 dP_imu = v3(0, 0, 0), dP_wheel = v3(0, 0, 0);
    int openExEstimation = 0, openExWheelEstimation = 0, openIxEstimation = 0;
    M3 back_R0 = m3_identity(), last_R = m3_identity(), last_R0 = m3_identity(); V3 back_P0 = v3(0, 0, 0), last_P = v3(0, 0, 0), last_P0 = v3(0, 0, 0);
    // para_* (estimator.h:335-341)
    std::vector<double> para_Pose, para_SpeedBias, para_Feature;
    double para_Ex_Pose[7], para_Ex_Pose_wheel[7], para_Ix[3], para_Td[1], para_Td_wheel[1];
    // last_marginalization_info in C-ABI form
    bool prior_valid = false, prior_resident = false; int prior_n = 0;   // prior_resident: J lives in the group's shared handle (slot group_slot), prior_J is empty
    std::vector<int> prior_block_id; std::vector<double> prior_J, prior_r, prior_x0;
    // feedback to the tracker (EST:1132-1136)
    std::vector<int> predict_ids, remove_ids; std::vector<double> predict_xyz;
    struct WinScratch {
        std::vector<int> imu_i, wh_i, vf, vi, vj;
        std::vector<double> imu_sum_dt, imu_dp, imu_dq, imu_dv, imu_ba, imu_bg, imu_J, imu_P, wh_sum_dt, wh_dp, wh_dq, wh_J, wh_P, wh_lin, wh_lv, wh_lg, wh_v1, wh_g1, vpi, vpj, vvi, vvj, vti, vtj;
        std::vector<unsigned char> fixed;
        void clear() {
            for (auto* v : {&imu_i, &wh_i, &vf, &vi, &vj}) v->clear();
            for (auto* v : {&imu_sum_dt, &imu_dp, &imu_dq, &imu_dv, &imu_ba, &imu_bg, &imu_J, &imu_P, &wh_sum_dt, &wh_dp, &wh_dq, &wh_J, &wh_P, &wh_lin, &wh_lv, &wh_lg, &wh_v1, &wh_g1, &vpi, &vpj, &vvi, &vvj, &vti, &vtj}) v->clear();
            fixed.clear();
        }
    } scratch;
    // GNSS (estimator.h:293-331): epoch queue, per-frame measurement buffers, receiver clock / anchor / yaw states
    struct GMsgObs { gf_gnss_obs o; gf_gnss_raw_obs raw; bool is_raw; };   // an observation of a queued message: with its satellite state, or raw (ephemeris resolved in processGNSS)
    std::deque<std::pair<double, std::vector<GMsgObs>>> GNSSBuf;
    std::vector<GMsgObs> gnss_msg;                           // member of the reference too: the last epoch taken stays until the next one (EST:503-508, :656)
    // inputEphem (EST:1428-1437): per satellite the ephemerides in arrival order and toe -> index; GLONASS in its own table
    std::map<int, std::vector<gf_gnss_ephem>> sat2ephem; std::map<int, std::vector<gf_gnss_glo_ephem>> sat2gephem;
    std::map<int, std::map<double, size_t>> sat2time_index;
    std::vector<std::vector<gf_gnss_obs>> gnss_meas_buf;     // [WINDOW_SIZE + 1]
    std::map<int, int> sat_track_status;
    bool gnss_ready = false, first_optimization = true, lowspeed = false, align_pending = false;
    double diff_t_gnss_local = 0, yaw_enu_local = 0;
    double align_anc[3] = {0, 0, 0}, align_yaw = 0, align_dt[4] = {0, 0, 0, 0}, align_ddt = 0;
    V3 anc_ecef = v3(0, 0, 0), ecef_pos = v3(0, 0, 0), enu_pos = v3(0, 0, 0); M3 R_ecef_enu = m3_identity();
    std::vector<double> para_rcv_dt, para_rcv_ddt, gnss_iono; double para_yaw_enu_local[1] = {0}, para_anc_ecef[3] = {0, 0, 0};
    std::vector<int> gn_frame, gn_lower, gn_sys; std::vector<double> gn_ratio, gn_data;
    gf_ba_summary last_summary{};
    std::vector<double> marg_J, marg_r, marg_x0; std::vector<int> marg_id;   // receive buffers of gf_ba_marginalize
    double t_sect[6] = {0, 0, 0, 0, 0, 0};   // host wall time [s]: before optimization(), window build, waiting for the solve, between solve and marginalisation, waiting for it, after
    double t_mark = 0;   // CPU time of the member's thread (not wall: the members of a group share the host's cores)
    static double cpu_now() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
    void lap(int i) { const double n = cpu_now(); t_sect[i] += n - t_mark; t_mark = n; }
    long long n_optimizations = 0;
    double imu_noise[4], wheel_noise[2];
    std::string err;

    explicit gf_estimator(const gf_estimator_cfg& c) : cfg(c), WINDOW_SIZE(c.window_size) {
        const int NP = WINDOW_SIZE + 1;
        Ps.assign(NP, v3(0, 0, 0)); Vs = Bas = Bgs = Ps; Rs.assign(NP, m3_identity()); Headers.assign(NP, 0.0);
        pre_integrations.assign(NP, nullptr); pre_integrations_wheel.assign(NP, nullptr);
        para_Pose.assign(7 * NP, 0); para_SpeedBias.assign(9 * NP, 0); para_Feature.assign(std::max(1, c.max_features), 0);
        tic = arr3(c.tic); ric = arr9(c.ric); tio = arr3(c.tio); rio = arr9(c.rio);   // setParameter EST:176-199
        td = c.td; td_wheel = c.td_wheel; sx = c.sx; sy = c.sy; sw = c.sw; g = v3(0, 0, c.g_norm);
        imu_noise[0] = c.acc_n; imu_noise[1] = c.gyr_n; imu_noise[2] = c.acc_w; imu_noise[3] = c.gyr_w;
        wheel_noise[0] = c.wheel_vel_n; wheel_noise[1] = c.wheel_gyr_n;
        f_manager.WINDOW_SIZE = WINDOW_SIZE; f_manager.FOCAL_LENGTH = c.focal_length; f_manager.MIN_PARALLAX = c.min_parallax_px / c.focal_length;
        f_manager.INIT_DEPTH = c.init_depth; f_manager.depth_threshold = c.depth_threshold;
        gnss_meas_buf.assign(NP, {}); para_rcv_dt.assign(4 * NP, 0.0); para_rcv_ddt.assign(NP, 0.0);
        gnss_iono.assign(c.gnss_iono, c.gnss_iono + 8);      // clearState EST:147-149
        diff_t_gnss_local = c.gnss_local_time_diff;          // rosNodeTest.cpp:703-708 hands GNSS_LOCAL_TIME_DIFF over at start-up
    }
    ~gf_estimator() { if (ba) gf_ba_destroy(ba); if (tracker) gf_tracker_destroy(tracker); }

    // ------------------------------------------------------------ measurement intake
    bool getInterval(std::deque<std::pair<double, V3>>& a, std::deque<std::pair<double, V3>>& b, double t0, double t1, std::vector<std::pair<double, V3>>& av,
                     std::vector<std::pair<double, V3>>& bv) {  // getIMUInterval EST:406-439 / getWheelInterval EST:440-474
        if (a.empty()) return false;
        if (!(t1 <= a.back().first)) return false;
        while (!a.empty() && a.front().first <= t0) { a.pop_front(); b.pop_front(); }
        while (!a.empty() && a.front().first < t1) { av.push_back(a.front()); a.pop_front(); bv.push_back(b.front()); b.pop_front(); }
        if (!a.empty()) { av.push_back(a.front()); bv.push_back(b.front()); }
        return true;
    }
    // ------------------------------------------------------------ GNSS intake
    bool getGNSSInterval(double /*t0*/, double t1) {  // EST:476-510: stale epochs are thrown away, then the front epoch is taken whatever its age
        if (GNSSBuf.empty()) return false;
        while (!GNSSBuf.empty() && GNSSBuf.front().second[0].o.time < t1 + diff_t_gnss_local - 0.1 /* MAX_GNSS_CAMERA_DELAY */) {
            GNSSBuf.pop_front();
            if (GNSSBuf.empty()) return false;
        }
        gnss_msg = std::move(GNSSBuf.front().second);
        GNSSBuf.pop_front();
        return true;
    }
    static V3 ecef2geo(V3 p) {   // gnss_comm ecef2geo: latitude [deg], longitude [deg], height [m] (the formulas of gf_ba_gnss.hpp, on the host)
        if (p.x == 0 && p.y == 0) return v3(0, 0, 0);
        const double a = 6378137.0, e2 = 6.69437999014e-3, a2 = a * a, b2 = a2 * (1 - e2), b = sqrt(b2), ep2 = (a2 - b2) / b2, rho = sqrt(p.x * p.x + p.y * p.y);
        double s1 = p.z * a, s2 = rho * b, h = sqrt(s1 * s1 + s2 * s2);
        const double st = s1 / h, ct = s2 / h;
        s1 = p.z + ep2 * b * st * st * st; s2 = rho - a * e2 * ct * ct * ct; h = sqrt(s1 * s1 + s2 * s2);
        const double sin_lat = s1 / h, cos_lat = s2 / h, N = a2 / sqrt(a2 * cos_lat * cos_lat + b2 * sin_lat * sin_lat);
        return v3(atan(s1 / s2) * 180.0 / M_PI, atan2(p.y, p.x) * 180.0 / M_PI, rho / cos_lat - N);
    }
    static M3 ecef2rotation(V3 p) {   // gnss_comm ecef2rotation = geo2rotation(ecef2geo(p)): R_ecef_enu
        const V3 lla = ecef2geo(p);
        const double lat = lla.x * M_PI / 180.0, lon = lla.y * M_PI / 180.0, sl = sin(lat), cl = cos(lat), so = sin(lon), co = cos(lon);
        M3 R;
        R.m[0] = -so; R.m[1] = -sl * co; R.m[2] = cl * co; R.m[3] = co; R.m[4] = -sl * so; R.m[5] = cl * so; R.m[6] = 0; R.m[7] = cl; R.m[8] = sl;
        return R;
    }
    static double sat_elevation(V3 rcv, V3 sat) {   // gnss_comm sat_azel, elevation only
        V3 dl = sat - rcv; dl = dl / norm(dl);
        return asin((transpose(ecef2rotation(rcv)) * dl).z);
    }
    // ---- GNSSVIInitializer (initial/gnss_vi_initializer.cpp) on the satellite states carried by gf_gnss_obs.  Its three gnss_comm callees -- psr_pos, psr_res,
    // dopp_res (gnss_spp.cpp; gnss_comm is not vendored by the reference) -- are restated from their published form with the measurement model of
    // GnssPsrDoppFactor::Evaluate (gnss_psr_dopp_factor.cpp:76-101): unweighted residuals, Jacobian rows [-unit(rcv -> sat), 1 on the system's clock].
    static double trop_delay(V3 lla, double el) {   // Saastamoinen, standard atmosphere, humidity 0.7 (as gf_ba_gnss.hpp)
        if (lla.z < -100.0 || 1e4 < lla.z || el <= 0) return 0.0;
        const double hgt = lla.z < 0.0 ? 0.0 : lla.z;
        const double pres = 1013.25 * pow(1.0 - 2.2557e-5 * hgt, 5.2568), temp = 15.0 - 6.5e-3 * hgt + 273.16;
        const double e = 6.108 * 0.7 * exp((17.15 * temp - 4684.0) / (temp - 38.45)), z = M_PI / 2.0 - el;
        return 0.0022768 * pres / (1.0 - 0.00266 * cos(2.0 * lla.x * M_PI / 180.0) - 0.00028 * hgt / 1e3) / cos(z) + 0.002277 * (1255.0 / temp + 0.05) * e / cos(z);
    }
    static double ion_delay(double tow, const double* ion_in, V3 lla, double az, double el) {   // Klobuchar (as gf_ba_gnss.hpp)
        const double ion_default[8] = {0.1118e-07, -0.7451e-08, -0.5961e-07, 0.1192e-06, 0.1167e+06, -0.2294e+06, -0.1311e+06, 0.1049e+07};
        if (lla.z < -1e3 || el <= 0) return 0.0;
        double nrm = 0;
        for (int i = 0; i < 8; i++) nrm += ion_in[i] * ion_in[i];
        double ion[8];
        for (int i = 0; i < 8; i++) ion[i] = nrm <= 0.0 ? ion_default[i] : ion_in[i];
        const double psi = 0.0137 / (el / M_PI + 0.11) - 0.022;
        double phi = lla.x / 180.0 + psi * cos(az);
        if (phi > 0.416) phi = 0.416; else if (phi < -0.416) phi = -0.416;
        const double lam = lla.y / 180.0 + psi * sin(az) / cos(phi * M_PI);
        phi += 0.064 * cos((lam - 1.617) * M_PI);
        double tt = 43200.0 * lam + tow;
        tt -= floor(tt / 86400.0) * 86400.0;
        const double f = 1.0 + 16.0 * pow(0.53 - el / M_PI, 3.0);
        double amp = ion[0] + phi * (ion[1] + phi * (ion[2] + phi * ion[3])), per = ion[4] + phi * (ion[5] + phi * (ion[6] + phi * ion[7]));
        amp = amp < 0.0 ? 0.0 : amp; per = per < 72000.0 ? 72000.0 : per;
        const double x = 2.0 * M_PI * (tt - 50400.0) / per;
        return 2.99792458e8 * f * (fabs(x) < 1.57 ? 5e-9 + amp * (1.0 + x * x * (-0.5 + x * x / 24.0)) : 5e-9);
    }
    static constexpr double kC = 2.99792458e8, kOmg = 7.2921151467e-5;
    // psr_res: residual and Jacobian row (7: position 3, clock bias per system 4) of every observation of an epoch at receiver state xyzt
    void psr_res(const double* xyzt, const std::vector<gf_gnss_obs>& meas, std::vector<double>& res, std::vector<double>& J) const {
        const V3 rcv = arr3(xyzt);
        for (const gf_gnss_obs& o : meas) {
            const V3 sv = arr3(o.sv_pos);
            double ion = 0, tro = 0;
            if (norm(rcv) > 0) {
                const V3 lla = ecef2geo(rcv);
                V3 dl = sv - rcv; dl = dl / norm(dl);
                const V3 enu = transpose(ecef2rotation(rcv)) * dl;
                double az = sqrt(dl.x * dl.x + dl.y * dl.y) < 1e-12 ? 0.0 : atan2(enu.x, enu.y);
                if (az < 0) az += 2 * M_PI;
                const double el = asin(enu.z);
                tro = trop_delay(lla, el); ion = ion_delay(o.tow, gnss_iono.data(), lla, az, el);
            }
            const V3 r2s = sv - rcv;
            const double rg = norm(r2s);
            const double est = rg + kOmg * (sv.x * rcv.y - sv.y * rcv.x) / kC + xyzt[3 + o.sys] - o.svdt * kC + ion + tro + o.tgd * kC;
            res.push_back(est - o.psr);
            double row[7] = {-r2s.x / rg, -r2s.y / rg, -r2s.z / rg, 0, 0, 0, 0};
            row[3 + o.sys] = 1.0;
            J.insert(J.end(), row, row + 7);
        }
    }
    // dopp_res: residual and Jacobian row (4: ECEF velocity 3, clock drift) of every observation of an epoch
    static void dopp_res(const double* vel_ddt, V3 rcv, const std::vector<gf_gnss_obs>& meas, std::vector<double>& res, std::vector<double>& J) {
        const V3 vel = arr3(vel_ddt);
        for (const gf_gnss_obs& o : meas) {
            const V3 sv = arr3(o.sv_pos), svv = arr3(o.sv_vel);
            const V3 r2s = sv - rcv;
            const V3 unit = r2s / norm(r2s);
            const double sag = kOmg / kC * (svv.x * rcv.y + sv.x * vel.y - svv.y * rcv.x - sv.y * vel.x);
            const double est = dot(svv - vel, unit) + vel_ddt[3] + sag - o.svddt * kC;
            res.push_back(est + o.dopp * o.wavelength);
            J.insert(J.end(), {-unit.x, -unit.y, -unit.z, 1.0});
        }
    }
    // dx = -(G^T G)^-1 G^T b for an m x n system (n <= 7), Gaussian elimination with partial pivoting on the normal equations
    static bool normal_solve(const std::vector<double>& G, const std::vector<double>& b, int n, double* dx) {
        const int m = (int)b.size();
        double A[7][8];
        for (int i = 0; i < n; i++) {
            for (int j = 0; j < n; j++) { double a = 0; for (int r = 0; r < m; r++) a += G[(size_t)r * n + i] * G[(size_t)r * n + j]; A[i][j] = a; }
            double g = 0; for (int r = 0; r < m; r++) g += G[(size_t)r * n + i] * b[r];
            A[i][n] = -g;
        }
        for (int c = 0; c < n; c++) {
            int p = c;
            for (int r = c + 1; r < n; r++) if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
            if (!(fabs(A[p][c]) > 0)) return false;
            if (p != c) for (int j = 0; j <= n; j++) std::swap(A[p][j], A[c][j]);
            for (int r = c + 1; r < n; r++) { const double f = A[r][c] / A[c][c]; for (int j = c; j <= n; j++) A[r][j] -= f * A[c][j]; }
        }
        for (int i = n - 1; i >= 0; i--) { double v = A[i][n]; for (int j = i + 1; j < n; j++) v -= A[i][j] * dx[j]; dx[i] = v / A[i][i]; }
        return true;
    }
    // GNSSVIAlign's three stages (EST:1972-2013) on gnss_meas_buf[0..W]: false = one of them failed
    bool gnssViInitialize(double* refined_xyzt, double* rough_xyzt, double& aligned_yaw, double& aligned_ddt) const {
        const int NPW = WINDOW_SIZE + 1;
        size_t num_all = 0;
        for (int i = 0; i < NPW; i++) num_all += gnss_meas_buf[i].size();
        // 1. coarse_localization = psr_pos on all measurements of the window at one receiver position (gnss_vi_initializer.cpp:16-41)
        {
            std::vector<gf_gnss_obs> accum;
            for (int i = 0; i < NPW; i++) accum.insert(accum.end(), gnss_meas_buf[i].begin(), gnss_meas_buf[i].end());
            if (accum.size() < 4) return false;
            double xyzt[7] = {0, 0, 0, 0, 0, 0, 0};
            bool seen[4] = {false, false, false, false};
            for (auto& o : accum) seen[o.sys] = true;
            double dxn = 1.0; int it = 0;
            while (it < 10 && dxn > 1e-4) {
                std::vector<double> b, G;
                psr_res(xyzt, accum, b, G);
                for (int k = 0; k < 4; k++) if (!seen[k]) { double row[7] = {0, 0, 0, 0, 0, 0, 0}; row[3 + k] = 1.0; G.insert(G.end(), row, row + 7); b.push_back(0.0); }   // pin unobserved clocks
                double dx[7];
                if (!normal_solve(G, b, 7, dx)) return false;
                dxn = 0; for (int k = 0; k < 7; k++) { xyzt[k] += dx[k]; dxn += dx[k] * dx[k]; }
                dxn = sqrt(dxn); it++;
            }
            if (it == 10 && dxn > 1e-4) return false;
            if (norm(arr3(xyzt)) == 0 || std::isnan(xyzt[0])) return false;
            for (int k = 0; k < 4; k++) if (fabs(xyzt[3 + k]) < 1) xyzt[3 + k] = 0;   // not observed yet
            memcpy(rough_xyzt, xyzt, sizeof(xyzt));
        }
        // 2. yaw_alignment (gnss_vi_initializer.cpp:43-104)
        {
            const V3 anchor = arr3(rough_xyzt);
            const M3 Ree = ecef2rotation(anchor);
            double est_yaw = 0, est_ddt = 0, dxn = 1.0; int it = 0;
            while (it < 10 && dxn > 1e-5) {
                std::vector<double> G, b;
                const double cy = cos(est_yaw), sy_ = sin(est_yaw);
                for (int i = 0; i < NPW; i++) {
                    const V3 v = Vs[i];
                    const V3 ve = Ree * v3(cy * v.x - sy_ * v.y, sy_ * v.x + cy * v.y, v.z);
                    const double vd[4] = {ve.x, ve.y, ve.z, est_ddt};
                    std::vector<double> r, Jd;
                    dopp_res(vd, anchor, gnss_meas_buf[i], r, Jd);
                    const V3 dv = Ree * v3(-sy_ * v.x - cy * v.y, cy * v.x - sy_ * v.y, 0.0);   // R_ecef_enu * tmp_M * local_v
                    for (size_t q = 0; q < r.size(); q++) { G.push_back(Jd[4 * q] * dv.x + Jd[4 * q + 1] * dv.y + Jd[4 * q + 2] * dv.z); G.push_back(1.0); b.push_back(r[q]); }
                }
                double dx[2];
                if (!normal_solve(G, b, 2, dx)) return false;
                est_yaw += dx[0]; est_ddt += dx[1]; dxn = sqrt(dx[0] * dx[0] + dx[1] * dx[1]); it++;
            }
            aligned_yaw = est_yaw;
            if (aligned_yaw > M_PI) aligned_yaw -= floor(est_yaw / (2.0 * M_PI) + 0.5) * (2.0 * M_PI);
            else if (aligned_yaw < -M_PI) aligned_yaw -= ceil(est_yaw / (2.0 * M_PI) - 0.5) * (2.0 * M_PI);
            aligned_ddt = est_ddt;
        }
        // 3. anchor_refinement (gnss_vi_initializer.cpp:106-172)
        {
            V3 anchor = arr3(rough_xyzt);
            double dt4[4] = {rough_xyzt[3], rough_xyzt[4], rough_xyzt[5], rough_xyzt[6]};
            const double cy = cos(aligned_yaw), sy_ = sin(aligned_yaw);
            double dxn = 1.0; int it = 0;
            while (it < 10 && dxn > 1e-5) {
                std::vector<double> G, b;
                const M3 Ree = ecef2rotation(anchor);
                for (int i = 0; i < NPW; i++) {
                    const V3 p = Ps[i];
                    const V3 pe = Ree * v3(cy * p.x - sy_ * p.y, sy_ * p.x + cy * p.y, p.z) + anchor;
                    const double st[7] = {pe.x, pe.y, pe.z, dt4[0] + aligned_ddt * i, dt4[1] + aligned_ddt * i, dt4[2] + aligned_ddt * i, dt4[3] + aligned_ddt * i};
                    psr_res(st, gnss_meas_buf[i], b, G);
                }
                for (int k = 0; k < 4; k++) if (rough_xyzt[3 + k] == 0) { double row[7] = {0, 0, 0, 0, 0, 0, 0}; row[3 + k] = 1.0; G.insert(G.end(), row, row + 7); b.push_back(0.0); }
                double dx[7];
                if (!normal_solve(G, b, 7, dx)) return false;
                anchor = anchor + arr3(dx);
                dxn = 0; for (int k = 0; k < 7; k++) dxn += dx[k] * dx[k];
                for (int k = 0; k < 4; k++) dt4[k] += dx[3 + k];
                dxn = sqrt(dxn); it++;
            }
            refined_xyzt[0] = anchor.x; refined_xyzt[1] = anchor.y; refined_xyzt[2] = anchor.z;
            for (int k = 0; k < 4; k++) refined_xyzt[3 + k] = dt4[k];
        }
        (void)num_all;
        return true;
    }
    void processGNSS(const std::vector<GMsgObs>& gnss_meas) {  // EST:1455-1535 (L1 selection happens before the C boundary)
        std::vector<gf_gnss_obs> valid_meas;
        for (const GMsgObs& m : gnss_meas) {
            const gf_gnss_obs& obs = m.o;
            if (obs.sys < 0 || obs.sys > 3) continue;                                             // :1463-1465
            const gf_gnss_ephem* eph = nullptr; const gf_gnss_glo_ephem* geph = nullptr;
            if (m.is_raw) {                                                                       // :1467-1495: the satellite's ephemeris nearest in toe, within EPH_VALID_SECONDS
                const auto ti = sat2time_index.find(obs.sat);
                if (ti == sat2time_index.end()) continue;
                double ephem_time = 7200.0; size_t ephem_index = 0;
                for (const auto& kv : ti->second) if (fabs(kv.first - obs.time) < ephem_time) { ephem_time = fabs(kv.first - obs.time); ephem_index = kv.second; }
                if (ephem_time >= 7200.0) continue;
                if (obs.sys == 1) geph = &sat2gephem.at(obs.sat)[ephem_index]; else eph = &sat2ephem.at(obs.sat)[ephem_index];
            }
            if (obs.psr_std > cfg.gnss_psr_std_thres || obs.dopp_std > cfg.gnss_dopp_std_thres) { sat_track_status[obs.sat] = 0; continue; }   // :1499-1504
            ++sat_track_status[obs.sat];                                                          // :1505-1510
            if (sat_track_status[obs.sat] < cfg.gnss_track_num_thres) continue;                   // :1511-1512
            if (gnss_ready) {                                                                     // :1515-1526: the satellite at the reception time when the ephemeris is here
                const V3 sat_ecef = m.is_raw ? (geph ? gnss_eph::geph2pos(obs.time, *geph, nullptr) : gnss_eph::eph2pos(obs.time, *eph, nullptr)) : arr3(obs.sv_pos);
                if (sat_elevation(ecef_pos, sat_ecef) < cfg.gnss_elevation_thres * M_PI / 180.0) continue;
            }
            if (m.is_raw) { gf_gnss_obs st; gnss_eph::sat_state(m.raw, eph, geph, &st); valid_meas.push_back(st); }   // what the factor's constructor derives (gnss_psr_dopp_factor.cpp:3-47)
            else valid_meas.push_back(obs);
        }
        gnss_meas_buf[frame_count] = std::move(valid_meas);
    }
    bool GNSSVIAlign() {  // EST:1928-2043 with the initialiser's result handed in (gf_estimator_set_gnss_alignment)
        if (!is_imu_excited && solver_flag == INITIAL) return false;
        if (gnss_ready) return true;
        double ax = 0, ay = 0;
        for (int i = 0; i <= WINDOW_SIZE; i++) { ax += fabs(Vs[i].x); ay += fabs(Vs[i].y); }
        ax /= WINDOW_SIZE + 1; ay /= WINDOW_SIZE + 1;
        if (sqrt(ax * ax + ay * ay) < 0.3) return false;
        double refined[7], rough[7], yaw = 0, ddt = 0;
        if (align_pending) {   // the caller ran its own initialiser (gf_estimator_set_gnss_alignment): take its result
            memcpy(refined, align_anc, 24); memcpy(refined + 3, align_dt, 32); memcpy(rough, refined, sizeof(rough)); yaw = align_yaw; ddt = align_ddt;
            align_pending = false;
        } else if (!gnssViInitialize(refined, rough, yaw, ddt)) return false;   // :1972-2013
        int one_observed_sys = -1;
        for (int k = 0; k < 4; k++) if (rough[3 + k] != 0) { one_observed_sys = k; break; }
        for (int i = 0; i <= WINDOW_SIZE; i++) {   // :2015-2036 (the drift is multiplied by the frame index, not by a time)
            para_rcv_ddt[i] = ddt;
            for (int k = 0; k < 4; k++) para_rcv_dt[4 * i + k] = (rough[3 + k] == 0 ? (one_observed_sys < 0 ? 0.0 : refined[3 + one_observed_sys]) : refined[3 + k]) + ddt * i;
        }
        anc_ecef = arr3(refined); R_ecef_enu = ecef2rotation(anc_ecef); yaw_enu_local = yaw;
        return true;
    }
    void updateGNSSStatistics() {  // EST:2045-2058
        const double c = cos(yaw_enu_local), s = sin(yaw_enu_local);
        const V3 p = Ps[WINDOW_SIZE];
        enu_pos = v3(c * p.x - s * p.y, s * p.x + c * p.y, p.z);
        ecef_pos = anc_ecef + R_ecef_enu * enu_pos;
    }
    void afterOptimizationGNSS() {  // EST:945-956, :1014-1025, :1113-1123
        if (!cfg.gnss_enable) return;
        if (!gnss_ready) gnss_ready = GNSSVIAlign();
        if (gnss_ready) updateGNSSStatistics();
    }

    void initFirstIMUPose(const std::vector<std::pair<double, V3>>& accVector) {  // EST:710-731
        initFirstPoseFlag = true;
        V3 averAcc = v3(0, 0, 0);
        for (auto& a : accVector) averAcc = averAcc + a.second;
        averAcc = averAcc / (double)(int)accVector.size();
        M3 R0 = g2R(averAcc);
        const double yaw = R2ypr(R0).x;
        R0 = ypr2R(v3(-yaw, 0, 0)) * R0;
        Rs[0] = R0 * arr9(cfg.rio);  // RIO: the configured wheel extrinsic rotation
    }
    void processIMU(double t, double dt, V3 linear_acceleration, V3 angular_velocity) {  // EST:743-785
        if (!first_imu) { first_imu = true; acc_0 = linear_acceleration; gyr_0 = angular_velocity; }
        if (!pre_integrations[frame_count]) pre_integrations[frame_count] = std::make_shared<ImuPre>(acc_0, gyr_0, Bas[frame_count], Bgs[frame_count]);
        if (frame_count != 0) {
            pre_integrations[frame_count]->push_back(dt, linear_acceleration, angular_velocity);
            tmp_pre_integration->push_back(dt, linear_acceleration, angular_velocity);
            const int j = frame_count;
            const V3 un_acc_0 = Rs[j] * (acc_0 - Bas[j]) - g;
            const V3 un_acc_1 = Rs[j] * (linear_acceleration - Bas[j]) - g;
            const V3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
            dP_imu = dP_imu + Vs[j] * dt + un_acc * (0.5 * dt * dt);
        }
        acc_0 = linear_acceleration; gyr_0 = angular_velocity;
    }
    void fastPredictIMU(double t, V3 linear_acceleration, V3 angular_velocity) {  // EST:4014-4028
        const double dt = t - latest_time;
        latest_time = t;
        const V3 un_acc_0 = latest_Q * (latest_acc_0 - latest_Ba) - g;
        const V3 un_gyr = (latest_gyr_0 + angular_velocity) * 0.5 - latest_Bg;
        latest_Q = latest_Q * qmat(deltaQ(un_gyr * dt));
        const V3 un_acc_1 = latest_Q * (linear_acceleration - latest_Ba) - g;
        const V3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
        latest_P = latest_P + latest_V * dt + un_acc * (0.5 * dt * dt);
        latest_V = latest_V + un_acc * dt;
        latest_acc_0 = linear_acceleration; latest_gyr_0 = angular_velocity;
    }
    void fastPredictWheel(double t, V3 linear_velocity, V3 angular_velocity) {  // EST:4079-4093 (un_gyr from the IMU's latest_gyr_0, as written there)
        const double dt = t - latest_time_wheel;
        latest_time_wheel = t;
        const V3 un_gyr = (latest_gyr_0 + angular_velocity) * (0.5 * latest_sw);
        const V3 un_vel_0 = latest_Q_wheel * latest_vel_wheel_0;
        latest_Q_wheel = latest_Q_wheel * qmat(deltaQ(un_gyr * dt));
        const V3 s = latest_Q_wheel * linear_velocity + un_vel_0;
        latest_V_wheel = v3(0.5 * latest_sx * s.x, 0.5 * latest_sy * s.y, 0.5 * s.z);
        latest_P_wheel = latest_P_wheel + latest_V_wheel * dt;
        latest_vel_wheel_0 = linear_velocity; latest_gyr_wheel_0 = angular_velocity;   // shared with processWheel (SURVEY.md 8a quirk list, DESIGN.md: quirk 15)
    }
    void updateLatestStates() {  // EST:4141-4198
        const int fc = frame_count;
        latest_time = Headers[fc] + td;
        latest_P = Ps[fc]; latest_Q = Rs[fc]; latest_V = Vs[fc]; latest_Ba = Bas[fc]; latest_Bg = Bgs[fc];
        latest_acc_0 = acc_0; latest_gyr_0 = gyr_0;
        for (size_t i = 0; i < accBuf.size() && i < gyrBuf.size(); i++) fastPredictIMU(accBuf[i].first, accBuf[i].second, gyrBuf[i].second);
        latest_time_wheel = Headers[fc] + td - td_wheel;
        latest_Q_wheel = Rs[fc] * rio;
        latest_P_wheel = Rs[fc] * tio + Ps[fc];
        latest_sx = sx; latest_sy = sy; latest_sw = sw;
        latest_vel_wheel_0 = vel_0_wheel; latest_gyr_wheel_0 = gyr_0_wheel;
        for (size_t i = 0; i < wheelVelBuf.size() && i < wheelGyrBuf.size(); i++) fastPredictWheel(wheelVelBuf[i].first, wheelVelBuf[i].second, wheelGyrBuf[i].second);
    }
    void processWheel(double t, double dt, V3 linear_velocity, V3 angular_velocity) {  // EST:786-842
        if (!first_wheel) { first_wheel = true; vel_0_wheel = linear_velocity; gyr_0_wheel = angular_velocity; }
        if (!pre_integrations_wheel[frame_count]) pre_integrations_wheel[frame_count] = std::make_shared<WheelPre>(vel_0_wheel, gyr_0_wheel, sx, sy, sw, td_wheel);
        if (frame_count != 0) {
            pre_integrations_wheel[frame_count]->push_back(dt, linear_velocity, angular_velocity);
            tmp_wheel_pre_integration->push_back(dt, linear_velocity, angular_velocity);
            const int j = frame_count;
            latest_time_wheel = t;
            const V3 un_gyr = (gyr_0_wheel + angular_velocity) * 0.5;
            const V3 un_vel_0 = Rs[j] * latest_vel_wheel_0;
            if (!systemstationary) {
                Rs[j] = Rs[j] * qmat(deltaQ(un_gyr * dt));
                Vs[j] = (Rs[j] * linear_velocity + un_vel_0) * 0.5;
                Ps[j] = Ps[j] + Vs[j] * dt;
            }
            if (systemstationary) Vs[j] = v3(0, 0, 0);
            latest_vel_wheel_0 = linear_velocity; latest_gyr_wheel_0 = angular_velocity;
            dP_wheel.x -= dt * Vs[j].y; dP_wheel.y += dt * Vs[j].x; dP_wheel.z -= dt * Vs[j].z;
        }
        vel_0_wheel = linear_velocity; gyr_0_wheel = angular_velocity;
    }
    // the processThread of multiple_thread: 1 (EST:209, :529-707) made deterministic: whenever an input arrives, take every queued frame
    // whose IMU / wheel interval is complete.  Which samples a frame integrates depends only on time stamps, not on when it is taken.
    int drain() {
        for (;;) {
            bool progressed = false;
            if (int rc = processMeasurements(&progressed)) return rc;
            if (!progressed) return GF_OK;
        }
    }
    int processMeasurements(bool* progressed = nullptr) {  // EST:526-709, single-thread form: one feature frame per call
        if (progressed) *progressed = false;
        if (featureBuf.empty()) return GF_OK;
        auto& feature = featureBuf.front();
        curTime = feature.first + td; curTime_wheel = curTime - td_wheel;
        if (cfg.use_imu && !(!accBuf.empty() && feature.first + td <= accBuf.back().first)) return GF_OK;                                  // wait for imu
        if (cfg.use_wheel && !(!wheelVelBuf.empty() && feature.first + td - td_wheel <= wheelVelBuf.back().first)) return GF_OK;           // wait for wheel
        std::vector<std::pair<double, V3>> accVector, gyrVector, velWheelVector, gyrWheelVector;
        if (cfg.use_imu) getInterval(accBuf, gyrBuf, prevTime, curTime, accVector, gyrVector);
        const double header = feature.first;
        std::vector<gf_feature_obs> image = std::move(feature.second);
        featureBuf.pop_front();
        if (cfg.gnss_enable) getGNSSInterval(prevTime, curTime);   // EST:584-587
        if (cfg.use_wheel) getInterval(wheelVelBuf, wheelGyrBuf, prevTime_wheel, curTime_wheel, velWheelVector, gyrWheelVector);
        if (cfg.use_imu) {
            dP_imu = v3(0, 0, 0);
            if (!initFirstPoseFlag) initFirstIMUPose(accVector);
            for (size_t i = 0; i < accVector.size(); i++) {
                double dt;
                if (i == 0) dt = accVector[i].first - prevTime;
                else if (i == accVector.size() - 1) dt = curTime - accVector[i - 1].first;
                else dt = accVector[i].first - accVector[i - 1].first;
                processIMU(accVector[i].first, dt, accVector[i].second, gyrVector[i].second);
            }
        }
        if (cfg.use_wheel) {
            dP_wheel = v3(0, 0, 0);
            for (size_t i = 0; i < velWheelVector.size(); i++) {
                double dt;
                if (i == 0) dt = velWheelVector[i].first - prevTime_wheel;
                else if (i == velWheelVector.size() - 1) dt = curTime_wheel - velWheelVector[i - 1].first;
                else dt = velWheelVector[i].first - velWheelVector[i - 1].first;
                processWheel(velWheelVector[i].first, dt, velWheelVector[i].second, gyrWheelVector[i].second);
            }
            const double dis = norm(dP_wheel - dP_imu);
            if (dis > 0.02 && cfg.wdetect) wheelanomaly = true;
            wheelstationary = norm(dP_wheel) < 0.001;
            preintegrationstationary = norm(dP_imu) < 0.001;
        }
        if (cfg.gnss_enable && !gnss_msg.empty()) processGNSS(gnss_msg);   // EST:656-660
        const int rc = processImage(image, header);
        prevTime = curTime; prevTime_wheel = curTime_wheel;
        if (progressed) *progressed = true;
        // pubOdometry(*this, header), EST:679 -> utility/visualization.cpp:287-357: the trajectory file gets a line once the window is live
        if (rc == GF_OK && !result_path.empty() && (solver_flag == NON_LINEAR || is_imu_excited))
            return gf_tum_append(result_path.c_str(), header, &Ps[WINDOW_SIZE].x, Rs[WINDOW_SIZE].m);
        return rc;
    }

    // ------------------------------------------------------------ stationarity votes
    void checkimu() {  // EST:2173-2216 (sum_g is an uninitialised local in the reference; zero here)
        const int n = (int)all_image_frame.size() - 1;
        V3 sum_g = v3(0, 0, 0);
        bool first = true;
        for (auto& kv : all_image_frame) {
            if (first) { first = false; continue; }
            kv.second.pre_integration->eval_state();   // delta_v and sum_dt are all this vote reads
            sum_g = sum_g + arr3(kv.second.pre_integration->delta_v) / kv.second.pre_integration->sum_dt;
        }
        const V3 aver_g = sum_g * 1.0 / (double)n;
        double var = 0;
        first = true;
        for (auto& kv : all_image_frame) {
            if (first) { first = false; continue; }
            const V3 tmp_g = arr3(kv.second.pre_integration->delta_v) / kv.second.pre_integration->sum_dt;
            var += sqn(tmp_g - aver_g);
        }
        var = sqrt(var / (double)n);
        varstationary = var < 0.1;  // NaN (n == 0) compares false, as in the reference
    }
    bool checkvisual() {  // EST:2218-2274; solveRelativeRT_PNP always succeeds, its pose is unused
        // The reference collects the correspondences (i, WINDOW_SIZE) frame by frame -- WINDOW_SIZE walks over the track list, a vector each -- and sums their parallax.  One
        // walk does the same: a track contributes to frame i's sum in list order either way, with the same operations per term (getCorrespondingWithDepth's depth-scaled
        // points, FM:219-247), so every sum has the bits it had; the decisions are then taken in frame order as before.  (round 6: this vote runs on every image of every member)
        double sum_parallax[32]; int nc[32];
        const int W_ = std::min(WINDOW_SIZE, 32);
        for (int i = 0; i < W_; i++) { sum_parallax[i] = 0; nc[i] = 0; }
        if (WINDOW_SIZE <= 32)
            for (auto& it : f_manager.feature) {
                if (it.endFrame() < WINDOW_SIZE) continue;
                const FeaturePerFrame& fb = it.feature_per_frame[WINDOW_SIZE - it.start_frame];
                if (fb.depth < 0.1 || fb.depth > 10) continue;
                const V3 b = fb.point * fb.depth;
                for (int i = std::max(it.start_frame, 0); i < WINDOW_SIZE; i++) {
                    const FeaturePerFrame& fa = it.feature_per_frame[i - it.start_frame];
                    if (fa.depth < 0.1 || fa.depth > 10) continue;
                    const V3 a = fa.point * fa.depth;
                    const double dx = a.x / a.z - b.x / b.z, dy = a.y / a.z - b.y / b.z;
                    sum_parallax[i] = sum_parallax[i] + sqrt(dx * dx + dy * dy);
                    nc[i]++;
                }
            }
        for (int i = 0; i < WINDOW_SIZE; i++) {
            double sp; int n;
            if (WINDOW_SIZE <= 32) { sp = sum_parallax[i]; n = nc[i]; }
            else {   // (longer windows than any configuration uses: the frame-by-frame route)
                const std::vector<double> corres = f_manager.getCorrespondingWithDepth(i, WINDOW_SIZE);
                n = (int)(corres.size() / 6); sp = 0;
                for (int j = 0; j < n; j++) { const double* c = &corres[6 * j]; const double dx = c[0] / c[2] - c[3] / c[5], dy = c[1] / c[2] - c[4] / c[5]; sp = sp + sqrt(dx * dx + dy * dy); }
            }
            if (n > 20) {
                const double average_parallax = 1.0 * sp / n;
                if (average_parallax * 460 < 0.5) return true;
                visualstationary = false;
            }
            visualstationary = false;
        }
        return false;
    }

    // ------------------------------------------------------------ initialisation shortcuts
    void solveGyroscopeBias() {  // initial/initial_aligment.cpp:14-47
        double A[9] = {0}, b[3] = {0};
        for (auto fi = all_image_frame.begin(); std::next(fi) != all_image_frame.end(); ++fi) {
            auto fj = std::next(fi);
            ImuPre& p = *fj->second.pre_integration;
            p.eval(imu_noise);
            const Q4 q_ij = rot_to_quat(transpose(fi->second.R) * fj->second.R);
            double tA[9];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) tA[3 * r + c] = p.jacobian[(3 + r) * 15 + 12 + c];  // block<3,3>(O_R, O_BG)
            const V3 tb = qvec(qmul(qinverse(Q4{p.delta_q[0], p.delta_q[1], p.delta_q[2], p.delta_q[3]}), q_ij)) * 2.0;
            const double tbv[3] = {tb.x, tb.y, tb.z};
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) { double s = 0; for (int k = 0; k < 3; k++) s += tA[3 * k + r] * tA[3 * k + c]; A[3 * r + c] += s; }
                double s = 0; for (int k = 0; k < 3; k++) s += tA[3 * k + r] * tbv[k];
                b[r] += s;
            }
        }
        // A.ldlt().solve(b) on a symmetric 3x3
        const double d0 = A[0], l10 = A[3] / d0, l20 = A[6] / d0, d1 = A[4] - l10 * l10 * d0, l21 = (A[7] - l20 * l10 * d0) / d1, d2 = A[8] - l20 * l20 * d0 - l21 * l21 * d1;
        const double y0 = b[0], y1 = b[1] - l10 * y0, y2 = b[2] - l20 * y0 - l21 * y1;
        const double z0 = y0 / d0, z1 = y1 / d1, z2 = y2 / d2;
        const double x2 = z2, x1 = z1 - l21 * x2, x0 = z0 - l10 * x1 - l20 * x2;
        const V3 delta_bg = v3(x0, x1, x2);
        for (int i = 0; i <= WINDOW_SIZE; i++) Bgs[i] = Bgs[i] + delta_bg;
        for (auto fi = all_image_frame.begin(); std::next(fi) != all_image_frame.end(); ++fi) std::next(fi)->second.pre_integration->repropagate(v3(0, 0, 0), Bgs[0]);
    }
    // EST:2087-2124: solveRelativeRT_PNP returns true whatever cv::solvePnPRansac found, so the first frame with more than 20 correspondences is taken
    // and the parallax test below it is never reached.  -1: no frame; -2: solvePnPRansac found no model (the reference would read unset matrices)
    int relativePoseWithDepth(M3& relative_R, V3& relative_T) {
        for (int i = 0; i < WINDOW_SIZE; i++) {
            const std::vector<double> c = f_manager.getCorrespondingWithDepth(i, WINDOW_SIZE);
            if ((int)c.size() / 6 > 20) {
                std::vector<std::array<double, 6>> corres(c.size() / 6);
                for (size_t k = 0; k < corres.size(); k++) for (int a = 0; a < 6; a++) corres[k][a] = c[6 * k + a];
                double T[3];
                if (!gfinit::solve_relative_rt_pnp(corres, relative_R.m, T)) return -2;
                relative_T = arr3(T);
                return i;
            }
        }
        return -1;
    }
    bool visualInitialAlign() {  // EST:1849-1926; VisualIMUAlignment initial_aligment.cpp:640-653 (depth variants)
        solveGyroscopeBias();
        std::vector<gfinit::AlignFrame> fr;
        for (auto& kv : all_image_frame) {
            gfinit::AlignFrame a{};
            for (int k = 0; k < 9; k++) a.R[k] = kv.second.R.m[k];
            a.T[0] = kv.second.T.x; a.T[1] = kv.second.T.y; a.T[2] = kv.second.T.z;
            if (!fr.empty()) {   // the first frame's pre-integration is never read
                kv.second.pre_integration->eval(imu_noise);
                a.sum_dt = kv.second.pre_integration->sum_dt;
                for (int k = 0; k < 3; k++) { a.delta_p[k] = kv.second.pre_integration->delta_p[k]; a.delta_v[k] = kv.second.pre_integration->delta_v[k]; }
                if (cfg.use_wheel) { kv.second.pre_integration_wheel->eval(wheel_noise); for (int k = 0; k < 3; k++) a.wheel_delta_p[k] = kv.second.pre_integration_wheel->delta_p[k]; }
            }
            fr.push_back(a);
        }
        const double TICa[3] = {tic.x, tic.y, tic.z}, TIOa[3] = {tio.x, tio.y, tio.z};
        double ga[3];
        std::vector<double> x;
        if (!gfinit::linear_alignment(fr, TICa, cfg.g_norm, cfg.use_wheel != 0, rio.m, TIOa, ga, x)) return false;
        g = init_g = arr3(ga);
        for (int i = 0; i <= frame_count; i++) {
            ImageFrame& f = all_image_frame[Headers[i]];
            Ps[i] = f.T; Rs[i] = f.R; f.is_key_frame = true;
        }
        const double s = init_s = x.back();   // EST:1871: the last entry of x -- in the depth variants a gravity-refinement component, not a scale
        for (int i = 0; i <= WINDOW_SIZE; i++) pre_integrations[i]->repropagate(v3(0, 0, 0), Bgs[i]);
        const V3 P0 = Ps[0];
        for (int i = frame_count; i >= 0; i--) Ps[i] = Ps[i] * s - Rs[i] * tic - (P0 * s - Rs[0] * tic);
        int kv = -1;
        for (auto& f : all_image_frame)
            if (f.second.is_key_frame) { kv++; Vs[kv] = f.second.R * v3(x[3 * kv], x[3 * kv + 1], x[3 * kv + 2]); }
        M3 R0 = g2R(g);
        const double yaw = R2ypr(R0 * Rs[0]).x;
        R0 = ypr2R(v3(-yaw, 0, 0)) * R0;
        g = R0 * g;
        for (int i = 0; i <= frame_count; i++) { Ps[i] = R0 * Ps[i]; Rs[i] = R0 * Rs[i]; Vs[i] = R0 * Vs[i]; }
        f_manager.clearDepth();
        f_manager.triangulateWithDepth(Ps.data(), Rs.data(), tic, ric);
        f_manager.triangulate(Ps.data(), Rs.data(), tic, ric);
        return true;
    }
    bool debug_skip_solve = false;
    int head_rc = GF_OK;   // batched pre-integration of this frame (estimator groups)
    int init_rc = GF_OK, init_l = -1, init_points = 0; double init_s = 0; V3 init_g = v3(0, 0, 0);   // what the SfM branch did (debug / tests)
    bool initialStructureSfM(V3 aver_g) {  // EST:1684-1847
        if (!cfg.depth) { init_rc = gf::set_err(GF_ERR_INVALID, "initialStructure: the monocular SfM path (construct / relativePose) is outside the RGB-D scope"); return false; }
        std::vector<gfinit::SfmFeature> sfm_f;
        for (auto& it : f_manager.feature) {
            gfinit::SfmFeature f;
            f.id = it.feature_id;
            int j = it.start_frame - 1;
            for (auto& pf : it.feature_per_frame) { j++; f.observation.push_back({j, {pf.point.x, pf.point.y}}); f.depth.push_back(pf.depth); }
            sfm_f.push_back(f);
        }
        M3 relative_R; V3 relative_T;
        const int l = relativePoseWithDepth(relative_R, relative_T);
        if (l == -2) { init_rc = gf::set_err(GF_ERR_INVALID, "initialStructure: solvePnPRansac found no model (the reference reads unset matrices here)"); return false; }
        if (l < 0) return false;
        std::vector<std::array<double, 4>> Q;
        std::vector<gfinit::P3> T;
        std::map<int, gfinit::P3> tracked;
        const double rT[3] = {relative_T.x, relative_T.y, relative_T.z};
        if (!gfinit::construct_with_depth(frame_count + 1, l, relative_R.m, rT, sfm_f, Q, T, tracked)) { marginalization_flag = MARGIN_OLD; return false; }
        init_l = l; init_points = (int)tracked.size();
        int i = 0;
        for (auto& kv : all_image_frame) {   // solve pnp for all frame, :1749-1813
            ImageFrame& fr = kv.second;
            M3 Ri;
            gfinit::quat_rot(Q[i].data(), Ri.m);
            if (kv.first == Headers[i]) {
                fr.is_key_frame = true;
                fr.R = Ri * transpose(ric);
                fr.T = v3(T[i][0], T[i][1], T[i][2]);
                i++;
                continue;
            }
            if (kv.first > Headers[i]) { i++; gfinit::quat_rot(Q[i].data(), Ri.m); }
            const M3 R_initial = transpose(Ri);
            const V3 P_initial = (R_initial * v3(T[i][0], T[i][1], T[i][2])) * -1.0;
            fr.is_key_frame = false;
            std::vector<gfinit::P3> X; std::vector<gfinit::P2> uv;
            for (auto& o : fr.points) {
                auto it = tracked.find(o.id);
                if (it != tracked.end()) { X.push_back(it->second); uv.push_back({o.v[0], o.v[1]}); }
            }
            if (X.size() < 6) return false;
            double rv[3], tv[3] = {P_initial.x, P_initial.y, P_initial.z};
            gfinit::rodrigues_inv(R_initial.m, rv);
            if (!gfinit::solve_pnp_iterative(X, uv, rv, tv, true)) return false;
            M3 r;
            gfinit::rodrigues(rv, r.m);
            const M3 R_pnp = transpose(r);
            fr.R = R_pnp * transpose(ric);
            fr.T = R_pnp * (arr3(tv) * -1.0);
        }
        if (visualInitialAlign()) {
            const V3 G = v3(0, 0, cfg.g_norm);
            const V3 tmp_Bas = aver_g - transpose(g2R(aver_g)) * G;
            for (int k = 0; k <= WINDOW_SIZE; k++) Bas[k] = tmp_Bas;
            return true;
        }
        return false;
    }
    bool initialStructure() {  // EST:1557-1682 here; the SfM / PnP branch (EST:1684-1847) in initialStructureSfM
        V3 aver_g;
        {
            const int n = (int)all_image_frame.size() - 1;
            V3 sum_g = v3(0, 0, 0);
            bool first = true;
            for (auto& kv : all_image_frame) { if (first) { first = false; continue; } kv.second.pre_integration->eval(imu_noise); sum_g = sum_g + arr3(kv.second.pre_integration->delta_v) / kv.second.pre_integration->sum_dt; }
            aver_g = sum_g * 1.0 / (double)n;
            double var = 0;
            first = true;
            for (auto& kv : all_image_frame) { if (first) { first = false; continue; } var += sqn(arr3(kv.second.pre_integration->delta_v) / kv.second.pre_integration->sum_dt - aver_g); }
            var = sqrt(var / (double)n);
            if (!(var < 0.35)) is_imu_excited = true;
        }
        const V3 G = v3(0, 0, cfg.g_norm);
        if (!Bas_calibok && systemstationary && solver_flag != NON_LINEAR) {
            const V3 tmp_Bas = aver_g - transpose(g2R(aver_g)) * G;   // R.inverse() of a rotation
            for (int i = 0; i <= WINDOW_SIZE; i++) Bas[i] = tmp_Bas;
            Bas_calibok = true;
            solveGyroscopeBias();
            return true;
        }
        if (!Bas_calibok && is_imu_excited) {
            const V3 tmp_Bas = aver_g - transpose(g2R(aver_g)) * G;
            for (int i = 0; i <= WINDOW_SIZE; i++) Bas[i] = tmp_Bas;
            Bas_calibok = true;
            solveGyroscopeBias();
            M3 R0 = g2R(g);
            const V3 ypr = R2ypr(R0 * Rs[0]);
            R0 = ypr2R(v3(-ypr.x, -ypr.y, -ypr.z)) * R0;
            g = R0 * g;
            for (int i = 0; i <= frame_count; i++) { Ps[i] = R0 * Ps[i]; Rs[i] = R0 * Rs[i]; Vs[i] = R0 * Vs[i]; }
            return true;
        }
        return initialStructureSfM(aver_g);
    }

    // ------------------------------------------------------------ processImage
    void processImageHead(const std::vector<gf_feature_obs>& image, double header) {  // EST:843-905: everything before the solver_flag switch
        marginalization_flag = f_manager.addFeatureCheckParallax(frame_count, image.data(), (int)image.size(), td) ? MARGIN_OLD : MARGIN_SECOND_NEW;
        Headers[frame_count] = header;
        ImageFrame imageframe;
        imageframe.pre_integration = tmp_pre_integration; imageframe.pre_integration_wheel = tmp_wheel_pre_integration;
        if (solver_flag == INITIAL) imageframe.points = image;   // read by initialStructure only
        all_image_frame.insert(std::make_pair(header, imageframe));
        tmp_pre_integration = std::make_shared<ImuPre>(acc_0, gyr_0, Bas[frame_count], Bgs[frame_count]);
        tmp_wheel_pre_integration = std::make_shared<WheelPre>(vel_0_wheel, gyr_0_wheel, sx, sy, sw, td_wheel);
        if (group && group->pre) {   // this frame's IMU intervals (the frame's own and the window's) join the group's batched launch; a failure leaves them to the host loop
            BatchSolver::Req r{};
            r.kind = 2; r.noise = imu_noise; r.rc = GF_OK;
            std::set<ImuPre*> uniq;
            auto want = [&](const std::shared_ptr<ImuPre>& p) { if (p && p->dirty && uniq.insert(p.get()).second) r.pres.push_back(p.get()); };
            want(all_image_frame[header].pre_integration);
            for (int i = 0; i <= frame_count; i++) want(pre_integrations[i]);
            head_rc = group->submit(r);
        }
        checkimu();
        imustationary = varstationary && preintegrationstationary;
        if (checkvisual()) visualstationary = true;
        systemstationary = (imustationary && wheelstationary) || (visualstationary && wheelstationary) || (imustationary && visualstationary);
        predict_ids.clear(); predict_xyz.clear(); remove_ids.clear();
    }
    int processImage(const std::vector<gf_feature_obs>& image, double header) {  // EST:843-1163
        processImageHead(image, header);
        if (head_rc != GF_OK) return head_rc;
        if (solver_flag == INITIAL) {
            if (frame_count == WINDOW_SIZE) {  // DEPTH && USE_IMU branch, EST:967-1037
                int i = 0;
                for (auto& kv : all_image_frame) { if (i <= WINDOW_SIZE) { kv.second.R = Rs[i]; kv.second.T = Ps[i]; } i++; }
                bool result = false;
                if (header - initial_timestamp > 0.1) { result = initialStructure(); initial_timestamp = header; }
                if (init_rc != GF_OK) return init_rc;
                if (result) {
                    solveGyroscopeBias();
                    for (int k = 0; k <= WINDOW_SIZE; k++) pre_integrations[k]->repropagate(v3(0, 0, 0), Bgs[k]);
                    solver_flag = NON_LINEAR;
                    if (int rc = optimization()) return rc;
                    afterOptimizationGNSS();
                    slideWindow();
                } else {
                    if (int rc = optimization()) return rc;
                    slideWindow();
                    updateLatestStates();   // EST:1030-1033: only on the branch without a successful initialisation
                }
            }
            if (frame_count < WINDOW_SIZE) {
                frame_count++;
                const int prev = frame_count - 1;
                Ps[frame_count] = Ps[prev]; Vs[frame_count] = Vs[prev]; Rs[frame_count] = Rs[prev]; Bas[frame_count] = Bas[prev]; Bgs[frame_count] = Bgs[prev];
            }
        } else {
            // member of a group with the device sweeps on: the two per-feature loops of all members run as one launch each (same bits as the host loops)
            const bool dev_sweeps = group && group->sweeps;
            if (dev_sweeps) { if (int rc = deviceSweep(0, nullptr)) return rc; }
            else f_manager.triangulateWithDepth(Ps.data(), Rs.data(), tic, ric);   // DEPTH, EST:1086-1089
            f_manager.triangulate(Ps.data(), Rs.data(), tic, ric);            // EST:1101
            std::set<int> removeIndex;
            if (cfg.use_mcc) {
                if (dev_sweeps) { if (int rc = deviceSweep(1, &removeIndex)) return rc; } else movingConsistencyCheckW(removeIndex);
                f_manager.removeOutlier(removeIndex);
            }
            if (int rc = optimization()) return rc;
            afterOptimizationGNSS();
            if (!cfg.use_mcc) {   // shadowing set, SURVEY.md quirk 13
                std::set<int> inner;
                if (dev_sweeps) { if (int rc = deviceSweep(1, &inner)) return rc; } else movingConsistencyCheckW(inner);
                f_manager.removeOutlier(inner);
            }
            if (!cfg.multiple_thread) {
                remove_ids.assign(removeIndex.begin(), removeIndex.end());
                predictPtsInNextFrame();
                if (tracker) {
                    if (int rc = gf_tracker_remove_outliers(tracker, 0, remove_ids.data(), (int)remove_ids.size())) return rc;
                    if (int rc = gf_tracker_set_prediction(tracker, 0, predict_ids.data(), predict_xyz.data(), (int)predict_ids.size())) return rc;
                }
            }
            slideWindow();
            f_manager.removeFailures();
            last_R = Rs[WINDOW_SIZE]; last_P = Ps[WINDOW_SIZE]; last_R0 = Rs[0]; last_P0 = Ps[0];
            updateLatestStates();   // EST:1161
        }
        return GF_OK;
    }

    // ------------------------------------------------------------ optimisation
    void vector2double() {  // EST:2276-2353
        for (int i = 0; i <= WINDOW_SIZE; i++) {
            double* p = &para_Pose[7 * i];
            p[0] = Ps[i].x; p[1] = Ps[i].y; p[2] = Ps[i].z;
            const Q4 q = rot_to_quat(Rs[i]);
            p[3] = q.x; p[4] = q.y; p[5] = q.z; p[6] = q.w;
            double* s = &para_SpeedBias[9 * i];
            s[0] = Vs[i].x; s[1] = Vs[i].y; s[2] = Vs[i].z; s[3] = Bas[i].x; s[4] = Bas[i].y; s[5] = Bas[i].z; s[6] = Bgs[i].x; s[7] = Bgs[i].y; s[8] = Bgs[i].z;
        }
        Q4 q = rot_to_quat(ric);
        para_Ex_Pose[0] = tic.x; para_Ex_Pose[1] = tic.y; para_Ex_Pose[2] = tic.z; para_Ex_Pose[3] = q.x; para_Ex_Pose[4] = q.y; para_Ex_Pose[5] = q.z; para_Ex_Pose[6] = q.w;
        q = rot_to_quat(rio);
        para_Ex_Pose_wheel[0] = tio.x; para_Ex_Pose_wheel[1] = tio.y; para_Ex_Pose_wheel[2] = tio.z;
        para_Ex_Pose_wheel[3] = q.x; para_Ex_Pose_wheel[4] = q.y; para_Ex_Pose_wheel[5] = q.z; para_Ex_Pose_wheel[6] = q.w;
        para_Ix[0] = sx; para_Ix[1] = sy; para_Ix[2] = sw;
        std::vector<double> dep;
        f_manager.getDepthVector(dep);
        if (dep.size() > para_Feature.size()) para_Feature.resize(dep.size());
        std::copy(dep.begin(), dep.end(), para_Feature.begin());
        para_Td[0] = td; para_Td_wheel[0] = td_wheel;
        if (gnss_ready) { para_yaw_enu_local[0] = yaw_enu_local; para_anc_ecef[0] = anc_ecef.x; para_anc_ecef[1] = anc_ecef.y; para_anc_ecef[2] = anc_ecef.z; }   // :2347-2352
    }
    int double2vector() {  // EST:2440-2569
        std::vector<double> R(9 * (WINDOW_SIZE + 1)), P(3 * (WINDOW_SIZE + 1)), V(P.size()), Ba(P.size()), Bg(P.size());
        if (int rc = gf_ba_double2vector(WINDOW_SIZE, Rs[0].m, &Ps[0].x, para_Pose.data(), para_SpeedBias.data(), R.data(), P.data(), V.data(), Ba.data(), Bg.data())) return rc;
        for (int i = 0; i <= WINDOW_SIZE; i++) { Rs[i] = arr9(&R[9 * i]); Ps[i] = arr3(&P[3 * i]); Vs[i] = arr3(&V[3 * i]); Bas[i] = arr3(&Ba[3 * i]); Bgs[i] = arr3(&Bg[3 * i]); }
        tic = arr3(para_Ex_Pose); ric = qmat(Q4{para_Ex_Pose[6], para_Ex_Pose[3], para_Ex_Pose[4], para_Ex_Pose[5]});
        if (cfg.use_wheel) {
            tio = arr3(para_Ex_Pose_wheel);
            rio = qmat(qnormalized(Q4{para_Ex_Pose_wheel[6], para_Ex_Pose_wheel[3], para_Ex_Pose_wheel[4], para_Ex_Pose_wheel[5]}));
            sx = para_Ix[0]; sy = para_Ix[1]; sw = para_Ix[2]; td_wheel = para_Td_wheel[0];
        }
        f_manager.setDepth(para_Feature.data());
        td = para_Td[0];
        if (gnss_ready) { yaw_enu_local = para_yaw_enu_local[0]; anc_ecef = arr3(para_anc_ecef); R_ecef_enu = ecef2rotation(anc_ecef); }   // :2562-2568
        return GF_OK;
    }
    int optimization() {  // EST:2890-3636
        if (debug_skip_solve) return GF_OK;
        if (!ba && !group) {
            gf_ba_cfg bc{WINDOW_SIZE, cfg.max_features, cfg.max_visual, 1, cfg.gnss_enable ? cfg.max_gnss_per_frame * (WINDOW_SIZE + 1) : 0};
            if (int rc = gf_ba_create(&bc, &ba)) return rc;
        }
        lap(0);
        vector2double();
        gf_ba_window w;
        memset(&w, 0, sizeof(w));
        w.W = WINDOW_SIZE; w.G[0] = g.x; w.G[1] = g.y; w.G[2] = g.z; w.vis_sqrt_info = cfg.focal_length / 1.5;
        const bool moving = norm(Vs[0]) > 0.2;
        if ((cfg.estimate_extrinsic && frame_count == WINDOW_SIZE && moving) || openExEstimation) openExEstimation = 1; else w.fix_ex_pose = 1;            // EST:2990-2999
        const bool wheel_on = cfg.use_wheel && !cfg.only_initial_with_wheel;
        if (wheel_on) {
            if ((cfg.estimate_wheel_extrinsic && frame_count == WINDOW_SIZE && moving) || openExWheelEstimation) openExWheelEstimation = 1; else w.fix_ex_wheel = 1;  // :3032-3041
            if ((cfg.estimate_wheel_intrinsic && frame_count == WINDOW_SIZE && moving) || openIxEstimation) openIxEstimation = 1; else w.fix_ix = 1;                 // :3045-3056
        } else { w.fix_ex_wheel = 1; w.fix_ix = 1; }
        // PoseSubsetParameterization of the two extrinsics (EST:2969-2985, :3010-3026): chosen by ESTIMATE_EXTRINSIC[_WHEEL], whether or not the block is free yet
        w.ex_pose_mask = cfg.estimate_extrinsic ? gf_pose_subset_mask(cfg.extrinsic_type) : 0;
        w.ex_wheel_mask = (wheel_on && cfg.estimate_wheel_extrinsic) ? gf_pose_subset_mask(cfg.extrinsic_type_wheel) : 0;
        w.fix_td = (!cfg.estimate_td || norm(Vs[0]) < 0.2) ? 1 : 0;                                                                                      // :3097-3100
        w.fix_td_wheel = (!cfg.estimate_td_wheel || norm(Vs[0]) < 0.2) ? 1 : 0;
        if (gnss_ready) {   // :2904-2941: the GNSS blocks; yaw_enu_local is held constant, lowspeed drops the factors of this solve
            double ax = 0, ay = 0;
            for (int i = 0; i <= WINDOW_SIZE; i++) { ax += fabs(Vs[i].x); ay += fabs(Vs[i].y); }
            ax /= WINDOW_SIZE + 1; ay /= WINDOW_SIZE + 1;
            lowspeed = sqrt(ax * ax + ay * ay) < 0.3;
        }
        if (first_optimization && cfg.gnss_enable) {   // :2943-2951 PoseAnchorFactor on the values of para_Pose[0]
            w.has_anchor = 1; memcpy(w.anchor_value, para_Pose.data(), 56);
            first_optimization = false;
        }
        if (cfg.gnss_enable) {
            gn_frame.clear(); gn_lower.clear(); gn_sys.clear(); gn_ratio.clear(); gn_data.clear();
            if (gnss_ready)   // :3178-3210 (built whenever gnss_ready: the MARGIN_OLD marginalisation takes the factors of frame 0 even when lowspeed, :3398)
                for (int i = 0; i <= WINDOW_SIZE; i++)
                    for (const gf_gnss_obs& o : gnss_meas_buf[i]) {
                        const double obs_local_ts = o.time - diff_t_gnss_local;
                        const int lower_idx = Headers[i] > obs_local_ts ? (i == 0 ? 0 : i - 1) : (i == WINDOW_SIZE ? WINDOW_SIZE - 1 : i);
                        const double lower_ts = Headers[lower_idx], upper_ts = Headers[lower_idx + 1];
                        gn_frame.push_back(i); gn_lower.push_back(lower_idx); gn_sys.push_back(o.sys); gn_ratio.push_back((upper_ts - obs_local_ts) / (upper_ts - lower_ts));
                        gn_data.insert(gn_data.end(), {o.sv_pos[0], o.sv_pos[1], o.sv_pos[2], o.sv_vel[0], o.sv_vel[1], o.sv_vel[2], o.svdt, o.svddt, o.tgd, o.pr_uura, o.dp_uura,
                                                       o.psr, o.dopp, o.wavelength, o.tow, 0.0});
                    }
            w.gnss_enabled = gnss_ready ? 1 : 0; w.gnss_lowspeed = lowspeed ? 1 : 0; w.n_gnss = (int)gn_frame.size();
            w.para_rcv_dt = para_rcv_dt.data(); w.para_rcv_ddt = para_rcv_ddt.data(); w.para_yaw_enu_local = para_yaw_enu_local; w.para_anc_ecef = para_anc_ecef;
            w.gnss_ddt_weight = 1.0 / cfg.gnss_ddt_sigma; w.gnss_iono = gnss_iono.data();
            w.gnss_frame = gn_frame.data(); w.gnss_lower = gn_lower.data(); w.gnss_sys = gn_sys.data(); w.gnss_ratio = gn_ratio.data(); w.gnss_data = gn_data.data();
            w.gnss_headers = Headers.data();
        }
        // IMU factors :3109-3119
        // the window's tables live in vectors of the estimator that keep their capacity from frame to frame (no allocation in the steady state)
        WinScratch& S_ = scratch; S_.clear();
        std::vector<int>&imu_i = S_.imu_i, &wh_i = S_.wh_i;
        std::vector<double>&imu_sum_dt = S_.imu_sum_dt, &imu_dp = S_.imu_dp, &imu_dq = S_.imu_dq, &imu_dv = S_.imu_dv, &imu_ba = S_.imu_ba, &imu_bg = S_.imu_bg, &imu_J = S_.imu_J, &imu_P = S_.imu_P;
        for (int i = 0; i < frame_count; i++) {
            ImuPre& p = *pre_integrations[i + 1];
            if (int rc = p.eval(imu_noise)) return rc;
            if (p.sum_dt > 10.0) continue;
            imu_i.push_back(i); imu_sum_dt.push_back(p.sum_dt);
            imu_dp.insert(imu_dp.end(), p.delta_p, p.delta_p + 3); imu_dq.insert(imu_dq.end(), p.delta_q, p.delta_q + 4); imu_dv.insert(imu_dv.end(), p.delta_v, p.delta_v + 3);
            imu_ba.insert(imu_ba.end(), {p.lin_ba.x, p.lin_ba.y, p.lin_ba.z}); imu_bg.insert(imu_bg.end(), {p.lin_bg.x, p.lin_bg.y, p.lin_bg.z});
            imu_J.insert(imu_J.end(), p.jacobian.begin(), p.jacobian.end()); imu_P.insert(imu_P.end(), p.covariance.begin(), p.covariance.end());
        }
        // wheel factors :3120-3151
        std::vector<double>&wh_sum_dt = S_.wh_sum_dt, &wh_dp = S_.wh_dp, &wh_dq = S_.wh_dq, &wh_J = S_.wh_J, &wh_P = S_.wh_P, &wh_lin = S_.wh_lin, &wh_lv = S_.wh_lv, &wh_lg = S_.wh_lg, &wh_v1 = S_.wh_v1, &wh_g1 = S_.wh_g1;
        if (wheel_on)
            for (int i = 0; i < frame_count; i++) {
                WheelPre& p = *pre_integrations_wheel[i + 1];
                if (int rc = p.eval(wheel_noise)) return rc;
                if (p.sum_dt > 10.0) continue;
                if (cfg.wdetect && wheelanomaly) continue;
                wh_i.push_back(i); wh_sum_dt.push_back(p.sum_dt);
                wh_dp.insert(wh_dp.end(), p.delta_p, p.delta_p + 3); wh_dq.insert(wh_dq.end(), p.delta_q, p.delta_q + 4);
                wh_J.insert(wh_J.end(), p.jacobian, p.jacobian + 18); wh_P.insert(wh_P.end(), p.covariance, p.covariance + 36); wh_lin.insert(wh_lin.end(), p.lin, p.lin + 4);
                const V3 v1 = p.vel_1(), g1 = p.gyr_1();
                wh_lv.insert(wh_lv.end(), {p.vel0.x, p.vel0.y, p.vel0.z}); wh_lg.insert(wh_lg.end(), {p.gyr0.x, p.gyr0.y, p.gyr0.z});
                wh_v1.insert(wh_v1.end(), {v1.x, v1.y, v1.z}); wh_g1.insert(wh_g1.end(), {g1.x, g1.y, g1.z});
            }
        // stationary: zero the velocities, hold every pose / speed-bias block :3233-3246
        if (systemstationary && cfg.stationary_detect) {
            for (int i = 0; i <= WINDOW_SIZE; i++) { para_SpeedBias[9 * i] = 0; para_SpeedBias[9 * i + 1] = 0; para_SpeedBias[9 * i + 2] = 0; }
            w.fix_poses = 1;
        }
        // visual factors :3262-3297
        std::vector<int>&vf = S_.vf, &vi = S_.vi, &vj = S_.vj; std::vector<double>&vpi = S_.vpi, &vpj = S_.vpj, &vvi = S_.vvi, &vvj = S_.vvj, &vti = S_.vti, &vtj = S_.vtj; std::vector<unsigned char>& fixed = S_.fixed;
        int feature_index = -1;
        for (auto& it : f_manager.feature) {
            it.used_num = (int)it.feature_per_frame.size();
            if (it.used_num < 4) continue;
            ++feature_index;
            fixed.push_back(it.estimate_flag == 1 ? 1 : 0);
            const int imu_i_ = it.start_frame; int imu_j_ = imu_i_ - 1;
            const FeaturePerFrame& f0 = it.feature_per_frame[0];
            for (auto& fr : it.feature_per_frame) {
                imu_j_++;
                if (imu_i_ == imu_j_) continue;
                vf.push_back(feature_index); vi.push_back(imu_i_); vj.push_back(imu_j_);
                vpi.insert(vpi.end(), {f0.point.x, f0.point.y, f0.point.z}); vpj.insert(vpj.end(), {fr.point.x, fr.point.y, fr.point.z});
                vvi.insert(vvi.end(), {f0.velocity[0], f0.velocity[1]}); vvj.insert(vvj.end(), {fr.velocity[0], fr.velocity[1]});
                vti.push_back(f0.cur_td); vtj.push_back(fr.cur_td);
            }
        }
        const int nf = feature_index + 1;
        if (nf > cfg.max_features || (int)vf.size() > cfg.max_visual)
            return gf::set_err(GF_ERR_CAPACITY, "window has %d features / %d visual factors, capacity %d / %d", nf, (int)vf.size(), cfg.max_features, cfg.max_visual);
        w.n_feature = nf; w.n_visual = (int)vf.size(); w.n_imu = (int)imu_i.size(); w.n_wheel = (int)wh_i.size();
        w.para_Pose = para_Pose.data(); w.para_SpeedBias = para_SpeedBias.data(); w.para_Ex_Pose = para_Ex_Pose; w.para_Ex_Pose_wheel = para_Ex_Pose_wheel;
        w.para_Ix = para_Ix; w.para_Td = para_Td; w.para_Td_wheel = para_Td_wheel; w.para_Feature = para_Feature.data(); w.feature_fixed = fixed.data();
        w.vis_feature = vf.data(); w.vis_i = vi.data(); w.vis_j = vj.data(); w.vis_pts_i = vpi.data(); w.vis_pts_j = vpj.data(); w.vis_vel_i = vvi.data(); w.vis_vel_j = vvj.data();
        w.vis_td_i = vti.data(); w.vis_td_j = vtj.data();
        w.imu_i = imu_i.data(); w.imu_sum_dt = imu_sum_dt.data(); w.imu_delta_p = imu_dp.data(); w.imu_delta_q = imu_dq.data(); w.imu_delta_v = imu_dv.data();
        w.imu_lin_ba = imu_ba.data(); w.imu_lin_bg = imu_bg.data(); w.imu_jacobian = imu_J.data(); w.imu_covariance = imu_P.data();
        w.wh_i = wh_i.data(); w.wh_sum_dt = wh_sum_dt.data(); w.wh_delta_p = wh_dp.data(); w.wh_delta_q = wh_dq.data(); w.wh_jacobian = wh_J.data(); w.wh_covariance = wh_P.data();
        w.wh_lin = wh_lin.data(); w.wh_lin_vel = wh_lv.data(); w.wh_lin_gyr = wh_lg.data(); w.wh_vel_1 = wh_v1.data(); w.wh_gyr_1 = wh_g1.data();
        if (prior_valid) { w.prior_n = prior_n; w.prior_nblocks = (int)prior_block_id.size(); w.prior_block_id = prior_block_id.data(); w.prior_J = prior_resident ? nullptr : prior_J.data(); w.prior_r = prior_r.data(); w.prior_x0 = prior_x0.data(); }
        lap(1);
        if (!group && cfg.max_solver_time > 0) gf_ba_set_max_solver_time(ba, marginalization_flag == MARGIN_OLD ? cfg.max_solver_time * 4.0 / 5.0 : cfg.max_solver_time);   // EST:3312-3315
        if (group) {   // ceres::Solve, EST:3303-3318
            // this thread packs its own window into the shared staging tables, the rendezvous only uploads and launches, and the result is unpacked here again
            const int prc = gf_ba_pack_slot(group->ba, group_slot, &w);
            BatchSolver::Req rq{0, &w, cfg.num_iterations, 0, &last_summary, nullptr, GF_OK, false, std::string()};
            rq.slot = group_slot;
            if (prc != GF_OK) { const std::string msg = gf_last_error(); group->leave_with_error(); return gf::set_err(prc, "%s", msg.c_str()); }
            if (int rc = group->submit(rq)) return rc;
            if (int rc = gf_ba_unpack_slot(group->ba, group_slot, &w, &last_summary)) return rc;
        } else if (int rc = gf_ba_solve(ba, &w, 1, cfg.num_iterations, &last_summary)) return rc;
        lap(2);
        n_optimizations++;
        while (para_yaw_enu_local[0] > M_PI) para_yaw_enu_local[0] -= 2.0 * M_PI;           // :3322-3325
        while (para_yaw_enu_local[0] < -M_PI) para_yaw_enu_local[0] += 2.0 * M_PI;
        if (int rc = double2vector()) return rc;                                            // :3327
        if (frame_count < WINDOW_SIZE) { wheelanomaly = false; return GF_OK; }
        bool run_marg = marginalization_flag == MARGIN_OLD;
        if (!run_marg) {  // :3538-3539: only when the prior mentions the second-newest pose
            run_marg = prior_valid && std::count(prior_block_id.begin(), prior_block_id.end(), GF_POSE * 4096 + WINDOW_SIZE - 1) > 0;
            // the reference tests last_marginalization_info != nullptr only; an invalid info still owns its block list
        }
        if (run_marg) {
            vector2double();                                                                // :3337 / :3543
            if (systemstationary && cfg.stationary_detect) { /* para_SpeedBias already refreshed from Vs by vector2double */ }
            const int gx = cfg.gnss_enable ? 5 * (WINDOW_SIZE + 1) + 4 : 0;
            const int cap_n = 16 * (WINDOW_SIZE + 1) + 64 + gx, cap_b = 2 * (WINDOW_SIZE + 1) + 16 + gx;
            // receive buffers of the prior: members of the estimator, sized once (half a megabyte of zero-filled std::vector per frame and sequence otherwise)
            if (marg_J.size() != (size_t)cap_n * cap_n) { marg_J.assign((size_t)cap_n * cap_n, 0.0); marg_r.assign(cap_n, 0.0); marg_x0.assign(16 * (WINDOW_SIZE + 1) + 64 + gx, 0.0); marg_id.assign(cap_b, 0); }
            std::vector<double>&pJ = marg_J, &pr = marg_r, &px0 = marg_x0; std::vector<int>& pid = marg_id;
            gf_ba_prior p{};
            p.cap_n = cap_n; p.cap_blocks = cap_b; p.block_id = pid.data(); p.J = group ? nullptr : pJ.data(); p.r = pr.data(); p.x0 = px0.data();   // group: J stays on the device
            lap(3);
            if (group) {
                BatchSolver::Req rq{1, &w, 0, marginalization_flag, nullptr, &p, GF_OK, false, std::string()};
                rq.slot = group_slot;
                if (int rc = group->submit(rq)) return rc;
                if (int rc = gf_ba_unpack_prior_slot(group->ba, group_slot, marginalization_flag, &p)) return rc;
            } else if (int rc = gf_ba_marginalize(ba, &w, 1, marginalization_flag, &p)) return rc;
            lap(4);
            prior_valid = p.valid != 0;
            if (p.valid) {
                prior_n = p.n;
                prior_block_id.assign(pid.begin(), pid.begin() + p.nblocks);
                prior_resident = group != nullptr;
                if (prior_resident) prior_J.clear(); else prior_J.assign(pJ.begin(), pJ.begin() + (size_t)p.n * p.n);
                prior_r.assign(pr.begin(), pr.begin() + p.n);
                int gs = 0;
                for (int id : prior_block_id) { const int k = id / 4096; gs += (k == GF_POSE || k == GF_EX_POSE || k == GF_EX_WHEEL) ? 7 : k == GF_SPEEDBIAS ? 9 : k == GF_ANC ? 3 : 1; }
                prior_x0.assign(px0.begin(), px0.begin() + gs);
            }
        }
        wheelanomaly = false;   // :3633
        return GF_OK;
    }

    // ------------------------------------------------------------ window bookkeeping
    void slideWindow() {  // EST:3638-3790
        if (debug_skip_solve) return;
        if (marginalization_flag == MARGIN_OLD) {
            const double t_0 = Headers[0];
            back_R0 = Rs[0]; back_P0 = Ps[0];
            if (frame_count != WINDOW_SIZE) return;
            for (int i = 0; i < WINDOW_SIZE; i++) {
                Headers[i] = Headers[i + 1];
                std::swap(Rs[i], Rs[i + 1]); std::swap(Ps[i], Ps[i + 1]);
                std::swap(pre_integrations[i], pre_integrations[i + 1]);
                std::swap(Vs[i], Vs[i + 1]); std::swap(Bas[i], Bas[i + 1]); std::swap(Bgs[i], Bgs[i + 1]);
                if (cfg.use_wheel) std::swap(pre_integrations_wheel[i], pre_integrations_wheel[i + 1]);
                if (cfg.gnss_enable) {   // :3674-3681
                    gnss_meas_buf[i].swap(gnss_meas_buf[i + 1]);
                    for (int k = 0; k < 4; k++) para_rcv_dt[4 * i + k] = para_rcv_dt[4 * (i + 1) + k];
                    para_rcv_ddt[i] = para_rcv_ddt[i + 1];
                }
            }
            if (cfg.gnss_enable) gnss_meas_buf[WINDOW_SIZE].clear();   // :3700-3704
            Headers[WINDOW_SIZE] = Headers[WINDOW_SIZE - 1]; Ps[WINDOW_SIZE] = Ps[WINDOW_SIZE - 1]; Rs[WINDOW_SIZE] = Rs[WINDOW_SIZE - 1];
            Vs[WINDOW_SIZE] = Vs[WINDOW_SIZE - 1]; Bas[WINDOW_SIZE] = Bas[WINDOW_SIZE - 1]; Bgs[WINDOW_SIZE] = Bgs[WINDOW_SIZE - 1];
            pre_integrations[WINDOW_SIZE] = std::make_shared<ImuPre>(acc_0, gyr_0, Bas[WINDOW_SIZE], Bgs[WINDOW_SIZE]);
            if (cfg.use_wheel) pre_integrations_wheel[WINDOW_SIZE] = std::make_shared<WheelPre>(vel_0_wheel, gyr_0_wheel, sx, sy, sw, td_wheel);
            auto it_0 = all_image_frame.find(t_0);
            if (it_0 != all_image_frame.end()) { it_0->second.pre_integration.reset(); it_0->second.pre_integration_wheel.reset(); all_image_frame.erase(all_image_frame.begin(), it_0); }
            slideWindowOld();
        } else {
            if (frame_count != WINDOW_SIZE) return;
            Headers[frame_count - 1] = Headers[frame_count]; Ps[frame_count - 1] = Ps[frame_count]; Rs[frame_count - 1] = Rs[frame_count];
            {
                if (pre_integrations[frame_count] && pre_integrations[frame_count - 1]) {  // always true once IMU data arrived for the newest frame
                    ImuPre& src = *pre_integrations[frame_count]; ImuPre& dst = *pre_integrations[frame_count - 1];
                    for (size_t i = 0; i < src.dt.size(); i++) dst.push_back(src.dt[i], arr3(&src.acc[3 * i]), arr3(&src.gyr[3 * i]));
                }
                Vs[frame_count - 1] = Vs[frame_count]; Bas[frame_count - 1] = Bas[frame_count]; Bgs[frame_count - 1] = Bgs[frame_count];
                pre_integrations[WINDOW_SIZE] = std::make_shared<ImuPre>(acc_0, gyr_0, Bas[WINDOW_SIZE], Bgs[WINDOW_SIZE]);
            }
            if (cfg.use_wheel) {
                if (pre_integrations_wheel[frame_count] && pre_integrations_wheel[frame_count - 1]) {
                    WheelPre& src = *pre_integrations_wheel[frame_count]; WheelPre& dst = *pre_integrations_wheel[frame_count - 1];
                    for (size_t i = 0; i < src.dt.size(); i++) dst.push_back(src.dt[i], arr3(&src.vel[3 * i]), arr3(&src.gyr[3 * i]));
                }
                pre_integrations_wheel[WINDOW_SIZE] = std::make_shared<WheelPre>(vel_0_wheel, gyr_0_wheel, sx, sy, sw, td_wheel);
            }
            {   // :3761-3768
                gnss_meas_buf[frame_count - 1] = gnss_meas_buf[frame_count];
                for (int k = 0; k < 4; k++) para_rcv_dt[4 * (frame_count - 1) + k] = para_rcv_dt[4 * frame_count + k];
                para_rcv_ddt[frame_count - 1] = para_rcv_ddt[frame_count];
                gnss_meas_buf[frame_count].clear();
            }
            sum_of_front++;
            f_manager.removeFront(frame_count);   // slideWindowNew, EST:3792-3802
        }
    }
    void slideWindowOld() {  // EST:3804-3837
        sum_of_back++;
        if (solver_flag == NON_LINEAR) {
            const M3 R0 = back_R0 * ric, R1 = Rs[0] * ric;
            const V3 P0 = back_P0 + back_R0 * tic, P1 = Ps[0] + Rs[0] * tic;
            f_manager.removeBackShiftDepth(R0, P0, R1, P1);
        } else f_manager.removeBack();
    }
    double reprojectionError(const M3& Ri, V3 Pi, const M3& Rj, V3 Pj, double depth, V3 uvi, V3 uvj) const {  // EST:3899-3910
        const V3 pts_w = Ri * (ric * (uvi * depth) + tic) + Pi;
        const V3 pts_cj = transpose(ric) * (transpose(Rj) * (pts_w - Pj) - tic);
        const double rx = pts_cj.x / pts_cj.z - uvj.x, ry = pts_cj.y / pts_cj.z - uvj.y;
        return sqrt(rx * rx + ry * ry);
    }
    double reprojectionError3D(const M3& Ri, V3 Pi, const M3& Rj, V3 Pj, double depth, V3 uvi, V3 uvj) const {  // EST:3912-3919
        const V3 pts_w = Ri * (ric * (uvi * depth) + tic) + Pi;
        const V3 pts_cj = transpose(ric) * (transpose(Rj) * (pts_w - Pj) - tic);
        return norm(pts_cj - uvj) / depth;
    }
    // FeatureManager::triangulateWithDepth (mode 0) or movingConsistencyCheckW (mode 1) of this member as part of the group's batched launch: the window's
    // poses and feature tables go into the member's SweepIn, the rendezvous runs all members' tables at once, the results come back in feature-list order.
    SweepIn sweep_in;
    int deviceSweep(int mode, std::set<int>* removeIndex) {
        SweepIn& in = sweep_in;
        const int NP = WINDOW_SIZE + 1;
        in.W = WINDOW_SIZE;
        in.Rs.resize(9 * NP); in.Ps.resize(3 * NP);
        for (int i = 0; i < NP; i++) { memcpy(&in.Rs[9 * i], Rs[i].m, 72); in.Ps[3 * i] = Ps[i].x; in.Ps[3 * i + 1] = Ps[i].y; in.Ps[3 * i + 2] = Ps[i].z; }
        in.tic[0] = tic.x; in.tic[1] = tic.y; in.tic[2] = tic.z; memcpy(in.ric, ric.m, 72);
        in.start_frame.clear(); in.first_obs.assign(1, 0); in.flag.clear(); in.obs.clear(); in.depth.clear();
        for (auto& it : f_manager.feature) {
            it.used_num = (int)it.feature_per_frame.size();   // both host loops leave this behind
            in.start_frame.push_back(it.start_frame);
            for (const FeaturePerFrame& fr : it.feature_per_frame) in.obs.insert(in.obs.end(), {fr.point.x, fr.point.y, fr.point.z, fr.depth});
            in.first_obs.push_back((int)(in.obs.size() / 4));
            in.depth.push_back(it.estimated_depth); in.flag.push_back(it.estimate_flag);
        }
        in.nf = (int)in.start_frame.size();
        in.remove.assign(in.nf, 0);
        in.depth_threshold = f_manager.depth_threshold; in.init_depth = f_manager.INIT_DEPTH; in.focal_length = cfg.focal_length;
        BatchSolver::Req rq{3, nullptr, 0, mode, nullptr, nullptr, GF_OK, false, std::string()};
        rq.sweep = &in;
        if (int rc = group->submit(rq)) return rc;
        int f = 0;
        for (auto& it : f_manager.feature) {
            if (mode == 0) { it.estimated_depth = in.depth[f]; it.estimate_flag = in.flag[f]; }
            else if (in.remove[f]) removeIndex->insert(it.feature_id);
            f++;
        }
        return GF_OK;
    }
    void movingConsistencyCheckW(std::set<int>& removeIndex) {  // EST:3955-3995
        for (auto& it : f_manager.feature) {
            it.used_num = (int)it.feature_per_frame.size();
            if (!(it.used_num >= 2 && it.start_frame < WINDOW_SIZE - 2)) continue;
            const double depth = it.estimated_depth;
            if (depth < 0) continue;
            double err = 0, err3D = 0; int errCnt = 0;
            const int wi = it.start_frame; int wj = wi - 1;
            const V3 pts_i = it.feature_per_frame[0].point;
            for (auto& fr : it.feature_per_frame) {
                wj++;
                if (wi == wj) continue;
                err += reprojectionError(Rs[wi], Ps[wi], Rs[wj], Ps[wj], depth, pts_i, fr.point);
                err3D += reprojectionError3D(Rs[wi], Ps[wi], Rs[wj], Ps[wj], depth, pts_i, fr.point);
                errCnt++;
            }
            if (errCnt > 0 && (cfg.focal_length * err / errCnt > 10 || err3D / errCnt > 2.0)) removeIndex.insert(it.feature_id);
        }
    }
    void predictPtsInNextFrame() {  // EST:3862-3897; nextT = curT * (prevT^-1 * curT) on rigid transforms
        if (frame_count < 2) return;
        const M3 Rc = Rs[frame_count], Rp = Rs[frame_count - 1];
        const V3 Pc = Ps[frame_count], Pp = Ps[frame_count - 1];
        const M3 Rrel = transpose(Rp) * Rc; const V3 Prel = transpose(Rp) * (Pc - Pp);
        const M3 Rn = Rc * Rrel; const V3 Pn = Rc * Prel + Pc;
        std::map<int, V3> predictPts;
        for (auto& it : f_manager.feature) {
            if (!(it.estimated_depth > 0)) continue;
            const int firstIndex = it.start_frame, lastIndex = it.start_frame + (int)it.feature_per_frame.size() - 1;
            if ((int)it.feature_per_frame.size() >= 2 && lastIndex == frame_count) {
                const V3 pts_j = ric * (it.feature_per_frame[0].point * it.estimated_depth) + tic;
                const V3 pts_w = Rs[firstIndex] * pts_j + Ps[firstIndex];
                const V3 pts_local = transpose(Rn) * (pts_w - Pn);
                predictPts[it.feature_id] = transpose(ric) * (pts_local - tic);
            }
        }
        for (auto& kv : predictPts) { predict_ids.push_back(kv.first); predict_xyz.insert(predict_xyz.end(), {kv.second.x, kv.second.y, kv.second.z}); }
    }
};

// ---------------------------------------------------------------- C-ABI
extern "C" int gf_ba_debug_upload_bytes(long long* bytes, long long* calls);
extern "C" {

int gf_estimator_default_cfg(gf_estimator_cfg* c) {
    if (!c) return gf::set_err(GF_ERR_INVALID, "null cfg");
    memset(c, 0, sizeof(*c));
    c->window_size = 10; c->max_features = 512; c->max_visual = 4096;
    c->use_imu = 1; c->use_wheel = 1; c->depth = 1;                                     // m2dgrp.yaml:4-6
    c->estimate_extrinsic = 0; c->estimate_wheel_extrinsic = 1; c->estimate_wheel_intrinsic = 0; c->estimate_td = 0; c->estimate_td_wheel = 0;
    c->use_mcc = 0; c->wdetect = 1; c->stationary_detect = 1; c->only_initial_with_wheel = 0; c->multiple_thread = 1;
    c->num_iterations = 8;
    c->acc_n = 1.2374091609523514e-02; c->gyr_n = 3.0032654435730201e-03; c->acc_w = 1.9218003442176448e-04; c->gyr_w = 5.4692100664858005e-05;
    c->g_norm = 9.805; c->wheel_vel_n = 0.01; c->wheel_gyr_n = 0.004;
    c->min_parallax_px = 10.0; c->depth_threshold = 3.0; c->init_depth = 5.0; c->focal_length = 600.0;
    c->sx = c->sy = c->sw = 1.0;
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(c->ric, I, sizeof(I));
    const double rio[9] = {0.352551, -0.935764, -0.00734672, 0.0145238, 0.0133214, -0.999806, 0.93568, 0.352375, 0.0182873};   // body_T_wheel, m2dgrp.yaml:107-114
    const double tio[3] = {0.0497956, 1.06332, -0.037465};
    memcpy(c->rio, rio, sizeof(rio)); memcpy(c->tio, tio, sizeof(tio));
    // GNSS off, thresholds as in m2dgrp.yaml:7, :38-48
    c->gnss_enable = 0; c->gnss_track_num_thres = 5; c->max_gnss_per_frame = 32;
    c->gnss_elevation_thres = 30.0; c->gnss_psr_std_thres = 2.0; c->gnss_dopp_std_thres = 2.0; c->gnss_ddt_sigma = 0.1; c->gnss_local_time_diff = 18.0;
    const double iono[8] = {0.1118e-07, 0.2235e-07, -0.4172e-06, 0.6557e-06, 0.1249e+06, -0.4424e+06, 0.1507e+07, -0.2621e+06};
    memcpy(c->gnss_iono, iono, sizeof(iono));
    return GF_OK;
}

int gf_estimator_create(const gf_estimator_cfg* c, gf_estimator** out) {
    if (!c || !out) return gf::set_err(GF_ERR_INVALID, "null argument");
    if (c->window_size < 2 || c->window_size > 30) return gf::set_err(GF_ERR_INVALID, "window_size must be in [2, 30]");
    if (!c->use_imu || !c->depth) return gf::set_err(GF_ERR_INVALID, "only the RGB-D + IMU configuration is built (USE_IMU=1, DEPTH=1)");
    if (c->estimate_extrinsic == 2) return gf::set_err(GF_ERR_INVALID, "ESTIMATE_EXTRINSIC=2 (online rotation calibration) is not built");
    if (c->gnss_enable && (c->max_gnss_per_frame < 1 || !(c->gnss_ddt_sigma > 0))) return gf::set_err(GF_ERR_INVALID, "gnss_enable needs max_gnss_per_frame >= 1 and gnss_ddt_sigma > 0");
    gf_estimator* e = new gf_estimator(*c);
    if (c->with_tracker) {
        gf_tracker_cfg tc = c->tracker; tc.batch = 1;
        if (int rc = gf_tracker_create(&tc, &e->tracker)) { delete e; return rc; }
    }
    *out = e;
    return GF_OK;
}
int gf_estimator_destroy(gf_estimator* e) { delete e; return GF_OK; }
int gf_estimator_set_result_path(gf_estimator* e, const char* vio_txt) {   // VINS_RESULT_PATH, created empty at start-up (parameters.cpp:347-352)
    if (!e) return gf::set_err(GF_ERR_INVALID, "null handle");
    e->result_path = vio_txt ? vio_txt : "";
    if (!e->result_path.empty()) {
        FILE* f = fopen(vio_txt, "w");
        if (!f) { e->result_path.clear(); return gf::set_err(GF_ERR_INVALID, "cannot create %s", vio_txt); }
        fclose(f);
    }
    return GF_OK;
}

int gf_estimator_input_imu(gf_estimator* e, double t, const double* acc, const double* gyr) {  // Estimator::inputIMU EST:330-346
    if (!e || !acc || !gyr) return gf::set_err(GF_ERR_INVALID, "null argument");
    e->accBuf.emplace_back(t, arr3(acc)); e->gyrBuf.emplace_back(t, arr3(gyr));
    e->fastPredictIMU(t, arr3(acc), arr3(gyr));   // EST:332: latest_P / latest_Q / latest_V for pubLatestOdometry (gf_estimator_get_latest)
    if (e->cfg.multiple_thread && !e->featureBuf.empty()) return e->drain();   // a frame was waiting for this sample ("wait for imu ...", EST:551-560)
    return GF_OK;
}
int gf_estimator_input_wheel(gf_estimator* e, double t, const double* vel, const double* gyr) {  // Estimator::inputWheel EST:347-360
    if (!e || !vel || !gyr) return gf::set_err(GF_ERR_INVALID, "null argument");
    e->wheelVelBuf.emplace_back(t, arr3(vel)); e->wheelGyrBuf.emplace_back(t, arr3(gyr));
    e->fastPredictWheel(t, arr3(vel), arr3(gyr));   // EST:363-366
    if (e->cfg.multiple_thread && !e->featureBuf.empty()) return e->drain();   // "wait for wheel ...", EST:562-573
    return GF_OK;
}
int gf_estimator_input_gnss(gf_estimator* e, double t, const gf_gnss_obs* obs, int n) {  // Estimator::inputGNSS EST:397-404
    if (!e || !obs || n < 1) return gf::set_err(GF_ERR_INVALID, "an epoch needs at least one observation (getGNSSInterval reads the first one's time, EST:489)");
    if (!e->cfg.gnss_enable) return gf::set_err(GF_ERR_INVALID, "estimator was created with gnss_enable 0");
    if (n > e->cfg.max_gnss_per_frame) return gf::set_err(GF_ERR_CAPACITY, "epoch with %d observations, max_gnss_per_frame %d", n, e->cfg.max_gnss_per_frame);
    std::vector<gf_estimator::GMsgObs> v(n);
    for (int i = 0; i < n; i++) { v[i].o = obs[i]; v[i].is_raw = false; }
    e->GNSSBuf.emplace_back(t, std::move(v));
    return GF_OK;
}
int gf_estimator_input_gnss_raw(gf_estimator* e, double t, const gf_gnss_raw_obs* obs, int n) {  // Estimator::inputGNSS with the observations as they come off the receiver
    if (!e || !obs || n < 1) return gf::set_err(GF_ERR_INVALID, "an epoch needs at least one observation (getGNSSInterval reads the first one's time, EST:489)");
    if (!e->cfg.gnss_enable) return gf::set_err(GF_ERR_INVALID, "estimator was created with gnss_enable 0");
    if (n > e->cfg.max_gnss_per_frame) return gf::set_err(GF_ERR_CAPACITY, "epoch with %d observations, max_gnss_per_frame %d", n, e->cfg.max_gnss_per_frame);
    std::vector<gf_estimator::GMsgObs> v(n);
    for (int i = 0; i < n; i++) {
        if (!(obs[i].freq > 0)) return gf::set_err(GF_ERR_INVALID, "observation %d: carrier frequency must be positive", i);
        memset(&v[i].o, 0, sizeof(v[i].o));
        v[i].raw = obs[i]; v[i].is_raw = true;
        v[i].o.sat = obs[i].sat; v[i].o.sys = obs[i].sys; v[i].o.time = obs[i].time; v[i].o.psr_std = obs[i].psr_std; v[i].o.dopp_std = obs[i].dopp_std;
    }
    e->GNSSBuf.emplace_back(t, std::move(v));
    return GF_OK;
}
int gf_estimator_input_ephem(gf_estimator* e, const gf_gnss_ephem* eph) {  // Estimator::inputEphem EST:1428-1437: a (satellite, toe) pair is taken once
    if (!e || !eph) return gf::set_err(GF_ERR_INVALID, "null argument");
    if (eph->sys != 0 && eph->sys != 2 && eph->sys != 3) return gf::set_err(GF_ERR_INVALID, "gf_gnss_ephem is for GPS (0), Galileo (2) and BeiDou (3); GLONASS goes through gf_estimator_input_glo_ephem");
    if (!(eph->A > 0)) return gf::set_err(GF_ERR_INVALID, "semi-major axis must be positive");
    auto& idx = e->sat2time_index[eph->sat];
    if (idx.count(eph->toe)) return GF_OK;
    e->sat2ephem[eph->sat].push_back(*eph);
    idx.emplace(eph->toe, e->sat2ephem[eph->sat].size() - 1);
    return GF_OK;
}
int gf_estimator_input_glo_ephem(gf_estimator* e, const gf_gnss_glo_ephem* g) {
    if (!e || !g) return gf::set_err(GF_ERR_INVALID, "null argument");
    auto& idx = e->sat2time_index[g->sat];
    if (idx.count(g->toe)) return GF_OK;
    e->sat2gephem[g->sat].push_back(*g);
    idx.emplace(g->toe, e->sat2gephem[g->sat].size() - 1);
    return GF_OK;
}
int gf_gnss_obs_from_ephem(const gf_gnss_raw_obs* raw, const gf_gnss_ephem* eph, const gf_gnss_glo_ephem* geph, gf_gnss_obs* out) {
    if (!raw || !out || (eph == nullptr) == (geph == nullptr) || !(raw->freq > 0)) return gf::set_err(GF_ERR_INVALID, "bad argument (exactly one ephemeris, positive frequency)");
    gnss_eph::sat_state(*raw, eph, geph, out);
    return GF_OK;
}
int gf_gnss_eph2pos(double t, const gf_gnss_ephem* eph, const gf_gnss_glo_ephem* geph, double* pos3, double* svdt) {
    if (!pos3 || (eph == nullptr) == (geph == nullptr)) return gf::set_err(GF_ERR_INVALID, "bad argument (exactly one ephemeris)");
    const V3 p = eph ? gnss_eph::eph2pos(t, *eph, svdt) : gnss_eph::geph2pos(t, *geph, svdt);
    pos3[0] = p.x; pos3[1] = p.y; pos3[2] = p.z;
    return GF_OK;
}
int gf_estimator_input_gnss_time_diff(gf_estimator* e, double t_diff) {  // EST:1450-1453
    if (!e) return gf::set_err(GF_ERR_INVALID, "null handle");
    e->diff_t_gnss_local = t_diff;
    return GF_OK;
}
int gf_estimator_input_iono_params(gf_estimator* e, const double* p) {  // EST:1438-1448
    if (!e || !p) return gf::set_err(GF_ERR_INVALID, "null argument");
    e->gnss_iono.assign(p, p + 8);
    return GF_OK;
}
int gf_estimator_set_gnss_alignment(gf_estimator* e, const double* anc_ecef, double yaw, const double* rcv_dt4, double rcv_ddt) {
    if (!e || !anc_ecef || !rcv_dt4) return gf::set_err(GF_ERR_INVALID, "null argument");
    if (!e->cfg.gnss_enable) return gf::set_err(GF_ERR_INVALID, "estimator was created with gnss_enable 0");
    memcpy(e->align_anc, anc_ecef, 24); memcpy(e->align_dt, rcv_dt4, 32); e->align_yaw = yaw; e->align_ddt = rcv_ddt; e->align_pending = true;
    return GF_OK;
}
int gf_estimator_get_gnss_state(gf_estimator* e, int* gnss, double* rcv_dt, double* rcv_ddt, double* yaw, double* anc, double* ecef, double* enu) {
    if (!e) return gf::set_err(GF_ERR_INVALID, "null handle");
    if (gnss) { const int v[8] = {e->gnss_ready ? 1 : 0, e->lowspeed ? 1 : 0, (int)e->gnss_meas_buf[e->WINDOW_SIZE].size(), e->first_optimization ? 1 : 0, (int)e->GNSSBuf.size(), 0, 0, 0}; memcpy(gnss, v, sizeof(v)); }
    if (rcv_dt) memcpy(rcv_dt, e->para_rcv_dt.data(), e->para_rcv_dt.size() * 8);
    if (rcv_ddt) memcpy(rcv_ddt, e->para_rcv_ddt.data(), e->para_rcv_ddt.size() * 8);
    if (yaw) *yaw = e->yaw_enu_local;
    if (anc) memcpy(anc, &e->anc_ecef.x, 24);
    if (ecef) memcpy(ecef, &e->ecef_pos.x, 24);
    if (enu) memcpy(enu, &e->enu_pos.x, 24);
    return GF_OK;
}
// Estimator::inputFeature (EST:362-375) + processMeasurements: `obs` is the tracker's map flattened in id order
int gf_estimator_input_feature(gf_estimator* e, double t, const gf_feature_obs* obs, int n) {
    if (!e || (n > 0 && !obs) || n < 0) return gf::set_err(GF_ERR_INVALID, "bad argument");
    std::vector<gf_feature_obs> v(obs, obs + n);
    std::stable_sort(v.begin(), v.end(), [](const gf_feature_obs& a, const gf_feature_obs& b) { return a.id < b.id; });
    e->featureBuf.emplace_back(t, std::move(v));
    return e->cfg.multiple_thread ? e->drain() : e->processMeasurements();   // inline call of the non-threaded mode: one frame, EST:239
}
int gf_estimator_process_image(gf_estimator* e, double header, const gf_feature_obs* obs, int n) {  // Estimator::processImage, estimator.h:110
    if (!e || (n > 0 && !obs) || n < 0) return gf::set_err(GF_ERR_INVALID, "bad argument");
    if (e->cfg.use_imu && !e->first_imu) return gf::set_err(GF_ERR_INVALID, "processImage before any IMU sample was processed (the reference dereferences a null pre-integration here)");
    std::vector<gf_feature_obs> v(obs, obs + n);
    std::stable_sort(v.begin(), v.end(), [](const gf_feature_obs& a, const gf_feature_obs& b) { return a.id < b.id; });
    return e->processImage(v, header);
}
// Estimator::inputImage (EST:213-242): track, then (multiple_thread: every second frame) hand the features to the back end
int gf_estimator_input_image(gf_estimator* e, double t, const uint8_t* gray, int stride, const uint16_t* depth, int dstride, gf_feature_obs* out, int cap, int* n_out) {
    if (!e || !e->tracker) return gf::set_err(GF_ERR_INVALID, "estimator was created without a tracker (cfg.with_tracker)");
    std::vector<gf_feature_obs> tmp;
    if (!out) { cap = e->cfg.tracker.max_cnt + 8; tmp.resize(cap); out = tmp.data(); }
    int n = 0;
    if (int rc = gf_tracker_track(e->tracker, 0, t, gray, stride, depth, dstride, out, cap, &n)) return rc;
    if (n_out) *n_out = n;
    e->inputImageCnt++;
    if (e->cfg.multiple_thread && e->inputImageCnt % 2 != 0) return GF_OK;
    return gf_estimator_input_feature(e, t, out, n);
}

int gf_estimator_get_state(gf_estimator* e, double* Ps, double* Rs, double* Vs, double* Bas, double* Bgs, double* Headers, int* info, double* extr) {
    if (!e) return gf::set_err(GF_ERR_INVALID, "null handle");
    for (int i = 0; i <= e->WINDOW_SIZE; i++) {
        if (Ps) memcpy(Ps + 3 * i, &e->Ps[i].x, 24);
        if (Rs) memcpy(Rs + 9 * i, e->Rs[i].m, 72);
        if (Vs) memcpy(Vs + 3 * i, &e->Vs[i].x, 24);
        if (Bas) memcpy(Bas + 3 * i, &e->Bas[i].x, 24);
        if (Bgs) memcpy(Bgs + 3 * i, &e->Bgs[i].x, 24);
        if (Headers) Headers[i] = e->Headers[i];
    }
    if (info) {
        const int v[16] = {e->frame_count, e->solver_flag, e->marginalization_flag, (int)e->f_manager.feature.size(), e->prior_valid ? 1 : 0, e->prior_valid ? e->prior_n : 0,
                           e->systemstationary ? 1 : 0, e->last_summary.iterations, e->last_summary.successful_steps, (int)e->n_optimizations, e->openExWheelEstimation,
                           e->f_manager.last_track_num, e->f_manager.long_track_num, e->f_manager.new_feature_num, e->sum_of_back, e->sum_of_front};
        memcpy(info, v, sizeof(v));
    }
    if (extr) {
        memcpy(extr, &e->tic.x, 24); memcpy(extr + 3, e->ric.m, 72); memcpy(extr + 12, &e->tio.x, 24); memcpy(extr + 15, e->rio.m, 72);
        extr[24] = e->sx; extr[25] = e->sy; extr[26] = e->sw; extr[27] = e->td; extr[28] = e->td_wheel; extr[29] = e->last_summary.initial_cost; extr[30] = e->last_summary.final_cost;
        extr[31] = e->f_manager.last_average_parallax;
    }
    return GF_OK;
}
int gf_estimator_get_latest(gf_estimator* e, double* imu, double* wheel) {
    if (!e) return gf::set_err(GF_ERR_INVALID, "null handle");
    if (imu) { imu[0] = e->latest_time; memcpy(imu + 1, &e->latest_P.x, 24); memcpy(imu + 4, e->latest_Q.m, 72); memcpy(imu + 13, &e->latest_V.x, 24); }
    if (wheel) { wheel[0] = e->latest_time_wheel; memcpy(wheel + 1, &e->latest_P_wheel.x, 24); memcpy(wheel + 4, e->latest_Q_wheel.m, 72); memcpy(wheel + 13, &e->latest_V_wheel.x, 24); }
    return GF_OK;
}
int gf_estimator_set_state(gf_estimator* e, int frame_count, int solver_flag, const double* Ps, const double* Rs, const double* Vs, const double* Bas, const double* Bgs) {
    if (!e || frame_count < 0 || frame_count > e->WINDOW_SIZE) return gf::set_err(GF_ERR_INVALID, "bad argument");
    e->frame_count = frame_count; e->solver_flag = solver_flag;
    for (int i = 0; i <= e->WINDOW_SIZE; i++) {
        if (Ps) e->Ps[i] = arr3(Ps + 3 * i);
        if (Rs) e->Rs[i] = arr9(Rs + 9 * i);
        if (Vs) e->Vs[i] = arr3(Vs + 3 * i);
        if (Bas) e->Bas[i] = arr3(Bas + 3 * i);
        if (Bgs) e->Bgs[i] = arr3(Bgs + 3 * i);
    }
    return GF_OK;
}
int gf_estimator_get_features(gf_estimator* e, int cap, int* id, int* start_frame, int* n_obs, double* estimated_depth, int* estimate_flag, int* solve_flag, int* n) {
    if (!e || !n) return gf::set_err(GF_ERR_INVALID, "null argument");
    *n = (int)e->f_manager.feature.size();
    if (*n > cap) return gf::set_err(GF_ERR_CAPACITY, "%d features, capacity %d", *n, cap);
    int k = 0;
    for (auto& it : e->f_manager.feature) {
        if (id) id[k] = it.feature_id;
        if (start_frame) start_frame[k] = it.start_frame;
        if (n_obs) n_obs[k] = (int)it.feature_per_frame.size();
        if (estimated_depth) estimated_depth[k] = it.estimated_depth;
        if (estimate_flag) estimate_flag[k] = it.estimate_flag;
        if (solve_flag) solve_flag[k] = it.solve_flag;
        k++;
    }
    return GF_OK;
}
int gf_estimator_get_feedback(gf_estimator* e, int cap, int* predict_ids, double* predict_xyz, int* n_predict, int* remove_ids, int* n_remove) {
    if (!e || !n_predict || !n_remove) return gf::set_err(GF_ERR_INVALID, "null argument");
    *n_predict = (int)e->predict_ids.size(); *n_remove = (int)e->remove_ids.size();
    if (*n_predict > cap || *n_remove > cap) return gf::set_err(GF_ERR_CAPACITY, "feedback lists need capacity %d", std::max(*n_predict, *n_remove));
    if (predict_ids) std::copy(e->predict_ids.begin(), e->predict_ids.end(), predict_ids);
    if (predict_xyz) std::copy(e->predict_xyz.begin(), e->predict_xyz.end(), predict_xyz);
    if (remove_ids) std::copy(e->remove_ids.begin(), e->remove_ids.end(), remove_ids);
    return GF_OK;
}
int gf_estimator_get_prior(gf_estimator* e, int cap_n, int cap_blocks, int* n, int* nblocks, int* block_id, double* J, double* r) {
    if (!e || !n || !nblocks) return gf::set_err(GF_ERR_INVALID, "null argument");
    *n = e->prior_valid ? e->prior_n : 0; *nblocks = e->prior_valid ? (int)e->prior_block_id.size() : 0;
    if (*n > cap_n || *nblocks > cap_blocks) return gf::set_err(GF_ERR_CAPACITY, "prior is %d x %d with %d blocks", *n, *n, *nblocks);
    if (block_id) std::copy(e->prior_block_id.begin(), e->prior_block_id.begin() + *nblocks, block_id);
    if (J && *n) {
        if (e->prior_resident && e->group) { if (int rc = gf_ba_fetch_resident_prior(e->group->ba, e->group_slot, *n, J)) return rc; }   // a group member's J lives on the device
        else std::copy(e->prior_J.begin(), e->prior_J.begin() + (size_t)*n * *n, J);
    }
    if (r && *n) std::copy(e->prior_r.begin(), e->prior_r.begin() + *n, r);
    return GF_OK;
}

// Host-only pieces of the estimator, callable one by one (tests; none of these touches the GPU):
//   "triangulate", "triangulateWithDepth", "removeBack", "removeFront" (in: frame_count), "removeFailures", "removeBackShiftDepth" (in: R0 9, P0 3, R1 9, P1 3),
//   "movingConsistencyCheckW" (out: ids), "predictPtsInNextFrame" (out: id,x,y,z ...), "slideWindow" (in: marginalization_flag), "setDepth" (in: inverse depths),
//   "getDepthVector" (out), "addFeature" (in: frame_count, td, then n x (id, 8 values); out: keyframe flag), "checkvisual" (out: flag), "getFeatureCount" (out)
int gf_estimator_debug(gf_estimator* e, const char* op, const double* in, int n_in, double* out, int cap_out, int* n_out) {
    if (!e || !op) return gf::set_err(GF_ERR_INVALID, "null argument");
    std::vector<double> o;
    const std::string s(op);
    if (s == "triangulate") e->f_manager.triangulate(e->Ps.data(), e->Rs.data(), e->tic, e->ric);
    else if (s == "triangulateWithDepth") e->f_manager.triangulateWithDepth(e->Ps.data(), e->Rs.data(), e->tic, e->ric);
    else if (s == "removeBack") e->f_manager.removeBack();
    else if (s == "updateLatestStates") e->updateLatestStates();
    else if (s == "removeFront" && n_in >= 1) e->f_manager.removeFront((int)in[0]);
    else if (s == "removeFailures") e->f_manager.removeFailures();
    else if (s == "removeBackShiftDepth" && n_in >= 24) e->f_manager.removeBackShiftDepth(arr9(in), arr3(in + 9), arr9(in + 12), arr3(in + 21));
    else if (s == "movingConsistencyCheckW") { std::set<int> r; e->movingConsistencyCheckW(r); for (int id : r) o.push_back(id); }
    else if (s == "predictPtsInNextFrame") {
        e->predict_ids.clear(); e->predict_xyz.clear(); e->predictPtsInNextFrame();
        for (size_t i = 0; i < e->predict_ids.size(); i++) o.insert(o.end(), {(double)e->predict_ids[i], e->predict_xyz[3 * i], e->predict_xyz[3 * i + 1], e->predict_xyz[3 * i + 2]});
    }
    else if (s == "slideWindow" && n_in >= 1) { e->marginalization_flag = (int)in[0]; e->slideWindow(); }
    else if (s == "setDepth") { if (n_in < e->f_manager.getFeatureCount()) return gf::set_err(GF_ERR_INVALID, "setDepth needs %d values", e->f_manager.getFeatureCount()); e->f_manager.setDepth(in); }
    else if (s == "getDepthVector") e->f_manager.getDepthVector(o);
    else if (s == "getFeatureCount") o.push_back(e->f_manager.getFeatureCount());
    else if (s == "checkvisual") o.push_back(e->checkvisual() ? 1 : 0);
    else if (s == "gnss_meas_buf") {   // per frame: count, then the satellite numbers (gnss_meas_buf[i], estimator.h:296)
        for (auto& b : e->gnss_meas_buf) { o.push_back((double)b.size()); for (auto& m : b) o.push_back(m.sat); }
    }
    else if (s == "sat_track_status") for (auto& kv : e->sat_track_status) { o.push_back(kv.first); o.push_back(kv.second); }
    else if (s == "solveRelativeRT_PNP" && n_in >= 6 && n_in % 6 == 0) {   // in: correspondences (6 values each); out: ok, Rotation (9), Translation (3)
        std::vector<std::array<double, 6>> corres(n_in / 6);
        for (size_t k = 0; k < corres.size(); k++) for (int a = 0; a < 6; a++) corres[k][a] = in[6 * k + a];
        double R[9] = {0}, T[3] = {0};
        std::vector<int> inl;
        const bool ok = gfinit::solve_relative_rt_pnp(corres, R, T, &inl);
        o.push_back(ok ? 1.0 : 0.0); o.insert(o.end(), R, R + 9); o.insert(o.end(), T, T + 3);
        for (int i : inl) o.push_back(i);   // RANSAC inliers (indices into the correspondences with both depths positive)
    }
    else if (s == "solvePnP" && n_in >= 8 && (n_in - 8) % 5 == 0) {   // in: n, use_guess, rvec, tvec, then (X, Y, Z, u, v) per point; out: ok, rvec, tvec
        const int n = (int)in[0];
        if (n_in != 8 + 5 * n) return gf::set_err(GF_ERR_INVALID, "solvePnP: %d values for %d points", n_in, n);
        std::vector<gfinit::P3> X(n); std::vector<gfinit::P2> uv(n);
        for (int k = 0; k < n; k++) { X[k] = {in[8 + 5 * k], in[9 + 5 * k], in[10 + 5 * k]}; uv[k] = {in[11 + 5 * k], in[12 + 5 * k]}; }
        double rv[3] = {in[2], in[3], in[4]}, tv[3] = {in[5], in[6], in[7]};
        const bool ok = gfinit::solve_pnp_iterative(X, uv, rv, tv, in[1] != 0);
        o.push_back(ok ? 1.0 : 0.0); o.insert(o.end(), rv, rv + 3); o.insert(o.end(), tv, tv + 3);
    }
    else if (s == "epnp" && n_in >= 1 && (n_in - 1) % 5 == 0) {   // in: n, then (X, Y, Z, u, v) per point; out: ok, rvec, tvec of the EPnP kernel solvePnPRansac runs on its subsets
        const int n = (int)in[0];
        if (n_in != 1 + 5 * n || n < 4) return gf::set_err(GF_ERR_INVALID, "epnp: %d values for %d points", n_in, n);
        std::vector<gfinit::P3> X(n); std::vector<gfinit::P2> uv(n);
        for (int k = 0; k < n; k++) { X[k] = {in[1 + 5 * k], in[2 + 5 * k], in[3 + 5 * k]}; uv[k] = {in[4 + 5 * k], in[5 + 5 * k]}; }
        double rv[3] = {0, 0, 0}, tv[3] = {0, 0, 0};
        const bool ok = gfinit::epnp(X, uv, rv, tv);
        o.push_back(ok ? 1.0 : 0.0); o.insert(o.end(), rv, rv + 3); o.insert(o.end(), tv, tv + 3);
    }
    else if (s == "skip_solve" && n_in >= 1) e->debug_skip_solve = in[0] != 0;   // optimization() and slideWindow() become no-ops: host-only parity tests of initialStructure
    else if (s == "init_info") o = {(double)e->init_l, (double)e->init_points, e->init_s, e->init_g.x, e->init_g.y, e->init_g.z, (double)e->init_rc, e->g.x, e->g.y, e->g.z};
    else if (s == "addFeature" && n_in >= 2 && (n_in - 2) % 9 == 0) {
        const int n = (n_in - 2) / 9;
        std::vector<gf_feature_obs> v(n);
        for (int i = 0; i < n; i++) { v[i].id = (int)in[2 + 9 * i]; v[i].camera_id = 0; memcpy(v[i].v, in + 3 + 9 * i, 64); }
        std::stable_sort(v.begin(), v.end(), [](const gf_feature_obs& a, const gf_feature_obs& b) { return a.id < b.id; });
        o.push_back(e->f_manager.addFeatureCheckParallax((int)in[0], v.data(), n, in[1]) ? 1 : 0);
    }
    else return gf::set_err(GF_ERR_INVALID, "unknown debug op '%s' or too few inputs", op);
    if (n_out) *n_out = (int)o.size();
    if ((int)o.size() > cap_out) return gf::set_err(GF_ERR_CAPACITY, "debug op '%s' returns %d values", op, (int)o.size());
    if (out) std::copy(o.begin(), o.end(), out);
    return GF_OK;
}

// ---------------------------------------------------------------- group of sequences on one batched solver
struct gf_estimator_group {
    std::vector<gf_estimator*> mem;
    BatchSolver solver;
    int device = 0;
    std::vector<std::thread> thr;
    std::mutex m;
    Gate go, all_done;    // a new step for the workers; the last worker of a step
    double t_input = 0;   // wall time inside gf_estimator_group_input_features [s]
    double t_tail = 0; long long n_steps = 0;   // from the end of a step's last batch to the return of input_features
    std::atomic<int> remaining{0};
    std::atomic<bool> stop{false};
    std::unique_ptr<std::atomic<int>[]> job_gen;   // generation of `go` in which member i has a frame to process (0 = never)
    std::vector<double> t;
    std::vector<const gf_feature_obs*> frame_ptr; std::vector<int> frame_n;   // this step's frame of member i inside the caller's buffer (no copy)
    std::vector<int> rcs;
    std::vector<std::string> errs;

    int n_threads = 1;
    bool in_flight = false; std::vector<int> flight_seq; std::chrono::steady_clock::time_point flight_t0;   // a submitted step until its wait
    std::atomic<long long> t_last_done_ns{0};   // when a member last finished its frame (steady clock; every worker stores before it counts itself out, the waiter reads behind the count)
    std::unique_ptr<Fiber[]> fib;

    void run_frame(int i) {   // one member's frame (inside its fiber)
        mem[i]->t_mark = gf_estimator::cpu_now();
        int rc;
        // an exception (std::bad_alloc in a member's bookkeeping) must not unwind past makecontext's frame -- that is std::terminate for the whole process: it
        // becomes this member's frame error, and the member still leaves the rendezvous so that the others' batches close
        try { rc = gf_estimator_input_feature(mem[i], t[i], frame_ptr[i], frame_n[i]); }   // the caller's buffer: input_features does not return before this is done
        catch (const std::exception& e) { rc = gf::set_err(GF_ERR_INVALID, "member %d: exception in processImage: %s", i, e.what()); }
        catch (...) { rc = gf::set_err(GF_ERR_INVALID, "member %d: unknown exception in processImage", i); }
        mem[i]->lap(5);
        rcs[i] = rc;
        if (rc != GF_OK) errs[i] = gf_last_error();
        solver.leave();
        t_last_done_ns.store(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(), std::memory_order_relaxed);
        if (remaining.fetch_sub(1, std::memory_order_acq_rel) == 1) all_done.bump();
    }
    static void fiber_entry(unsigned lo, unsigned hi, int i) {
        gf_estimator_group* g = reinterpret_cast<gf_estimator_group*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
        g->run_frame(i);
        g->fib[i].state = 3;   // returning switches to uc_link = the worker's scheduler context
    }
    void worker(int w) {   // worker w owns the members w, w + n_threads, ...
        (void)hipSetDevice(device);   // the device is a per-thread setting; whichever member closes a rendezvous launches the batch
        gf::pin_thread_to_device_node(device);   // this rank's workers onto the cores next to its GPU (GF_NUMA_PIN=0: off)
        ucontext_t sched;
        std::vector<int> mine;
        int seen = 0;   // go starts at generation 0 and is only bumped by input_features / the destructor
        for (;;) {
            while (go.now() == seen && !stop.load(std::memory_order_acquire)) go.wait_while(seen);
            if (stop.load(std::memory_order_acquire)) return;
            seen = go.now();   // one step at a time: input_features does not return before every listed member is done
            // The work list is published per generation: job_gen[i] == seen means "member i is listed in the step this thread just woke for" and, by the
            // release / acquire pair on it, that t[i] / frame_ptr[i] are completely written.  A thread that only gets here while step k + 1 is being set up
            // sees job_gen[i] == k + 1 != seen for its members, goes round the loop, picks up generation k + 1 and runs every frame exactly once.
            mine.clear();
            for (int i = w; i < (int)mem.size(); i += n_threads) if (job_gen[i].load(std::memory_order_acquire) == seen) mine.push_back(i);
            for (int i : mine) {
                Fiber& f = fib[i];
                getcontext(&f.ctx);
                f.ctx.uc_stack.ss_sp = f.stack + Fiber::kGuard; f.ctx.uc_stack.ss_size = f.kStack; f.ctx.uc_link = &sched;
                const uintptr_t self = reinterpret_cast<uintptr_t>(this);
                makecontext(&f.ctx, reinterpret_cast<void (*)()>(&gf_estimator_group::fiber_entry), 3, (unsigned)(self & 0xffffffffu), (unsigned)(self >> 32), i);
                f.state = 1;
            }
            size_t left = mine.size();
            while (left) {
                const int fin_seen = solver.finished.now();   // before the scan: a batch closed during the scan must not be slept through
                bool all_waiting = true;
                for (int i : mine) {
                    Fiber& f = fib[i];
                    if (f.state == 3) continue;
                    tl_sched = &sched; tl_fiber = &f; f.state = 1;
                    swapcontext(&sched, &f.ctx);
                    tl_fiber = nullptr;
                    if (f.state == 3) { left--; all_waiting = false; }
                }
                if (left && all_waiting) solver.finished.wait_while(fin_seen);   // every member of this thread waits for a batch somebody else will close
            }
        }
    }
    ~gf_estimator_group() {
        stop.store(true, std::memory_order_release);
        go.bump();
        for (auto& th : thr) if (th.joinable()) th.join();
        for (gf_estimator* e : mem) delete e;
        if (solver.ba) gf_ba_destroy(solver.ba);
        if (solver.pre) gf::preint_batch_destroy(solver.pre);
        if (solver.sweeps) gf_featsweep_destroy(solver.sweeps);
    }
};

int gf_estimator_group_create(const gf_estimator_cfg* c, int n, gf_estimator_group** out) {
    if (!c || !out || n < 1 || n > 4096) return gf::set_err(GF_ERR_INVALID, "bad argument (1 <= n <= 4096)");
    if (c->with_tracker) return gf::set_err(GF_ERR_INVALID, "group members take feature frames (cfg.with_tracker = 0); run one batched gf_tracker next to the group");
    if (c->max_solver_time > 0) return gf::set_err(GF_ERR_INVALID, "max_solver_time %.3g s: the wall-clock cut of ceres::Solve is per solve, not per batch; set it to 0 for a group (iteration cap only)", c->max_solver_time);
    gf_estimator_group* g = new gf_estimator_group();
    for (int i = 0; i < n; i++) {
        gf_estimator* e = nullptr;
        if (int rc = gf_estimator_create(c, &e)) { delete g; return rc; }
        e->group = &g->solver; e->group_slot = i;
        g->mem.push_back(e);
    }
    gf_ba_cfg bc{c->window_size, c->max_features, c->max_visual, n, c->gnss_enable ? c->max_gnss_per_frame * (c->window_size + 1) : 0};
    if (int rc = gf_ba_create(&bc, &g->solver.ba)) { delete g; return rc; }
    g->solver.mem_count = (size_t)n;
    (void)hipGetDevice(&g->device);
    // worker threads: half of the hardware threads this PROCESS may run on (its affinity mask: a process confined by taskset / cgroups to 8 threads sizes its pool for
    // 8, not for the 256 the box has), divided among the ranks that share the node
    int hw_box = 0, hw = 0;
    gf::host_cpus(hw_box, hw);   // affinity mask AND the container's CPU quota (gf_host_cpus.hpp): the boxes of round 6 show 256 hardware threads and grant 16
    int share = 1;
    // a launcher that pins every rank to its part of the node (numactl, cgroups, torchrun binding) has divided already: the mask IS the rank's share (round-5 advisor)
    if (const char* e = getenv("LOCAL_WORLD_SIZE")) if (hw_box <= 0 || hw >= hw_box) share = std::max(1, atoi(e));
    // Twice as many workers as the rank can run in parallel, at most 32: a group step is three short host phases between two device batches and a worker mostly sleeps on
    // the batch its members wait for, so a modest oversubscription pays; far beyond the quota the threads are throttled, not run (round 6, honest clock, 16-core quota, one
    // group of 256: 8 workers 15.6 k window-solves/s, 16: 22.8 k, 32: 33.3 k, 128: 20.0 k; two groups: 2 x 8: 22.6 k, 2 x 16: 34.6-45.4 k, 2 x 32: 40.6-44.6 k, 2 x 64: 28.3 k)
    int nt = std::min(n, std::max(1, std::min(32, 2 * hw / share)));
    if (const char* e = getenv("GF_GROUP_THREADS")) if (atoi(e) > 0) nt = std::min(n, atoi(e));
    // the members' IMU pre-integration as one device launch per camera frame.  It costs one more rendezvous per frame and pays where host threads are
    // scarce (round 5, 256 members on 8 hardware threads = 4 workers, two alternating groups: 20.9 k against 19.4 k window-solves/s; one group 11.9 k against 11.6 k;
    // on the 256-thread box it loses: 30.4 k against 47.3 k, round 3).  Default: on when a worker carries sixteen members or more; GF_GROUP_DEVICE_PREINT=0|1 decides otherwise.
    // The device feature sweeps (triangulateWithDepth, movingConsistencyCheckW) lose on both hosts (8 threads: 14.7 k against 19.4 k) and stay opt-in.
    bool want_pre = n >= 16 * nt;
    if (const char* e = getenv("GF_GROUP_DEVICE_PREINT")) want_pre = atoi(e) != 0;
    const bool pre_forced = getenv("GF_GROUP_DEVICE_PREINT") != nullptr;
    if (want_pre) if (int rc = gf::preint_batch_create(&g->solver.pre)) {
        if (pre_forced) { delete g; return rc; }   // asked for by name: its failure is the caller's to see
        g->solver.pre = nullptr;                   // chosen by the heuristic: the host loops give the same bits
        fprintf(stderr, "gf_estimator_group: device pre-integration could not be set up (%s); the members pre-integrate on the host\n", gf_last_error());
    }
    if (getenv("GF_GROUP_TIMING")) fprintf(stderr, "gf_estimator_group: %d members, %d worker threads (%d usable hardware threads, node share 1/%d), pre-integration on the %s\n", n, nt, hw, share, g->solver.pre ? "device" : "host");
    if (const char* e = getenv("GF_GROUP_DEVICE_SWEEPS")) if (atoi(e) != 0) if (int rc = gf_featsweep_create(&g->solver.sweeps)) { delete g; return rc; }
    g->job_gen.reset(new std::atomic<int>[n]); for (int i = 0; i < n; i++) g->job_gen[i].store(0, std::memory_order_relaxed);
    g->t.assign(n, 0.0); g->frame_ptr.assign(n, nullptr); g->frame_n.assign(n, 0); g->rcs.assign(n, GF_OK); g->errs.resize(n);
    {
        // default: half the hardware threads of this rank's share of the node (measured on a 256-thread host with 256 members: 8 threads 13 k window-solves/s end to end,
        // 32: 31 k, 128: 36 k, 256: 30 k -- the tracker's and the runtime's threads want cores too)
        g->n_threads = nt;
        g->fib.reset(new Fiber[n]);
        for (int i = 0; i < n; i++) if (!g->fib[i].alloc()) { delete g; return gf::set_err(GF_ERR_INVALID, "cannot map the stack of member %d", i); }
        for (int w = 0; w < nt; w++) g->thr.emplace_back([g, w] { g->worker(w); });
    }
    *out = g;
    return GF_OK;
}
int gf_estimator_group_destroy(gf_estimator_group* g) {
    if (g && g->in_flight) (void)gf_estimator_group_wait(g);
    if (g && getenv("GF_GROUP_TIMING")) {
        double ts[6] = {0, 0, 0, 0, 0, 0};
        for (gf_estimator* e : g->mem) for (int q = 0; q < 6; q++) ts[q] += e->t_sect[q];
        // thread CPU time between the marks of a member: the two "waiting" entries also contain what the OTHER members of the same worker thread computed meanwhile
        fprintf(stderr, "gf_estimator_group: %zu members on %d worker threads, CPU time summed over the members [ms]: before optimization %.1f, window build %.1f, waiting for the solve %.1f, "
                        "double2vector .. marginalisation request %.1f, waiting for the marginalisation %.1f, rest of the frame %.1f\n", g->mem.size(), g->n_threads, 1e3 * ts[0], 1e3 * ts[1],
                1e3 * ts[2], 1e3 * ts[3], 1e3 * ts[4], 1e3 * ts[5]);
    }
    if (g && getenv("GF_GROUP_TIMING") && g->solver.pre)
        fprintf(stderr, "gf_estimator_group: %lld batched pre-integration launches, %lld intervals, %.1f ms inside them\n", g->solver.pre_batches, g->solver.pre_intervals, 1e3 * g->solver.t_pre);
    if (g && getenv("GF_GROUP_TIMING") && g->solver.sweeps)
        fprintf(stderr, "gf_estimator_group: %lld batched feature sweeps, %lld features, %.1f ms inside them\n", g->solver.sweep_batches, g->solver.sweep_features, 1e3 * g->solver.t_sweep);
    if (g && getenv("GF_GROUP_TIMING") && g->solver.ba) {
        gf_ba_stats bs{};
        if (gf_ba_get_stats(g->solver.ba, &bs) == GF_OK)
            fprintf(stderr, "gf_estimator_group: device time of the shared handle [ms]: upload %.1f, solve %.1f, marginalise %.1f\n", bs.ms_upload, bs.ms_solve, bs.ms_marginalize);
        long long ub = 0, uc = 0; gf_ba_debug_upload_bytes(&ub, &uc);
        fprintf(stderr, "gf_estimator_group: %.1f MB host -> device in %lld copies since process start\n", ub / 1e6, uc);
    }
    if (g && getenv("GF_GROUP_TIMING"))
        fprintf(stderr, "gf_estimator_group: %lld batches, %lld windows; %.1f ms inside batched solves, %.1f ms inside batched marginalisations, %.1f ms inside input_features\n",
                g->solver.batches, g->solver.windows, 1e3 * g->solver.t_solve, 1e3 * g->solver.t_marg, 1e3 * g->t_input);
    if (g && getenv("GF_GROUP_TIMING") && g->n_steps) {
        const BatchSolver& S = g->solver;
        fprintf(stderr, "gf_estimator_group: wall-clock anatomy of %lld steps [ms per step]:", g->n_steps);
        for (int q = 0; q < 4; q++)
            if (S.w_n[q]) fprintf(stderr, " rendezvous %d (%lld): to first request %.3f, first..last request %.3f, batch %.3f;", q, S.w_n[q], 1e3 * S.w_to_first[q] / g->n_steps,
                                  1e3 * S.w_spread[q] / g->n_steps, 1e3 * S.w_batch[q] / g->n_steps);
        fprintf(stderr, " last batch .. return %.3f; whole step %.3f\n", 1e3 * g->t_tail / g->n_steps, 1e3 * g->t_input / g->n_steps);
    }
    delete g;
    return GF_OK;
}
int gf_estimator_group_set_device_preint(gf_estimator_group* g, int on) {
    if (!g) return gf::set_err(GF_ERR_INVALID, "null handle");
    std::unique_lock<std::mutex> lk(g->solver.m);
    if (on && !g->solver.pre) return gf::preint_batch_create(&g->solver.pre);
    if (!on && g->solver.pre) { gf::preint_batch_destroy(g->solver.pre); g->solver.pre = nullptr; }
    return GF_OK;
}
int gf_estimator_group_set_device_sweeps(gf_estimator_group* g, int on) {
    if (!g) return gf::set_err(GF_ERR_INVALID, "null handle");
    std::unique_lock<std::mutex> lk(g->solver.m);
    if (on && !g->solver.sweeps) return gf_featsweep_create(&g->solver.sweeps);
    if (!on && g->solver.sweeps) { gf_featsweep_destroy(g->solver.sweeps); g->solver.sweeps = nullptr; }
    return GF_OK;
}
int gf_estimator_group_member(gf_estimator_group* g, int i, gf_estimator** out) {
    if (!g || !out || i < 0 || i >= (int)g->mem.size()) return gf::set_err(GF_ERR_INVALID, "bad argument");
    *out = g->mem[i];
    return GF_OK;
}
// One step of the group in two halves, as the reference's inputFeature (estimator.cpp:447-459: push to featureBuf, return) and its processThread: submit publishes
// the frames to the workers and returns; wait blocks until every listed member has finished its frame.  `stride` >= 0: sequence k's observations start at
// obs + k * stride (a tracker's padded output table as it lies); < 0: back to back.  `obs`, `seq`, `n_obs` stay the caller's until wait returns.
int gf_estimator_group_submit_features(gf_estimator_group* g, int count, const int* seq, const double* t, const gf_feature_obs* obs, const int* n_obs, long long stride) {
    if (!g || count < 0 || (count > 0 && (!seq || !t || !n_obs))) return gf::set_err(GF_ERR_INVALID, "bad argument");
    if (g->in_flight) return gf::set_err(GF_ERR_INVALID, "a step of this group is in flight: gf_estimator_group_wait first");
    if (count == 0) return GF_OK;
    const auto tc0 = std::chrono::steady_clock::now();
    const int n = (int)g->mem.size();
    std::vector<char> seen(n, 0);
    size_t off = 0;
    {
        std::unique_lock<std::mutex> lk(g->m);
        for (int k = 0; k < count; k++) {   // validate first: nothing is published for a call that is refused
            const int i = seq[k];
            if (i < 0 || i >= n || seen[i] || n_obs[k] < 0 || (n_obs[k] > 0 && !obs) || (stride >= 0 && n_obs[k] > stride))
                return gf::set_err(GF_ERR_INVALID, "sequence index %d out of range, listed twice, without observations, or with more than the stride holds", i);
            seen[i] = 1;
        }
        const int next = g->go.now() + 1;   // only this function (serialised by g->m) and the destructor bump `go`
        for (int k = 0; k < count; k++) {
            const int i = seq[k];
            g->t[i] = t[k]; g->frame_ptr[i] = n_obs[k] > 0 ? obs + (stride >= 0 ? (size_t)k * (size_t)stride : off) : nullptr; g->frame_n[i] = n_obs[k]; g->rcs[i] = GF_OK;
            g->job_gen[i].store(next, std::memory_order_release);
            off += (size_t)n_obs[k];
        }
        { std::unique_lock<std::mutex> sl(g->solver.m); g->solver.active = count; g->solver.t_mark = tc0; g->solver.rdv = 0; }
        g->remaining.store(count, std::memory_order_release);
        g->flight_seq.assign(seq, seq + count); g->flight_t0 = tc0; g->in_flight = true;
    }
    g->go.bump();
    return GF_OK;
}
int gf_estimator_group_wait(gf_estimator_group* g) {
    if (!g) return gf::set_err(GF_ERR_INVALID, "null handle");
    if (!g->in_flight) return GF_OK;
    for (;;) {
        const int seen = g->all_done.now();
        if (g->remaining.load(std::memory_order_acquire) == 0) break;
        g->all_done.wait_while(seen);
    }
    g->in_flight = false;
    // (with submit / wait apart the step's clock also contains whatever the caller did in between once the members were done: GF_GROUP_TIMING's "whole step" is
    //  the caller's view, the rendezvous clocks are the group's)
    g->t_input += std::chrono::duration<double>(std::chrono::steady_clock::now() - g->flight_t0).count();
    g->t_tail += 1e-9 * (double)(g->t_last_done_ns.load(std::memory_order_relaxed) - std::chrono::duration_cast<std::chrono::nanoseconds>(g->solver.t_mark.time_since_epoch()).count()); g->n_steps++;
    for (int i : g->flight_seq) if (g->rcs[i] != GF_OK) return gf::set_err(g->rcs[i], "sequence %d: %s", i, g->errs[i].c_str());
    return GF_OK;
}
int gf_estimator_group_input_features(gf_estimator_group* g, int count, const int* seq, const double* t, const gf_feature_obs* obs, const int* n_obs) {
    if (int rc = gf_estimator_group_submit_features(g, count, seq, t, obs, n_obs, -1)) return rc;
    return gf_estimator_group_wait(g);
}
int gf_estimator_group_stats(gf_estimator_group* g, long long* batches, long long* windows, long long* largest) {
    if (!g) return gf::set_err(GF_ERR_INVALID, "null handle");
    std::unique_lock<std::mutex> lk(g->solver.m);
    if (batches) *batches = g->solver.batches;
    if (windows) *windows = g->solver.windows;
    if (largest) *largest = g->solver.largest;
    return GF_OK;
}

}  // extern "C"
