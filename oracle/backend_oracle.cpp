// ============================================================================
// TEST INFRASTRUCTURE — CPU ORACLE for the Ground-Fusion back end (sliding-window optimisation).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
//
// PARITY UNPINNED: the reference holds no tests for this path and its solver arithmetic lives in un-vendored
// Ceres 1.14 / Eigen 3.3.7 / Sophus.  This file restates
//   * the reference's own code (cited file:line, relative to vins_estimator/src/):
//       factor/integration_base.h:39-195, factor/imu_factor.h:28-191,
//       factor/wheel_integration_base.h:41-219, factor/wheel_factor.h:28-247,
//       factor/projectionTwoFrameOneCamFactor.cpp:43-151,
//       factor/marginalization_factor.cpp:12-392, factor/pose_local_parameterization.cpp:12-36,
//       estimator/estimator.cpp:2890-3631 (which factors/blocks enter the problem),
//   * the published Ceres 1.14 algorithms it drives (cited by upstream file name):
//       internal/ceres/trust_region_minimizer.cc  (step acceptance, tolerances, Jacobi scaling)
//       internal/ceres/dogleg_strategy.cc         (TRADITIONAL_DOGLEG; radius 1e4, mu in [1e-8,1], diag in [1e-6,1e32])
//       internal/ceres/schur_eliminator_impl.h + dense_schur / LLT (DENSE_SCHUR)
//       internal/ceres/corrector.cc, include/ceres/loss_function.h (HuberLoss)
//     Solver defaults used: function_tolerance 1e-6, gradient_tolerance 1e-10, parameter_tolerance 1e-8,
//     min_relative_decrease 1e-3, jacobi_scaling true, max_num_consecutive_invalid_steps 5.
// Documented choices: (a) the Schur-eliminated group is exactly the free inverse-depth blocks (Ceres picks a maximal
// independent set; any choice gives the same step up to rounding); (b) marginalisation orders parameter blocks by first
// appearance instead of the reference's address-keyed unordered_map iteration order (marginalization_factor.cpp:186-201);
// (c) wall-clock termination (max_solver_time) is disabled, iteration count is the only budget.
// ============================================================================
#include <cfloat>
#include <cstdio>
#include <map>
#include <thread>
#include <vector>

#include "gf_math.h"
#include "gf_oracle.h"


using namespace gfm;

namespace gfo_be {

// ------------------------------------------------------------------ pre-integration (B2)
struct ImuPre {
    V3 acc_0, gyr_0, lin_ba, lin_bg, delta_p, delta_v;
    Quat delta_q;
    Mat<15, 15> jacobian, covariance;
    Mat<18, 18> noise;
    double sum_dt = 0;
    void init(const V3& a0, const V3& g0, const V3& ba, const V3& bg, double ACC_N, double GYR_N, double ACC_W, double GYR_W) {
        acc_0 = a0; gyr_0 = g0; lin_ba = ba; lin_bg = bg; jacobian = Mat<15, 15>::Identity(); covariance = Mat<15, 15>();
        for (int i = 0; i < 3; i++) {
            noise(i, i) = ACC_N * ACC_N; noise(3 + i, 3 + i) = GYR_N * GYR_N; noise(6 + i, 6 + i) = ACC_N * ACC_N; noise(9 + i, 9 + i) = GYR_N * GYR_N;
            noise(12 + i, 12 + i) = ACC_W * ACC_W; noise(15 + i, 15 + i) = GYR_W * GYR_W;
        }
    }
    void propagate(double dt, const V3& acc_1, const V3& gyr_1) {  // integration_base.h:63-167
        V3 un_acc_0 = delta_q * (acc_0 - lin_ba);
        V3 un_gyr = (gyr_0 + gyr_1) * 0.5 - lin_bg;
        Quat rq = delta_q * Quat(1, un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2);
        V3 un_acc_1 = rq * (acc_1 - lin_ba);
        V3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
        V3 rp = delta_p + delta_v * dt + un_acc * (0.5 * dt * dt);
        V3 rv = delta_v + un_acc * dt;
        {
            V3 w_x = (gyr_0 + gyr_1) * 0.5 - lin_bg, a_0_x = acc_0 - lin_ba, a_1_x = acc_1 - lin_ba;
            M3 R_w_x = skew(w_x), R_a_0_x = skew(a_0_x), R_a_1_x = skew(a_1_x);
            M3 Rd = delta_q.toRotationMatrix(), Rr = rq.toRotationMatrix(), I = M3::Identity();
            Mat<15, 15> F;
            F.setBlock(0, 0, I);
            F.setBlock(0, 3, Rd * R_a_0_x * (-0.25 * dt * dt) + Rr * R_a_1_x * (I - R_w_x * dt) * (-0.25 * dt * dt));
            F.setBlock(0, 6, I * dt);
            F.setBlock(0, 9, (Rd + Rr) * (-0.25 * dt * dt));
            F.setBlock(0, 12, Rr * R_a_1_x * (-0.25 * dt * dt * -dt));
            F.setBlock(3, 3, I - R_w_x * dt);
            F.setBlock(3, 12, I * (-1.0 * dt));
            F.setBlock(6, 3, Rd * R_a_0_x * (-0.5 * dt) + Rr * R_a_1_x * (I - R_w_x * dt) * (-0.5 * dt));
            F.setBlock(6, 6, I);
            F.setBlock(6, 9, (Rd + Rr) * (-0.5 * dt));
            F.setBlock(6, 12, Rr * R_a_1_x * (-0.5 * dt * -dt));
            F.setBlock(9, 9, I);
            F.setBlock(12, 12, I);
            Mat<15, 18> V;
            V.setBlock(0, 0, Rd * (0.25 * dt * dt));
            M3 v03 = (-Rr) * R_a_1_x * (0.25 * dt * dt * 0.5 * dt);
            V.setBlock(0, 3, v03);
            V.setBlock(0, 6, Rr * (0.25 * dt * dt));
            V.setBlock(0, 9, v03);
            V.setBlock(3, 3, I * (0.5 * dt));
            V.setBlock(3, 9, I * (0.5 * dt));
            V.setBlock(6, 0, Rd * (0.5 * dt));
            M3 v63 = (-Rr) * R_a_1_x * (0.5 * dt * 0.5 * dt);
            V.setBlock(6, 3, v63);
            V.setBlock(6, 6, Rr * (0.5 * dt));
            V.setBlock(6, 9, v63);
            V.setBlock(9, 12, I * dt);
            V.setBlock(12, 15, I * dt);
            jacobian = F * jacobian;
            covariance = F * covariance * F.T() + V * noise * V.T();
        }
        delta_p = rp; delta_q = rq; delta_v = rv;
        delta_q.normalize();
        sum_dt += dt;
        acc_0 = acc_1; gyr_0 = gyr_1;
    }
};

struct WheelPre {
    V3 vel_0, gyr_0, delta_p;
    Quat delta_q;
    double sx, sy, sw, sum_dt = 0;
    Mat<6, 3> jacobian;
    Mat<6, 6> covariance;
    Mat<12, 12> noise;
    void init(const V3& v0, const V3& g0, double sx_, double sy_, double sw_, double VEL_N, double GYR_N) {
        vel_0 = v0; gyr_0 = g0; sx = sx_; sy = sy_; sw = sw_;
        for (int i = 0; i < 3; i++) { noise(i, i) = VEL_N * VEL_N; noise(3 + i, 3 + i) = GYR_N * GYR_N; noise(6 + i, 6 + i) = VEL_N * VEL_N; noise(9 + i, 9 + i) = GYR_N * GYR_N; }
    }
    void propagate(double dt, const V3& vel_1, const V3& gyr_1) {  // wheel_integration_base.h:67-178
        M3 sv = diag3(sx, sy, 1);
        V3 un_vel_0 = delta_q * (sv * vel_0);
        V3 un_gyr = (gyr_0 + gyr_1) * (0.5 * sw);
        Quat ddq(1, un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2);
        Quat rq = delta_q * ddq;
        V3 un_vel_1 = rq * (sv * vel_1);
        V3 un_vel = (un_vel_0 + un_vel_1) * 0.5;
        V3 rp = delta_p + un_vel * dt;
        {
            V3 vel_0_x = sv * vel_0, vel_1_x = sv * vel_1;
            M3 R_vel_0_x = skew(vel_0_x), R_vel_1_x = skew(vel_1_x);
            M3 Rd = delta_q.toRotationMatrix(), Rr = rq.toRotationMatrix(), Rdd = ddq.toRotationMatrix(), I = M3::Identity();
            Mat<6, 6> F;
            F.setBlock(0, 0, I);
            F.setBlock(0, 3, (Rd * R_vel_0_x + Rr * R_vel_1_x * Rdd.T()) * (-0.5 * dt));
            F.setBlock(3, 3, Rdd.T());
            M3 Jr = rightJacobianSO3(un_gyr * dt);
            Mat<6, 12> V;
            V.setBlock(0, 0, Rd * sv * (0.5 * dt));
            V.setBlock(0, 3, Rr * R_vel_1_x * Jr * (-0.25 * dt * dt));
            V.setBlock(0, 6, Rr * sv * (0.5 * dt));
            V.setBlock(0, 9, Rr * R_vel_1_x * Jr * (-0.25 * dt * dt));
            V.setBlock(3, 3, Jr * (0.5 * sw * dt));
            V.setBlock(3, 9, Jr * (0.5 * sw * dt));
            M3 I1 = diag3(1, 0, 0), I2 = diag3(0, 1, 0);
            V3 j00 = jacobian.block<3, 1>(0, 0) + (Rd * (I1 * vel_0) + Rr * (I1 * vel_1)) * (0.5 * dt);
            V3 j01 = jacobian.block<3, 1>(0, 1) + (Rd * (I2 * vel_0) + Rr * (I2 * vel_1)) * (0.5 * dt);
            V3 dr_dsw_last = jacobian.block<3, 1>(3, 2);
            V3 j32 = dr_dsw_last + Jr * ((gyr_0 + gyr_1) * 0.5) * dt;
            jacobian.setBlock(0, 0, j00); jacobian.setBlock(0, 1, j01); jacobian.setBlock(3, 2, j32);
            V3 j02 = jacobian.block<3, 1>(0, 2) + (Rd * (skew(dr_dsw_last) * (sv * vel_0)) + Rr * (skew(j32) * (sv * vel_1))) * (0.5 * dt);
            jacobian.setBlock(0, 2, j02);
            covariance = F * covariance * F.T() + V * noise * V.T();
        }
        delta_p = rp; delta_q = rq;
        delta_q.normalize();
        sum_dt += dt;
        vel_0 = vel_1; gyr_0 = gyr_1;
    }
};

// ------------------------------------------------------------------ parameter blocks
enum Kind { POSE = 0, SPEEDBIAS = 1, EX_POSE = 2, EX_WHEEL = 3, SX = 4, SY = 5, SW = 6, TD = 7, TD_WHEEL = 8, FEATURE = 9, RCV_DT = 10, RCV_DDT = 11, YAW = 12, ANC = 13 };
inline int bid(int kind, int idx) { return kind * 4096 + idx; }
inline int gsize_of(int kind) { return (kind == POSE || kind == EX_POSE || kind == EX_WHEEL) ? 7 : kind == SPEEDBIAS ? 9 : kind == ANC ? 3 : 1; }
inline int lsize_of(int kind) { int g = gsize_of(kind); return g == 7 ? 6 : g; }

struct State {  // values of every parameter block (copied from / to the window)
    int W, F;
    std::vector<double> pose, sb, feat;
    double ex[7], exw[7], ix[3], td, tdw;
    std::vector<double> rcv_dt, rcv_ddt; double yaw = 0, anc[3] = {0, 0, 0}; bool gnss = false;
    double* ptr(int id) {
        int k = id / 4096, i = id % 4096;
        switch (k) {
            case POSE: return &pose[7 * i]; case SPEEDBIAS: return &sb[9 * i]; case EX_POSE: return ex; case EX_WHEEL: return exw;
            case SX: return &ix[0]; case SY: return &ix[1]; case SW: return &ix[2]; case TD: return &td; case TD_WHEEL: return &tdw;
            case RCV_DT: return &rcv_dt[i]; case RCV_DDT: return &rcv_ddt[i]; case YAW: return &yaw; case ANC: return anc;
            default: return &feat[i];
        }
    }
    const double* ptr(int id) const { return const_cast<State*>(this)->ptr(id); }
    void load(const gfo_window* w) {
        W = w->W; F = w->n_feature;
        pose.assign(w->para_Pose, w->para_Pose + 7 * (W + 1)); sb.assign(w->para_SpeedBias, w->para_SpeedBias + 9 * (W + 1));
        feat.assign(w->para_Feature, w->para_Feature + F);
        memcpy(ex, w->para_Ex_Pose, 56); memcpy(exw, w->para_Ex_Pose_wheel, 56); memcpy(ix, w->para_Ix, 24); td = w->para_Td[0]; tdw = w->para_Td_wheel[0];
        gnss = w->gnss_enabled != 0;
        if (gnss) { rcv_dt.assign(w->para_rcv_dt, w->para_rcv_dt + 4 * (W + 1)); rcv_ddt.assign(w->para_rcv_ddt, w->para_rcv_ddt + W + 1); yaw = w->para_yaw_enu_local[0]; memcpy(anc, w->para_anc_ecef, 24); }
    }
    void store(gfo_window* w) const {
        memcpy(w->para_Pose, pose.data(), pose.size() * 8); memcpy(w->para_SpeedBias, sb.data(), sb.size() * 8);
        if (F) memcpy(w->para_Feature, feat.data(), F * 8);
        memcpy(w->para_Ex_Pose, ex, 56); memcpy(w->para_Ex_Pose_wheel, exw, 56); memcpy(w->para_Ix, ix, 24); w->para_Td[0] = td; w->para_Td_wheel[0] = tdw;
        if (gnss) { memcpy(w->para_rcv_dt, rcv_dt.data(), rcv_dt.size() * 8); memcpy(w->para_rcv_ddt, rcv_ddt.data(), rcv_ddt.size() * 8); w->para_yaw_enu_local[0] = yaw; memcpy(w->para_anc_ecef, anc, 24); }
    }
};

inline V3 P_of(const double* p) { return v3(p[0], p[1], p[2]); }
inline Quat Q_of(const double* p) { return Quat(p[6], p[3], p[4], p[5]); }

// ------------------------------------------------------------------ factors.  J[b] = row-major nres x gsize(b)
struct FactorOut { int nres; double r[160]; std::vector<std::vector<double>> J; };

template <int R, int C> static void put(std::vector<double>& J, int cols, int r0, int c0, const Mat<R, C>& m) {
    for (int r = 0; r < R; r++) for (int c = 0; c < C; c++) J[(size_t)(r0 + r) * cols + c0 + c] = m(r, c);
}

// projectionTwoFrameOneCamFactor.cpp:43-151
static void eval_visual(const gfo_window* w, int k, const double* const* p, FactorOut& o, bool jac) {
    V3 Pi = P_of(p[0]), Pj = P_of(p[1]), tic = P_of(p[2]);
    Quat Qi = Q_of(p[0]), Qj = Q_of(p[1]), qic = Q_of(p[2]);
    const double inv_dep_i = p[3][0], td = p[4][0];
    V3 pts_i = v3(w->vis_pts_i[3 * k], w->vis_pts_i[3 * k + 1], w->vis_pts_i[3 * k + 2]), pts_j = v3(w->vis_pts_j[3 * k], w->vis_pts_j[3 * k + 1], w->vis_pts_j[3 * k + 2]);
    V3 vel_i = v3(w->vis_vel_i[2 * k], w->vis_vel_i[2 * k + 1], 0), vel_j = v3(w->vis_vel_j[2 * k], w->vis_vel_j[2 * k + 1], 0);
    const double td_i = w->vis_td_i[k], td_j = w->vis_td_j[k], si = w->vis_sqrt_info;
    V3 pts_i_td = pts_i - vel_i * (td - td_i), pts_j_td = pts_j - vel_j * (td - td_j);
    V3 pts_camera_i = pts_i_td / inv_dep_i;
    V3 pts_imu_i = qic * pts_camera_i + tic;
    V3 pts_w = Qi * pts_imu_i + Pi;
    V3 pts_imu_j = Qj.inverse() * (pts_w - Pj);
    V3 pts_camera_j = qic.inverse() * (pts_imu_j - tic);
    const double dep_j = pts_camera_j[2];
    o.nres = 2;
    o.r[0] = si * (pts_camera_j[0] / dep_j - pts_j_td[0]);
    o.r[1] = si * (pts_camera_j[1] / dep_j - pts_j_td[1]);
    if (!jac) return;
    M3 Ri = Qi.toRotationMatrix(), Rj = Qj.toRotationMatrix(), ric = qic.toRotationMatrix();
    Mat<2, 3> reduce;
    reduce(0, 0) = 1. / dep_j; reduce(0, 2) = -pts_camera_j[0] / (dep_j * dep_j);
    reduce(1, 1) = 1. / dep_j; reduce(1, 2) = -pts_camera_j[1] / (dep_j * dep_j);
    reduce = reduce * si;  // sqrt_info = si * I2
    o.J.assign(5, {});
    {
        Mat<3, 6> ji;
        ji.setBlock(0, 0, ric.T() * Rj.T());
        ji.setBlock(0, 3, ric.T() * Rj.T() * Ri * (-skew(pts_imu_i)));
        o.J[0].assign(14, 0.0); put(o.J[0], 7, 0, 0, reduce * ji);
    }
    {
        Mat<3, 6> jj;
        jj.setBlock(0, 0, ric.T() * (-Rj.T()));
        jj.setBlock(0, 3, ric.T() * skew(pts_imu_j));
        o.J[1].assign(14, 0.0); put(o.J[1], 7, 0, 0, reduce * jj);
    }
    {
        Mat<3, 6> je;
        je.setBlock(0, 0, ric.T() * (Rj.T() * Ri - M3::Identity()));
        M3 tmp_r = ric.T() * Rj.T() * Ri * ric;
        je.setBlock(0, 3, (-tmp_r) * skew(pts_camera_i) + skew(tmp_r * pts_camera_i) + skew(ric.T() * (Rj.T() * (Ri * tic + Pi - Pj) - tic)));
        o.J[2].assign(14, 0.0); put(o.J[2], 7, 0, 0, reduce * je);
    }
    {
        Mat<2, 1> jf = reduce * (ric.T() * Rj.T() * Ri * ric * pts_i_td) * (-1.0 / (inv_dep_i * inv_dep_i));
        o.J[3] = {jf[0], jf[1]};
    }
    {
        Mat<2, 1> jt = reduce * (ric.T() * Rj.T() * Ri * ric * vel_i) / inv_dep_i * -1.0;
        o.J[4] = {jt[0] + si * vel_j[0], jt[1] + si * vel_j[1]};
    }
}

// imu_factor.h:28-191 + integration_base.h:169-195
static void eval_imu(const gfo_window* w, int k, const double* const* p, FactorOut& o, bool jac) {
    V3 Pi = P_of(p[0]), Vi = v3(p[1][0], p[1][1], p[1][2]), Bai = v3(p[1][3], p[1][4], p[1][5]), Bgi = v3(p[1][6], p[1][7], p[1][8]);
    V3 Pj = P_of(p[2]), Vj = v3(p[3][0], p[3][1], p[3][2]), Baj = v3(p[3][3], p[3][4], p[3][5]), Bgj = v3(p[3][6], p[3][7], p[3][8]);
    Quat Qi = Q_of(p[0]), Qj = Q_of(p[2]);
    Mat<15, 15> jacobian, covariance;
    memcpy(jacobian.a, w->imu_jacobian + 225 * k, 225 * 8); memcpy(covariance.a, w->imu_covariance + 225 * k, 225 * 8);
    V3 delta_p = v3(w->imu_delta_p[3 * k], w->imu_delta_p[3 * k + 1], w->imu_delta_p[3 * k + 2]), delta_v = v3(w->imu_delta_v[3 * k], w->imu_delta_v[3 * k + 1], w->imu_delta_v[3 * k + 2]);
    Quat delta_q(w->imu_delta_q[4 * k], w->imu_delta_q[4 * k + 1], w->imu_delta_q[4 * k + 2], w->imu_delta_q[4 * k + 3]);
    V3 lin_ba = v3(w->imu_lin_ba[3 * k], w->imu_lin_ba[3 * k + 1], w->imu_lin_ba[3 * k + 2]), lin_bg = v3(w->imu_lin_bg[3 * k], w->imu_lin_bg[3 * k + 1], w->imu_lin_bg[3 * k + 2]);
    const double sum_dt = w->imu_sum_dt[k];
    V3 G = v3(w->G[0], w->G[1], w->G[2]);
    M3 dp_dba = jacobian.block<3, 3>(0, 9), dp_dbg = jacobian.block<3, 3>(0, 12), dq_dbg = jacobian.block<3, 3>(3, 12), dv_dba = jacobian.block<3, 3>(6, 9),
       dv_dbg = jacobian.block<3, 3>(6, 12);
    V3 dba = Bai - lin_ba, dbg = Bgi - lin_bg;
    Quat corrected_delta_q = delta_q * deltaQ(dq_dbg * dbg);
    V3 corrected_delta_v = delta_v + dv_dba * dba + dv_dbg * dbg;
    V3 corrected_delta_p = delta_p + dp_dba * dba + dp_dbg * dbg;
    Mat<15, 1> res;
    res.setBlock(0, 0, Qi.inverse() * (G * (0.5 * sum_dt * sum_dt) + Pj - Pi - Vi * sum_dt) - corrected_delta_p);
    res.setBlock(3, 0, (corrected_delta_q.inverse() * (Qi.inverse() * Qj)).vec() * 2.0);
    res.setBlock(6, 0, Qi.inverse() * (G * sum_dt + Vj - Vi) - corrected_delta_v);
    res.setBlock(9, 0, Baj - Bai);
    res.setBlock(12, 0, Bgj - Bgi);
    Mat<15, 15> sqrt_info = llt_upper(inverse(covariance));  // imu_factor.h:73
    res = sqrt_info * res;
    o.nres = 15;
    for (int i = 0; i < 15; i++) o.r[i] = res[i];
    if (!jac) return;
    o.J.assign(4, {});
    {
        Mat<15, 6> J;
        J.setBlock(0, 0, -(Qi.inverse().toRotationMatrix()));
        J.setBlock(0, 3, skew(Qi.inverse() * (G * (0.5 * sum_dt * sum_dt) + Pj - Pi - Vi * sum_dt)));
        J.setBlock(3, 3, -((Qleft(Qj.inverse() * Qi) * Qright(corrected_delta_q)).block<3, 3>(1, 1)));
        J.setBlock(6, 3, skew(Qi.inverse() * (G * sum_dt + Vj - Vi)));
        o.J[0].assign(15 * 7, 0.0); put(o.J[0], 7, 0, 0, sqrt_info * J);
    }
    {
        Mat<15, 9> J;
        J.setBlock(0, 0, -(Qi.inverse().toRotationMatrix()) * sum_dt);
        J.setBlock(0, 3, -dp_dba); J.setBlock(0, 6, -dp_dbg);
        J.setBlock(3, 6, -(Qleft(Qj.inverse() * Qi * delta_q).block<3, 3>(1, 1)) * dq_dbg);
        J.setBlock(6, 0, -(Qi.inverse().toRotationMatrix()));
        J.setBlock(6, 3, -dv_dba); J.setBlock(6, 6, -dv_dbg);
        J.setBlock(9, 3, -M3::Identity()); J.setBlock(12, 6, -M3::Identity());
        o.J[1].assign(15 * 9, 0.0); put(o.J[1], 9, 0, 0, sqrt_info * J);
    }
    {
        Mat<15, 6> J;
        J.setBlock(0, 0, Qi.inverse().toRotationMatrix());
        J.setBlock(3, 3, Qleft(corrected_delta_q.inverse() * Qi.inverse() * Qj).block<3, 3>(1, 1));
        o.J[2].assign(15 * 7, 0.0); put(o.J[2], 7, 0, 0, sqrt_info * J);
    }
    {
        Mat<15, 9> J;
        J.setBlock(6, 0, Qi.inverse().toRotationMatrix());
        J.setBlock(9, 3, M3::Identity()); J.setBlock(12, 6, M3::Identity());
        o.J[3].assign(15 * 9, 0.0); put(o.J[3], 9, 0, 0, sqrt_info * J);
    }
}

static inline Quat so3mul(const Quat& a, const Quat& b) { return (a * b).normalized(); }

// wheel_factor.h:28-247 + wheel_integration_base.h:180-219.  Blocks: pose_i, pose_j, T_io, sx, sy, sw, td_wheel
static void eval_wheel(const gfo_window* w, int k, const double* const* p, FactorOut& o, bool jac) {
    V3 Pi = P_of(p[0]), Pj = P_of(p[1]), tio = P_of(p[2]);
    Quat Qi = Q_of(p[0]), Qj = Q_of(p[1]), qio = Q_of(p[2]);
    const double sx = p[3][0], sy = p[4][0], sw = p[5][0], td = p[6][0];
    M3 sv = diag3(sx, sy, 1);
    Mat<6, 3> jacobian; Mat<6, 6> covariance;
    memcpy(jacobian.a, w->wh_jacobian + 18 * k, 18 * 8); memcpy(covariance.a, w->wh_covariance + 36 * k, 36 * 8);
    V3 delta_p = v3(w->wh_delta_p[3 * k], w->wh_delta_p[3 * k + 1], w->wh_delta_p[3 * k + 2]);
    Quat delta_q(w->wh_delta_q[4 * k], w->wh_delta_q[4 * k + 1], w->wh_delta_q[4 * k + 2], w->wh_delta_q[4 * k + 3]);
    const double lsx = w->wh_lin[4 * k], lsy = w->wh_lin[4 * k + 1], lsw = w->wh_lin[4 * k + 2], ltd = w->wh_lin[4 * k + 3];
    V3 lin_vel = v3(w->wh_lin_vel[3 * k], w->wh_lin_vel[3 * k + 1], w->wh_lin_vel[3 * k + 2]), lin_gyr = v3(w->wh_lin_gyr[3 * k], w->wh_lin_gyr[3 * k + 1], w->wh_lin_gyr[3 * k + 2]);
    V3 vel_1 = v3(w->wh_vel_1[3 * k], w->wh_vel_1[3 * k + 1], w->wh_vel_1[3 * k + 2]), gyr_1 = v3(w->wh_gyr_1[3 * k], w->wh_gyr_1[3 * k + 1], w->wh_gyr_1[3 * k + 2]);
    V3 dp_dsx = jacobian.block<3, 1>(0, 0), dp_dsy = jacobian.block<3, 1>(0, 1), dp_dsw = jacobian.block<3, 1>(0, 2), dq_dsw = jacobian.block<3, 1>(3, 2);
    const double dsx = sx - lsx, dsy = sy - lsy, dsw = sw - lsw;
    M3 Ri = Qi.toRotationMatrix(), Rj = Qj.toRotationMatrix(), rio = qio.toRotationMatrix();
    V3 corrected_delta_p = delta_p + dp_dsx * dsx + dp_dsy * dsy + dp_dsw * dsw;
    Quat corrected_delta_q = so3mul(delta_q.normalized(), so3_exp(dq_dsw * dsw));
    const double dtd = td - ltd;
    Quat e_fw = so3_exp(lin_gyr * (sw * dtd));
    Quat delta_q_time = so3mul(so3mul(e_fw, corrected_delta_q), so3_exp(gyr_1 * (-sw * dtd)));
    V3 delta_p_time = e_fw.toRotationMatrix() * (sv * lin_vel * dtd + corrected_delta_p - corrected_delta_q * (sv * vel_1 * dtd));
    Mat<6, 1> res;
    res.setBlock(0, 0, (Ri * rio).T() * (Rj * tio + Pj - Ri * tio - Pi) - delta_p_time);
    res.setBlock(3, 0, so3_log(delta_q_time.inverse() * (Qi * qio).inverse() * Qj * qio));
    Mat<6, 1> raw = res;
    Mat<6, 6> sqrt_info = llt_upper(inverse(covariance));
    res = sqrt_info * res;
    o.nres = 6;
    for (int i = 0; i < 6; i++) o.r[i] = res[i];
    if (!jac) return;
    V3 raw_r = raw.block<3, 1>(3, 0);
    M3 Jr_inv = rightJacobianInvSO3(raw_r);
    V3 drdsw = dq_dsw * (sw - lsw);
    M3 Jr_drdsw = rightJacobianSO3(drdsw);
    M3 Rcq = corrected_delta_q.toRotationMatrix();
    o.J.assign(7, {});
    {
        Mat<6, 6> J;
        J.setBlock(0, 0, -((Qi * qio).inverse().toRotationMatrix()));
        J.setBlock(0, 3, (Ri * rio).T() * (Ri * skew(tio)) + rio.T() * skew(Ri.T() * (Rj * tio + Pj - Ri * tio - Pi)));
        J.setBlock(3, 3, -(Jr_inv * ((Qj * qio).inverse() * Qi).toRotationMatrix()));
        o.J[0].assign(42, 0.0); put(o.J[0], 7, 0, 0, sqrt_info * J);
    }
    {
        Mat<6, 6> J;
        J.setBlock(0, 0, (Qi * qio).inverse().toRotationMatrix());
        J.setBlock(0, 3, -(((Qi * qio).inverse() * Qj).toRotationMatrix()) * skew(tio));
        J.setBlock(3, 3, Jr_inv * qio.inverse().toRotationMatrix());
        o.J[1].assign(42, 0.0); put(o.J[1], 7, 0, 0, sqrt_info * J);
    }
    {
        Mat<6, 6> J;
        J.setBlock(0, 0, (Qi * qio).inverse().toRotationMatrix() * (Rj - Ri));
        J.setBlock(0, 3, skew((Qi * qio).inverse() * (Qj * tio + Pj - Qi * tio - Pi)));
        J.setBlock(3, 3, Jr_inv * (M3::Identity() - ((Qj * qio).inverse() * Qi * qio).toRotationMatrix()));
        o.J[2].assign(42, 0.0); put(o.J[2], 7, 0, 0, sqrt_info * J);
    }
    V3 forward_compensate_w = lin_gyr * (sw * dtd), forward_compensate_v = sv * lin_vel * dtd, back_compensate_v = sv * vel_1 * dtd, back_compensate_w = gyr_1 * (sw * dtd);
    M3 Jrtd = rightJacobianSO3(forward_compensate_w), Jr_minus_td = rightJacobianSO3(-forward_compensate_w);
    M3 I1 = diag3(1, 0, 0), I2 = diag3(0, 1, 0);
    {   // wheel_factor.h:199 uses exp(forward_compensate_v) (a velocity) — kept as in the reference
        Mat<6, 1> J;
        J.setBlock(0, 0, -(so3_exp(forward_compensate_v).toRotationMatrix() * (I1 * lin_vel * dtd + dp_dsx - Rcq * (I1 * vel_1) * dtd)));
        Mat<6, 1> s = sqrt_info * J; o.J[3].assign(s.a, s.a + 6);
    }
    {
        Mat<6, 1> J;
        J.setBlock(0, 0, -(so3_exp(forward_compensate_v).toRotationMatrix() * (I2 * lin_vel * dtd + dp_dsy - Rcq * (I2 * vel_1) * dtd)));
        Mat<6, 1> s = sqrt_info * J; o.J[4].assign(s.a, s.a + 6);
    }
    {
        Mat<6, 1> J;
        J.setBlock(0, 0, -(so3_exp(forward_compensate_w).toRotationMatrix() *
                           (dp_dsw - Rcq * skew(Jr_drdsw * dq_dsw) * (sv * vel_1) * dtd +
                            skew(Jrtd * lin_gyr * dtd) * (forward_compensate_v + corrected_delta_p - corrected_delta_q * back_compensate_v))));
        J.setBlock(3, 0, -(Jr_inv * so3_exp(-raw_r).toRotationMatrix() * so3_exp(back_compensate_w).toRotationMatrix() *
                           (corrected_delta_q.inverse().toRotationMatrix() * (Jrtd * lin_gyr) * dtd + Jr_drdsw * dq_dsw)));
        Mat<6, 1> s = sqrt_info * J; o.J[5].assign(s.a, s.a + 6);
    }
    {
        Mat<6, 1> J;
        J.setBlock(0, 0, -(so3_exp(forward_compensate_w).toRotationMatrix() *
                           (sv * lin_vel - Rcq * (sv * vel_1) + skew(Jrtd * lin_gyr * sw) * (forward_compensate_v + corrected_delta_p - Rcq * back_compensate_v))));
        J.setBlock(3, 0, -(Jr_inv * so3_exp(-raw_r).toRotationMatrix() *
                           (so3_exp(back_compensate_w).toRotationMatrix() * corrected_delta_q.inverse().toRotationMatrix() * (Jrtd * lin_gyr) * sw - Jr_minus_td * gyr_1 * sw)));
        Mat<6, 1> s = sqrt_info * J; o.J[6].assign(s.a, s.a + 6);
    }
}

// MarginalizationFactor::Evaluate, marginalization_factor.cpp:344-392.  Jacobian = columns of linearized_jacobians.

// ------------------------------------------------------------------ GNSS factors (F4).  gnss_comm (un-vendored, unpinned git HEAD) supplies
// ecef2geo / ecef2rotation / sat_azel / calculate_trop_delay / calculate_ion_delay to gnss_psr_dopp_factor.cpp:70-83.  Restated here from the
// published algorithms gnss_comm takes them from (RTKLIB: Saastamoinen troposphere with standard atmosphere and 70 % humidity, Klobuchar broadcast
// ionosphere; closed-form Bowring geodetic conversion, latitude / longitude in DEGREES).  Constants: gnss_comm/gnss_constant.hpp.
static const double GN_C = 2.99792458e8, GN_OMG = 7.2921151467e-5, GN_A = 6378137.0, GN_E2 = 6.69437999014e-3, GN_PI = 3.14159265358979323846;
static V3 gn_ecef2geo(const V3& xyz) {   // lat [deg], lon [deg], alt [m]
    if (xyz[0] == 0 && xyz[1] == 0) return v3(0, 0, 0);
    const double a = GN_A, a2 = a * a, b2 = a2 * (1 - GN_E2), b = std::sqrt(b2), ep2 = (a2 - b2) / b2, p = std::sqrt(xyz[0] * xyz[0] + xyz[1] * xyz[1]);
    double s1 = xyz[2] * a, s2 = p * b, h = std::sqrt(s1 * s1 + s2 * s2);
    const double sin_theta = s1 / h, cos_theta = s2 / h;
    s1 = xyz[2] + ep2 * b * sin_theta * sin_theta * sin_theta;
    s2 = p - a * GN_E2 * cos_theta * cos_theta * cos_theta;
    h = std::sqrt(s1 * s1 + s2 * s2);
    const double tan_lat = s1 / s2, sin_lat = s1 / h, cos_lat = s2 / h;
    const double N = a2 / std::sqrt(a2 * cos_lat * cos_lat + b2 * sin_lat * sin_lat);
    return v3(std::atan(tan_lat) * 180.0 / GN_PI, std::atan2(xyz[1], xyz[0]) * 180.0 / GN_PI, p / cos_lat - N);
}
static M3 gn_geo2rotation(const V3& lla) {   // R_ecef_enu
    const double lat = lla[0] * GN_PI / 180.0, lon = lla[1] * GN_PI / 180.0, sl = std::sin(lat), cl = std::cos(lat), so = std::sin(lon), co = std::cos(lon);
    M3 R;
    R(0, 0) = -so; R(0, 1) = -sl * co; R(0, 2) = cl * co;
    R(1, 0) = co;  R(1, 1) = -sl * so; R(1, 2) = cl * so;
    R(2, 0) = 0;   R(2, 1) = cl;       R(2, 2) = sl;
    return R;
}
static M3 gn_ecef2rotation(const V3& ecef) { return gn_geo2rotation(gn_ecef2geo(ecef)); }
static void gn_sat_azel(const V3& rcv, const V3& sat, double azel[2]) {
    V3 d = sat - rcv; d = d * (1.0 / d.norm());
    const V3 enu = gn_ecef2rotation(rcv).T() * d;
    azel[0] = (std::sqrt(d[0] * d[0] + d[1] * d[1]) < 1e-12) ? 0.0 : std::atan2(enu[0], enu[1]);
    if (azel[0] < 0) azel[0] += 2 * GN_PI;
    azel[1] = std::asin(enu[2]);
}
static double gn_trop_delay(const V3& lla, const double azel[2]) {   // Saastamoinen, standard atmosphere, relative humidity 0.7
    if (lla[2] < -100.0 || 1e4 < lla[2] || azel[1] <= 0) return 0.0;
    const double hgt = lla[2] < 0.0 ? 0.0 : lla[2];
    const double pres = 1013.25 * std::pow(1.0 - 2.2557e-5 * hgt, 5.2568), temp = 15.0 - 6.5e-3 * hgt + 273.16;
    const double e = 6.108 * 0.7 * std::exp((17.15 * temp - 4684.0) / (temp - 38.45)), z = GN_PI / 2.0 - azel[1];
    const double trph = 0.0022768 * pres / (1.0 - 0.00266 * std::cos(2.0 * lla[0] * GN_PI / 180.0) - 0.00028 * hgt / 1e3) / std::cos(z);
    const double trpw = 0.002277 * (1255.0 / temp + 0.05) * e / std::cos(z);
    return trph + trpw;
}
static double gn_ion_delay(double tow, const double* ion_in, const V3& lla, const double azel[2]) {   // Klobuchar
    static const double ion_default[8] = {0.1118e-07, -0.7451e-08, -0.5961e-07, 0.1192e-06, 0.1167e+06, -0.2294e+06, -0.1311e+06, 0.1049e+07};
    if (lla[2] < -1e3 || azel[1] <= 0) return 0.0;
    double nrm = 0; for (int i = 0; i < 8; i++) nrm += ion_in[i] * ion_in[i];
    const double* ion = nrm <= 0.0 ? ion_default : ion_in;
    const double psi = 0.0137 / (azel[1] / GN_PI + 0.11) - 0.022;
    double phi = lla[0] / 180.0 + psi * std::cos(azel[0]);
    if (phi > 0.416) phi = 0.416; else if (phi < -0.416) phi = -0.416;
    const double lam = lla[1] / 180.0 + psi * std::sin(azel[0]) / std::cos(phi * GN_PI);
    phi += 0.064 * std::cos((lam - 1.617) * GN_PI);
    double tt = 43200.0 * lam + tow;
    tt -= std::floor(tt / 86400.0) * 86400.0;
    const double f = 1.0 + 16.0 * std::pow(0.53 - azel[1] / GN_PI, 3.0);
    double amp = ion[0] + phi * (ion[1] + phi * (ion[2] + phi * ion[3])), per = ion[4] + phi * (ion[5] + phi * (ion[6] + phi * ion[7]));
    amp = amp < 0.0 ? 0.0 : amp; per = per < 72000.0 ? 72000.0 : per;
    const double x = 2.0 * GN_PI * (tt - 50400.0) / per;
    return GN_C * f * (std::fabs(x) < 1.57 ? 5e-9 + amp * (1.0 + x * x * (-0.5 + x * x / 24.0)) : 5e-9);
}
// GnssPsrDoppFactor::Evaluate, gnss_psr_dopp_factor.cpp:49-208.  Blocks: Pose_i(7) SpeedBias_i(9) Pose_j(7) SpeedBias_j(9) rcv_dt(1) rcv_ddt(1) yaw(1) anc(3)
static void eval_gnss(const gfo_window* w, int k, const double* const* p, FactorOut& o, bool jac) {
    const double* d = w->gnss_data + (size_t)k * 16;
    const V3 sv_pos = v3(d[0], d[1], d[2]), sv_vel = v3(d[3], d[4], d[5]);
    const double svdt = d[6], svddt = d[7], tgd = d[8], pr_uura = d[9], dp_uura = d[10], psr = d[11], dopp = d[12], wavelength = d[13], tow = d[14];
    const double ratio = w->gnss_ratio[k];
    const V3 Pi = P_of(p[0]), Vi = v3(p[1][0], p[1][1], p[1][2]), Pj = P_of(p[2]), Vj = v3(p[3][0], p[3][1], p[3][2]);
    const double rcv_dt = p[4][0], rcv_ddt = p[5][0], yaw_diff = p[6][0];
    const V3 ref_ecef = v3(p[7][0], p[7][1], p[7][2]);
    const V3 local_pos = Pi * ratio + Pj * (1.0 - ratio), local_vel = Vi * ratio + Vj * (1.0 - ratio);
    const double sy = std::sin(yaw_diff), cy = std::cos(yaw_diff);
    M3 R_enu_local; R_enu_local(0, 0) = cy; R_enu_local(0, 1) = -sy; R_enu_local(0, 2) = 0; R_enu_local(1, 0) = sy; R_enu_local(1, 1) = cy; R_enu_local(1, 2) = 0;
    R_enu_local(2, 0) = 0; R_enu_local(2, 1) = 0; R_enu_local(2, 2) = 1;
    const M3 R_ecef_enu = gn_ecef2rotation(ref_ecef), R_ecef_local = R_ecef_enu * R_enu_local;
    const V3 P_ecef = R_ecef_local * local_pos + ref_ecef, V_ecef = R_ecef_local * local_vel;
    double ion_delay = 0, tro_delay = 0, azel[2] = {0, GN_PI / 2.0};
    if (P_ecef.norm() > 0) {
        gn_sat_azel(P_ecef, sv_pos, azel);
        const V3 lla = gn_ecef2geo(P_ecef);
        tro_delay = gn_trop_delay(lla, azel);
        ion_delay = gn_ion_delay(tow, w->gnss_iono, lla, azel);
    }
    const double sin_el = std::sin(azel[1]), sin_el_2 = sin_el * sin_el;
    const double pr_weight = sin_el_2 / pr_uura * 10.0, dp_weight = sin_el_2 / dp_uura * 10.0 * 5.0;   // relative_sqrt_info 10, PSR_TO_DOPP_RATIO 5
    const V3 rcv2sat = sv_pos - P_ecef;
    const double rng = rcv2sat.norm();
    const V3 unit = rcv2sat * (1.0 / rng);
    const double psr_sagnac = GN_OMG * (sv_pos[0] * P_ecef[1] - sv_pos[1] * P_ecef[0]) / GN_C;
    const double psr_est = rng + psr_sagnac + rcv_dt - svdt * GN_C + ion_delay + tro_delay + tgd * GN_C;
    o.nres = 2;
    o.r[0] = (psr_est - psr) * pr_weight;
    const double dopp_sagnac = GN_OMG / GN_C * (sv_vel[0] * P_ecef[1] + sv_pos[0] * V_ecef[1] - sv_vel[1] * P_ecef[0] - sv_pos[1] * V_ecef[0]);
    const double dopp_est = (sv_vel - V_ecef).dot(unit) + dopp_sagnac + rcv_ddt - svddt * GN_C;
    o.r[1] = (dopp_est + dopp * wavelength) * dp_weight;
    if (!jac) return;
    o.J.assign(8, {});
    const int gs[8] = {7, 9, 7, 9, 1, 1, 1, 3};
    for (int b = 0; b < 8; b++) o.J[b].assign((size_t)2 * gs[b], 0.0);
    const double norm3 = rng * rng * rng, norm2 = rcv2sat.dot(rcv2sat);
    M3 u2p;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) u2p(i, j) = -((i == j) ? (norm2 - rcv2sat[i] * rcv2sat[i]) / norm3 : (-rcv2sat[i] * rcv2sat[j]) / norm3);
    const V3 dv = sv_vel - V_ecef;
    V3 row_pr, row_dp, row_v;   // unit^T R, (sv_vel - V)^T u2p R, unit^T R
    for (int c = 0; c < 3; c++) {
        double a = 0, bsum = 0;
        for (int r = 0; r < 3; r++) { a += unit[r] * R_ecef_local(r, c); double t = 0; for (int q = 0; q < 3; q++) t += dv[q] * u2p(q, r); bsum += t * R_ecef_local(r, c); }
        row_pr[c] = a; row_dp[c] = bsum; row_v[c] = a;
    }
    for (int c = 0; c < 3; c++) {
        o.J[0][c] = -row_pr[c] * pr_weight * ratio;          o.J[0][7 + c] = row_dp[c] * dp_weight * ratio;
        o.J[1][9 + c] = -row_v[c] * dp_weight * ratio;
        o.J[2][c] = -row_pr[c] * pr_weight * (1.0 - ratio);  o.J[2][7 + c] = row_dp[c] * dp_weight * (1.0 - ratio);
        o.J[3][9 + c] = -row_v[c] * dp_weight * (1.0 - ratio);
        o.J[7][c] = -unit[c] * pr_weight;
    }
    o.J[4][0] = pr_weight; o.J[4][1] = 0;
    o.J[5][0] = 0; o.J[5][1] = dp_weight;
    M3 d_yaw; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) d_yaw(i, j) = 0;
    d_yaw(0, 0) = -sy; d_yaw(0, 1) = -cy; d_yaw(1, 0) = cy; d_yaw(1, 1) = -sy;
    o.J[6][0] = -unit.dot(R_ecef_enu * (d_yaw * local_pos)) * pr_weight;
    o.J[6][1] = -unit.dot(R_ecef_enu * (d_yaw * local_vel)) * dp_weight;
}
// DtDdtFactor (gnss_dt_ddt_factor.cpp): blocks rcv_dt_i, rcv_dt_j, rcv_ddt_i, rcv_ddt_j; dt_info_coeff 50
static void eval_dt_ddt(double delta_t, const double* const* p, FactorOut& o, bool jac) {
    o.nres = 1;
    o.r[0] = (p[1][0] - p[0][0] - 0.5 * (p[2][0] + p[3][0]) * delta_t) * 50.0;
    if (!jac) return;
    o.J.assign(4, std::vector<double>(1, 0.0));
    o.J[0][0] = -50.0; o.J[1][0] = 50.0; o.J[2][0] = -0.5 * delta_t * 50.0; o.J[3][0] = -0.5 * delta_t * 50.0;
}
// DdtSmoothFactor (gnss_ddt_smooth_factor.cpp)
static void eval_ddt_smooth(double weight, const double* const* p, FactorOut& o, bool jac) {
    o.nres = 1;
    o.r[0] = (p[0][0] - p[1][0]) * weight;
    if (!jac) return;
    o.J.assign(2, std::vector<double>(1, 0.0));
    o.J[0][0] = weight; o.J[1][0] = -weight;
}
// PoseAnchorFactor (pose_anchor_factor.cpp), sqrt_info 120
static void eval_anchor(const double* anchor, const double* const* p, FactorOut& o, bool jac) {
    const double si = 120.0;
    o.nres = 6;
    for (int i = 0; i < 3; i++) o.r[i] = (p[0][i] - anchor[i]) * si;
    const Quat cq = Q_of(p[0]), aq(anchor[6], anchor[3], anchor[4], anchor[5]);
    const Quat ai = aq.inverse(), e = cq * ai;
    o.r[3] = 2.0 * e.x * si; o.r[4] = 2.0 * e.y * si; o.r[5] = 2.0 * e.z * si;
    if (!jac) return;
    o.J.assign(1, std::vector<double>(42, 0.0));
    for (int i = 0; i < 3; i++) o.J[0][(size_t)i * 7 + i] = 2.0 * si;
    const double Jq[9] = {ai.w, ai.z, -ai.y, -ai.z, ai.w, ai.x, ai.y, -ai.x, ai.w};
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o.J[0][(size_t)(3 + r) * 7 + 3 + c] = Jq[3 * r + c] * 2.0 * si;
}

static void prior_dx(const gfo_window* w, const State& s, std::vector<double>& dx) {
    dx.assign(w->prior_n, 0.0);
    int idx = 0, off0 = 0;
    for (int b = 0; b < w->prior_nblocks; b++) {
        const int id = w->prior_block_id[b], kind = id / 4096, gs = gsize_of(kind);
        const double* x = s.ptr(id);
        const double* x0 = w->prior_x0 + off0;
        if (gs != 7) for (int i = 0; i < gs; i++) dx[idx + i] = x[i] - x0[i];
        else {
            for (int i = 0; i < 3; i++) dx[idx + i] = x[i] - x0[i];
            Quat dq = Quat(x0[6], x0[3], x0[4], x0[5]).inverse() * Quat(x[6], x[3], x[4], x[5]);
            V3 v = dq.vec() * 2.0;
            if (!(dq.w >= 0)) v = dq.vec() * -2.0;
            for (int i = 0; i < 3; i++) dx[idx + 3 + i] = v[i];
        }
        idx += lsize_of(kind); off0 += gs;
    }
}

// HuberLoss(1.0) + Corrector (loss_function.h, corrector.cc; same code at marginalization_factor.cpp:27-57)
static void huber(double s, double rho[3]) {
    if (s > 1.0) { const double r = std::sqrt(s); rho[0] = 2.0 * r - 1.0; rho[1] = std::max(DBL_MIN, 1.0 / r); rho[2] = -rho[1] / (2.0 * s); }
    else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
}

// ------------------------------------------------------------------ problem assembly
struct RowBlock { int nres = 0; int nb = 0; std::vector<int> id, lsz; std::vector<double> r; std::vector<std::vector<double>> J; /* J[q]: nres x lsz[q], row-major (local) */
    int add(int bid_, int ls) { id.push_back(bid_); lsz.push_back(ls); J.emplace_back(); return nb++; } };

struct Problem {
    const gfo_window* w;
    std::vector<int> ids;                // free parameter blocks in column order: e-blocks (free features) first
    std::map<int, int> col_of;           // id -> first column
    int n_e = 0, n_cols = 0;
    bool is_const(int id) const {
        const int kind = id / 4096, i = id % 4096;
        switch (kind) {
            case POSE: case SPEEDBIAS: return w->fix_poses != 0;
            case EX_POSE: return w->fix_ex_pose != 0; case EX_WHEEL: return w->fix_ex_wheel != 0;
            case SX: case SY: case SW: return w->fix_ix != 0;
            case TD: return w->fix_td != 0; case TD_WHEEL: return w->fix_td_wheel != 0;
            case YAW: return true;   // problem.SetParameterBlockConstant(para_yaw_enu_local), estimator.cpp:2932
            case RCV_DT: case RCV_DDT: case ANC: return false;
            default: return w->feature_fixed && w->feature_fixed[i] != 0;
        }
    }
    void build() {
        std::vector<char> used_feat(w->n_feature, 0);
        for (int k = 0; k < w->n_visual; k++) used_feat[w->vis_feature[k]] = 1;
        ids.clear(); col_of.clear(); n_cols = 0;
        auto add = [&](int id) { if (is_const(id)) return; col_of[id] = n_cols; ids.push_back(id); n_cols += lsize_of(id / 4096); };
        for (int f = 0; f < w->n_feature; f++) if (used_feat[f]) add(bid(FEATURE, f));
        n_e = n_cols;
        for (int i = 0; i <= w->W; i++) { add(bid(POSE, i)); add(bid(SPEEDBIAS, i)); }
        add(bid(EX_POSE, 0));
        // wheel blocks take part when a wheel factor or the prior mentions them (Ceres drops parameter blocks without residuals)
        auto in_prior = [&](int id) { for (int q = 0; q < w->prior_nblocks; q++) if (w->prior_block_id[q] == id) return true; return false; };
        for (int id : {bid(EX_WHEEL, 0), bid(SX, 0), bid(SY, 0), bid(SW, 0)}) if (w->n_wheel > 0 || in_prior(id)) add(id);
        add(bid(TD, 0));
        if (w->n_wheel > 0 || in_prior(bid(TD_WHEEL, 0))) add(bid(TD_WHEEL, 0));
        if (w->gnss_enabled) {   // receiver clock / anchor blocks that some residual block (or the prior) mentions
            const bool fac = !w->gnss_lowspeed;
            for (int i = 0; i <= w->W; i++) for (int q = 0; q < 4; q++) if (fac || in_prior(bid(RCV_DT, 4 * i + q))) add(bid(RCV_DT, 4 * i + q));
            for (int i = 0; i <= w->W; i++) if (fac || in_prior(bid(RCV_DDT, i))) add(bid(RCV_DDT, i));
            if ((fac && w->n_gnss > 0) || in_prior(bid(ANC, 0))) add(bid(ANC, 0));
        }
    }
    // Evaluates all residual blocks at s.  Returns cost = 1/2 sum rho(|r|^2).  If rows != null also the corrected residuals/Jacobians
    // of the free blocks (local parameterisation: first 6 columns of a 7-column pose Jacobian, pose_local_parameterization.cpp:30-36).
    double evaluate(const State& s, std::vector<RowBlock>* rows) const {
        double cost = 0;
        if (rows) rows->clear();
        FactorOut o;
        auto emit = [&](const int* fid, int nb, bool robust) {
            double sq = 0;
            for (int i = 0; i < o.nres; i++) sq += o.r[i] * o.r[i];
            double rho[3] = {sq, 1.0, 0.0};
            if (robust) huber(sq, rho);
            cost += 0.5 * rho[0];
            if (!rows) return;
            RowBlock rb; rb.nres = o.nres; rb.nb = 0;
            double sqrt_rho1 = 1.0, residual_scaling = 1.0, alpha_sq_norm = 0.0;
            if (robust) {  // Corrector
                sqrt_rho1 = std::sqrt(rho[1]);
                if (sq == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; }
                else {
                    const double D = 1.0 + 2.0 * sq * rho[2] / rho[1], alpha = 1.0 - std::sqrt(D);
                    residual_scaling = sqrt_rho1 / (1 - alpha); alpha_sq_norm = alpha / sq;
                }
            }
            for (int b = 0; b < nb; b++) {
                if (is_const(fid[b])) continue;
                const int kind = fid[b] / 4096, gs = gsize_of(kind), ls = lsize_of(kind), q = rb.add(fid[b], ls);
                rb.J[q].assign((size_t)o.nres * ls, 0.0);
                for (int r = 0; r < o.nres; r++) for (int c = 0; c < ls; c++) rb.J[q][(size_t)r * ls + c] = o.J[b][(size_t)r * gs + c];
                if (robust) {  // J = sqrt_rho1 * (J - alpha_sq_norm * r * (r^T J))
                    for (int c = 0; c < ls; c++) {
                        double rtj = 0;
                        for (int r = 0; r < o.nres; r++) rtj += o.r[r] * rb.J[q][(size_t)r * ls + c];
                        for (int r = 0; r < o.nres; r++) rb.J[q][(size_t)r * ls + c] = sqrt_rho1 * (rb.J[q][(size_t)r * ls + c] - alpha_sq_norm * o.r[r] * rtj);
                    }
                }
            }
            rb.r.assign(o.r, o.r + o.nres);
            for (auto& v : rb.r) v *= residual_scaling;
            if (rb.nb > 0) rows->push_back(std::move(rb));
        };
        const bool jac = rows != nullptr;
        if (w->has_anchor) {  // estimator.cpp:2943-2951
            const int fid[1] = {bid(POSE, 0)};
            const double* p[1] = {s.ptr(fid[0])};
            eval_anchor(w->anchor_value, p, o, jac);
            emit(fid, 1, false);
        }
        if (w->prior_n > 0) {  // estimator.cpp:3102-3108
            std::vector<double> dx;
            prior_dx(w, s, dx);
            const int n = w->prior_n;
            // residual = r0 + J0 * dx (n can exceed FactorOut's inline buffer; handle separately)
            std::vector<double> r(n);
            double sq = 0;
            for (int i = 0; i < n; i++) { double v = w->prior_r[i]; for (int j = 0; j < n; j++) v += w->prior_J[(size_t)i * n + j] * dx[j]; r[i] = v; sq += v * v; }
            cost += 0.5 * sq;
            if (rows) {
                RowBlock rb; rb.nres = n; rb.r = r;
                int idx = 0;
                for (int b = 0; b < w->prior_nblocks; b++) {
                    const int id = w->prior_block_id[b], ls = lsize_of(id / 4096);
                    if (!is_const(id)) {
                        const int q = rb.add(id, ls);
                        rb.J[q].assign((size_t)n * ls, 0.0);
                        for (int r2 = 0; r2 < n; r2++) for (int c = 0; c < ls; c++) rb.J[q][(size_t)r2 * ls + c] = w->prior_J[(size_t)r2 * n + idx + c];
                    }
                    idx += ls;
                }
                if (rb.nb > 0) rows->push_back(std::move(rb));
            }
        }
        for (int k = 0; k < w->n_imu; k++) {  // estimator.cpp:3109-3119
            const int i = w->imu_i[k], j = i + 1;
            const int fid[4] = {bid(POSE, i), bid(SPEEDBIAS, i), bid(POSE, j), bid(SPEEDBIAS, j)};
            const double* p[4] = {s.ptr(fid[0]), s.ptr(fid[1]), s.ptr(fid[2]), s.ptr(fid[3])};
            eval_imu(w, k, p, o, jac);
            emit(fid, 4, false);
        }
        for (int k = 0; k < w->n_wheel; k++) {  // estimator.cpp:3120-3151
            const int i = w->wh_i[k], j = i + 1;
            const int fid[7] = {bid(POSE, i), bid(POSE, j), bid(EX_WHEEL, 0), bid(SX, 0), bid(SY, 0), bid(SW, 0), bid(TD_WHEEL, 0)};
            const double* p[7];
            for (int q = 0; q < 7; q++) p[q] = s.ptr(fid[q]);
            eval_wheel(w, k, p, o, jac);
            emit(fid, 7, false);
        }
        if (w->gnss_enabled && !w->gnss_lowspeed) {  // estimator.cpp:3178-3229
            for (int k = 0; k < w->n_gnss; k++) {
                const int i = w->gnss_frame[k], l = w->gnss_lower[k];
                const int fid[8] = {bid(POSE, l), bid(SPEEDBIAS, l), bid(POSE, l + 1), bid(SPEEDBIAS, l + 1), bid(RCV_DT, 4 * i + w->gnss_sys[k]), bid(RCV_DDT, i), bid(YAW, 0), bid(ANC, 0)};
                const double* p[8];
                for (int q = 0; q < 8; q++) p[q] = s.ptr(fid[q]);
                eval_gnss(w, k, p, o, jac);
                emit(fid, 8, false);
            }
            for (int q = 0; q < 4; q++)
                for (int i = 0; i < w->W; i++) {
                    const int fid[4] = {bid(RCV_DT, 4 * i + q), bid(RCV_DT, 4 * (i + 1) + q), bid(RCV_DDT, i), bid(RCV_DDT, i + 1)};
                    const double* p[4] = {s.ptr(fid[0]), s.ptr(fid[1]), s.ptr(fid[2]), s.ptr(fid[3])};
                    eval_dt_ddt(w->gnss_headers[i + 1] - w->gnss_headers[i], p, o, jac);
                    emit(fid, 4, false);
                }
            for (int i = 0; i < w->W; i++) {
                const int fid[2] = {bid(RCV_DDT, i), bid(RCV_DDT, i + 1)};
                const double* p[2] = {s.ptr(fid[0]), s.ptr(fid[1])};
                eval_ddt_smooth(w->gnss_ddt_weight, p, o, jac);
                emit(fid, 2, false);
            }
        }
        for (int k = 0; k < w->n_visual; k++) {  // estimator.cpp:3269-3297, Huber(1.0)
            const int fid[5] = {bid(POSE, w->vis_i[k]), bid(POSE, w->vis_j[k]), bid(EX_POSE, 0), bid(FEATURE, w->vis_feature[k]), bid(TD, 0)};
            const double* p[5];
            for (int q = 0; q < 5; q++) p[q] = s.ptr(fid[q]);
            eval_visual(w, k, p, o, jac);
            emit(fid, 5, true);
        }
        return cost;
    }
};

// PoseLocalParameterization::Plus (pose_local_parameterization.cpp:12-28) and plain addition for the others
static void plus_block(int kind, const double* x, const double* d, double* out) {
    const int gs = gsize_of(kind);
    if (gs != 7) { for (int i = 0; i < gs; i++) out[i] = x[i] + d[i]; return; }
    for (int i = 0; i < 3; i++) out[i] = x[i] + d[i];
    Quat q = (Quat(x[6], x[3], x[4], x[5]) * deltaQ(v3(d[3], d[4], d[5]))).normalized();
    out[3] = q.x; out[4] = q.y; out[5] = q.z; out[6] = q.w;
}

// ------------------------------------------------------------------ block-sparse Jacobian ops (ceres BlockSparseMatrix subset)
struct Jac {
    std::vector<RowBlock>* rows; const Problem* P;
    void squaredColumnNorm(std::vector<double>& n) const {
        n.assign(P->n_cols, 0.0);
        for (auto& rb : *rows) for (int q = 0; q < rb.nb; q++) { const int c0 = P->col_of.at(rb.id[q]); for (int r = 0; r < rb.nres; r++) for (int c = 0; c < rb.lsz[q]; c++) { double v = rb.J[q][(size_t)r * rb.lsz[q] + c]; n[c0 + c] += v * v; } }
    }
    void scaleColumns(const std::vector<double>& s) {
        for (auto& rb : *rows) for (int q = 0; q < rb.nb; q++) { const int c0 = P->col_of.at(rb.id[q]); for (int r = 0; r < rb.nres; r++) for (int c = 0; c < rb.lsz[q]; c++) rb.J[q][(size_t)r * rb.lsz[q] + c] *= s[c0 + c]; }
    }
    int numRows() const { int n = 0; for (auto& rb : *rows) n += rb.nres; return n; }
    void residuals(std::vector<double>& r) const { r.clear(); for (auto& rb : *rows) r.insert(r.end(), rb.r.begin(), rb.r.end()); }
    void leftMultiply(const std::vector<double>& x, std::vector<double>& y) const {  // y += J^T x
        int r0 = 0;
        for (auto& rb : *rows) { for (int q = 0; q < rb.nb; q++) { const int c0 = P->col_of.at(rb.id[q]); for (int r = 0; r < rb.nres; r++) for (int c = 0; c < rb.lsz[q]; c++) y[c0 + c] += rb.J[q][(size_t)r * rb.lsz[q] + c] * x[r0 + r]; } r0 += rb.nres; }
    }
    void rightMultiply(const std::vector<double>& x, std::vector<double>& y) const {  // y += J x
        int r0 = 0;
        for (auto& rb : *rows) { for (int q = 0; q < rb.nb; q++) { const int c0 = P->col_of.at(rb.id[q]); for (int r = 0; r < rb.nres; r++) { double s = 0; for (int c = 0; c < rb.lsz[q]; c++) s += rb.J[q][(size_t)r * rb.lsz[q] + c] * x[c0 + c]; y[r0 + r] += s; } } r0 += rb.nres; }
    }
};

// DENSE_SCHUR: minimise |J y - r|^2 + |D y|^2 ; e-blocks are the first P->n_e (scalar) columns.  Returns false on Cholesky failure.
static bool dense_schur_solve(const Jac& J, const std::vector<double>& D, std::vector<double>& y) {
    const Problem& P = *J.P;
    const int ne = P.n_e, nf = P.n_cols - ne;
    std::vector<double> ete(ne, 0.0), etb(ne, 0.0);
    DMat etf(ne, nf), S(nf, nf);
    std::vector<double> rhs(nf, 0.0);
    for (int i = 0; i < ne; i++) ete[i] = D[i] * D[i];
    for (int i = 0; i < nf; i++) S(i, i) = D[ne + i] * D[ne + i];
    for (auto& rb : *J.rows) {
        int eq = -1;
        for (int q = 0; q < rb.nb; q++) if (P.col_of.at(rb.id[q]) < ne) eq = q;
        for (int q = 0; q < rb.nb; q++) {
            if (q == eq) continue;
            const int c0 = P.col_of.at(rb.id[q]) - ne, lq = rb.lsz[q];
            for (int q2 = 0; q2 < rb.nb; q2++) {
                if (q2 == eq) continue;
                const int c1 = P.col_of.at(rb.id[q2]) - ne, l2 = rb.lsz[q2];
                if (c1 < c0) continue;  // upper triangle by blocks (diagonal block computed in full)
                for (int a = 0; a < lq; a++) for (int b = 0; b < l2; b++) {
                    double s = 0;
                    for (int r = 0; r < rb.nres; r++) s += rb.J[q][(size_t)r * lq + a] * rb.J[q2][(size_t)r * l2 + b];
                    S(c0 + a, c1 + b) += s;
                }
            }
            for (int a = 0; a < lq; a++) { double s = 0; for (int r = 0; r < rb.nres; r++) s += rb.J[q][(size_t)r * lq + a] * rb.r[r]; rhs[c0 + a] += s; }
        }
        if (eq >= 0) {
            const int e = P.col_of.at(rb.id[eq]);
            for (int r = 0; r < rb.nres; r++) { const double ev = rb.J[eq][r]; ete[e] += ev * ev; etb[e] += ev * rb.r[r]; }
            for (int q = 0; q < rb.nb; q++) {
                if (q == eq) continue;
                const int c0 = P.col_of.at(rb.id[q]) - ne, lq = rb.lsz[q];
                for (int a = 0; a < lq; a++) { double s = 0; for (int r = 0; r < rb.nres; r++) s += rb.J[eq][r] * rb.J[q][(size_t)r * lq + a]; etf(e, c0 + a) += s; }
            }
        }
    }
    for (int i = 0; i < nf; i++) for (int j = 0; j < i; j++) S(i, j) = S(j, i);  // mirror (blocks below the diagonal were skipped)
    for (int e = 0; e < ne; e++) {  // S -= (E^T F)^T ete^-1 (E^T F)
        const double inv = 1.0 / ete[e];
        const double* row = &etf.a[(size_t)e * nf];
        std::vector<int> nz;
        for (int a = 0; a < nf; a++) if (row[a] != 0.0) nz.push_back(a);
        for (int a : nz) { const double fa = row[a] * inv; rhs[a] -= fa * etb[e]; for (int b : nz) S(a, b) -= fa * row[b]; }
    }
    if (!cholesky_lower(S.a.data(), nf, nf)) return false;
    cholesky_solve(S.a.data(), nf, nf, rhs.data());
    y.assign(P.n_cols, 0.0);
    for (int i = 0; i < nf; i++) y[ne + i] = rhs[i];
    for (int e = 0; e < ne; e++) { double s = etb[e]; for (int a = 0; a < nf; a++) s -= etf(e, a) * rhs[a]; y[e] = s / ete[e]; }
    for (double v : y) if (!std::isfinite(v)) return false;
    return true;
}

struct Dogleg {  // dogleg_strategy.cc, TRADITIONAL_DOGLEG
    double radius = 1e4, mu = 1e-8, alpha = 0, dogleg_step_norm = 0;
    const double min_diagonal = 1e-6, max_diagonal = 1e32, min_mu = 1e-8, max_mu = 1.0, mu_increase = 10.0;
    bool reuse = false;
    std::vector<double> diagonal, gradient, gn;
    bool computeStep(const Jac& J, const std::vector<double>& res, std::vector<double>& step) {
        const int n = J.P->n_cols;
        if (!reuse) {
            reuse = true;
            J.squaredColumnNorm(diagonal);
            for (auto& d : diagonal) d = std::sqrt(std::min(std::max(d, min_diagonal), max_diagonal));
            gradient.assign(n, 0.0);
            J.leftMultiply(res, gradient);
            for (int i = 0; i < n; i++) gradient[i] /= diagonal[i];
            {   // Cauchy point
                std::vector<double> sg(n), Jg(res.size(), 0.0);
                for (int i = 0; i < n; i++) sg[i] = gradient[i] / diagonal[i];
                J.rightMultiply(sg, Jg);
                double gn2 = 0, jg2 = 0;
                for (double v : gradient) gn2 += v * v;
                for (double v : Jg) jg2 += v * v;
                alpha = gn2 / jg2;
            }
            bool ok = false;
            while (mu < max_mu) {
                std::vector<double> lm(n);
                for (int i = 0; i < n; i++) lm[i] = diagonal[i] * std::sqrt(mu);
                if (dense_schur_solve(J, lm, gn)) { ok = true; break; }
                mu *= mu_increase;
            }
            if (!ok) return false;
            for (int i = 0; i < n; i++) gn[i] *= -diagonal[i];
        }
        step.assign(n, 0.0);
        double gnorm = 0, gnn = 0;
        for (double v : gradient) gnorm += v * v;
        for (double v : gn) gnn += v * v;
        gnorm = std::sqrt(gnorm); gnn = std::sqrt(gnn);
        if (gnn <= radius) { step = gn; dogleg_step_norm = gnn; }
        else if (gnorm * alpha >= radius) { for (int i = 0; i < n; i++) step[i] = -(radius / gnorm) * gradient[i]; dogleg_step_norm = radius; }
        else {
            double gdot = 0;
            for (int i = 0; i < n; i++) gdot += gradient[i] * gn[i];
            const double b_dot_a = -alpha * gdot, a_sq = std::pow(alpha * gnorm, 2.0), bma = a_sq - 2 * b_dot_a + std::pow(gnn, 2);
            const double c = b_dot_a - a_sq, d = std::sqrt(c * c + bma * (std::pow(radius, 2.0) - a_sq));
            const double beta = (c <= 0) ? (d - c) / bma : (radius * radius - a_sq) / (d + c);
            double nn = 0;
            for (int i = 0; i < n; i++) { step[i] = (-alpha * (1.0 - beta)) * gradient[i] + beta * gn[i]; nn += step[i] * step[i]; }
            dogleg_step_norm = std::sqrt(nn);
        }
        for (int i = 0; i < n; i++) step[i] /= diagonal[i];
        return true;
    }
    void accepted(double q) {
        if (q < 0.25) radius *= 0.5;
        if (q > 0.75) radius = std::max(radius, 3.0 * dogleg_step_norm);
        mu = std::max(min_mu, 2.0 * mu / mu_increase);
        reuse = false;
    }
    void rejected() { radius *= 0.5; reuse = true; }
    void invalid() { mu *= mu_increase; reuse = false; }
};

static void gather_x(const Problem& P, const State& s, std::vector<double>& x) {
    x.clear();
    for (int id : P.ids) { const double* p = s.ptr(id); x.insert(x.end(), p, p + gsize_of(id / 4096)); }
}

// trust_region_minimizer.cc
static int solve(gfo_window* w, int max_iters, gfo_summary* sum) {
    Problem P; P.w = w; P.build();
    State x; x.load(w);
    if (w->fix_poses) for (int i = 0; i <= w->W; i++) { x.sb[9 * i] = x.sb[9 * i + 1] = x.sb[9 * i + 2] = 0; }  // estimator.cpp:3233-3246
    std::vector<RowBlock> rows;
    Jac J{&rows, &P};
    double x_cost = P.evaluate(x, &rows);
    std::vector<double> scale;
    J.squaredColumnNorm(scale);
    for (auto& v : scale) v = 1.0 / (1.0 + std::sqrt(v));
    J.scaleColumns(scale);
    std::vector<double> res;
    J.residuals(res);
    auto grad_max = [&]() {  // unscaled gradient max-norm = |S^-1 J_s^T r|_inf
        std::vector<double> g(P.n_cols, 0.0);
        J.leftMultiply(res, g);
        double m = 0;
        for (int i = 0; i < P.n_cols; i++) m = std::max(m, std::abs(g[i] / scale[i]));
        return m;
    };
    sum->initial_cost = x_cost; sum->iterations = 0; sum->successful_steps = 0; sum->termination = 0;
    Dogleg tr;
    int invalid_run = 0;
    bool last_successful = true;
    double gmax = grad_max();
    if (P.n_cols == 0) { sum->final_cost = x_cost; x.store(w); return 0; }
    for (int iter = 1;; iter++) {
        // FinalizeIterationAndCheckIfMinimizerCanContinue
        if (iter - 1 >= max_iters) { sum->termination = 0; break; }
        if (last_successful && gmax <= 1e-10) { sum->termination = 3; break; }
        if (tr.radius <= 1e-32) { sum->termination = 4; break; }
        sum->iterations = iter;
        std::vector<double> step;
        bool valid = tr.computeStep(J, res, step);
        double model_cost_change = 0;
        if (valid) {
            std::vector<double> mr(res.size(), 0.0);
            J.rightMultiply(step, mr);
            for (size_t i = 0; i < res.size(); i++) model_cost_change -= mr[i] * (res[i] + mr[i] / 2.0);
            valid = model_cost_change > 0.0;
        }
        if (!valid) {
            last_successful = false;
            if (++invalid_run >= 5) { sum->termination = 4; break; }
            tr.invalid();
            continue;
        }
        invalid_run = 0;
        State cand = x;
        std::vector<double> xv, cv;
        gather_x(P, x, xv);
        for (int id : P.ids) {
            const int kind = id / 4096, c0 = P.col_of.at(id), ls = lsize_of(kind);
            double d[9];
            for (int i = 0; i < ls; i++) d[i] = step[c0 + i] * scale[c0 + i];
            // PoseSubsetParameterization::Plus (pose_subset_parameterization.cpp:27-56): masked components of the increment are dropped; the Jacobian
            // the solver saw is the full one (ComputeJacobian :57-64 is the identity whatever the mask), so step, model cost change and radius are those of
            // the unmasked block -- only the evaluated point differs
            if (kind == EX_POSE || kind == EX_WHEEL) { const int mask = kind == EX_POSE ? w->ex_pose_mask : w->ex_wheel_mask; for (int i = 0; i < 6; i++) if ((mask >> i) & 1) d[i] = 0.0; }
            plus_block(kind, x.ptr(id), d, cand.ptr(id));
        }
        const double cand_cost = P.evaluate(cand, nullptr);
        gather_x(P, cand, cv);
        double step_norm = 0, x_norm = 0;
        for (size_t i = 0; i < xv.size(); i++) { step_norm += (xv[i] - cv[i]) * (xv[i] - cv[i]); x_norm += xv[i] * xv[i]; }
        step_norm = std::sqrt(step_norm); x_norm = std::sqrt(x_norm);
        if (step_norm <= 1e-8 * (x_norm + 1e-8)) { sum->termination = 2; break; }          // ParameterToleranceReached
        if (std::abs(x_cost - cand_cost) <= 1e-6 * x_cost) { sum->termination = 1; break; }  // FunctionToleranceReached (cost_change vs current cost)
        const double rel = (x_cost - cand_cost) / model_cost_change;
        if (rel > 1e-3) {
            x = cand; x_cost = cand_cost;
            P.evaluate(x, &rows);
            J.scaleColumns(scale);
            J.residuals(res);
            gmax = grad_max();
            tr.accepted(rel);
            last_successful = true;
            sum->successful_steps++;
        } else { tr.rejected(); last_successful = false; }
    }
    sum->final_cost = x_cost; sum->radius = tr.radius;
    x.store(w);
    return 0;
}

// ------------------------------------------------------------------ marginalisation (marginalization_factor.cpp:119-308, estimator.cpp:3334-3631)
// debug sink (tests only): when set, marginalize() copies the assembled system A (pos x pos, row-major), b, and the Schur complement A_r, b_r
struct MargDebug { int cap; int pos, m, n; double* A; double* b; double* Ar; double* br; };
static thread_local MargDebug* g_marg_debug = nullptr;
static int marginalize(const gfo_window* w, int mode, int cap_n, int* out_n, int* out_nblocks, int* out_block_id, double* out_J, double* out_r, double* out_x0, int* out_m) {
    Problem P; P.w = w;  // only for evaluation helpers: in marginalisation NO block is constant (ResidualBlockInfo has no such notion)
    gfo_window wf = *w;
    wf.fix_ex_pose = wf.fix_ex_wheel = wf.fix_ix = wf.fix_td = wf.fix_td_wheel = wf.fix_poses = 0; wf.feature_fixed = nullptr;
    P.w = &wf;
    State s; s.load(w);
    const int W = w->W;
    std::vector<int> drop, keep;  // block ids by first appearance
    std::map<int, int> seen;
    auto touch = [&](int id, bool dropped) {
        auto it = seen.find(id);
        if (it == seen.end()) { seen[id] = dropped ? 1 : 0; (dropped ? drop : keep).push_back(id); }
        else if (dropped && it->second == 0) { it->second = 1; keep.erase(std::find(keep.begin(), keep.end(), id)); drop.push_back(id); }
    };
    struct Fac { int nres; int nb; std::vector<int> id; std::vector<double> r; std::vector<std::vector<double>> J; };  // J local
    std::vector<Fac> facs;
    FactorOut o;
    auto push = [&](const int* fid, int nb, bool robust, const std::vector<int>& dropset) {
        Fac f; f.nres = o.nres; f.nb = nb; f.id.assign(fid, fid + nb); f.r.assign(o.r, o.r + o.nres); f.J.resize(nb);
        double sq = 0;
        for (double v : f.r) sq += v * v;
        double sqrt_rho1 = 1, residual_scaling = 1, alpha_sq_norm = 0;
        if (robust) {
            double rho[3]; huber(sq, rho);
            sqrt_rho1 = std::sqrt(rho[1]);
            if (sq == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0; }
            else { const double D = 1.0 + 2.0 * sq * rho[2] / rho[1], alpha = 1.0 - std::sqrt(D); residual_scaling = sqrt_rho1 / (1 - alpha); alpha_sq_norm = alpha / sq; }
        }
        for (int b = 0; b < nb; b++) {
            const int kind = fid[b] / 4096, gs = gsize_of(kind), ls = lsize_of(kind);
            f.J[b].assign((size_t)o.nres * ls, 0.0);
            for (int r = 0; r < o.nres; r++) for (int c = 0; c < ls; c++) f.J[b][(size_t)r * ls + c] = o.J[b][(size_t)r * gs + c];
            if (robust) for (int c = 0; c < ls; c++) {
                double rtj = 0;
                for (int r = 0; r < o.nres; r++) rtj += o.r[r] * f.J[b][(size_t)r * ls + c];
                for (int r = 0; r < o.nres; r++) f.J[b][(size_t)r * ls + c] = sqrt_rho1 * (f.J[b][(size_t)r * ls + c] - alpha_sq_norm * o.r[r] * rtj);
            }
        }
        for (auto& v : f.r) v *= residual_scaling;
        for (int b = 0; b < nb; b++) touch(fid[b], std::find(dropset.begin(), dropset.end(), b) != dropset.end());
        facs.push_back(std::move(f));
    };
    // prior first (estimator.cpp:3339-3353 / :3543-3558)
    if (w->prior_n > 0) {
        const int n = w->prior_n;
        std::vector<double> dx; prior_dx(w, s, dx);
        Fac f; f.nres = n; f.nb = w->prior_nblocks; f.r.resize(n); f.J.resize(f.nb);
        for (int i = 0; i < n; i++) { double v = w->prior_r[i]; for (int j = 0; j < n; j++) v += w->prior_J[(size_t)i * n + j] * dx[j]; f.r[i] = v; }
        int idx = 0;
        for (int b = 0; b < w->prior_nblocks; b++) {
            const int id = w->prior_block_id[b], ls = lsize_of(id / 4096);
            f.id.push_back(id); f.J[b].assign((size_t)n * ls, 0.0);
            for (int r = 0; r < n; r++) for (int c = 0; c < ls; c++) f.J[b][(size_t)r * ls + c] = w->prior_J[(size_t)r * n + idx + c];
            idx += ls;
            const bool dropped = mode == 0 ? (id == bid(POSE, 0) || id == bid(SPEEDBIAS, 0)) : (id == bid(POSE, W - 1));
            touch(id, dropped);
        }
        facs.push_back(std::move(f));
    }
    if (mode == 0) {
        for (int k = 0; k < w->n_imu; k++) if (w->imu_i[k] == 0 && w->imu_sum_dt[k] < 10.0) {  // estimator.cpp:3354-3364
            const int fid[4] = {bid(POSE, 0), bid(SPEEDBIAS, 0), bid(POSE, 1), bid(SPEEDBIAS, 1)};
            const double* p[4] = {s.ptr(fid[0]), s.ptr(fid[1]), s.ptr(fid[2]), s.ptr(fid[3])};
            eval_imu(w, k, p, o, true); push(fid, 4, false, {0, 1});
        }
        for (int k = 0; k < w->n_wheel; k++) if (w->wh_i[k] == 0 && w->wh_sum_dt[k] < 10.0) {  // :3365-3375
            const int fid[7] = {bid(POSE, 0), bid(POSE, 1), bid(EX_WHEEL, 0), bid(SX, 0), bid(SY, 0), bid(SW, 0), bid(TD_WHEEL, 0)};
            const double* p[7]; for (int q = 0; q < 7; q++) p[q] = s.ptr(fid[q]);
            eval_wheel(w, k, p, o, true); push(fid, 7, false, {0});
        }
        if (w->gnss_enabled) {  // :3390-3431: whenever gnss_ready, also at low speed
            for (int k = 0; k < w->n_gnss; k++) if (w->gnss_frame[k] == 0) {
                const int fid[8] = {bid(POSE, 0), bid(SPEEDBIAS, 0), bid(POSE, 1), bid(SPEEDBIAS, 1), bid(RCV_DT, w->gnss_sys[k]), bid(RCV_DDT, 0), bid(YAW, 0), bid(ANC, 0)};
                const double* p[8]; for (int q = 0; q < 8; q++) p[q] = s.ptr(fid[q]);
                eval_gnss(w, k, p, o, true); push(fid, 8, false, {0, 1, 4, 5});
            }
            for (int q = 0; q < 4; q++) {
                const int fid[4] = {bid(RCV_DT, q), bid(RCV_DT, 4 + q), bid(RCV_DDT, 0), bid(RCV_DDT, 1)};
                const double* p[4] = {s.ptr(fid[0]), s.ptr(fid[1]), s.ptr(fid[2]), s.ptr(fid[3])};
                eval_dt_ddt(w->gnss_headers[1] - w->gnss_headers[0], p, o, true); push(fid, 4, false, {0, 2});
            }
            const int fid[2] = {bid(RCV_DDT, 0), bid(RCV_DDT, 1)};
            const double* p[2] = {s.ptr(fid[0]), s.ptr(fid[1])};
            eval_ddt_smooth(w->gnss_ddt_weight, p, o, true); push(fid, 2, false, {0});
        }
        for (int k = 0; k < w->n_visual; k++) if (w->vis_i[k] == 0) {  // :3433-3462 (features starting at frame 0)
            const int fid[5] = {bid(POSE, 0), bid(POSE, w->vis_j[k]), bid(EX_POSE, 0), bid(FEATURE, w->vis_feature[k]), bid(TD, 0)};
            const double* p[5]; for (int q = 0; q < 5; q++) p[q] = s.ptr(fid[q]);
            eval_visual(w, k, p, o, true); push(fid, 5, true, {0, 3});
        }
    }
    // marginalize(): positions
    std::map<int, int> pos_of;
    int pos = 0;
    for (int id : drop) { pos_of[id] = pos; pos += lsize_of(id / 4096); }
    const int m = pos;
    for (int id : keep) { pos_of[id] = pos; pos += lsize_of(id / 4096); }
    const int n = pos - m;
    *out_m = m; *out_n = 0; *out_nblocks = 0;
    if (m == 0) return 1;  // valid = false (marginalization_factor.cpp:205-210)
    if (n > cap_n) return -1;
    DMat A(pos, pos);
    std::vector<double> b(pos, 0.0);
    auto construct = [&](DMat& A, std::vector<double>& b, size_t first, size_t stride) {
    for (size_t fi = first; fi < facs.size(); fi += stride) {
        auto& f = facs[fi];
        for (int i = 0; i < f.nb; i++) {
            const int pi = pos_of.at(f.id[i]), li = lsize_of(f.id[i] / 4096);
            for (int j = i; j < f.nb; j++) {
                const int pj = pos_of.at(f.id[j]), lj = lsize_of(f.id[j] / 4096);
                for (int a = 0; a < li; a++) for (int c = 0; c < lj; c++) {
                    double sacc = 0;
                    for (int r = 0; r < f.nres; r++) sacc += f.J[i][(size_t)r * li + a] * f.J[j][(size_t)r * lj + c];
                    A(pi + a, pj + c) += sacc;
                    if (i != j) A(pj + c, pi + a) = A(pi + a, pj + c);
                }
            }
            for (int a = 0; a < li; a++) { double sacc = 0; for (int r = 0; r < f.nres; r++) sacc += f.J[i][(size_t)r * li + a] * f.r[r]; b[pi + a] += sacc; }
        }
    }
    };
    if (gfo_get_threads() > 1) {   // MARG:232-262: NUM_THREADS = 4 pthreads, factor k to thread k % 4, A += A_thread in thread order
        const int NT = 4;
        std::vector<DMat> At; std::vector<std::vector<double>> bt(NT, std::vector<double>(pos, 0.0));
        for (int t = 0; t < NT; t++) At.emplace_back(pos, pos);
        std::vector<std::thread> th;
        for (int t = 0; t < NT; t++) th.emplace_back([&, t] { construct(At[t], bt[t], (size_t)t, (size_t)NT); });
        for (auto& x : th) x.join();
        for (int t = 0; t < NT; t++) { for (size_t i = 0; i < A.a.size(); i++) A.a[i] += At[t].a[i]; for (int i = 0; i < pos; i++) b[i] += bt[t][i]; }
    } else construct(A, b, 0, 1);
    if (g_marg_debug && pos <= g_marg_debug->cap) {
        MargDebug& dbg = *g_marg_debug; dbg.pos = pos; dbg.m = m; dbg.n = n;
        for (int i = 0; i < pos; i++) { for (int j = 0; j < pos; j++) dbg.A[(size_t)i * pos + j] = A(i, j); dbg.b[i] = b[i]; }
    }
    const double eps = 1e-8;
    DMat Amm(m, m), V(m, m);
    for (int i = 0; i < m; i++) for (int j = 0; j < m; j++) Amm(i, j) = 0.5 * (A(i, j) + A(j, i));
    std::vector<double> ev(m);
    sym_eig(m, Amm.a.data(), ev.data(), V.a.data());
    DMat Ainv(m, m);
    for (int k = 0; k < m; k++) if (ev[k] > eps) { const double iv = 1.0 / ev[k]; for (int i = 0; i < m; i++) { const double vi = V(i, k) * iv; if (vi != 0) for (int j = 0; j < m; j++) Ainv(i, j) += vi * V(j, k); } }
    // A = Arr - Arm Amm_inv Amr ; b = brr - Arm Amm_inv bmm
    DMat T(n, m);
    for (int i = 0; i < n; i++) for (int k = 0; k < m; k++) { const double a = A(m + i, k); if (a != 0) for (int j = 0; j < m; j++) T(i, j) += a * Ainv(k, j); }
    DMat Ar(n, n);
    std::vector<double> br(n);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < n; j++) { double sacc = A(m + i, m + j); for (int k = 0; k < m; k++) sacc -= T(i, k) * A(k, m + j); Ar(i, j) = sacc; }
        double sb = b[m + i];
        for (int k = 0; k < m; k++) sb -= T(i, k) * b[k];
        br[i] = sb;
    }
    if (g_marg_debug && pos <= g_marg_debug->cap) {
        MargDebug& dbg = *g_marg_debug;
        for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) dbg.Ar[(size_t)i * n + j] = Ar(i, j); dbg.br[i] = br[i]; }
    }
    DMat V2(n, n);
    std::vector<double> ev2(n);
    // NOTE: Eigen::SelfAdjointEigenSolver reads only the lower triangle of its argument; Ar is symmetric up to rounding
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) Ar(i, j) = Ar(j, i);
    sym_eig(n, Ar.a.data(), ev2.data(), V2.a.data());
    for (int k = 0; k < n; k++) {
        const double S = ev2[k] > eps ? ev2[k] : 0.0, Sinv = ev2[k] > eps ? 1.0 / ev2[k] : 0.0;
        const double ss = std::sqrt(S), sis = std::sqrt(Sinv);
        double vb = 0;
        for (int j = 0; j < n; j++) { out_J[(size_t)k * n + j] = ss * V2(j, k); vb += V2(j, k) * br[j]; }
        out_r[k] = sis * vb;
    }
    // getParameterBlocks with addr_shift (estimator.cpp:3471-3500 / :3583-3626)
    int nb = 0, x0off = 0;
    for (int id : keep) {
        int kind = id / 4096, i = id % 4096, nid = id;
        if (kind == POSE || kind == SPEEDBIAS || kind == RCV_DDT) {
            if (mode == 0) nid = bid(kind, i - 1);
            else nid = (i == W) ? bid(kind, W - 1) : id;
        } else if (kind == RCV_DT) {
            if (mode == 0) nid = bid(kind, i - 4);
            else nid = (i / 4 == W) ? bid(kind, i - 4) : id;
        }
        out_block_id[nb++] = nid;
        const double* p = s.ptr(id);
        for (int q = 0; q < gsize_of(kind); q++) out_x0[x0off++] = p[q];
    }
    *out_n = n; *out_nblocks = nb;
    return 0;
}

}  // namespace gfo_be

using namespace gfo_be;

extern "C" {
int gfo_ba_solve(gfo_window* w, int max_iters, gfo_summary* s) { return solve(w, max_iters, s); }
int gfo_ba_marginalize(const gfo_window* w, int mode, int cap_n, int* out_n, int* out_nblocks, int* out_block_id, double* out_J, double* out_r, double* out_x0,
                       int* out_m) {
    return marginalize(w, mode, cap_n, out_n, out_nblocks, out_block_id, out_J, out_r, out_x0, out_m);
}
/* test helper: the assembled marginalisation system before the Schur complement (A pos x pos, b), and A_r, b_r as the oracle forms them */
int gfo_ba_marg_system(const gfo_window* w, int mode, int cap, double* A, double* b, double* Ar, double* br, int* pos, int* m, int* n) {
    MargDebug dbg{cap, 0, 0, 0, A, b, Ar, br};
    g_marg_debug = &dbg;
    std::vector<int> bidv(512); std::vector<double> J((size_t)cap * cap), r(cap), x0(2 * (size_t)cap);
    int on = 0, onb = 0, om = 0;
    const int rc = marginalize(w, mode, cap, &on, &onb, bidv.data(), J.data(), r.data(), x0.data(), &om);
    g_marg_debug = nullptr;
    *pos = dbg.pos; *m = dbg.m; *n = dbg.n;
    return rc;
}
int gfo_factor_eval(const gfo_window* w, int kind, int k, double* residuals, double* jacobians, int* nres, int* ncols) {
    State s; s.load(w);
    FactorOut o;
    int nb = 0, fid[8];
    if (kind == 0) { const int f[5] = {bid(POSE, w->vis_i[k]), bid(POSE, w->vis_j[k]), bid(EX_POSE, 0), bid(FEATURE, w->vis_feature[k]), bid(TD, 0)}; nb = 5; memcpy(fid, f, sizeof f); }
    else if (kind == 1) { const int i = w->imu_i[k]; const int f[4] = {bid(POSE, i), bid(SPEEDBIAS, i), bid(POSE, i + 1), bid(SPEEDBIAS, i + 1)}; nb = 4; memcpy(fid, f, sizeof f); }
    else if (kind == 2) { const int i = w->wh_i[k]; const int f[7] = {bid(POSE, i), bid(POSE, i + 1), bid(EX_WHEEL, 0), bid(SX, 0), bid(SY, 0), bid(SW, 0), bid(TD_WHEEL, 0)}; nb = 7; memcpy(fid, f, sizeof f); }
    else if (kind == 3) { const int i = w->gnss_frame[k], l = w->gnss_lower[k]; const int f[8] = {bid(POSE, l), bid(SPEEDBIAS, l), bid(POSE, l + 1), bid(SPEEDBIAS, l + 1), bid(RCV_DT, 4 * i + w->gnss_sys[k]), bid(RCV_DDT, i), bid(YAW, 0), bid(ANC, 0)}; nb = 8; memcpy(fid, f, sizeof f); }
    else if (kind == 4) { const int i = k / 4, q = k % 4; const int f[4] = {bid(RCV_DT, 4 * i + q), bid(RCV_DT, 4 * (i + 1) + q), bid(RCV_DDT, i), bid(RCV_DDT, i + 1)}; nb = 4; memcpy(fid, f, sizeof f); }
    else if (kind == 5) { const int f[2] = {bid(RCV_DDT, k), bid(RCV_DDT, k + 1)}; nb = 2; memcpy(fid, f, sizeof f); }
    else if (kind == 6) { fid[0] = bid(POSE, 0); nb = 1; }
    else return -1;
    const double* p[8];
    for (int q = 0; q < nb; q++) p[q] = s.ptr(fid[q]);
    if (kind == 0) eval_visual(w, k, p, o, true); else if (kind == 1) eval_imu(w, k, p, o, true); else if (kind == 2) eval_wheel(w, k, p, o, true);
    else if (kind == 3) eval_gnss(w, k, p, o, true); else if (kind == 4) eval_dt_ddt(w->gnss_headers[k / 4 + 1] - w->gnss_headers[k / 4], p, o, true);
    else if (kind == 5) eval_ddt_smooth(w->gnss_ddt_weight, p, o, true); else eval_anchor(w->anchor_value, p, o, true);
    int cols = 0;
    for (int q = 0; q < nb; q++) cols += gsize_of(fid[q] / 4096);
    *nres = o.nres; *ncols = cols;
    for (int r = 0; r < o.nres; r++) residuals[r] = o.r[r];
    int c0 = 0;
    for (int q = 0; q < nb; q++) {
        const int gs = gsize_of(fid[q] / 4096);
        for (int r = 0; r < o.nres; r++) for (int c = 0; c < gs; c++) jacobians[(size_t)r * cols + c0 + c] = o.J[q][(size_t)r * gs + c];
        c0 += gs;
    }
    return 0;
}
void gfo_imu_preintegrate(int n, const double* dt, const double* acc, const double* gyr, const double* acc0, const double* gyr0, const double* ba, const double* bg,
                          const double* noise, double* delta_p, double* delta_q, double* delta_v, double* jacobian, double* covariance, double* sum_dt) {
    ImuPre p;
    p.init(v3(acc0[0], acc0[1], acc0[2]), v3(gyr0[0], gyr0[1], gyr0[2]), v3(ba[0], ba[1], ba[2]), v3(bg[0], bg[1], bg[2]), noise[0], noise[1], noise[2], noise[3]);
    for (int i = 0; i < n; i++) p.propagate(dt[i], v3(acc[3 * i], acc[3 * i + 1], acc[3 * i + 2]), v3(gyr[3 * i], gyr[3 * i + 1], gyr[3 * i + 2]));
    for (int i = 0; i < 3; i++) { delta_p[i] = p.delta_p[i]; delta_v[i] = p.delta_v[i]; }
    delta_q[0] = p.delta_q.w; delta_q[1] = p.delta_q.x; delta_q[2] = p.delta_q.y; delta_q[3] = p.delta_q.z;
    memcpy(jacobian, p.jacobian.a, 225 * 8); memcpy(covariance, p.covariance.a, 225 * 8);
    *sum_dt = p.sum_dt;
}
void gfo_wheel_preintegrate(int n, const double* dt, const double* vel, const double* gyr, const double* vel0, const double* gyr0, const double* lin, const double* noise,
                            double* delta_p, double* delta_q, double* jacobian, double* covariance, double* sum_dt) {
    WheelPre p;
    p.init(v3(vel0[0], vel0[1], vel0[2]), v3(gyr0[0], gyr0[1], gyr0[2]), lin[0], lin[1], lin[2], noise[0], noise[1]);
    for (int i = 0; i < n; i++) p.propagate(dt[i], v3(vel[3 * i], vel[3 * i + 1], vel[3 * i + 2]), v3(gyr[3 * i], gyr[3 * i + 1], gyr[3 * i + 2]));
    for (int i = 0; i < 3; i++) delta_p[i] = p.delta_p[i];
    delta_q[0] = p.delta_q.w; delta_q[1] = p.delta_q.x; delta_q[2] = p.delta_q.y; delta_q[3] = p.delta_q.z;
    memcpy(jacobian, p.jacobian.a, 18 * 8); memcpy(covariance, p.covariance.a, 36 * 8);
    *sum_dt = p.sum_dt;
}
void gfo_sym_eig(int n, const double* A, double* d, double* V) { sym_eig(n, A, d, V); }

// Estimator::double2vector pose part, estimator.cpp:2440-2497 (USE_IMU); Utility::R2ypr / ypr2R (degrees), utility.h:78-118
static V3 o_R2ypr(const M3& R) {
    V3 n = v3(R(0, 0), R(1, 0), R(2, 0)), o = v3(R(0, 1), R(1, 1), R(2, 1)), a = v3(R(0, 2), R(1, 2), R(2, 2));
    double y = atan2(n[1], n[0]);
    double p = atan2(-n[2], n[0] * cos(y) + n[1] * sin(y));
    double r = atan2(a[0] * sin(y) - a[1] * cos(y), -o[0] * sin(y) + o[1] * cos(y));
    return v3(y, p, r) / M_PI * 180.0;
}
static M3 o_ypr2R(const V3& ypr) {
    double y = ypr[0] / 180.0 * M_PI, p = ypr[1] / 180.0 * M_PI, r = ypr[2] / 180.0 * M_PI;
    M3 Rz, Ry, Rx;
    Rz(0, 0) = cos(y); Rz(0, 1) = -sin(y); Rz(1, 0) = sin(y); Rz(1, 1) = cos(y); Rz(2, 2) = 1;
    Ry(0, 0) = cos(p); Ry(0, 2) = sin(p); Ry(1, 1) = 1; Ry(2, 0) = -sin(p); Ry(2, 2) = cos(p);
    Rx(0, 0) = 1; Rx(1, 1) = cos(r); Rx(1, 2) = -sin(r); Rx(2, 1) = sin(r); Rx(2, 2) = cos(r);
    return Rz * Ry * Rx;
}
void gfo_double2vector(int W, const double* R0_before, const double* P0_before, const double* para_Pose, const double* para_SpeedBias, double* Rs, double* Ps,
                       double* Vs, double* Bas, double* Bgs) {
    M3 R0; memcpy(R0.a, R0_before, 72);
    V3 origin_R0 = o_R2ypr(R0), origin_P0 = v3(P0_before[0], P0_before[1], P0_before[2]);
    M3 R00 = Quat(para_Pose[6], para_Pose[3], para_Pose[4], para_Pose[5]).toRotationMatrix();
    V3 origin_R00 = o_R2ypr(R00);
    double y_diff = origin_R0[0] - origin_R00[0];
    M3 rot_diff = o_ypr2R(v3(y_diff, 0, 0));
    if (std::abs(std::abs(origin_R0[1]) - 90) < 1.0 || std::abs(std::abs(origin_R00[1]) - 90) < 1.0) rot_diff = R0 * R00.T();
    for (int i = 0; i <= W; i++) {
        const double* pp = para_Pose + 7 * i; const double* sb = para_SpeedBias + 9 * i;
        M3 Ri = rot_diff * Quat(pp[6], pp[3], pp[4], pp[5]).normalized().toRotationMatrix();
        memcpy(Rs + 9 * i, Ri.a, 72);
        V3 P = rot_diff * v3(pp[0] - para_Pose[0], pp[1] - para_Pose[1], pp[2] - para_Pose[2]) + origin_P0;
        V3 Vv = rot_diff * v3(sb[0], sb[1], sb[2]);
        for (int k = 0; k < 3; k++) { Ps[3 * i + k] = P[k]; Vs[3 * i + k] = Vv[k]; Bas[3 * i + k] = sb[3 + k]; Bgs[3 * i + k] = sb[6 + k]; }
    }
}

// Debug/inspection: H = J^T J, g = J^T r (loss-corrected, UNscaled) and cost at the window's state, dense, in the canonical
// column order [free f-blocks: pose0, sb0, pose1, ..., ex, exw, sx, sy, sw, td, tdw | free features by index].
int gfo_ba_linearize(const gfo_window* w, int cap, double* H, double* g, double* cost, int* n_f, int* n_e, int* col_block_id) {
    Problem P; P.w = w; P.build();
    State x; x.load(w);
    std::vector<RowBlock> rows;
    *cost = P.evaluate(x, &rows);
    const int n = P.n_cols, ne = P.n_e, nf = n - ne;
    if (n > cap) return -1;
    auto canon = [&](int c) { return c < ne ? nf + c : c - ne; };
    for (int i = 0; i < n * n; i++) H[i] = 0;
    for (int i = 0; i < n; i++) g[i] = 0;
    for (auto& rb : rows)
        for (int q = 0; q < rb.nb; q++) {
            const int c0 = P.col_of.at(rb.id[q]);
            for (int a = 0; a < rb.lsz[q]; a++) {
                double sg = 0;
                for (int r = 0; r < rb.nres; r++) sg += rb.J[q][(size_t)r * rb.lsz[q] + a] * rb.r[r];
                g[canon(c0 + a)] += sg;
            }
            for (int q2 = 0; q2 < rb.nb; q2++) {
                const int c1 = P.col_of.at(rb.id[q2]);
                for (int a = 0; a < rb.lsz[q]; a++) for (int b = 0; b < rb.lsz[q2]; b++) {
                    double sh = 0;
                    for (int r = 0; r < rb.nres; r++) sh += rb.J[q][(size_t)r * rb.lsz[q] + a] * rb.J[q2][(size_t)r * rb.lsz[q2] + b];
                    H[(size_t)canon(c0 + a) * n + canon(c1 + b)] += sh;
                }
            }
        }
    *n_f = nf; *n_e = ne;
    for (int id : P.ids) { const int c0 = P.col_of.at(id); for (int a = 0; a < lsize_of(id / 4096); a++) col_block_id[canon(c0 + a)] = id; }
    return 0;
}
}
