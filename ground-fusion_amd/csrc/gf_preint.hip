// gf_preint.hip — pre-integration: the host loops (~40 kFLOP per IMU sample) and the batched device kernel for the IMU intervals (8(f)4).
// IntegrationBase::push_back/propagate/midPointIntegration (factor/integration_base.h:39-167) and
// WheelIntegrationBase (factor/wheel_integration_base.h:41-178), restated on the small matrix type below.
#include <algorithm>
#include <cstring>
#include <vector>
#include <hip/hip_runtime.h>
#include "../../include/groundfusion_hip.h"
#include "gf_dmath.hpp"
#include "gf_preint.hpp"

namespace gf { int set_err(int code, const char* fmt, ...); }
using namespace gfd;

namespace {
template <int R, int C> struct DM {  // tiny dense row-major matrix on the stack (the estimator group calls this from hundreds of host threads: no heap traffic)
    double a[R * C];
    DM() { for (int i = 0; i < R * C; i++) a[i] = 0.0; }
    double& operator()(int i, int j) { return a[i * C + j]; }
    double operator()(int i, int j) const { return a[i * C + j]; }
    void set3(int r0, int c0, const M3& m) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) (*this)(r0 + i, c0 + j) = m.m[3 * i + j]; }
};
template <int R, int K, int C> DM<R, C> mul(const DM<R, K>& x, const DM<K, C>& y) { DM<R, C> o; for (int i = 0; i < R; i++) for (int k = 0; k < K; k++) { const double v = x(i, k); if (v != 0.0) for (int j = 0; j < C; j++) o(i, j) += v * y(k, j); } return o; }
template <int R, int C> DM<C, R> tr(const DM<R, C>& x) { DM<C, R> o; for (int i = 0; i < R; i++) for (int j = 0; j < C; j++) o(j, i) = x(i, j); return o; }
template <int R, int C> DM<R, C> add(const DM<R, C>& x, const DM<R, C>& y) { DM<R, C> o; for (int i = 0; i < R * C; i++) o.a[i] = x.a[i] + y.a[i]; return o; }
V3 arr(const double* p) { return v3(p[0], p[1], p[2]); }
}  // namespace

namespace {
// Utility::R2ypr / ypr2R work in DEGREES (utility/utility.h:78-118)
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
}  // namespace

extern "C" {

// Estimator::double2vector, pose part (estimator.cpp:2440-2497, USE_IMU branch): the optimised window is rotated about the
// vertical and shifted so that yaw and position of pose 0 keep their pre-optimisation values (the 4 unobservable DoF).
int gf_ba_double2vector(int W, const double* R0_before, const double* P0_before, const double* para_Pose, const double* para_SpeedBias, double* Rs, double* Ps,
                        double* Vs, double* Bas, double* Bgs) {
    if (W < 0 || !R0_before || !P0_before || !para_Pose || !para_SpeedBias || !Rs || !Ps || !Vs || !Bas || !Bgs) return gf::set_err(GF_ERR_INVALID, "bad argument");
    M3 R0; for (int i = 0; i < 9; i++) R0.m[i] = R0_before[i];
    const V3 origin_R0 = R2ypr(R0), origin_P0 = arr(P0_before);
    const M3 R00m = qmat(Q4{para_Pose[6], para_Pose[3], para_Pose[4], para_Pose[5]});
    const V3 origin_R00 = R2ypr(R00m);
    const double y_diff = origin_R0.x - origin_R00.x;
    M3 rot_diff = ypr2R(v3(y_diff, 0, 0));
    if (fabs(fabs(origin_R0.y) - 90) < 1.0 || fabs(fabs(origin_R00.y) - 90) < 1.0) rot_diff = R0 * transpose(R00m);  // euler singular point (:2462-2471)
    for (int i = 0; i <= W; i++) {
        const double* pp = para_Pose + 7 * i;
        const double* sb = para_SpeedBias + 9 * i;
        const M3 Ri = rot_diff * qmat(qnormalized(Q4{pp[6], pp[3], pp[4], pp[5]}));
        for (int k = 0; k < 9; k++) Rs[9 * i + k] = Ri.m[k];
        const V3 P = rot_diff * v3(pp[0] - para_Pose[0], pp[1] - para_Pose[1], pp[2] - para_Pose[2]) + origin_P0;
        Ps[3 * i] = P.x; Ps[3 * i + 1] = P.y; Ps[3 * i + 2] = P.z;
        const V3 Vv = rot_diff * v3(sb[0], sb[1], sb[2]);
        Vs[3 * i] = Vv.x; Vs[3 * i + 1] = Vv.y; Vs[3 * i + 2] = Vv.z;
        for (int k = 0; k < 3; k++) { Bas[3 * i + k] = sb[3 + k]; Bgs[3 * i + k] = sb[6 + k]; }
    }
    return GF_OK;
}

}  // extern "C"

namespace gf {
// IntegrationBase::push_back for samples [s0, s1) on top of the running state `st` (the estimator appends samples to an interval frame after
// frame; integrating only the new ones is the same sequence of operations as starting over, bit for bit)
void imu_preint_reset(ImuPreState& st, const double* acc0, const double* gyr0) {
    memset(&st, 0, sizeof st);
    for (int i = 0; i < 3; i++) { st.acc_0[i] = acc0[i]; st.gyr_0[i] = gyr0[i]; }
    st.dq[0] = 1;
    for (int i = 0; i < 15; i++) st.J[i * 15 + i] = 1;
}
// the 3-vector / quaternion part of the recursion alone: delta_p, delta_q, delta_v and sum_dt do not depend on the Jacobian and the covariance, which are 99 % of the
// arithmetic (two 15 x 15 x 15 and two 15 x 18 products per sample).  Same operations in the same order as imu_preint_range: the same bits.  For callers that read nothing
// else -- Estimator::checkimu (estimator.cpp:2173-2216) looks at delta_v / sum_dt of every frame of all_image_frame on every image.
void imu_preint_state_range(ImuPreState& st, const double* ba, const double* bg, const double* dt, const double* acc, const double* gyr, int s0, int s1) {
    V3 acc_0 = arr(st.acc_0), gyr_0 = arr(st.gyr_0), lba = arr(ba), lbg = arr(bg), dp = arr(st.dp), dv = arr(st.dv);
    Q4 dq{st.dq[0], st.dq[1], st.dq[2], st.dq[3]};
    double sdt = st.sum_dt;
    for (int s = s0; s < s1; s++) {
        const double t = dt[s];
        const V3 acc_1 = arr(acc + 3 * s), gyr_1 = arr(gyr + 3 * s);
        const V3 un_acc_0 = qrot(dq, acc_0 - lba);
        const V3 un_gyr = (gyr_0 + gyr_1) * 0.5 - lbg;
        const Q4 rq = qmul(dq, Q4{1, un_gyr.x * t / 2, un_gyr.y * t / 2, un_gyr.z * t / 2});
        const V3 un_acc_1 = qrot(rq, acc_1 - lba);
        const V3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
        const V3 rp = dp + dv * t + un_acc * (0.5 * t * t), rv = dv + un_acc * t;
        dp = rp; dq = qnormalized(rq); dv = rv;
        sdt += t; acc_0 = acc_1; gyr_0 = gyr_1;
    }
    st.dp[0] = dp.x; st.dp[1] = dp.y; st.dp[2] = dp.z; st.dv[0] = dv.x; st.dv[1] = dv.y; st.dv[2] = dv.z;
    st.dq[0] = dq.w; st.dq[1] = dq.x; st.dq[2] = dq.y; st.dq[3] = dq.z;
    st.acc_0[0] = acc_0.x; st.acc_0[1] = acc_0.y; st.acc_0[2] = acc_0.z; st.gyr_0[0] = gyr_0.x; st.gyr_0[1] = gyr_0.y; st.gyr_0[2] = gyr_0.z;
    st.sum_dt = sdt; st.n_done = s1;
}
void imu_preint_range(ImuPreState& st, const double* ba, const double* bg, const double* noise, const double* dt, const double* acc, const double* gyr, int s0, int s1) {
    V3 acc_0 = arr(st.acc_0), gyr_0 = arr(st.gyr_0), lba = arr(ba), lbg = arr(bg), dp = arr(st.dp), dv = arr(st.dv);
    Q4 dq{st.dq[0], st.dq[1], st.dq[2], st.dq[3]};
    DM<15, 15> J, P; DM<18, 18> N;
    memcpy(J.a, st.J, sizeof J.a); memcpy(P.a, st.P, sizeof P.a);
    for (int i = 0; i < 3; i++) {
        N(i, i) = noise[0] * noise[0]; N(3 + i, 3 + i) = noise[1] * noise[1]; N(6 + i, 6 + i) = noise[0] * noise[0]; N(9 + i, 9 + i) = noise[1] * noise[1];
        N(12 + i, 12 + i) = noise[2] * noise[2]; N(15 + i, 15 + i) = noise[3] * noise[3];
    }
    double sdt = st.sum_dt;
    for (int s = s0; s < s1; s++) {
        const double t = dt[s];
        const V3 acc_1 = arr(acc + 3 * s), gyr_1 = arr(gyr + 3 * s);
        const V3 un_acc_0 = qrot(dq, acc_0 - lba);
        const V3 un_gyr = (gyr_0 + gyr_1) * 0.5 - lbg;
        const Q4 rq = qmul(dq, Q4{1, un_gyr.x * t / 2, un_gyr.y * t / 2, un_gyr.z * t / 2});
        const V3 un_acc_1 = qrot(rq, acc_1 - lba);
        const V3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
        const V3 rp = dp + dv * t + un_acc * (0.5 * t * t), rv = dv + un_acc * t;
        const M3 Rwx = skew(un_gyr), Ra0 = skew(acc_0 - lba), Ra1 = skew(acc_1 - lba), Rd = qmat(dq), Rr = qmat(rq), I = m3_identity();
        DM<15, 15> F; DM<15, 18> V;
        F.set3(0, 0, I);
        F.set3(0, 3, Rd * Ra0 * (-0.25 * t * t) + Rr * Ra1 * (I - Rwx * t) * (-0.25 * t * t));
        F.set3(0, 6, I * t);
        F.set3(0, 9, (Rd + Rr) * (-0.25 * t * t));
        F.set3(0, 12, Rr * Ra1 * (-0.25 * t * t * -t));
        F.set3(3, 3, I - Rwx * t);
        F.set3(3, 12, I * (-1.0 * t));
        F.set3(6, 3, Rd * Ra0 * (-0.5 * t) + Rr * Ra1 * (I - Rwx * t) * (-0.5 * t));
        F.set3(6, 6, I);
        F.set3(6, 9, (Rd + Rr) * (-0.5 * t));
        F.set3(6, 12, Rr * Ra1 * (-0.5 * t * -t));
        F.set3(9, 9, I); F.set3(12, 12, I);
        V.set3(0, 0, Rd * (0.25 * t * t));
        const M3 v03 = (-Rr) * Ra1 * (0.25 * t * t * 0.5 * t);
        V.set3(0, 3, v03); V.set3(0, 6, Rr * (0.25 * t * t)); V.set3(0, 9, v03);
        V.set3(3, 3, I * (0.5 * t)); V.set3(3, 9, I * (0.5 * t));
        V.set3(6, 0, Rd * (0.5 * t));
        const M3 v63 = (-Rr) * Ra1 * (0.5 * t * 0.5 * t);
        V.set3(6, 3, v63); V.set3(6, 6, Rr * (0.5 * t)); V.set3(6, 9, v63);
        V.set3(9, 12, I * t); V.set3(12, 15, I * t);
        J = mul(F, J);
        P = add(mul(mul(F, P), tr(F)), mul(mul(V, N), tr(V)));
        dp = rp; dq = qnormalized(rq); dv = rv;
        sdt += t; acc_0 = acc_1; gyr_0 = gyr_1;
    }
    st.dp[0] = dp.x; st.dp[1] = dp.y; st.dp[2] = dp.z; st.dv[0] = dv.x; st.dv[1] = dv.y; st.dv[2] = dv.z;
    st.dq[0] = dq.w; st.dq[1] = dq.x; st.dq[2] = dq.y; st.dq[3] = dq.z;
    st.acc_0[0] = acc_0.x; st.acc_0[1] = acc_0.y; st.acc_0[2] = acc_0.z; st.gyr_0[0] = gyr_0.x; st.gyr_0[1] = gyr_0.y; st.gyr_0[2] = gyr_0.z;
    memcpy(st.J, J.a, sizeof J.a); memcpy(st.P, P.a, sizeof P.a);
    st.sum_dt = sdt; st.n_done = s1;
}
}  // namespace gf

// ---------------------------------------------------------------- many intervals at once on the device (SURVEY.md 8(f)4, row B2 on the GPU)
// One wavefront per interval; the lanes own the entries of the 15 x 15 results and evaluate each as the host code does -- the same products in the same
// order, no contraction, structural zeros skipped the way mul() skips them -- so that a device interval is bit-identical to gf::imu_preint_range of the same
// samples (tests/test_preint_gpu.py).  The 3-vector / quaternion part of a sample and the 3 x 3 blocks of F and V are evaluated by every lane (uniform).
namespace gf {
struct PreintJobDev { int s0, s1; double acc0[3], gyr0[3], ba[3], bg[3]; };
constexpr int PREINT_OUT = 16 + 225 + 225;   // dp 3, dq 4, dv 3, sum_dt, pad 5 | jacobian | covariance

__global__ __launch_bounds__(64) void imu_preint_batch_kernel(int n, const PreintJobDev* __restrict__ jobs, const double* __restrict__ dt, const double* __restrict__ acc,
                                                              const double* __restrict__ gyr, double n0, double n1, double n2, double n3, double* __restrict__ out) {
    __shared__ double sJ[225], sP[225], sF[225], sV[270], sT[225], sVN[270];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= n) return;
    const PreintJobDev jb = jobs[b];
    for (int e = lane; e < 225; e += 64) { sJ[e] = (e % 16 == 0) ? 1.0 : 0.0; sP[e] = 0.0; }
    V3 acc_0 = v3(jb.acc0[0], jb.acc0[1], jb.acc0[2]), gyr_0 = v3(jb.gyr0[0], jb.gyr0[1], jb.gyr0[2]);
    const V3 lba = v3(jb.ba[0], jb.ba[1], jb.ba[2]), lbg = v3(jb.bg[0], jb.bg[1], jb.bg[2]);
    V3 dp = v3(0, 0, 0), dv = v3(0, 0, 0);
    Q4 dq{1, 0, 0, 0};
    double sdt = 0;
    const double Nd[6] = {n0 * n0, n1 * n1, n0 * n0, n1 * n1, n2 * n2, n3 * n3};   // diagonal of the 18 x 18 noise matrix, 3 entries each
    __syncthreads();
    for (int s = jb.s0; s < jb.s1; s++) {
        const double t = dt[s];
        const V3 acc_1 = v3(acc[3 * s], acc[3 * s + 1], acc[3 * s + 2]), gyr_1 = v3(gyr[3 * s], gyr[3 * s + 1], gyr[3 * s + 2]);
        const V3 un_acc_0 = qrot(dq, acc_0 - lba);
        const V3 un_gyr = (gyr_0 + gyr_1) * 0.5 - lbg;
        const Q4 rq = qmul(dq, Q4{1, un_gyr.x * t / 2, un_gyr.y * t / 2, un_gyr.z * t / 2});
        const V3 un_acc_1 = qrot(rq, acc_1 - lba);
        const V3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
        const V3 rp = dp + dv * t + un_acc * (0.5 * t * t), rv = dv + un_acc * t;
        for (int e = lane; e < 225; e += 64) sF[e] = 0.0;
        for (int e = lane; e < 270; e += 64) sV[e] = 0.0;
        __syncthreads();
        {   // the 3 x 3 blocks of F and V: evaluated by every lane (uniform, the host's expressions), entry p of each block written by lane p
            const M3 Rwx = skew(un_gyr), Ra0 = skew(acc_0 - lba), Ra1 = skew(acc_1 - lba), Rd = qmat(dq), Rr = qmat(rq), I = m3_identity();
            const M3 A = Rd * Ra0, B = Rr * Ra1, Cm = B * (I - Rwx * t), nB = (-Rr) * Ra1, RdRr = Rd + Rr;
            const M3 f03 = A * (-0.25 * t * t) + Cm * (-0.25 * t * t), f06 = I * t, f09 = RdRr * (-0.25 * t * t), f012 = B * (-0.25 * t * t * -t), f33 = I - Rwx * t,
                     f312 = I * (-1.0 * t), f63 = A * (-0.5 * t) + Cm * (-0.5 * t), f69 = RdRr * (-0.5 * t), f612 = B * (-0.5 * t * -t);
            const M3 v00 = Rd * (0.25 * t * t), v03 = nB * (0.25 * t * t * 0.5 * t), v06 = Rr * (0.25 * t * t), v33 = I * (0.5 * t), v60 = Rd * (0.5 * t),
                     v63 = nB * (0.5 * t * 0.5 * t), v66 = Rr * (0.5 * t), v912 = I * t;
            if (lane < 9) {
                const int i = lane / 3, j = lane % 3;
                auto pick = [&](const M3& m) { double r = m.m[0]; for (int q = 1; q < 9; q++) r = lane == q ? m.m[q] : r; return r; };
                auto putF = [&](int r0, int c0, const M3& m) { sF[(r0 + i) * 15 + c0 + j] = pick(m); };
                auto putV = [&](int r0, int c0, const M3& m) { sV[(r0 + i) * 18 + c0 + j] = pick(m); };
                putF(0, 0, I); putF(0, 3, f03); putF(0, 6, f06); putF(0, 9, f09); putF(0, 12, f012);
                putF(3, 3, f33); putF(3, 12, f312);
                putF(6, 3, f63); putF(6, 6, I); putF(6, 9, f69); putF(6, 12, f612);
                putF(9, 9, I); putF(12, 12, I);
                putV(0, 0, v00); putV(0, 3, v03); putV(0, 6, v06); putV(0, 9, v03);
                putV(3, 3, v33); putV(3, 9, v33);
                putV(6, 0, v60); putV(6, 3, v63); putV(6, 6, v66); putV(6, 9, v63);
                putV(9, 12, v912); putV(12, 15, v912);
            }
        }
        __syncthreads();
        // J' = F J and T = F P (entries owned by lanes; sums in k order, zero factors of the left matrix skipped as mul() does)
        double jn[4], pn[4];
        for (int q = 0; q < 4; q++) {
            const int e = lane + 64 * q;
            if (e < 225) {
                const int i = e / 15, j = e % 15;
                double a = 0.0, c = 0.0;
#pragma unroll
                for (int k = 0; k < 15; k++) { const double v = sF[i * 15 + k], a1 = a + v * sJ[k * 15 + j], c1 = c + v * sP[k * 15 + j]; a = v != 0.0 ? a1 : a; c = v != 0.0 ? c1 : c; }
                jn[q] = a; sT[e] = c;
            }
        }
        for (int e = lane; e < 270; e += 64) { const double v = sV[e]; sVN[e] = v != 0.0 ? v * Nd[(e % 18) / 3] : 0.0; }   // V N, N diagonal
        __syncthreads();
        for (int q = 0; q < 4; q++) {
            const int e = lane + 64 * q;
            if (e < 225) {
                const int i = e / 15, j = e % 15;
                double a = 0.0, c = 0.0;
#pragma unroll
                for (int k = 0; k < 15; k++) { const double v = sT[i * 15 + k], a1 = a + v * sF[j * 15 + k]; a = v != 0.0 ? a1 : a; }      // (F P) F^T
#pragma unroll
                for (int k = 0; k < 18; k++) { const double v = sVN[i * 18 + k], c1 = c + v * sV[j * 18 + k]; c = v != 0.0 ? c1 : c; }    // (V N) V^T
                pn[q] = a + c;
            }
        }
        __syncthreads();
        for (int q = 0; q < 4; q++) { const int e = lane + 64 * q; if (e < 225) { sJ[e] = jn[q]; sP[e] = pn[q]; } }
        dp = rp; dq = qnormalized(rq); dv = rv;
        sdt += t; acc_0 = acc_1; gyr_0 = gyr_1;
        __syncthreads();
    }
    double* o = out + (size_t)b * PREINT_OUT;
    if (lane == 0) {
        o[0] = dp.x; o[1] = dp.y; o[2] = dp.z; o[3] = dq.w; o[4] = dq.x; o[5] = dq.y; o[6] = dq.z; o[7] = dv.x; o[8] = dv.y; o[9] = dv.z; o[10] = sdt;
        o[11] = acc_0.x; o[12] = acc_0.y; o[13] = acc_0.z;
    }
    for (int e = lane; e < 225; e += 64) { o[16 + e] = sJ[e]; o[16 + 225 + e] = sP[e]; }
}

struct PreintBatch {
    hipStream_t stream = nullptr;
    PreintJobDev* d_jobs = nullptr; double *d_dt = nullptr, *d_acc = nullptr, *d_gyr = nullptr, *d_out = nullptr;
    PreintJobDev* h_jobs = nullptr; double *h_dt = nullptr, *h_acc = nullptr, *h_gyr = nullptr, *h_out = nullptr;   // pinned
    int cap_jobs = 0, cap_samples = 0;
    double t_kernel_ms = 0; long long launches = 0, intervals = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    ~PreintBatch() {
        (void)hipFree(d_jobs); (void)hipFree(d_dt); (void)hipFree(d_acc); (void)hipFree(d_gyr); (void)hipFree(d_out);
        (void)hipHostFree(h_jobs); (void)hipHostFree(h_dt); (void)hipHostFree(h_acc); (void)hipHostFree(h_gyr); (void)hipHostFree(h_out);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
    }
};
#define GF_PRE_HIP(x) do { const hipError_t e_ = (x); if (e_ != hipSuccess) return gf::set_err(e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice ? GF_ERR_NO_DEVICE : GF_ERR_HIP, "%s: %s", #x, hipGetErrorString(e_)); } while (0)
int preint_batch_create(PreintBatch** out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return gf::set_err(GF_ERR_NO_DEVICE, "no HIP device: batched pre-integration has no CPU fallback (use gf_imu_preintegrate on the host)");
    PreintBatch* b = new PreintBatch();
    if (hipStreamCreate(&b->stream) != hipSuccess || hipEventCreate(&b->ev0) != hipSuccess || hipEventCreate(&b->ev1) != hipSuccess) { delete b; return gf::set_err(GF_ERR_HIP, "hipStreamCreate / hipEventCreate failed"); }
    *out = b;
    return GF_OK;
}
void preint_batch_destroy(PreintBatch* b) { delete b; }
static int preint_reserve(PreintBatch* b, int jobs, int samples) {
    if (jobs > b->cap_jobs) {
        const int c = std::max(jobs, 2 * b->cap_jobs);
        (void)hipFree(b->d_jobs); (void)hipFree(b->d_out); (void)hipHostFree(b->h_jobs); (void)hipHostFree(b->h_out);
        b->d_jobs = nullptr; b->d_out = nullptr; b->h_jobs = nullptr; b->h_out = nullptr; b->cap_jobs = 0;
        GF_PRE_HIP(hipMalloc(&b->d_jobs, sizeof(PreintJobDev) * c)); GF_PRE_HIP(hipMalloc(&b->d_out, sizeof(double) * PREINT_OUT * c));
        GF_PRE_HIP(hipHostMalloc(&b->h_jobs, sizeof(PreintJobDev) * c)); GF_PRE_HIP(hipHostMalloc(&b->h_out, sizeof(double) * PREINT_OUT * c));
        b->cap_jobs = c;
    }
    if (samples > b->cap_samples) {
        const int c = std::max(samples, 2 * b->cap_samples);
        (void)hipFree(b->d_dt); (void)hipFree(b->d_acc); (void)hipFree(b->d_gyr); (void)hipHostFree(b->h_dt); (void)hipHostFree(b->h_acc); (void)hipHostFree(b->h_gyr);
        b->d_dt = b->d_acc = b->d_gyr = nullptr; b->h_dt = b->h_acc = b->h_gyr = nullptr; b->cap_samples = 0;
        GF_PRE_HIP(hipMalloc(&b->d_dt, sizeof(double) * c)); GF_PRE_HIP(hipMalloc(&b->d_acc, sizeof(double) * 3 * c)); GF_PRE_HIP(hipMalloc(&b->d_gyr, sizeof(double) * 3 * c));
        GF_PRE_HIP(hipHostMalloc(&b->h_dt, sizeof(double) * c)); GF_PRE_HIP(hipHostMalloc(&b->h_acc, sizeof(double) * 3 * c)); GF_PRE_HIP(hipHostMalloc(&b->h_gyr, sizeof(double) * 3 * c));
        b->cap_samples = c;
    }
    return GF_OK;
}
// every job from scratch over its samples [0, n): fills st as imu_preint_reset + imu_preint_range(st, ..., 0, n) would
int preint_batch_run(PreintBatch* b, const std::vector<PreintJob>& jobs, const double* noise) {
    const int n = (int)jobs.size();
    if (n == 0) return GF_OK;
    int total = 0;
    for (auto& j : jobs) total += j.n;
    if (int rc = preint_reserve(b, n, std::max(total, 1))) return rc;
    int off = 0;
    for (int i = 0; i < n; i++) {
        const PreintJob& j = jobs[i];
        PreintJobDev& d = b->h_jobs[i];
        d.s0 = off; d.s1 = off + j.n;
        for (int k = 0; k < 3; k++) { d.acc0[k] = j.acc0[k]; d.gyr0[k] = j.gyr0[k]; d.ba[k] = j.ba[k]; d.bg[k] = j.bg[k]; }
        if (j.n > 0) { memcpy(b->h_dt + off, j.dt, sizeof(double) * j.n); memcpy(b->h_acc + 3 * off, j.acc, sizeof(double) * 3 * j.n); memcpy(b->h_gyr + 3 * off, j.gyr, sizeof(double) * 3 * j.n); }
        off += j.n;
    }
    GF_PRE_HIP(hipMemcpyAsync(b->d_jobs, b->h_jobs, sizeof(PreintJobDev) * n, hipMemcpyHostToDevice, b->stream));
    if (total > 0) {
        GF_PRE_HIP(hipMemcpyAsync(b->d_dt, b->h_dt, sizeof(double) * total, hipMemcpyHostToDevice, b->stream));
        GF_PRE_HIP(hipMemcpyAsync(b->d_acc, b->h_acc, sizeof(double) * 3 * total, hipMemcpyHostToDevice, b->stream));
        GF_PRE_HIP(hipMemcpyAsync(b->d_gyr, b->h_gyr, sizeof(double) * 3 * total, hipMemcpyHostToDevice, b->stream));
    }
    GF_PRE_HIP(hipEventRecord(b->ev0, b->stream));
    hipLaunchKernelGGL(imu_preint_batch_kernel, dim3(n), dim3(64), 0, b->stream, n, b->d_jobs, b->d_dt, b->d_acc, b->d_gyr, noise[0], noise[1], noise[2], noise[3], b->d_out);
    GF_PRE_HIP(hipGetLastError());
    GF_PRE_HIP(hipEventRecord(b->ev1, b->stream));
    GF_PRE_HIP(hipMemcpyAsync(b->h_out, b->d_out, sizeof(double) * PREINT_OUT * n, hipMemcpyDeviceToHost, b->stream));
    GF_PRE_HIP(hipStreamSynchronize(b->stream));
    float ms = 0;
    if (hipEventElapsedTime(&ms, b->ev0, b->ev1) == hipSuccess) b->t_kernel_ms += ms;
    b->launches++; b->intervals += n;
    for (int i = 0; i < n; i++) {
        ImuPreState& st = *jobs[i].st;
        const double* o = b->h_out + (size_t)i * PREINT_OUT;
        memcpy(st.dp, o, 24); memcpy(st.dq, o + 3, 32); memcpy(st.dv, o + 7, 24);
        st.sum_dt = o[10];
        const int last = jobs[i].n - 1;
        for (int k = 0; k < 3; k++) { st.acc_0[k] = last >= 0 ? jobs[i].acc[3 * last + k] : jobs[i].acc0[k]; st.gyr_0[k] = last >= 0 ? jobs[i].gyr[3 * last + k] : jobs[i].gyr0[k]; }
        memcpy(st.J, o + 16, 225 * 8); memcpy(st.P, o + 16 + 225, 225 * 8);
        st.n_done = jobs[i].n;
    }
    return GF_OK;
}
}  // namespace gf

extern "C" {
struct gf_preint { gf::PreintBatch* b; };
int gf_preint_create(gf_preint** out) {
    if (!out) return gf::set_err(GF_ERR_INVALID, "null argument");
    gf::PreintBatch* b = nullptr;
    if (int rc = gf::preint_batch_create(&b)) return rc;
    *out = new gf_preint{b};
    return GF_OK;
}
int gf_preint_destroy(gf_preint* h) { if (h) { gf::preint_batch_destroy(h->b); delete h; } return GF_OK; }
int gf_imu_preintegrate_batch(gf_preint* h, int n, const int* first, const double* dt, const double* acc, const double* gyr, const double* acc0, const double* gyr0,
                              const double* ba, const double* bg, const double* noise, double* delta_p, double* delta_q, double* delta_v, double* jacobian, double* covariance,
                              double* sum_dt) {
    if (!h || n < 0 || (n > 0 && (!first || !acc0 || !gyr0 || !ba || !bg || !noise || !delta_p || !delta_q || !delta_v || !jacobian || !covariance || !sum_dt)))
        return gf::set_err(GF_ERR_INVALID, "bad argument");
    std::vector<gf::ImuPreState> st(n);
    std::vector<gf::PreintJob> jobs(n);
    for (int i = 0; i < n; i++) {
        if (first[i + 1] < first[i]) return gf::set_err(GF_ERR_INVALID, "interval %d: first[] must not decrease", i);
        gf::PreintJob& j = jobs[i];
        j.st = &st[i]; j.n = first[i + 1] - first[i];
        j.dt = dt + first[i]; j.acc = acc + 3 * (size_t)first[i]; j.gyr = gyr + 3 * (size_t)first[i];
        j.ba = ba + 3 * i; j.bg = bg + 3 * i;
        for (int k = 0; k < 3; k++) { j.acc0[k] = acc0[3 * i + k]; j.gyr0[k] = gyr0[3 * i + k]; }
    }
    if (int rc = gf::preint_batch_run(h->b, jobs, noise)) return rc;
    for (int i = 0; i < n; i++) {
        memcpy(delta_p + 3 * i, st[i].dp, 24); memcpy(delta_q + 4 * i, st[i].dq, 32); memcpy(delta_v + 3 * i, st[i].dv, 24);
        memcpy(jacobian + 225 * (size_t)i, st[i].J, 225 * 8); memcpy(covariance + 225 * (size_t)i, st[i].P, 225 * 8);
        sum_dt[i] = st[i].sum_dt;
    }
    return GF_OK;
}
int gf_preint_stats(gf_preint* h, long long* launches, long long* intervals, double* kernel_ms) {
    if (!h) return gf::set_err(GF_ERR_INVALID, "null handle");
    if (launches) *launches = h->b->launches;
    if (intervals) *intervals = h->b->intervals;
    if (kernel_ms) *kernel_ms = h->b->t_kernel_ms;
    return GF_OK;
}
}  // extern "C"

extern "C" {
int gf_imu_preintegrate(int n, const double* dt, const double* acc, const double* gyr, const double* acc0, const double* gyr0, const double* ba, const double* bg,
                        const double* noise, double* delta_p, double* delta_q, double* delta_v, double* jacobian, double* covariance, double* sum_dt) {
    if (n < 0 || !acc0 || !gyr0 || !ba || !bg || !noise) return gf::set_err(GF_ERR_INVALID, "bad argument");
    gf::ImuPreState st;
    gf::imu_preint_reset(st, acc0, gyr0);
    gf::imu_preint_range(st, ba, bg, noise, dt, acc, gyr, 0, n);
    memcpy(delta_p, st.dp, 24); memcpy(delta_q, st.dq, 32); memcpy(delta_v, st.dv, 24);
    memcpy(jacobian, st.J, 225 * 8); memcpy(covariance, st.P, 225 * 8);
    *sum_dt = st.sum_dt;
    return GF_OK;
}

int gf_imu_preintegrate_state(int n, const double* dt, const double* acc, const double* gyr, const double* acc0, const double* gyr0, const double* ba, const double* bg,
                              double* delta_p, double* delta_q, double* delta_v, double* sum_dt) {
    if (n < 0 || !acc0 || !gyr0 || !ba || !bg || !delta_p || !delta_q || !delta_v || !sum_dt) return gf::set_err(GF_ERR_INVALID, "bad argument");
    gf::ImuPreState st;
    gf::imu_preint_reset(st, acc0, gyr0);
    gf::imu_preint_state_range(st, ba, bg, dt, acc, gyr, 0, n);
    memcpy(delta_p, st.dp, 24); memcpy(delta_q, st.dq, 32); memcpy(delta_v, st.dv, 24);
    *sum_dt = st.sum_dt;
    return GF_OK;
}

int gf_wheel_preintegrate(int n, const double* dt, const double* vel, const double* gyr, const double* vel0, const double* gyr0, const double* lin, const double* noise,
                          double* delta_p, double* delta_q, double* jacobian, double* covariance, double* sum_dt) {
    if (n < 0 || !vel0 || !gyr0 || !lin || !noise) return gf::set_err(GF_ERR_INVALID, "bad argument");
    V3 vel_0 = arr(vel0), gyr_0 = arr(gyr0), dp = v3(0, 0, 0);
    Q4 dq{1, 0, 0, 0};
    const double sx = lin[0], sy = lin[1], sw = lin[2];
    DM<6, 3> Jm; DM<6, 6> P; DM<12, 12> N;
    for (int i = 0; i < 3; i++) { N(i, i) = noise[0] * noise[0]; N(3 + i, 3 + i) = noise[1] * noise[1]; N(6 + i, 6 + i) = noise[0] * noise[0]; N(9 + i, 9 + i) = noise[1] * noise[1]; }
    double sdt = 0;
    const M3 sv = m3_diag(sx, sy, 1), I = m3_identity(), I1 = m3_diag(1, 0, 0), I2 = m3_diag(0, 1, 0);
    for (int s = 0; s < n; s++) {
        const double t = dt[s];
        const V3 vel_1 = arr(vel + 3 * s), gyr_1 = arr(gyr + 3 * s);
        const V3 un_vel_0 = qrot(dq, sv * vel_0);
        const V3 un_gyr = (gyr_0 + gyr_1) * (0.5 * sw);
        const Q4 ddq{1, un_gyr.x * t / 2, un_gyr.y * t / 2, un_gyr.z * t / 2};
        const Q4 rq = qmul(dq, ddq);
        const V3 un_vel_1 = qrot(rq, sv * vel_1);
        const V3 rp = dp + (un_vel_0 + un_vel_1) * 0.5 * t;
        const M3 Rv0 = skew(sv * vel_0), Rv1 = skew(sv * vel_1), Rd = qmat(dq), Rr = qmat(rq), Rdd = qmat(ddq);
        DM<6, 6> F; DM<6, 12> V;
        F.set3(0, 0, I);
        F.set3(0, 3, (Rd * Rv0 + Rr * Rv1 * transpose(Rdd)) * (-0.5 * t));
        F.set3(3, 3, transpose(Rdd));
        const M3 Jr = rightJacobianSO3(un_gyr * t);
        V.set3(0, 0, Rd * sv * (0.5 * t));
        V.set3(0, 3, Rr * Rv1 * Jr * (-0.25 * t * t));
        V.set3(0, 6, Rr * sv * (0.5 * t));
        V.set3(0, 9, Rr * Rv1 * Jr * (-0.25 * t * t));
        V.set3(3, 3, Jr * (0.5 * sw * t));
        V.set3(3, 9, Jr * (0.5 * sw * t));
        auto col = [&](int r0, int c) { return v3(Jm(r0, c), Jm(r0 + 1, c), Jm(r0 + 2, c)); };
        auto setc = [&](int r0, int c, V3 v) { Jm(r0, c) = v.x; Jm(r0 + 1, c) = v.y; Jm(r0 + 2, c) = v.z; };
        const V3 j00 = col(0, 0) + (Rd * (I1 * vel_0) + Rr * (I1 * vel_1)) * (0.5 * t);
        const V3 j01 = col(0, 1) + (Rd * (I2 * vel_0) + Rr * (I2 * vel_1)) * (0.5 * t);
        const V3 last = col(3, 2);
        const V3 j32 = last + Jr * ((gyr_0 + gyr_1) * 0.5) * t;
        setc(0, 0, j00); setc(0, 1, j01); setc(3, 2, j32);
        const V3 j02 = col(0, 2) + (Rd * (skew(last) * (sv * vel_0)) + Rr * (skew(j32) * (sv * vel_1))) * (0.5 * t);
        setc(0, 2, j02);
        P = add(mul(mul(F, P), tr(F)), mul(mul(V, N), tr(V)));
        dp = rp; dq = qnormalized(rq);
        sdt += t; vel_0 = vel_1; gyr_0 = gyr_1;
    }
    delta_p[0] = dp.x; delta_p[1] = dp.y; delta_p[2] = dp.z;
    delta_q[0] = dq.w; delta_q[1] = dq.x; delta_q[2] = dq.y; delta_q[3] = dq.z;
    memcpy(jacobian, Jm.a, 18 * 8); memcpy(covariance, P.a, 36 * 8);
    *sum_dt = sdt;
    return GF_OK;
}
}
