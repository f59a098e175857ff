"""Golden normal equations of small windows, evaluated with 60 digits (mpmath) from the REFERENCE's formulas -- not from the oracle's or the library's code.

Why: the oracle (oracle/backend_oracle.cpp) and the HIP kernels were written by the same hand from the same reading of the reference; the reference cannot be
built here (no Eigen / Ceres / ROS) and ships no test vectors, so nothing tied either of them to numbers they did not produce themselves ("parity unpinned").
This script is a third, independent evaluation route: every formula below is transcribed from the cited reference lines into exact-enough arithmetic (60
significant digits, own quaternion / matrix algebra, no shared helper with the oracle or the product), and its output is committed as tests/golden/ref_*.json.gz.
tests/test_golden.py then holds the oracle (CPU) and tests/test_backend_gpu.py the HIP path (through gf_ba_linearize) to these numbers at 1e-11.

What is evaluated (paths under /root/reference/vins_estimator/src):
  factor/projectionTwoFrameOneCamFactor.cpp:43-151   residual and the five Jacobian blocks of ProjectionTwoFrameOneCamFactor::Evaluate
  factor/imu_factor.h:28-191 + factor/integration_base.h:169-195   IMUFactor::Evaluate on top of IntegrationBase::evaluate, sqrt_info = LLT(cov^-1).matrixL()^T
  factor/wheel_factor.h:28-247 + factor/wheel_integration_base.h:180-219 + utility/sophus_utils.hpp:155-236   WheelFactor::Evaluate (all seven blocks)
  factor/marginalization_factor.cpp:12-78, :119-308 + estimator/estimator.cpp:3334-3560   the marginalisation itself: MARGIN_OLD / MARGIN_SECOND_NEW factor sets, loss
                                                     scaling, A / b, eigen pseudo-inverse of A_mm, eigen square root of the kept system (eps 1e-8)  -> ref_marg_*.json.gz
  factor/marginalization_factor.cpp:344-392          MarginalizationFactor::Evaluate: r = r0 + J0 dx, dx of pose blocks 2 vec(q0^-1 q) with the sign of its w
  utility/utility.h:23-76                             deltaQ, skewSymmetric, Qleft, Qright (positify returns its argument: line 49-57)
  estimator/estimator.cpp:3269-3297                   which factors get the loss: ceres::HuberLoss(1.0) on the visual factors only
  Ceres 1.14 (not vendored by the reference; published algorithm): loss_function.cc HuberLoss::Evaluate, corrector.cc Corrector (robustified r and J)
The local parameterisation is PoseLocalParameterization (pose_local_parameterization.cpp:12-45): ComputeJacobian is [I6; 0], so the columns of a pose block are
the first six columns of the factor's 7-column "global" Jacobian.  Normal equations: H = sum J^T J, g = sum J^T r, cost = sum rho(|r|^2) / 2 -- what Ceres'
evaluator hands to the trust-region step (before Jacobi scaling).

The window itself (poses, pre-integrations, observations) is INPUT data: it is synthesised with ground-fusion_amd/synth_window.py and stored in the fixture, so
the fixture stays valid when the synthesiser changes.  Run here (CPU container):  python tests/golden/make_ref_golden.py"""
import json
import os
import sys

import numpy as np
import mpmath as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "ground-fusion_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

mp.mp.dps = 60
M = mp.matrix


# ---------------------------------------------------------------- small algebra (quaternions as (w, x, y, z))
def mpf(x):
    return mp.mpf(float(x))          # a double, exactly


def vec(a):
    return M([mpf(x) for x in a])


def qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return (aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw)


def qinv(q):       # Eigen's Quaternion::inverse(): conjugate / squared norm
    n2 = q[0] ** 2 + q[1] ** 2 + q[2] ** 2 + q[3] ** 2
    return (q[0] / n2, -q[1] / n2, -q[2] / n2, -q[3] / n2)


def qnormalized(q):
    n = mp.sqrt(q[0] ** 2 + q[1] ** 2 + q[2] ** 2 + q[3] ** 2)
    return tuple(c / n for c in q)


def qrot(q, v):    # Eigen: q * v  (q v q^-1 for a unit quaternion; Eigen's _transformVector: v + 2 w (u x v) + 2 u x (u x v))
    u = M([q[1], q[2], q[3]])
    uv = cross(u, v)
    return v + 2 * q[0] * uv + 2 * cross(u, uv)


def qmat(q):       # Eigen: toRotationMatrix()
    w, x, y, z = q
    return M([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
              [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
              [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def cross(a, b):
    return M([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])


def skew(q):       # utility.h:39-46
    return M([[0, -q[2], q[1]], [q[2], 0, -q[0]], [-q[1], q[0], 0]])


def delta_q(theta):   # utility.h:23-36
    return qnormalized((mp.mpf(1), theta[0] / 2, theta[1] / 2, theta[2] / 2))


def q_left(q):     # utility.h:59-66 (positify returns q)
    w, v = q[0], M([q[1], q[2], q[3]])
    A = mp.zeros(4, 4)
    A[0, 0] = w
    for k in range(3):
        A[0, 1 + k] = -v[k]; A[1 + k, 0] = v[k]
    B = w * mp.eye(3) + skew(v)
    for r in range(3):
        for c in range(3):
            A[1 + r, 1 + c] = B[r, c]
    return A


def q_right(p):    # utility.h:69-76
    w, v = p[0], M([p[1], p[2], p[3]])
    A = mp.zeros(4, 4)
    A[0, 0] = w
    for k in range(3):
        A[0, 1 + k] = -v[k]; A[1 + k, 0] = v[k]
    B = w * mp.eye(3) - skew(v)
    for r in range(3):
        for c in range(3):
            A[1 + r, 1 + c] = B[r, c]
    return A


def br33(A):       # bottomRightCorner<3, 3>()
    return A[1:4, 1:4]


def pose_of(p7):   # para_Pose layout: px py pz qx qy qz qw  (estimator.cpp:2276-2290)
    return M([mpf(p7[0]), mpf(p7[1]), mpf(p7[2])]), (mpf(p7[6]), mpf(p7[3]), mpf(p7[4]), mpf(p7[5]))


def setblock(J, r0, c0, B):
    for r in range(B.rows):
        for c in range(B.cols):
            J[r0 + r, c0 + c] = B[r, c]


# ---------------------------------------------------------------- the factors
def visual_factor(Pose_i, Pose_j, Ex, inv_dep_i, td, pts_i, pts_j, vel_i, vel_j, td_i, td_j, si):
    """projectionTwoFrameOneCamFactor.cpp:43-151 (UNIT_SPHERE_ERROR not defined).  sqrt_info = si * I2 (estimator.cpp:244: FOCAL_LENGTH / 1.5 * Identity).
    Returns r (2), dict block -> J (2 x local size): 'pi' 6, 'pj' 6, 'ex' 6, 'f' 1, 'td' 1"""
    Pi, Qi = Pose_i
    Pj, Qj = Pose_j
    tic, qic = Ex
    velocity_i, velocity_j = M([vel_i[0], vel_i[1], 0]), M([vel_j[0], vel_j[1], 0])          # :22-27
    pts_i_td = pts_i - (td - td_i) * velocity_i                                                  # :61
    pts_j_td = pts_j - (td - td_j) * velocity_j                                                  # :62
    pts_camera_i = pts_i_td / inv_dep_i                                                          # :63
    pts_imu_i = qrot(qic, pts_camera_i) + tic                                                    # :64
    pts_w = qrot(Qi, pts_imu_i) + Pi                                                             # :65
    pts_imu_j = qrot(qinv(Qj), pts_w - Pj)                                                       # :66
    pts_camera_j = qrot(qinv(qic), pts_imu_j - tic)                                              # :67
    dep_j = pts_camera_j[2]                                                                      # :73
    r = si * M([pts_camera_j[0] / dep_j - pts_j_td[0], pts_camera_j[1] / dep_j - pts_j_td[1]])   # :74, :77
    Ri, Rj, ric = qmat(Qi), qmat(Qj), qmat(qic)                                                  # :81-83
    reduce = si * M([[1 / dep_j, 0, -pts_camera_j[0] / (dep_j * dep_j)], [0, 1 / dep_j, -pts_camera_j[1] / (dep_j * dep_j)]])   # :96-99
    J = {}
    jaco_i = mp.zeros(3, 6)
    setblock(jaco_i, 0, 0, ric.T * Rj.T)                                                         # :106
    setblock(jaco_i, 0, 3, ric.T * Rj.T * Ri * (-skew(pts_imu_i)))                               # :107
    J["pi"] = reduce * jaco_i
    jaco_j = mp.zeros(3, 6)
    setblock(jaco_j, 0, 0, ric.T * (-Rj.T))                                                      # :118
    setblock(jaco_j, 0, 3, ric.T * skew(pts_imu_j))                                              # :119
    J["pj"] = reduce * jaco_j
    jaco_ex = mp.zeros(3, 6)
    setblock(jaco_ex, 0, 0, ric.T * (Rj.T * Ri - mp.eye(3)))                                     # :128
    tmp_r = ric.T * Rj.T * Ri * ric                                                              # :129
    setblock(jaco_ex, 0, 3, -tmp_r * skew(pts_camera_i) + skew(tmp_r * pts_camera_i) + skew(ric.T * (Rj.T * (Ri * tic + Pi - Pj) - tic)))   # :130-131
    J["ex"] = reduce * jaco_ex
    J["f"] = reduce * (tmp_r * pts_i_td) * (-1 / (inv_dep_i * inv_dep_i))                        # :138
    J["td"] = reduce * (tmp_r * velocity_i) / inv_dep_i * -1 + si * M([velocity_j[0], velocity_j[1]])   # :143-144
    return r, J


def huber_correct(r, Js):
    """ceres::HuberLoss(1.0) + Corrector (Ceres 1.14 loss_function.cc:49-65, corrector.cc:43-145): returns rho(s), corrected r and Jacobians"""
    s = sum(x * x for x in r)
    if s > 1:
        rt = mp.sqrt(s)
        rho0, rho1 = 2 * rt - 1, max(mp.mpf(np.finfo(float).tiny), 1 / rt)
        rho2 = -rho1 / (2 * s)
    else:
        rho0, rho1, rho2 = s, mp.mpf(1), mp.mpf(0)
    sqrt_rho1 = mp.sqrt(rho1)
    if s == 0 or rho2 <= 0:
        residual_scaling, alpha_sq_norm = sqrt_rho1, mp.mpf(0)
    else:
        D = 1 + 2 * s * rho2 / rho1
        alpha = 1 - mp.sqrt(D)
        residual_scaling, alpha_sq_norm = sqrt_rho1 / (1 - alpha), alpha / s
    out = {}
    for k, Jm in Js.items():      # J = sqrt_rho1 (J - alpha_sq_norm r (r^T J))
        out[k] = sqrt_rho1 * (Jm - alpha_sq_norm * r * (r.T * Jm))
    return rho0, residual_scaling * r, out


def imu_factor(Pose_i, SBi, Pose_j, SBj, pre, G):
    """imu_factor.h:28-191 on integration_base.h:169-195.  Block order / local columns: pose_i 6, speedbias_i 9, pose_j 6, speedbias_j 9.  O_P 0, O_R 3, O_V 6,
    O_BA 9, O_BG 12 (parameters.h:85-92).  Returns the whitened r (15) and Jacobians."""
    Pi, Qi = Pose_i
    Pj, Qj = Pose_j
    Vi, Bai, Bgi = SBi[0:3], SBi[3:6], SBi[6:9]
    Vj, Baj, Bgj = SBj[0:3], SBj[3:6], SBj[6:9]
    jac, cov, sum_dt = pre["jacobian"], pre["covariance"], pre["sum_dt"]
    dq0, dp0, dv0, lba, lbg = pre["delta_q"], pre["delta_p"], pre["delta_v"], pre["lin_ba"], pre["lin_bg"]
    dp_dba, dp_dbg, dq_dbg, dv_dba, dv_dbg = jac[0:3, 9:12], jac[0:3, 12:15], jac[3:6, 12:15], jac[6:9, 9:12], jac[6:9, 12:15]   # integration_base.h:174-180
    dba, dbg = Bai - lba, Bgi - lbg                                                              # :182-183
    corrected_delta_q = qmul(dq0, delta_q(dq_dbg * dbg))                                         # :185
    corrected_delta_v = dv0 + dv_dba * dba + dv_dbg * dbg                                        # :186
    corrected_delta_p = dp0 + dp_dba * dba + dp_dbg * dbg                                        # :187
    Qi_inv = qinv(Qi)
    r = mp.zeros(15, 1)
    rp = qrot(Qi_inv, mp.mpf("0.5") * G * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt) - corrected_delta_p      # :189
    qe = qmul(qinv(corrected_delta_q), qmul(Qi_inv, Qj))
    rq = 2 * M([qe[1], qe[2], qe[3]])                                                            # :190
    rv = qrot(Qi_inv, G * sum_dt + Vj - Vi) - corrected_delta_v                                  # :191
    for k in range(3):
        r[k], r[3 + k], r[6 + k], r[9 + k], r[12 + k] = rp[k], rq[k], rv[k], Baj[k] - Bai[k], Bgj[k] - Bgi[k]   # :189-193
    sqrt_info = mp.cholesky(cov ** -1).T                                                         # imu_factor.h:73: LLT(cov^-1).matrixL().transpose()
    Rit = qmat(Qi_inv)
    Ji, Jsi, Jj, Jsj = mp.zeros(15, 6), mp.zeros(15, 9), mp.zeros(15, 6), mp.zeros(15, 9)
    setblock(Ji, 0, 0, -Rit)                                                                     # :98
    setblock(Ji, 0, 3, skew(qrot(Qi_inv, mp.mpf("0.5") * G * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt)))   # :99
    setblock(Ji, 3, 3, -br33(q_left(qmul(qinv(Qj), Qi)) * q_right(corrected_delta_q)))           # :104-105
    setblock(Ji, 6, 3, skew(qrot(Qi_inv, G * sum_dt + Vj - Vi)))                                 # :108
    setblock(Jsi, 0, 0, -Rit * sum_dt)                                                           # :123
    setblock(Jsi, 0, 3, -dp_dba); setblock(Jsi, 0, 6, -dp_dbg)                                   # :124-125
    setblock(Jsi, 3, 6, -br33(q_left(qmul(qmul(qinv(Qj), Qi), dq0))) * dq_dbg)                   # :132 (delta_q, not the corrected one)
    setblock(Jsi, 6, 0, -Rit); setblock(Jsi, 6, 3, -dv_dba); setblock(Jsi, 6, 6, -dv_dbg)        # :135-137
    setblock(Jsi, 9, 3, -mp.eye(3)); setblock(Jsi, 12, 6, -mp.eye(3))                            # :139, :141
    setblock(Jj, 0, 0, Rit)                                                                      # :153
    setblock(Jj, 3, 3, br33(q_left(qmul(qmul(qinv(corrected_delta_q), Qi_inv), Qj))))            # :158-159
    setblock(Jsj, 6, 0, Rit); setblock(Jsj, 9, 3, mp.eye(3)); setblock(Jsj, 12, 6, mp.eye(3))    # :172-176
    return sqrt_info * r, {"pi": sqrt_info * Ji, "sbi": sqrt_info * Jsi, "pj": sqrt_info * Jj, "sbj": sqrt_info * Jsj}


# Sophus pieces the wheel factor uses (Sophus itself is not vendored by the reference; factor/wheel_factor.h includes utility/sophus_utils.hpp for the Jacobians)
SOPHUS_EPS = mp.mpf("1e-10")        # Sophus::Constants<double>::epsilon(); epsilonSqrt() = 1e-5


def so3_exp(v):                     # Sophus::SO3d::exp -> unit quaternion (exact exponential; Sophus switches to a Taylor series below |v|^2 < eps^2, same value to 1e-40)
    th = mp.sqrt(v[0] ** 2 + v[1] ** 2 + v[2] ** 2)
    if th == 0:
        return (mp.mpf(1), mp.mpf(0), mp.mpf(0), mp.mpf(0))
    s = mp.sin(th / 2) / th
    return (mp.cos(th / 2), s * v[0], s * v[1], s * v[2])


def so3_log(q):                     # Sophus::SO3d::log of the NORMALISED quaternion (SO3d(q) normalises): 2 atan(|v| / w) / |v| * v
    q = qnormalized(q)
    n = mp.sqrt(q[1] ** 2 + q[2] ** 2 + q[3] ** 2)
    if n == 0:
        return M([0, 0, 0])
    f = 2 * mp.atan(n / q[0]) / n
    return M([f * q[1], f * q[2], f * q[3]])


def right_jacobian_so3(phi):        # sophus_utils.hpp:155-184
    n2 = phi[0] ** 2 + phi[1] ** 2 + phi[2] ** 2
    h = skew(phi); h2 = h * h
    J = mp.eye(3)
    if n2 > SOPHUS_EPS:
        n = mp.sqrt(n2)
        J = J - h * (1 - mp.cos(n)) / n2 + h2 * (n - mp.sin(n)) / (n2 * n)
    else:
        J = J - h / 2 + h2 / 6
    return J


def right_jacobian_inv_so3(phi):    # sophus_utils.hpp:195-236
    n2 = phi[0] ** 2 + phi[1] ** 2 + phi[2] ** 2
    h = skew(phi); h2 = h * h
    J = mp.eye(3) + h / 2
    if n2 > SOPHUS_EPS:
        n = mp.sqrt(n2)
        if n < mp.pi - mp.mpf("1e-5"):
            J = J + h2 * (1 / n2 - (1 + mp.cos(n)) / (2 * n * mp.sin(n)))
        else:
            J = J + h2 / (mp.pi * mp.pi)
    else:
        J = J + h2 / 12
    return J


def wheel_factor(Pose_i, Pose_j, Exw, sx, sy, sw, td, pre):
    """wheel_factor.h:28-247 on wheel_integration_base.h:180-219.  Blocks / local columns: pose_i 6, pose_j 6, wheel extrinsic 6, sx, sy, sw, td_wheel (1 each); O_P 0, O_R 3.
    Returns the whitened r (6) and Jacobians."""
    Pi, Qi = Pose_i
    Pj, Qj = Pose_j
    tio, qio = Exw
    jac, cov = pre["jacobian"], pre["covariance"]                     # 6 x 3, 6 x 6
    dp_dsx, dp_dsy, dp_dsw, dq_dsw = jac[0:3, 0], jac[0:3, 1], jac[0:3, 2], jac[3:6, 2]      # wheel_integration_base.h:186-191
    lsx, lsy, lsw, ltd = pre["lin"]
    lin_vel, lin_gyr, vel_1, gyr_1 = pre["lin_vel"], pre["lin_gyr"], pre["vel_1"], pre["gyr_1"]
    dsx, dsy, dsw = sx - lsx, sy - lsy, sw - lsw                                                # :193-195
    sv = mp.diag([sx, sy, 1])                                                                   # :196
    Ri, Rj, rio = qmat(Qi), qmat(Qj), qmat(qio)                                                 # :198-200
    corrected_delta_p = pre["delta_p"] + dp_dsx * dsx + dp_dsy * dsy + dp_dsw * dsw              # :202
    corrected_delta_q = qnormalized(qmul(qnormalized(pre["delta_q"]), so3_exp(dq_dsw * dsw)))   # :203
    dtd = td - ltd                                                                              # :204
    e_fw = so3_exp(sw * lin_gyr * dtd)
    delta_q_time = qnormalized(qmul(qmul(e_fw, corrected_delta_q), so3_exp(-sw * gyr_1 * dtd)))                          # :206
    delta_p_time = qmat(e_fw) * (sv * lin_vel * dtd + corrected_delta_p - qrot(corrected_delta_q, sv * vel_1 * dtd))    # :207
    rp = (Ri * rio).T * (Rj * tio + Pj - Ri * tio - Pi) - delta_p_time                                                    # :212
    rr = so3_log(qmul(qmul(qmul(qinv(delta_q_time), qinv(qmul(Qi, qio))), Qj), qio))                                      # :213
    r = M([rp[0], rp[1], rp[2], rr[0], rr[1], rr[2]])
    sqrt_info = mp.cholesky(cov ** -1).T                                                        # wheel_factor.h:85
    Jr_delta_q_inv = right_jacobian_inv_so3(rr)                                                 # :106-108 (raw residual)
    drdsw = dq_dsw * (sw - lsw)
    Jr_drdsw = right_jacobian_so3(drdsw)                                                        # :110-112
    Qio = qmul(Qi, qio)
    Ji, Jj, Jex = mp.zeros(6, 6), mp.zeros(6, 6), mp.zeros(6, 6)
    setblock(Ji, 0, 0, -qmat(qinv(Qio)))                                                        # :122
    setblock(Ji, 0, 3, (Ri * rio).T * (Ri * skew(tio)) + rio.T * skew(Ri.T * (Rj * tio + Pj - Ri * tio - Pi)))           # :124
    setblock(Ji, 3, 3, -Jr_delta_q_inv * qmat(qmul(qinv(qmul(Qj, qio)), Qi)))                   # :133
    setblock(Jj, 0, 0, qmat(qinv(Qio)))                                                         # :153
    setblock(Jj, 0, 3, -qmat(qmul(qinv(Qio), Qj)) * skew(tio))                                  # :154
    setblock(Jj, 3, 3, Jr_delta_q_inv * qmat(qinv(qio)))                                        # :160
    setblock(Jex, 0, 0, qmat(qinv(Qio)) * (Rj - Ri))                                            # :173
    setblock(Jex, 0, 3, skew(qrot(qinv(Qio), qrot(Qj, tio) + Pj - qrot(Qi, tio) - Pi)))         # :175
    setblock(Jex, 3, 3, Jr_delta_q_inv * (mp.eye(3) - qmat(qmul(qmul(qinv(qmul(Qj, qio)), Qi), qio))))                   # :177
    fcw, fcv, bcv, bcw = sw * lin_gyr * dtd, sv * lin_vel * dtd, sv * vel_1 * dtd, sw * gyr_1 * dtd                        # :184-187
    Jrtd, Jr_minus_td = right_jacobian_so3(fcw), right_jacobian_so3(-fcw)                       # :189-192
    I1, I2 = mp.diag([1, 0, 0]), mp.diag([0, 1, 0])                                             # :193-194
    Rcq = qmat(corrected_delta_q)
    Efv, Efw = qmat(so3_exp(fcv)), qmat(so3_exp(fcw))
    Em, Ebw, Rcq_inv = qmat(so3_exp(-rr)), qmat(so3_exp(bcw)), qmat(qinv(corrected_delta_q))
    Jsx, Jsy, Jsw, Jtd = mp.zeros(6, 1), mp.zeros(6, 1), mp.zeros(6, 1), mp.zeros(6, 1)
    setblock(Jsx, 0, 0, -(Efv * (I1 * lin_vel * dtd + dp_dsx - Rcq * (I1 * vel_1) * dtd)))     # :199 (exp(forward_compensate_v): as written)
    setblock(Jsy, 0, 0, -(Efv * (I2 * lin_vel * dtd + dp_dsy - Rcq * (I2 * vel_1) * dtd)))     # :211
    setblock(Jsw, 0, 0, -(Efw * (dp_dsw - Rcq * skew(Jr_drdsw * dq_dsw) * (sv * vel_1) * dtd + skew(Jrtd * lin_gyr * dtd) * (fcv + corrected_delta_p - qrot(corrected_delta_q, bcv)))))   # :223
    setblock(Jsw, 3, 0, -(Jr_delta_q_inv * Em * Ebw * (Rcq_inv * (Jrtd * lin_gyr) * dtd + Jr_drdsw * dq_dsw)))          # :225
    setblock(Jtd, 0, 0, -(Efw * (sv * lin_vel - Rcq * (sv * vel_1) + skew(Jrtd * sw * lin_gyr) * (fcv + corrected_delta_p - Rcq * bcv))))   # :236
    setblock(Jtd, 3, 0, -(Jr_delta_q_inv * Em * (Ebw * Rcq_inv * (Jrtd * sw * lin_gyr) - Jr_minus_td * sw * gyr_1)))   # :237
    S = sqrt_info
    return S * r, {"pi": S * Ji, "pj": S * Jj, "exw": S * Jex, "sx": S * Jsx, "sy": S * Jsy, "sw": S * Jsw, "tdw": S * Jtd}


# ---- GNSS (factor/gnss_psr_dopp_factor.cpp, gnss_dt_ddt_factor.cpp, gnss_ddt_smooth_factor.cpp are in the reference; ecef2geo / ecef2rotation / sat_azel / the
# Saastamoinen and Klobuchar delays come from gnss_comm, which the reference does not vendor: those five are restated from their published form -- the SAME reading the
# oracle and the kernels follow, so they are only pinned against transcription slips here, not against gnss_comm)
GN_C, GN_OMG, GN_A, GN_E2 = mp.mpf("2.99792458e8"), mp.mpf("7.2921151467e-5"), mp.mpf("6378137.0"), mp.mpf("6.69437999014e-3")


def gn_ecef2geo(p):   # latitude [deg], longitude [deg], height [m] (closed form, RTKLIB lineage)
    if p[0] == 0 and p[1] == 0:
        return M([0, 0, 0])
    a = GN_A; a2 = a * a; b2 = a2 * (1 - GN_E2); b = mp.sqrt(b2); ep2 = (a2 - b2) / b2; rho = mp.sqrt(p[0] ** 2 + p[1] ** 2)
    s1, s2 = p[2] * a, rho * b
    h = mp.sqrt(s1 * s1 + s2 * s2)
    st, ct = s1 / h, s2 / h
    s1 = p[2] + ep2 * b * st ** 3
    s2 = rho - a * GN_E2 * ct ** 3
    h = mp.sqrt(s1 * s1 + s2 * s2)
    sin_lat, cos_lat = s1 / h, s2 / h
    N = a2 / mp.sqrt(a2 * cos_lat ** 2 + b2 * sin_lat ** 2)
    return M([mp.atan(s1 / s2) * 180 / mp.pi, mp.atan2(p[1], p[0]) * 180 / mp.pi, rho / cos_lat - N])


def gn_geo2rotation(lla):   # R_ecef_enu
    lat, lon = lla[0] * mp.pi / 180, lla[1] * mp.pi / 180
    sl, cl, so, co = mp.sin(lat), mp.cos(lat), mp.sin(lon), mp.cos(lon)
    return M([[-so, -sl * co, cl * co], [co, -sl * so, cl * so], [0, cl, sl]])


def gn_sat_azel(rcv, sat):
    dl = sat - rcv
    dl = dl / mp.sqrt(dl[0] ** 2 + dl[1] ** 2 + dl[2] ** 2)
    enu = gn_geo2rotation(gn_ecef2geo(rcv)).T * dl
    az = mp.mpf(0) if mp.sqrt(dl[0] ** 2 + dl[1] ** 2) < mp.mpf("1e-12") else mp.atan2(enu[0], enu[1])
    if az < 0:
        az += 2 * mp.pi
    return az, mp.asin(enu[2])


def gn_trop_delay(lla, el):   # Saastamoinen, standard atmosphere, relative humidity 0.7
    if lla[2] < -100 or lla[2] > 1e4 or el <= 0:
        return mp.mpf(0)
    hgt = mp.mpf(0) if lla[2] < 0 else lla[2]
    pres = mp.mpf("1013.25") * (1 - mp.mpf("2.2557e-5") * hgt) ** mp.mpf("5.2568")
    temp = 15 - mp.mpf("6.5e-3") * hgt + mp.mpf("273.16")
    e = mp.mpf("6.108") * mp.mpf("0.7") * mp.exp((mp.mpf("17.15") * temp - 4684) / (temp - mp.mpf("38.45")))
    z = mp.pi / 2 - el
    trph = mp.mpf("0.0022768") * pres / (1 - mp.mpf("0.00266") * mp.cos(2 * lla[0] * mp.pi / 180) - mp.mpf("0.00028") * hgt / 1000) / mp.cos(z)
    trpw = mp.mpf("0.002277") * (1255 / temp + mp.mpf("0.05")) * e / mp.cos(z)
    return trph + trpw


def gn_ion_delay(tow, ion_in, lla, az, el):   # Klobuchar
    ion_default = [mp.mpf(x) for x in ("0.1118e-07", "-0.7451e-08", "-0.5961e-07", "0.1192e-06", "0.1167e+06", "-0.2294e+06", "-0.1311e+06", "0.1049e+07")]
    if lla[2] < -1000 or el <= 0:
        return mp.mpf(0)
    nrm = sum(x * x for x in ion_in)
    ion = ion_default if nrm <= 0 else ion_in
    psi = mp.mpf("0.0137") / (el / mp.pi + mp.mpf("0.11")) - mp.mpf("0.022")
    phi = lla[0] / 180 + psi * mp.cos(az)
    phi = min(max(phi, mp.mpf("-0.416")), mp.mpf("0.416"))
    lam = lla[1] / 180 + psi * mp.sin(az) / mp.cos(phi * mp.pi)
    phi += mp.mpf("0.064") * mp.cos((lam - mp.mpf("1.617")) * mp.pi)
    tt = 43200 * lam + tow
    tt -= mp.floor(tt / 86400) * 86400
    f = 1 + 16 * (mp.mpf("0.53") - el / mp.pi) ** 3
    amp = ion[0] + phi * (ion[1] + phi * (ion[2] + phi * ion[3]))
    per = ion[4] + phi * (ion[5] + phi * (ion[6] + phi * ion[7]))
    amp = max(amp, mp.mpf(0)); per = max(per, mp.mpf(72000))
    x = 2 * mp.pi * (tt - 50400) / per
    return GN_C * f * ((mp.mpf("5e-9") + amp * (1 + x * x * (mp.mpf("-0.5") + x * x / 24))) if abs(x) < mp.mpf("1.57") else mp.mpf("5e-9"))


def gnss_psr_dopp_factor(Pi, Vi, Pj, Vj, rcv_dt, rcv_ddt, yaw_diff, ref_ecef, dat, ratio, iono):
    """GnssPsrDoppFactor::Evaluate, gnss_psr_dopp_factor.cpp:49-208.  dat: what the constructor (:3-47) leaves in the factor -- sv_pos, sv_vel, svdt, svddt, tgd, pr_uura,
    dp_uura -- and the observation (psr, dopp, wavelength, time of week).  Returns r (2) and Jacobians: 'pi' 2x6, 'vi' 2x9, 'pj', 'vj', 'dt' 2x1, 'ddt' 2x1, 'yaw' 2x1, 'anc' 2x3"""
    sv_pos, sv_vel = dat["sv_pos"], dat["sv_vel"]
    local_pos, local_vel = ratio * Pi + (1 - ratio) * Pj, ratio * Vi + (1 - ratio) * Vj          # :61-62
    sy, cy = mp.sin(yaw_diff), mp.cos(yaw_diff)
    R_enu_local = M([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])                                    # :66-69
    R_ecef_enu = gn_geo2rotation(gn_ecef2geo(ref_ecef))                                          # :70 ecef2rotation
    R_ecef_local = R_ecef_enu * R_enu_local
    P_ecef, V_ecef = R_ecef_local * local_pos + ref_ecef, R_ecef_local * local_vel              # :73-74
    ion_delay = tro_delay = mp.mpf(0)
    az, el = mp.mpf(0), mp.pi / 2
    if P_ecef[0] ** 2 + P_ecef[1] ** 2 + P_ecef[2] ** 2 > 0:                                     # :78-84
        az, el = gn_sat_azel(P_ecef, sv_pos)
        lla = gn_ecef2geo(P_ecef)
        tro_delay = gn_trop_delay(lla, el)
        ion_delay = gn_ion_delay(dat["tow"], iono, lla, az, el)
    sin_el_2 = mp.sin(el) ** 2
    pr_weight = sin_el_2 / dat["pr_uura"] * 10                                                   # :87 relative_sqrt_info 10 (:46)
    dp_weight = sin_el_2 / dat["dp_uura"] * 10 * 5                                               # :88 PSR_TO_DOPP_RATIO 5
    rcv2sat = sv_pos - P_ecef
    norm2 = rcv2sat[0] ** 2 + rcv2sat[1] ** 2 + rcv2sat[2] ** 2
    rng = mp.sqrt(norm2)
    unit = rcv2sat / rng
    psr_sagnac = GN_OMG * (sv_pos[0] * P_ecef[1] - sv_pos[1] * P_ecef[0]) / GN_C                 # :93
    psr_est = rng + psr_sagnac + rcv_dt - dat["svdt"] * GN_C + ion_delay + tro_delay + dat["tgd"] * GN_C   # :94-95
    dopp_sagnac = GN_OMG / GN_C * (sv_vel[0] * P_ecef[1] + sv_pos[0] * V_ecef[1] - sv_vel[1] * P_ecef[0] - sv_pos[1] * V_ecef[0])   # :99-100
    dv = sv_vel - V_ecef
    dopp_est = (dv.T * unit)[0] + dopp_sagnac + rcv_ddt - dat["svddt"] * GN_C                    # :101
    r = M([(psr_est - dat["psr"]) * pr_weight, (dopp_est + dat["dopp"] * dat["wavelength"]) * dp_weight])   # :97, :103
    norm3 = rng ** 3
    u2r = mp.zeros(3, 3)                                                                         # :116-128
    for a in range(3):
        for c in range(3):
            u2r[a, c] = -(((norm2 - rcv2sat[a] * rcv2sat[a]) / norm3) if a == c else ((-rcv2sat[a] * rcv2sat[c]) / norm3))
    Jpi, Jvi, Jpj, Jvj = mp.zeros(2, 6), mp.zeros(2, 9), mp.zeros(2, 6), mp.zeros(2, 9)
    top = -(unit.T * R_ecef_local)                                                               # 1 x 3
    bot = dv.T * u2r * R_ecef_local
    for c in range(3):
        Jpi[0, c] = top[0, c] * pr_weight * ratio; Jpi[1, c] = bot[0, c] * dp_weight * ratio                 # :113, :129-130
        Jvi[1, c] = top[0, c] * dp_weight * ratio                                                                # :138-139
        Jpj[0, c] = top[0, c] * pr_weight * (1 - ratio); Jpj[1, c] = bot[0, c] * dp_weight * (1 - ratio)         # :147, :163-164
        Jvj[1, c] = top[0, c] * dp_weight * (1 - ratio)                                                          # :172-173
    d_yaw = M([[-sy, -cy, 0], [cy, -sy, 0], [0, 0, 0]])                                          # :193-196
    Jyaw = M([-(unit.T * (R_ecef_enu * (d_yaw * local_pos)))[0] * pr_weight, -(unit.T * (R_ecef_enu * (d_yaw * local_vel)))[0] * dp_weight])   # :197-198
    Janc = mp.zeros(2, 3)
    for c in range(3):
        Janc[0, c] = -unit[c] * pr_weight                                                        # :206 ("approximation for simplicity")
    return r, {"pi": Jpi, "vi": Jvi, "pj": Jpj, "vj": Jvj, "dt": M([pr_weight, 0]), "ddt": M([0, dp_weight]), "yaw": Jyaw, "anc": Janc}


def gnss_factors(w, frames=None):
    """every GNSS residual block of the window as (r, [(block id, J)]): GnssPsrDoppFactor per observation (estimator.cpp:3182-3208), DtDdtFactor x 4 and DdtSmoothFactor
    per frame pair (:3211-3229).  frames: only the blocks MARGIN_OLD takes (estimator.cpp:3397-3434: observations of frame 0, clock factors of the pair (0, 1))."""
    import gfwindow as gw
    NP = int(w["W"]) + 1
    out = []
    P = [vec(w["para_Pose"][7 * i:7 * i + 3]) for i in range(NP)]
    V = [vec(w["para_SpeedBias"][9 * i:9 * i + 3]) for i in range(NP)]
    dt = [mpf(x) for x in w["para_rcv_dt"]]
    ddt = [mpf(x) for x in w["para_rcv_ddt"]]
    yaw, anc = mpf(w["para_yaw_enu_local"][0]), vec(w["para_anc_ecef"])
    iono = [mpf(x) for x in w["gnss_iono"]]
    for k in range(int(w["n_gnss"])):
        i, lo, sys = int(w["gnss_frame"][k]), int(w["gnss_lower"][k]), int(w["gnss_sys"][k])
        if frames is not None and i not in frames:
            continue
        d = [mpf(x) for x in w["gnss_data"][16 * k:16 * k + 16]]
        dat = {"sv_pos": M(d[0:3]), "sv_vel": M(d[3:6]), "svdt": d[6], "svddt": d[7], "tgd": d[8], "pr_uura": d[9], "dp_uura": d[10], "psr": d[11], "dopp": d[12],
               "wavelength": d[13], "tow": d[14]}
        r, J = gnss_psr_dopp_factor(P[lo], V[lo], P[lo + 1], V[lo + 1], dt[4 * i + sys], ddt[i], yaw, anc, dat, mpf(w["gnss_ratio"][k]), iono)
        out.append((r, [(gw.bid(gw.POSE, lo), J["pi"]), (gw.bid(gw.SPEEDBIAS, lo), J["vi"]), (gw.bid(gw.POSE, lo + 1), J["pj"]), (gw.bid(gw.SPEEDBIAS, lo + 1), J["vj"]),
                        (gw.bid(gw.RCV_DT, 4 * i + sys), J["dt"]), (gw.bid(gw.RCV_DDT, i), J["ddt"]), (gw.bid(gw.YAW), J["yaw"]), (gw.bid(gw.ANC), J["anc"])]))
    hdr = [mpf(x) for x in w["gnss_headers"]]
    wt = mpf(w["gnss_ddt_weight"])
    for i in range(NP - 1):
        if frames is not None and i not in frames:
            continue
        delta_t = hdr[i + 1] - hdr[i]
        for q in range(4):   # gnss_dt_ddt_factor.cpp: dt_info_coeff 50
            r = M([(dt[4 * (i + 1) + q] - dt[4 * i + q] - (ddt[i] + ddt[i + 1]) / 2 * delta_t) * 50])
            out.append((r, [(gw.bid(gw.RCV_DT, 4 * i + q), M([[-50]])), (gw.bid(gw.RCV_DT, 4 * (i + 1) + q), M([[50]])), (gw.bid(gw.RCV_DDT, i), M([[-delta_t * 25]])),
                            (gw.bid(gw.RCV_DDT, i + 1), M([[-delta_t * 25]]))]))
        out.append((M([(ddt[i] - ddt[i + 1]) * wt]), [(gw.bid(gw.RCV_DDT, i), M([[wt]])), (gw.bid(gw.RCV_DDT, i + 1), M([[-wt]]))]))   # gnss_ddt_smooth_factor.cpp
    return out


def prior_dx(kind_is_pose, x, x0):
    """marginalization_factor.cpp:356-372"""
    if not kind_is_pose:
        return [mpf(a) - mpf(b) for a, b in zip(x, x0)]
    d = [mpf(x[k]) - mpf(x0[k]) for k in range(3)]
    q0, q = (mpf(x0[6]), mpf(x0[3]), mpf(x0[4]), mpf(x0[5])), (mpf(x[6]), mpf(x[3]), mpf(x[4]), mpf(x[5]))
    e = qmul(qinv(q0), q)
    sgn = 1 if e[0] >= 0 else -1
    return d + [2 * sgn * e[1], 2 * sgn * e[2], 2 * sgn * e[3]]


# ---------------------------------------------------------------- a window's normal equations
def window_normal_equations(w, ids):
    import gfwindow as gw
    n = len(ids)
    col0 = {}
    for c, b in enumerate(ids):
        col0.setdefault(int(b), c)
    H, g, cost = mp.zeros(n, n), mp.zeros(n, 1), mp.mpf(0)

    def add(r, blocks):       # blocks: list of (block id, J); constant blocks (no column) drop out
        live = [(col0[b], Jm) for b, Jm in blocks if b in col0]
        for ca, Ja in live:
            ga = Ja.T * r
            for q in range(Ja.cols):
                g[ca + q] += ga[q]
            for cb, Jb in live:
                Hab = Ja.T * Jb
                for p in range(Ja.cols):
                    for q in range(Jb.cols):
                        H[ca + p, cb + q] += Hab[p, q]
    NP = w["W"] + 1
    pose = [pose_of(w["para_Pose"][7 * i:7 * i + 7]) for i in range(NP)]
    sb = [vec(w["para_SpeedBias"][9 * i:9 * i + 9]) for i in range(NP)]
    ex = pose_of(w["para_Ex_Pose"])
    td = mpf(w["para_Td"][0])
    G = vec(w["G"])
    si = mpf(w["vis_sqrt_info"])
    # prior (estimator.cpp:2947-2953: added first, no loss)
    if w["prior_n"] > 0:
        npr = int(w["prior_n"])
        J0 = M(npr, npr)
        for a in range(npr):
            for c in range(npr):
                J0[a, c] = mpf(w["prior_J"][a * npr + c])
        r0 = vec(w["prior_r"])
        dx, blocks, idx, xo = mp.zeros(npr, 1), [], 0, 0
        for bid_ in w["prior_block_id"]:
            kind, i = int(bid_) // 4096, int(bid_) % 4096
            gs, ls = gw.gsize(kind), gw.lsize(kind)
            off = {gw.POSE: ("para_Pose", 7 * i), gw.SPEEDBIAS: ("para_SpeedBias", 9 * i), gw.EX_POSE: ("para_Ex_Pose", 0), gw.EX_WHEEL: ("para_Ex_Pose_wheel", 0),
                   gw.SX: ("para_Ix", 0), gw.SY: ("para_Ix", 1), gw.SW: ("para_Ix", 2), gw.TD: ("para_Td", 0), gw.TD_WHEEL: ("para_Td_wheel", 0),
                   gw.RCV_DT: ("para_rcv_dt", i), gw.RCV_DDT: ("para_rcv_ddt", i), gw.YAW: ("para_yaw_enu_local", 0), gw.ANC: ("para_anc_ecef", 0)}[kind]
            x = w[off[0]][off[1]:off[1] + gs]
            d = prior_dx(gs == 7, x, w["prior_x0"][xo:xo + gs])
            for q in range(ls):
                dx[idx + q] = d[q]
            blocks.append((int(bid_), J0[:, idx:idx + ls]))
            idx += ls; xo += gs
        r = r0 + J0 * dx                                                                         # marginalization_factor.cpp:374
        cost += (r.T * r)[0] / 2
        add(r, blocks)
    # IMU factors (estimator.cpp:3048-3062, no loss)
    for k in range(int(w["n_imu"])):
        i = int(w["imu_i"][k]); j = i + 1
        pre = {"sum_dt": mpf(w["imu_sum_dt"][k]), "delta_p": vec(w["imu_delta_p"][3 * k:3 * k + 3]), "delta_v": vec(w["imu_delta_v"][3 * k:3 * k + 3]),
               "delta_q": tuple(mpf(x) for x in w["imu_delta_q"][4 * k:4 * k + 4]), "lin_ba": vec(w["imu_lin_ba"][3 * k:3 * k + 3]), "lin_bg": vec(w["imu_lin_bg"][3 * k:3 * k + 3]),
               "jacobian": M(15, 15), "covariance": M(15, 15)}
        for a in range(15):
            for c in range(15):
                pre["jacobian"][a, c] = mpf(w["imu_jacobian"][225 * k + 15 * a + c]); pre["covariance"][a, c] = mpf(w["imu_covariance"][225 * k + 15 * a + c])
        r, J = imu_factor(pose[i], sb[i], pose[j], sb[j], pre, G)
        cost += (r.T * r)[0] / 2
        add(r, [(gw.bid(gw.POSE, i), J["pi"]), (gw.bid(gw.SPEEDBIAS, i), J["sbi"]), (gw.bid(gw.POSE, j), J["pj"]), (gw.bid(gw.SPEEDBIAS, j), J["sbj"])])
    # wheel factors (estimator.cpp:3064-3080, no loss)
    exw = pose_of(w["para_Ex_Pose_wheel"])
    for k in range(int(w["n_wheel"])):
        i = int(w["wh_i"][k]); j = i + 1
        pre = {"delta_p": vec(w["wh_delta_p"][3 * k:3 * k + 3]), "delta_q": tuple(mpf(x) for x in w["wh_delta_q"][4 * k:4 * k + 4]), "jacobian": M(6, 3), "covariance": M(6, 6),
               "lin": [mpf(x) for x in w["wh_lin"][4 * k:4 * k + 4]], "lin_vel": vec(w["wh_lin_vel"][3 * k:3 * k + 3]), "lin_gyr": vec(w["wh_lin_gyr"][3 * k:3 * k + 3]),
               "vel_1": vec(w["wh_vel_1"][3 * k:3 * k + 3]), "gyr_1": vec(w["wh_gyr_1"][3 * k:3 * k + 3])}
        for a in range(6):
            for c in range(3):
                pre["jacobian"][a, c] = mpf(w["wh_jacobian"][18 * k + 3 * a + c])
            for c in range(6):
                pre["covariance"][a, c] = mpf(w["wh_covariance"][36 * k + 6 * a + c])
        r, J = wheel_factor(pose[i], pose[j], exw, mpf(w["para_Ix"][0]), mpf(w["para_Ix"][1]), mpf(w["para_Ix"][2]), mpf(w["para_Td_wheel"][0]), pre)
        cost += (r.T * r)[0] / 2
        add(r, [(gw.bid(gw.POSE, i), J["pi"]), (gw.bid(gw.POSE, j), J["pj"]), (gw.bid(gw.EX_WHEEL), J["exw"]), (gw.bid(gw.SX), J["sx"]), (gw.bid(gw.SY), J["sy"]), (gw.bid(gw.SW), J["sw"]),
                (gw.bid(gw.TD_WHEEL), J["tdw"])])
    # GNSS blocks (estimator.cpp:3178-3229: only when gnss_ready and not lowspeed; no loss)
    if int(w.get("gnss_enabled", 0)) and not int(w.get("gnss_lowspeed", 0)):
        for r, blocks in gnss_factors(w):
            cost += (r.T * r)[0] / 2
            add(r, blocks)
    # visual factors (estimator.cpp:3269-3297, loss_function = HuberLoss(1.0))
    for k in range(int(w["n_visual"])):
        f, i, j = int(w["vis_feature"][k]), int(w["vis_i"][k]), int(w["vis_j"][k])
        r, J = visual_factor(pose[i], pose[j], ex, mpf(w["para_Feature"][f]), td, vec(w["vis_pts_i"][3 * k:3 * k + 3]), vec(w["vis_pts_j"][3 * k:3 * k + 3]),
                             vec(w["vis_vel_i"][2 * k:2 * k + 2]), vec(w["vis_vel_j"][2 * k:2 * k + 2]), mpf(w["vis_td_i"][k]), mpf(w["vis_td_j"][k]), si)
        rho0, rc, Jc = huber_correct(r, J)
        cost += rho0 / 2
        add(rc, [(gw.bid(gw.POSE, i), Jc["pi"]), (gw.bid(gw.POSE, j), Jc["pj"]), (gw.bid(gw.EX_POSE), Jc["ex"]), (gw.bid(gw.FEATURE, f), Jc["f"]), (gw.bid(gw.TD), Jc["td"])])
    return H, g, cost


def marg_golden(w, mode=0):
    """mode 0 -- MARGIN_OLD as the reference runs it: the factor set of estimator.cpp:3334-3475 (prior, IMU factor 0 -> 1, wheel factor 0 -> 1, the visual factors of the features
    that start in frame 0; dropped: pose 0, speed-bias 0, those features), ResidualBlockInfo::Evaluate with the loss scaling (marginalization_factor.cpp:12-78),
    A = sum J^T J, b = sum J^T r over [dropped | kept] columns (:119-239), the eigen pseudo-inverse of A_mm and the eigen square root of the kept system with
    eps = 1e-8 (:278-302).  Returns the kept block ids (first-appearance order) and J^T J = V S V^T, J^T r = V V^T b_r (what the next window's solver sees: they do
    not depend on the eigenvector basis nor on the order of the dropped columns), plus the kept system's eigenvalues around the cut.
    mode 1 -- MARGIN_SECOND_NEW (estimator.cpp:3506-3560): the prior alone, dropped: pose WINDOW_SIZE - 1."""
    import gfwindow as gw
    NP = w["W"] + 1
    pose = [pose_of(w["para_Pose"][7 * i:7 * i + 7]) for i in range(NP)]
    sb = [vec(w["para_SpeedBias"][9 * i:9 * i + 9]) for i in range(NP)]
    ex, exw = pose_of(w["para_Ex_Pose"]), pose_of(w["para_Ex_Pose_wheel"])
    td, G, si = mpf(w["para_Td"][0]), vec(w["G"]), mpf(w["vis_sqrt_info"])
    factors = []      # (r, [(block id, J)])
    drop_ids = [gw.bid(gw.POSE, 0), gw.bid(gw.SPEEDBIAS, 0)] if mode == 0 else [gw.bid(gw.POSE, int(w["W"]) - 1)]
    if w["prior_n"] > 0:
        npr = int(w["prior_n"])
        J0 = M(npr, npr)
        for a in range(npr):
            for c in range(npr):
                J0[a, c] = mpf(w["prior_J"][a * npr + c])
        dx, blocks, idx, xo = mp.zeros(npr, 1), [], 0, 0
        for bid_ in w["prior_block_id"]:
            kind, i = int(bid_) // 4096, int(bid_) % 4096
            gs, ls = gw.gsize(kind), gw.lsize(kind)
            off = {gw.POSE: ("para_Pose", 7 * i), gw.SPEEDBIAS: ("para_SpeedBias", 9 * i), gw.EX_POSE: ("para_Ex_Pose", 0), gw.EX_WHEEL: ("para_Ex_Pose_wheel", 0),
                   gw.SX: ("para_Ix", 0), gw.SY: ("para_Ix", 1), gw.SW: ("para_Ix", 2), gw.TD: ("para_Td", 0), gw.TD_WHEEL: ("para_Td_wheel", 0),
                   gw.RCV_DT: ("para_rcv_dt", i), gw.RCV_DDT: ("para_rcv_ddt", i), gw.YAW: ("para_yaw_enu_local", 0), gw.ANC: ("para_anc_ecef", 0)}[kind]
            d = prior_dx(gs == 7, w[off[0]][off[1]:off[1] + gs], w["prior_x0"][xo:xo + gs])
            for q in range(ls):
                dx[idx + q] = d[q]
            blocks.append((int(bid_), J0[:, idx:idx + ls]))
            idx += ls; xo += gs
        factors.append((vec(w["prior_r"]) + J0 * dx, blocks))
    for k in range(int(w["n_imu"]) if mode == 0 else 0):
        if int(w["imu_i"][k]) != 0 or float(w["imu_sum_dt"][k]) >= 10.0:
            continue
        pre = {"sum_dt": mpf(w["imu_sum_dt"][k]), "delta_p": vec(w["imu_delta_p"][3 * k:3 * k + 3]), "delta_v": vec(w["imu_delta_v"][3 * k:3 * k + 3]),
               "delta_q": tuple(mpf(x) for x in w["imu_delta_q"][4 * k:4 * k + 4]), "lin_ba": vec(w["imu_lin_ba"][3 * k:3 * k + 3]), "lin_bg": vec(w["imu_lin_bg"][3 * k:3 * k + 3]),
               "jacobian": M(15, 15), "covariance": M(15, 15)}
        for a in range(15):
            for c in range(15):
                pre["jacobian"][a, c] = mpf(w["imu_jacobian"][225 * k + 15 * a + c]); pre["covariance"][a, c] = mpf(w["imu_covariance"][225 * k + 15 * a + c])
        r, J = imu_factor(pose[0], sb[0], pose[1], sb[1], pre, G)
        factors.append((r, [(gw.bid(gw.POSE, 0), J["pi"]), (gw.bid(gw.SPEEDBIAS, 0), J["sbi"]), (gw.bid(gw.POSE, 1), J["pj"]), (gw.bid(gw.SPEEDBIAS, 1), J["sbj"])]))
    for k in range(int(w["n_wheel"]) if mode == 0 else 0):
        if int(w["wh_i"][k]) != 0 or float(w["wh_sum_dt"][k]) >= 10.0:
            continue
        pre = {"delta_p": vec(w["wh_delta_p"][3 * k:3 * k + 3]), "delta_q": tuple(mpf(x) for x in w["wh_delta_q"][4 * k:4 * k + 4]), "jacobian": M(6, 3), "covariance": M(6, 6),
               "lin": [mpf(x) for x in w["wh_lin"][4 * k:4 * k + 4]], "lin_vel": vec(w["wh_lin_vel"][3 * k:3 * k + 3]), "lin_gyr": vec(w["wh_lin_gyr"][3 * k:3 * k + 3]),
               "vel_1": vec(w["wh_vel_1"][3 * k:3 * k + 3]), "gyr_1": vec(w["wh_gyr_1"][3 * k:3 * k + 3])}
        for a in range(6):
            for c in range(3):
                pre["jacobian"][a, c] = mpf(w["wh_jacobian"][18 * k + 3 * a + c])
            for c in range(6):
                pre["covariance"][a, c] = mpf(w["wh_covariance"][36 * k + 6 * a + c])
        r, J = wheel_factor(pose[0], pose[1], exw, mpf(w["para_Ix"][0]), mpf(w["para_Ix"][1]), mpf(w["para_Ix"][2]), mpf(w["para_Td_wheel"][0]), pre)
        factors.append((r, [(gw.bid(gw.POSE, 0), J["pi"]), (gw.bid(gw.POSE, 1), J["pj"]), (gw.bid(gw.EX_WHEEL), J["exw"]), (gw.bid(gw.SX), J["sx"]), (gw.bid(gw.SY), J["sy"]),
                            (gw.bid(gw.SW), J["sw"]), (gw.bid(gw.TD_WHEEL), J["tdw"])]))
    if mode == 0 and int(w.get("gnss_enabled", 0)):   # estimator.cpp:3397-3434 (taken whenever gnss_ready, lowspeed or not): dropped with them the frame-0 clocks
        for r, blocks in gnss_factors(w, frames=(0,)):
            factors.append((r, blocks))
        drop_ids += [gw.bid(gw.RCV_DT, q) for q in range(4)] + [gw.bid(gw.RCV_DDT, 0)]
    for k in range(int(w["n_visual"]) if mode == 0 else 0):
        f, i, j = int(w["vis_feature"][k]), int(w["vis_i"][k]), int(w["vis_j"][k])
        if i != 0:
            continue
        r, J = visual_factor(pose[0], pose[j], ex, mpf(w["para_Feature"][f]), td, vec(w["vis_pts_i"][3 * k:3 * k + 3]), vec(w["vis_pts_j"][3 * k:3 * k + 3]),
                             vec(w["vis_vel_i"][2 * k:2 * k + 2]), vec(w["vis_vel_j"][2 * k:2 * k + 2]), mpf(w["vis_td_i"][k]), mpf(w["vis_td_j"][k]), si)
        _, rc, Jc = huber_correct(r, J)                                                         # marginalization_factor.cpp:49-77: the same corrector arithmetic
        factors.append((rc, [(gw.bid(gw.POSE, 0), Jc["pi"]), (gw.bid(gw.POSE, j), Jc["pj"]), (gw.bid(gw.EX_POSE), Jc["ex"]), (gw.bid(gw.FEATURE, f), Jc["f"]), (gw.bid(gw.TD), Jc["td"])]))
        if gw.bid(gw.FEATURE, f) not in drop_ids:
            drop_ids.append(gw.bid(gw.FEATURE, f))
    present = []
    for _, blocks in factors:
        for b, _ in blocks:
            if b not in present:
                present.append(b)
    dropped = [b for b in drop_ids if b in present]
    kept = [b for b in present if b not in dropped]
    col, pos = {}, 0
    for b in dropped + kept:
        col[b] = pos
        pos += gw.lsize(b // 4096)
    m = sum(gw.lsize(b // 4096) for b in dropped)
    n = pos - m
    A, bb = mp.zeros(pos, pos), mp.zeros(pos, 1)
    for r, blocks in factors:
        for ba, Ja in blocks:
            ga = Ja.T * r
            for q in range(Ja.cols):
                bb[col[ba] + q] += ga[q]
            for bc, Jc_ in blocks:
                Hab = Ja.T * Jc_
                for p_ in range(Ja.cols):
                    for q in range(Jc_.cols):
                        A[col[ba] + p_, col[bc] + q] += Hab[p_, q]
    eps = mp.mpf("1e-8")                                                                         # marginalization_factor.h:28
    Amm = (A[0:m, 0:m] + A[0:m, 0:m].T) / 2                                                      # :278
    ev, V = mp.eigsy(Amm)                                                                        # :279
    Amm_inv = mp.zeros(m, m)
    for k in range(m):
        if ev[k] > eps:                                                                          # :283
            vk = V[:, k]
            Amm_inv += vk * vk.T / ev[k]
    Arm, Amr = A[m:pos, 0:m], A[0:m, m:pos]
    Ar = A[m:pos, m:pos] - Arm * Amm_inv * Amr                                                   # :291
    br = bb[m:pos, 0] - Arm * Amm_inv * bb[0:m, 0]                                               # :292
    ev2, V2 = mp.eigsy((Ar + Ar.T) / 2)   # SelfAdjointEigenSolver reads the lower triangle of a matrix that is symmetric up to rounding: symmetrised here  (:294)
    JtJ, proj = mp.zeros(n, n), mp.zeros(n, n)
    Jlin, rlin = mp.zeros(n, n), mp.zeros(n, 1)                                                  # linearized_jacobians / linearized_residuals themselves (one valid basis)
    for k in range(n):
        if ev2[k] > eps:                                                                         # :295-296
            vk = V2[:, k]
            JtJ += vk * vk.T * ev2[k]                                                            # J = sqrt(S) V^T  (:301)  ->  J^T J = V S V^T
            proj += vk * vk.T                                                                    # r = S^-1/2 V^T b (:302)  ->  J^T r = V V^T b
            sq = mp.sqrt(ev2[k])
            for c in range(n):
                Jlin[k, c] = sq * vk[c]
            rlin[k] = (vk.T * br)[0] / sq
    Jtr = proj * br
    evs = sorted(float(x) for x in ev2)
    # linearisation point of the prior: the kept blocks' states (keep_block_data, :214-217), in kept order
    x0 = []
    for b in kept:
        kind, i = b // 4096, b % 4096
        gs = gw.gsize(kind)
        off = {gw.POSE: ("para_Pose", 7 * i), gw.SPEEDBIAS: ("para_SpeedBias", 9 * i), gw.EX_POSE: ("para_Ex_Pose", 0), gw.EX_WHEEL: ("para_Ex_Pose_wheel", 0),
               gw.SX: ("para_Ix", 0), gw.SY: ("para_Ix", 1), gw.SW: ("para_Ix", 2), gw.TD: ("para_Td", 0), gw.TD_WHEEL: ("para_Td_wheel", 0),
               gw.RCV_DT: ("para_rcv_dt", i), gw.RCV_DDT: ("para_rcv_ddt", i), gw.YAW: ("para_yaw_enu_local", 0), gw.ANC: ("para_anc_ecef", 0)}[kind]
        x0 += [float(v) for v in w[off[0]][off[1]:off[1] + gs]]
    return {"kept": kept, "m": m, "n": n, "JtJ": JtJ, "Jtr": Jtr, "J": Jlin, "r": rlin, "x0": x0, "eigenvalues_kept": evs, "eigenvalues_dropped": sorted(float(x) for x in ev)}


def first_step_golden(w, ids, H, g):
    """The first trust-region step at 60 digits: Jacobi scaling s = 1 / (1 + sqrt(diag H)) (trust_region_minimizer.cc), the dogleg's regularised Gauss-Newton step
    (s H s + mu D^2) y = s g with D^2 = clamp(s^2 diag H, 1e-6, 1e32), mu = 1e-8 (dogleg_strategy.cc: min_diagonal / max_diagonal, min_mu), taken whole when |D y| <= the
    initial radius 1e4, then x (+) delta, delta = -s o y, with PoseLocalParameterization::Plus (pose_local_parameterization.cpp:12-28) on the poses.  Solved by LU on
    the FULL system (no Schur complement, no Cholesky).  Returns None when the step is clipped by the radius (the dogleg would interpolate)."""
    import gfwindow as gw
    n = len(ids)
    s = [1 / (1 + mp.sqrt(H[c, c])) for c in range(n)]
    S = mp.matrix(n, n)
    for a in range(n):
        for c in range(n):
            S[a, c] = s[a] * H[a, c] * s[c]
    mu = mp.mpf("1e-8")
    D2 = [min(max(s[c] * s[c] * H[c, c], mp.mpf("1e-6")), mp.mpf("1e32")) for c in range(n)]
    for c in range(n):
        S[c, c] += mu * D2[c]
    y = mp.lu_solve(S, mp.matrix([s[c] * g[c] for c in range(n)]))
    if mp.sqrt(sum(D2[c] * y[c] * y[c] for c in range(n))) > mp.mpf("1e4"):
        return None
    delta = [-s[c] * y[c] for c in range(n)]
    col0 = {}
    for c, b in enumerate(ids):
        col0.setdefault(int(b), c)

    def plus(p7, c0):
        P, Q = pose_of(p7)
        q = qnormalized(qmul(Q, delta_q(mp.matrix(delta[c0 + 3:c0 + 6]))))
        return [float(P[0] + delta[c0]), float(P[1] + delta[c0 + 1]), float(P[2] + delta[c0 + 2]), float(q[1]), float(q[2]), float(q[3]), float(q[0])]
    out = {"para_Pose": [], "para_SpeedBias": []}
    for i in range(int(w["W"]) + 1):
        out["para_Pose"] += plus(w["para_Pose"][7 * i:7 * i + 7], col0[gw.bid(gw.POSE, i)])
        c0 = col0[gw.bid(gw.SPEEDBIAS, i)]
        out["para_SpeedBias"] += [float(mpf(w["para_SpeedBias"][9 * i + q]) + delta[c0 + q]) for q in range(9)]
    if gw.bid(gw.EX_POSE) in col0:
        out["para_Ex_Pose"] = plus(w["para_Ex_Pose"], col0[gw.bid(gw.EX_POSE)])
    if gw.bid(gw.EX_WHEEL) in col0:
        out["para_Ex_Pose_wheel"] = plus(w["para_Ex_Pose_wheel"], col0[gw.bid(gw.EX_WHEEL)])
    feat = [float(x) for x in w["para_Feature"]]
    for f in range(int(w["n_feature"])):
        if gw.bid(gw.FEATURE, f) in col0:
            feat[f] = float(mpf(w["para_Feature"][f]) + delta[col0[gw.bid(gw.FEATURE, f)]])
    out["para_Feature"] = feat
    return out


def to_list(a):
    return np.asarray(a).reshape(-1).tolist()


def main():
    import oracle_py as O     # used to SYNTHESISE the window (its pre-integration is input data) and to read the column order; its factor code is what gets compared
    import synth_window as SW
    import gfwindow as gw
    out_dir = HERE
    cases = [("ref_window_free_ex_td", dict(seed=7, max_features=8, n_landmarks=12, use_wheel=False, fix_ex_pose=0, fix_td=0), False),
             ("ref_window_with_prior", dict(seed=8, max_features=10, n_landmarks=15, use_wheel=False), True),
             ("ref_window_wheel", dict(seed=9, max_features=6, n_landmarks=9), False),                                            # the shipped configuration: wheel extrinsic free, intrinsics / td_wheel fixed
             ("ref_window_wheel_free_ix_td", dict(seed=10, max_features=6, n_landmarks=9, fix_ix=0, fix_td_wheel=0), False),    # every column of the wheel factor
             ("ref_window_gnss", dict(seed=11, max_features=6, n_landmarks=9, gnss=True), False)]                                # + 132 pseudorange / Doppler blocks, the clock factors
    for name, kw, with_prior in cases:
        seed = kw.pop("seed")
        w = SW.make_window(seed, O, **kw)
        if name == "ref_window_free_ex_td":
            w["para_Td"][0] = 0.004
        if name == "ref_window_wheel_free_ix_td":     # away from the linearisation point of the wheel pre-integration (sx = sy = sw = 1, td = 0): every correction term is live
            w["para_Ix"][:] = [1.013, 0.991, 1.007]
            w["para_Td_wheel"][0] = 0.006
        if with_prior:     # the prior is input data as well: the oracle's MARGIN_OLD of the window before, renamed to this window
            w0 = SW.make_window(seed, O, **kw)
            O.ba_solve(w0, 4)
            p0 = O.ba_marginalize(w0, 0)
            w = SW.make_window(seed, O, frame0=1, prior=p0, **kw)
        w.finalize()
        lin = O.ba_linearize(w.copy())
        ids = [int(x) for x in lin["ids"]]
        H, g, cost = window_normal_equations(w, ids)
        n = len(ids)
        Hd = np.array([[float(H[a, c]) for c in range(n)] for a in range(n)])
        gd = np.array([float(g[a]) for a in range(n)])
        hs = np.sqrt(np.outer(np.abs(np.diag(Hd)), np.abs(np.diag(Hd)))) + 1e-300
        print("%s: %d columns (%d eliminated), %d visual / %d IMU / %d wheel factors, prior %d; cost %.17g" % (name, n, lin["n_e"], w["n_visual"], w["n_imu"], w["n_wheel"], w["prior_n"], float(cost)))
        print("   oracle vs 60 digits: cost rel %.2e, H scaled %.2e, g rel %.2e" % (abs(lin["cost"] - float(cost)) / float(cost), np.abs((lin["H"] - Hd) / hs).max(),
                                                                                    np.abs(lin["g"] - gd).max() / np.abs(gd).max()))
        fx = {"about": "normal equations of a small sliding window from the reference's formulas at 60 digits (tests/golden/make_ref_golden.py); inputs = the window, "
                       "expected = H (its lower triangle, row by row), g, cost in the column order `ids` (block id = kind * 4096 + index, kinds as in gfwindow.py)",
              "window": {k: (to_list(v) if isinstance(v, np.ndarray) else v) for k, v in dict(w).items()},
              "ids": ids, "n_f": int(lin["n_f"]), "n_e": int(lin["n_e"]), "H_lower": [float(Hd[a, c]) for a in range(n) for c in range(a + 1)], "g": gd.tolist(), "cost": float(cost),
              "cost_30_digits": mp.nstr(cost, 30)}
        if name in ("ref_window_free_ex_td", "ref_window_wheel"):     # the state after the first trust-region step, when that step is the whole Gauss-Newton step
            st1 = first_step_golden(w, ids, H, g)
            if st1 is not None:
                a1 = w.copy()
                so = O.ba_solve(a1, 1)
                dev = max(np.abs(np.array(st1[k]) - a1[k]).max() for k in st1)
                print("   first step (exact, LU on the full system at 60 digits) vs the oracle's first iteration (successful steps %d): %.2e" % (so["successful_steps"], dev))
                fx["first_step"] = st1
        import gzip
        path = os.path.join(out_dir, name + ".json.gz")
        with gzip.GzipFile(path, "wb", mtime=0) as f:      # mtime 0: the same bytes on every run
            f.write(json.dumps(fx).encode())
        print("   wrote", path, os.path.getsize(path), "bytes")
    # ---- marginalisation priors: the window AFTER a solve (input data: the oracle's), MARGIN_OLD without and with a prior, MARGIN_SECOND_NEW
    import gzip
    kwm = dict(max_features=10, n_landmarks=15)
    w0 = SW.make_window(8, O, **kwm)
    O.ba_solve(w0, 4)
    p0 = O.ba_marginalize(w0, 0)
    w1 = SW.make_window(8, O, frame0=1, prior=p0, **kwm)
    O.ba_solve(w1, 4)
    wg = SW.make_window(12, O, gnss=True, **kwm)
    O.ba_solve(wg, 4)
    for name, w, mode in (("ref_marg_old_first_window", w0, 0), ("ref_marg_old_with_prior", w1, 0), ("ref_marg_second_new", w1, 1), ("ref_marg_old_gnss", wg, 0)):
        w.finalize()
        g = marg_golden(w, mode)
        n = g["n"]
        JtJ = np.array([[float(g["JtJ"][a, c]) for c in range(n)] for a in range(n)])
        Jtr = np.array([float(g["Jtr"][a]) for a in range(n)])
        po = O.ba_marginalize(w.copy(), mode)
        Jo = po["J"].reshape(n, n)
        sc = np.sqrt(np.maximum(np.diag(JtJ), 1e-300))
        print("%s: %d dropped + %d kept columns; kept eigenvalues around the 1e-8 cut: %s" % (name, g["m"], n, ["%.1e" % x for x in g["eigenvalues_kept"] if 1e-12 < abs(x) < 1e-4][:8]))
        print("   oracle vs 60 digits: J^T J scaled %.2e, J^T r rel %.2e" % (np.abs((Jo.T @ Jo - JtJ) / np.outer(sc, sc)).max(), np.abs(Jo.T @ po["r"] - Jtr).max() / np.abs(Jtr).max()))
        fx = {"about": "marginalisation prior of a small window by the reference's route (marginalization_factor.cpp:119-308; factor set estimator.cpp:3334-3560) at 60 digits "
                       "(tests/golden/make_ref_golden.py marg_golden): inputs = the window and the mode, expected = J^T J (lower triangle) and J^T r of the prior over the kept "
                       "blocks `kept` (ids BEFORE the address shift, first-appearance order)",
              "window": {k: (to_list(v) if isinstance(v, np.ndarray) else v) for k, v in dict(w).items()}, "mode": mode, "kept": [int(x) for x in g["kept"]], "m": int(g["m"]), "n": int(n),
              "JtJ_lower": [float(JtJ[a, c]) for a in range(n) for c in range(a + 1)], "Jtr": Jtr.tolist(), "eigenvalues_kept": g["eigenvalues_kept"]}
        path = os.path.join(out_dir, name + ".json.gz")
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(json.dumps(fx).encode())
        print("   wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
