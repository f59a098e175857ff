"""Golden normal equations of small windows, evaluated with 60 digits (mpmath) from the REFERENCE's formulas -- not from the oracle's or the library's code.

Why: the oracle (oracle/backend_oracle.cpp) and the HIP kernels were written by the same hand from the same reading of the reference; the reference cannot be
built here (no Eigen / Ceres / ROS) and ships no test vectors, so nothing tied either of them to numbers they did not produce themselves ("parity unpinned").
This script is a third, independent evaluation route: every formula below is transcribed from the cited reference lines into exact-enough arithmetic (60
significant digits, own quaternion / matrix algebra, no shared helper with the oracle or the product), and its output is committed as tests/golden/ref_*.json.
tests/test_golden.py then holds the oracle (CPU) and tests/test_backend_gpu.py the HIP path (through gf_ba_linearize) to these numbers at 1e-11.

What is evaluated (paths under /root/reference/vins_estimator/src):
  factor/projectionTwoFrameOneCamFactor.cpp:43-151   residual and the five Jacobian blocks of ProjectionTwoFrameOneCamFactor::Evaluate
  factor/imu_factor.h:28-191 + factor/integration_base.h:169-195   IMUFactor::Evaluate on top of IntegrationBase::evaluate, sqrt_info = LLT(cov^-1).matrixL()^T
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
                   gw.SX: ("para_Ix", 0), gw.SY: ("para_Ix", 1), gw.SW: ("para_Ix", 2), gw.TD: ("para_Td", 0), gw.TD_WHEEL: ("para_Td_wheel", 0)}[kind]
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
    # visual factors (estimator.cpp:3269-3297, loss_function = HuberLoss(1.0))
    for k in range(int(w["n_visual"])):
        f, i, j = int(w["vis_feature"][k]), int(w["vis_i"][k]), int(w["vis_j"][k])
        r, J = visual_factor(pose[i], pose[j], ex, mpf(w["para_Feature"][f]), td, vec(w["vis_pts_i"][3 * k:3 * k + 3]), vec(w["vis_pts_j"][3 * k:3 * k + 3]),
                             vec(w["vis_vel_i"][2 * k:2 * k + 2]), vec(w["vis_vel_j"][2 * k:2 * k + 2]), mpf(w["vis_td_i"][k]), mpf(w["vis_td_j"][k]), si)
        rho0, rc, Jc = huber_correct(r, J)
        cost += rho0 / 2
        add(rc, [(gw.bid(gw.POSE, i), Jc["pi"]), (gw.bid(gw.POSE, j), Jc["pj"]), (gw.bid(gw.EX_POSE), Jc["ex"]), (gw.bid(gw.FEATURE, f), Jc["f"]), (gw.bid(gw.TD), Jc["td"])])
    return H, g, cost


def to_list(a):
    return np.asarray(a).reshape(-1).tolist()


def main():
    import oracle_py as O     # used to SYNTHESISE the window (its pre-integration is input data) and to read the column order; its factor code is what gets compared
    import synth_window as SW
    import gfwindow as gw
    out_dir = HERE
    cases = [("ref_window_free_ex_td", dict(seed=7, max_features=8, n_landmarks=12, use_wheel=False, fix_ex_pose=0, fix_td=0), False),
             ("ref_window_with_prior", dict(seed=8, max_features=10, n_landmarks=15, use_wheel=False), True)]
    for name, kw, with_prior in cases:
        seed = kw.pop("seed")
        w = SW.make_window(seed, O, **kw)
        if name == "ref_window_free_ex_td":
            w["para_Td"][0] = 0.004
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
        print("%s: %d columns (%d eliminated), %d visual / %d IMU factors, prior %d; cost %.17g" % (name, n, lin["n_e"], w["n_visual"], w["n_imu"], w["prior_n"], float(cost)))
        print("   oracle vs 60 digits: cost rel %.2e, H scaled %.2e, g rel %.2e" % (abs(lin["cost"] - float(cost)) / float(cost), np.abs((lin["H"] - Hd) / hs).max(),
                                                                                    np.abs(lin["g"] - gd).max() / np.abs(gd).max()))
        fx = {"about": "normal equations of a small sliding window from the reference's formulas at 60 digits (tests/golden/make_ref_golden.py); inputs = the window, "
                       "expected = H (its lower triangle, row by row), g, cost in the column order `ids` (block id = kind * 4096 + index, kinds as in gfwindow.py)",
              "window": {k: (to_list(v) if isinstance(v, np.ndarray) else v) for k, v in dict(w).items()},
              "ids": ids, "n_f": int(lin["n_f"]), "n_e": int(lin["n_e"]), "H_lower": [float(Hd[a, c]) for a in range(n) for c in range(a + 1)], "g": gd.tolist(), "cost": float(cost),
              "cost_30_digits": mp.nstr(cost, 30)}
        with open(os.path.join(out_dir, name + ".json"), "w") as f:
            json.dump(fx, f)
        print("   wrote", os.path.join(out_dir, name + ".json"), os.path.getsize(os.path.join(out_dir, name + ".json")), "bytes")


if __name__ == "__main__":
    main()
