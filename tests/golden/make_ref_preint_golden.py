"""Row B2 at 60 digits: tests/golden/ref_preint.json.gz.

`IntegrationBase::push_back -> propagate -> midPointIntegration` (vins_estimator/src/factor/integration_base.h:41-166) and `WheelIntegrationBase::push_back -> propagate ->
midPointIntegration` (wheel_integration_base.h:42-179, with `Sophus::rightJacobianSO3` of sophus_utils.hpp:155-184) transcribed into mpmath, line by cited line, on the
quaternion / matrix algebra of make_ref_golden.py (nothing shared with oracle/ or the library).  Inputs: a few seeded sample streams (dt, accelerometer / gyroscope or wheel
velocity / gyroscope, the linearisation point); expected: delta_p, delta_q (w, x, y, z), delta_v, the 15 x 15 (6 x 3) Jacobian, the 15 x 15 (6 x 6) covariance, sum_dt.

  python tests/golden/make_ref_preint_golden.py            (seconds)"""
import gzip
import json
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_ref_golden as G  # noqa: E402

mp.mp.dps = 60
M, mpf, vec = G.M, G.mpf, G.vec


def setblock(A, r0, c0, B):
    for r in range(B.rows):
        for c in range(B.cols):
            A[r0 + r, c0 + c] = B[r, c]


def imu_preintegrate(dt, acc, gyr, acc0, gyr0, ba, bg, noise):
    noise4 = noise
    ACC_N, GYR_N, ACC_W, GYR_W = (mpf(x) for x in noise4)
    noise = mp.zeros(18, 18)                                              # integration_base.h:30-36
    for blk, v in enumerate((ACC_N, GYR_N, ACC_N, GYR_N, ACC_W, GYR_W)):
        for k in range(3):
            noise[3 * blk + k, 3 * blk + k] = v * v
    acc_0, gyr_0 = vec(acc0), vec(gyr0)
    lba, lbg = vec(ba), vec(bg)
    delta_p, delta_v, delta_q = mp.zeros(3, 1), mp.zeros(3, 1), (mpf(1), mpf(0), mpf(0), mpf(0))
    jac, cov, sum_dt = mp.eye(15), mp.zeros(15, 15), mpf(0)
    I3 = mp.eye(3)
    for k in range(len(dt)):
        _dt, acc_1, gyr_1 = mpf(dt[k]), vec(acc[k]), vec(gyr[k])
        # midPointIntegration, integration_base.h:73-84
        un_acc_0 = G.qrot(delta_q, acc_0 - lba)
        un_gyr = (gyr_0 + gyr_1) / 2 - lbg
        res_q = G.qmul(delta_q, (mpf(1), un_gyr[0] * _dt / 2, un_gyr[1] * _dt / 2, un_gyr[2] * _dt / 2))
        un_acc_1 = G.qrot(res_q, acc_1 - lba)
        un_acc = (un_acc_0 + un_acc_1) / 2
        res_p = delta_p + delta_v * _dt + un_acc * _dt * _dt / 2
        res_v = delta_v + un_acc * _dt
        # :86-140
        R_w_x, R_a_0_x, R_a_1_x = G.skew((gyr_0 + gyr_1) / 2 - lbg), G.skew(acc_0 - lba), G.skew(acc_1 - lba)
        Rq, Rr = G.qmat(delta_q), G.qmat(res_q)
        F = mp.zeros(15, 15)
        setblock(F, 0, 0, I3)
        setblock(F, 0, 3, -(Rq * R_a_0_x) * _dt * _dt / 4 - (Rr * R_a_1_x * (I3 - R_w_x * _dt)) * _dt * _dt / 4)
        setblock(F, 0, 6, I3 * _dt)
        setblock(F, 0, 9, -(Rq + Rr) * _dt * _dt / 4)
        setblock(F, 0, 12, -(Rr * R_a_1_x) * _dt * _dt * (-_dt) / 4)
        setblock(F, 3, 3, I3 - R_w_x * _dt)
        setblock(F, 3, 12, -I3 * _dt)
        setblock(F, 6, 3, -(Rq * R_a_0_x) * _dt / 2 - (Rr * R_a_1_x * (I3 - R_w_x * _dt)) * _dt / 2)
        setblock(F, 6, 6, I3)
        setblock(F, 6, 9, -(Rq + Rr) * _dt / 2)
        setblock(F, 6, 12, -(Rr * R_a_1_x) * _dt * (-_dt) / 2)
        setblock(F, 9, 9, I3)
        setblock(F, 12, 12, I3)
        V = mp.zeros(15, 18)
        V03 = -(Rr * R_a_1_x) * _dt * _dt * _dt / 8
        V63 = -(Rr * R_a_1_x) * _dt * _dt / 4
        setblock(V, 0, 0, Rq * _dt * _dt / 4); setblock(V, 0, 3, V03); setblock(V, 0, 6, Rr * _dt * _dt / 4); setblock(V, 0, 9, V03)
        setblock(V, 3, 3, I3 * _dt / 2); setblock(V, 3, 9, I3 * _dt / 2)
        setblock(V, 6, 0, Rq * _dt / 2); setblock(V, 6, 3, V63); setblock(V, 6, 6, Rr * _dt / 2); setblock(V, 6, 9, V63)
        setblock(V, 9, 12, I3 * _dt); setblock(V, 12, 15, I3 * _dt)
        jac = F * jac
        cov = F * cov * F.T + V * noise * V.T
        # propagate, :158-165
        delta_p, delta_v, delta_q = res_p, res_v, G.qnormalized(res_q)
        sum_dt += _dt
        acc_0, gyr_0 = acc_1, gyr_1
    return {"delta_p": delta_p, "delta_q": delta_q, "delta_v": delta_v, "jacobian": jac, "covariance": cov, "sum_dt": sum_dt}


def wheel_preintegrate(dt, vel, gyr, vel0, gyr0, lin, noise):
    noise2 = noise
    VEL_N, GYR_N = (mpf(x) for x in noise2)
    noise = mp.zeros(12, 12)                                              # wheel_integration_base.h:32-36
    for blk, v in enumerate((VEL_N, GYR_N, VEL_N, GYR_N)):
        for k in range(3):
            noise[3 * blk + k, 3 * blk + k] = v * v
    sx, sy, sw = (mpf(x) for x in lin[:3])
    sv = mp.diag([sx, sy, mpf(1)])
    vel_0, gyr_0 = vec(vel0), vec(gyr0)
    delta_p, delta_q = mp.zeros(3, 1), (mpf(1), mpf(0), mpf(0), mpf(0))
    jac, cov, sum_dt = mp.zeros(6, 3), mp.zeros(6, 6), mpf(0)
    I1, I2 = mp.diag([1, 0, 0]), mp.diag([0, 1, 0])
    for k in range(len(dt)):
        _dt, vel_1, gyr_1 = mpf(dt[k]), vec(vel[k]), vec(gyr[k])
        # midPointIntegration, wheel_integration_base.h:78-86
        un_vel_0 = G.qrot(delta_q, sv * vel_0)
        un_gyr = (gyr_0 + gyr_1) * sw / 2
        ddq = (mpf(1), un_gyr[0] * _dt / 2, un_gyr[1] * _dt / 2, un_gyr[2] * _dt / 2)
        res_q = G.qmul(delta_q, ddq)
        un_vel_1 = G.qrot(res_q, sv * vel_1)
        res_p = delta_p + (un_vel_0 + un_vel_1) / 2 * _dt
        # :94-141
        R_vel_0_x, R_vel_1_x = G.skew(sv * vel_0), G.skew(sv * vel_1)
        Rq, Rr, Rdd = G.qmat(delta_q), G.qmat(res_q), G.qmat(ddq)
        F = mp.zeros(6, 6)
        setblock(F, 0, 0, mp.eye(3))
        setblock(F, 0, 3, -(Rq * R_vel_0_x + Rr * R_vel_1_x * Rdd.T) * _dt / 2)
        setblock(F, 3, 3, Rdd.T)
        Jr = G.right_jacobian_so3(un_gyr * _dt)
        V = mp.zeros(6, 12)
        V03 = -(Rr * R_vel_1_x * Jr) * _dt * _dt / 4
        setblock(V, 0, 0, Rq * sv * _dt / 2); setblock(V, 0, 3, V03); setblock(V, 0, 6, Rr * sv * _dt / 2); setblock(V, 0, 9, V03)
        setblock(V, 3, 3, Jr * sw * _dt / 2); setblock(V, 3, 9, Jr * sw * _dt / 2)
        j00 = jac[0:3, 0] + (Rq * I1 * vel_0 + Rr * I1 * vel_1) * _dt / 2
        j01 = jac[0:3, 1] + (Rq * I2 * vel_0 + Rr * I2 * vel_1) * _dt / 2
        dr_dsw_last = jac[3:6, 2]
        j32 = jac[3:6, 2] + Jr * (gyr_0 + gyr_1) / 2 * _dt
        j02 = jac[0:3, 2] + (Rq * G.skew(dr_dsw_last) * sv * vel_0 + Rr * G.skew(j32) * sv * vel_1) * _dt / 2
        setblock(jac, 0, 0, j00); setblock(jac, 0, 1, j01); setblock(jac, 3, 2, j32); setblock(jac, 0, 2, j02)
        cov = F * cov * F.T + V * noise * V.T
        # propagate, :168-177
        delta_p, delta_q = res_p, G.qnormalized(res_q)
        sum_dt += _dt
        vel_0, gyr_0 = vel_1, gyr_1
    return {"delta_p": delta_p, "delta_q": delta_q, "jacobian": jac, "covariance": cov, "sum_dt": sum_dt}


def flat(x):
    if isinstance(x, tuple):
        return [float(v) for v in x]
    if isinstance(x, mp.matrix):
        return [float(x[r, c]) for r in range(x.rows) for c in range(x.cols)]
    return float(x)


def main():
    rng = np.random.default_rng(20260926)
    cases = {"imu": [], "wheel": []}
    for n, rate in ((7, 100.0), (14, 200.0), (40, 400.0)):       # one camera interval at three IMU rates (the last one 0.1 s: the longest the estimator integrates)
        t = np.arange(n + 1) / rate
        dt = np.diff(t) * (1 + rng.normal(0, 1e-3, n))
        acc = np.stack([0.8 * np.sin(3 * t + 0.3), 0.5 * np.cos(2 * t), 9.805 + 0.3 * np.sin(5 * t)], 1) + rng.normal(0, 0.02, (n + 1, 3))
        gyr = np.stack([0.05 * np.sin(4 * t), 0.08 * np.cos(3 * t), 0.6 + 0.2 * np.sin(2 * t)], 1) + rng.normal(0, 0.002, (n + 1, 3))
        inp = {"dt": dt.tolist(), "acc": acc[1:].tolist(), "gyr": gyr[1:].tolist(), "acc0": acc[0].tolist(), "gyr0": gyr[0].tolist(),
               "ba": rng.normal(0, 0.02, 3).tolist(), "bg": rng.normal(0, 0.002, 3).tolist(), "noise": [0.1, 0.01, 0.001, 0.0001]}
        out = imu_preintegrate(**inp)
        cases["imu"].append({"input": inp, "expected": {k: flat(v) for k, v in out.items()}})
    for n, lin in ((4, [1.0, 1.0, 1.0]), (5, [1.013, 0.991, 1.007]), (10, [0.97, 1.02, 0.985])):
        t = np.arange(n + 1) / 50.0
        dt = np.diff(t) * (1 + rng.normal(0, 1e-3, n))
        vel = np.stack([1.5 + 0.4 * np.sin(2 * t), 0.05 * np.cos(3 * t), 0.01 * np.sin(t)], 1) + rng.normal(0, 0.005, (n + 1, 3))
        gyr = np.stack([0.01 * np.sin(4 * t), 0.02 * np.cos(3 * t), 0.5 + 0.2 * np.sin(2 * t)], 1) + rng.normal(0, 0.0005, (n + 1, 3))
        inp = {"dt": dt.tolist(), "vel": vel[1:].tolist(), "gyr": gyr[1:].tolist(), "vel0": vel[0].tolist(), "gyr0": gyr[0].tolist(), "lin": lin, "noise": [0.1, 0.01]}
        out = wheel_preintegrate(**inp)
        cases["wheel"].append({"input": inp, "expected": {k: flat(v) for k, v in out.items()}})
    fx = {"about": "IMU and wheel pre-integration of seeded sample streams from the reference's formulas at 60 digits (tests/golden/make_ref_preint_golden.py); quaternions w, x, y, z; "
                   "matrices row by row", **cases}
    path = os.path.join(HERE, "ref_preint.json.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps(fx).encode())
    print("wrote", path, os.path.getsize(path), "bytes")
    # the oracle next to it
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
    import oracle_py as O
    for kind, fn in (("imu", O.imu_preintegrate), ("wheel", O.wheel_preintegrate)):
        for c in cases[kind]:
            got = fn(**c["input"])
            dev = {k: float(np.abs(np.asarray(got[k]).reshape(-1) - np.asarray(v).reshape(-1)).max() / max(1e-300, np.abs(np.asarray(v)).max())) for k, v in c["expected"].items()}
            print(kind, len(c["input"]["dt"]), "samples: oracle vs 60 digits (relative to the largest entry):", {k: "%.1e" % v for k, v in dev.items()})


if __name__ == "__main__":
    main()
