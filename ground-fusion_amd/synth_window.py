"""Seeded synthetic sliding windows: planar ground-vehicle trajectory, 15 Hz frames, 200 Hz IMU, 50 Hz
wheel odometer, landmarks inside and beyond depth_threshold (3 m) so both depth-fixed (estimate_flag 1) and free inverse
depths occur.  Noise levels from config/realsense/m2dgrp.yaml:144-153.  `pre` supplies the pre-integration
(imu_preintegrate / wheel_preintegrate): the product's host implementation in bench/product use, the oracle's in oracle tests."""
import numpy as np
from gfwindow import Window

ACC_N, GYR_N, ACC_W, GYR_W = 0.1, 0.01, 0.001, 0.0001      # m2dgrp.yaml:144-147
VEL_N_WHEEL, GYR_N_WHEEL = 0.1, 0.01                        # :151-152 (order of magnitude)
G = np.array([0.0, 0.0, 9.805])
FOCAL_LENGTH = 600.0
R_IC = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])  # camera z forward = body x
T_IC = np.array([0.05, 0.0, 0.10])
FX, FY, CX, CY = 603.95556640625, 603.1257934570312, 324.0858154296875, 232.72303771972656


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


def quat_from_R(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        w, x, y, z = 0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        v = [0, 0, 0]
        v[i] = 0.25 * s
        w = (R[k, j] - R[j, k]) / s
        v[j] = (R[j, i] + R[i, j]) / s
        v[k] = (R[k, i] + R[i, k]) / s
        x, y, z = v
    q = np.array([x, y, z, w])
    return q / np.linalg.norm(q)


def small_rot(v):
    th = np.linalg.norm(v)
    if th < 1e-12:
        return np.eye(3)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


class Trajectory:
    def __init__(self, seed, T=3.0, v=1.0):
        rng = np.random.default_rng(seed)
        self.v = v
        self.w_amp, self.w_f, self.ph = rng.uniform(0.15, 0.3), rng.uniform(0.5, 1.2), rng.uniform(0, 6.28)
        self.t = np.arange(0, T + 1e-9, 1e-4)
        psi = -self.w_amp / self.w_f * (np.cos(self.w_f * self.t + self.ph) - np.cos(self.ph))
        vx, vy = v * np.cos(psi), v * np.sin(psi)
        self.px = np.concatenate([[0], np.cumsum(0.5 * (vx[1:] + vx[:-1]) * 1e-4)])
        self.py = np.concatenate([[0], np.cumsum(0.5 * (vy[1:] + vy[:-1]) * 1e-4)])
        self.psi = psi

    def at(self, t):
        psi = np.interp(t, self.t, self.psi)
        p = np.array([np.interp(t, self.t, self.px), np.interp(t, self.t, self.py), 0.0])
        w = self.w_amp * np.sin(self.w_f * t + self.ph)
        vel = self.v * np.array([np.cos(psi), np.sin(psi), 0.0])
        acc = self.v * w * np.array([-np.sin(psi), np.cos(psi), 0.0])
        return p, rot_z(psi), vel, acc, np.array([0, 0, w])


def make_window(seed, pre, W=10, n_landmarks=220, max_features=150, use_wheel=True, frame0=0, perturb=True, prior=None,
                fix_ex_pose=1, fix_ex_wheel=0, fix_ix=1, fix_td=1, fix_td_wheel=1, gnss=False, gnss_lowspeed=0, anchor=False):
    """Window over frames frame0 .. frame0+W of trajectory `seed`."""
    rng = np.random.default_rng(seed * 7919 + 13 + frame0)
    traj = Trajectory(seed, T=(frame0 + W + 3) / 15.0 + 0.5)
    dt_f = 1.0 / 15.0
    times = [(frame0 + k) * dt_f + 0.2 for k in range(W + 1)]
    brng = np.random.default_rng(seed * 31 + 5)
    ba, bg = brng.normal(0, 0.02, 3), brng.normal(0, 0.002, 3)
    R_io, t_io = small_rot(np.array([0.0, 0.0, 0.01])), np.array([0.1, 0.0, -0.05])
    # landmarks are fixed in the world for a given seed (independent of frame0)
    lrng = np.random.default_rng(seed * 104729 + 7)
    ahead = lrng.uniform(1.2, 9.0, n_landmarks)
    side = lrng.uniform(-0.55, 0.55, n_landmarks) * ahead
    up = lrng.uniform(-0.35, 0.35, n_landmarks) * ahead
    p_mid, R_mid, _, _, _ = traj.at(0.2 + 0.5 * W * dt_f)
    land = (R_mid @ np.stack([ahead, side, up + 0.1], 0)).T + p_mid
    w = Window()
    w["W"] = W
    Ps, Rs, Vs = [], [], []
    for t in times:
        p, R, v, _, _ = traj.at(t)
        Ps.append(p); Rs.append(R); Vs.append(v)
    # ---- IMU / wheel pre-integration between consecutive frames
    imu = {k: [] for k in ("i", "sum_dt", "delta_p", "delta_q", "delta_v", "lin_ba", "lin_bg", "jacobian", "covariance")}
    wh = {k: [] for k in ("i", "sum_dt", "delta_p", "delta_q", "jacobian", "covariance", "lin", "lin_vel", "lin_gyr", "vel_1", "gyr_1")}
    lin_ba, lin_bg = ba + brng.normal(0, 0.005, 3), bg + brng.normal(0, 0.0005, 3)
    for k in range(W):
        ts = np.arange(times[k], times[k + 1] + 1e-9, 0.005)
        acc, gyr = [], []
        for t in ts:
            _, R, _, a, om = traj.at(t)
            acc.append(R.T @ (a + G) + ba + rng.normal(0, ACC_N * 0.1, 3))
            gyr.append(om + bg + rng.normal(0, GYR_N * 0.1, 3))
        acc, gyr = np.array(acc), np.array(gyr)
        dts = np.diff(ts)
        r = pre.imu_preintegrate(dts, acc[1:], gyr[1:], acc[0], gyr[0], lin_ba, lin_bg, [ACC_N, GYR_N, ACC_W, GYR_W])
        imu["i"].append(k); imu["lin_ba"].append(lin_ba); imu["lin_bg"].append(lin_bg)
        for key in ("sum_dt", "delta_p", "delta_q", "delta_v", "jacobian", "covariance"):
            imu[key].append(r[key])
        if use_wheel:
            tw = np.arange(times[k], times[k + 1] + 1e-9, 0.02)
            if tw[-1] < times[k + 1] - 1e-9:
                tw = np.append(tw, times[k + 1])
            vel, wg = [], []
            for t in tw:
                _, R, v, _, om = traj.at(t)
                v_o = R_io.T @ (R.T @ v + np.cross(om, t_io))
                vel.append(v_o + rng.normal(0, VEL_N_WHEEL * 0.05, 3)); wg.append(R_io.T @ om + rng.normal(0, GYR_N_WHEEL * 0.05, 3))
            vel, wg = np.array(vel), np.array(wg)
            r = pre.wheel_preintegrate(np.diff(tw), vel[1:], wg[1:], vel[0], wg[0], [1.0, 1.0, 1.0], [VEL_N_WHEEL, GYR_N_WHEEL])
            wh["i"].append(k); wh["lin"].append([1.0, 1.0, 1.0, 0.0]); wh["lin_vel"].append(vel[0]); wh["lin_gyr"].append(wg[0]); wh["vel_1"].append(vel[-1]); wh["gyr_1"].append(wg[-1])
            for key in ("sum_dt", "delta_p", "delta_q", "jacobian", "covariance"):
                wh[key].append(r[key])
    for key, v in imu.items():
        w["imu_" + key] = np.array(v)
    for key, v in wh.items():
        w["wh_" + key] = np.array(v)
    # ---- visual observations
    orng = np.random.default_rng(seed * 977 + 3)
    noise_uv = orng.normal(0, 0.3 / FOCAL_LENGTH, (n_landmarks, 64, 2))  # per (landmark, absolute frame) so windows of one seed agree
    obs = {}
    for k in range(W + 1):
        Rc = Rs[k] @ R_IC
        pc = Ps[k] + Rs[k] @ T_IC
        Xc = (land - pc) @ Rc
        z = Xc[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            u, v = FX * Xc[:, 0] / z + CX, FY * Xc[:, 1] / z + CY
        vis = (z > 0.3) & (u > 5) & (u < 635) & (v > 5) & (v < 475)
        for l in np.nonzero(vis)[0]:
            xn = np.array([Xc[l, 0] / z[l], Xc[l, 1] / z[l]]) + noise_uv[l, (frame0 + k) % 64]
            obs.setdefault(l, []).append((k, xn, z[l]))
    feats = []
    for l, ob in obs.items():
        frames = [o[0] for o in ob]
        if len(ob) >= 4 and frames == list(range(frames[0], frames[0] + len(frames))):
            feats.append((l, ob))
    feats.sort(key=lambda f: (f[1][0][0], f[0]))
    feats = feats[:max_features]
    vf, vi, vj, pi_, pj_, vli, vlj, inv_dep, fixed = [], [], [], [], [], [], [], [], []
    for fi, (l, ob) in enumerate(feats):
        k0, x0, z0 = ob[0]
        vel = [np.zeros(2)] + [(ob[q][1] - ob[q - 1][1]) / dt_f for q in range(1, len(ob))]
        for q in range(1, len(ob)):
            vf.append(fi); vi.append(k0); vj.append(ob[q][0])
            pi_.append([x0[0], x0[1], 1.0]); pj_.append([ob[q][1][0], ob[q][1][1], 1.0]); vli.append(vel[0]); vlj.append(vel[q])
        inv_dep.append(1.0 / (z0 * (1.0 + (rng.normal(0, 0.05) if perturb else 0.0))))
        fixed.append(1 if z0 < 3.0 else 0)
    w["vis_feature"], w["vis_i"], w["vis_j"] = vf, vi, vj
    w["vis_pts_i"], w["vis_pts_j"], w["vis_vel_i"], w["vis_vel_j"] = np.array(pi_), np.array(pj_), np.array(vli), np.array(vlj)
    w["vis_td_i"], w["vis_td_j"] = np.zeros(len(vf)), np.zeros(len(vf))
    w["vis_sqrt_info"] = FOCAL_LENGTH / 1.5
    w["para_Feature"], w["feature_fixed"] = np.array(inv_dep), np.array(fixed, np.uint8)
    # ---- states (truth + perturbation)
    pose, sb = [], []
    for k in range(W + 1):
        dp = rng.normal(0, 0.02, 3) if (perturb and k > 0) else np.zeros(3)
        dr = rng.normal(0, 0.01, 3) if (perturb and k > 0) else np.zeros(3)
        q = quat_from_R(Rs[k] @ small_rot(dr))
        pose.append(np.concatenate([Ps[k] + dp, q]))
        sb.append(np.concatenate([Vs[k] + (rng.normal(0, 0.05, 3) if perturb else 0), lin_ba, lin_bg]))
    w["para_Pose"], w["para_SpeedBias"] = np.array(pose), np.array(sb)
    w["para_Ex_Pose"] = np.concatenate([T_IC, quat_from_R(R_IC)])
    w["para_Ex_Pose_wheel"] = np.concatenate([t_io, quat_from_R(R_io)])
    w["para_Ix"], w["para_Td"], w["para_Td_wheel"] = np.ones(3), np.zeros(1), np.zeros(1)
    w["fix_ex_pose"], w["fix_ex_wheel"], w["fix_ix"], w["fix_td"], w["fix_td_wheel"], w["fix_poses"] = fix_ex_pose, fix_ex_wheel, fix_ix, fix_td, fix_td_wheel, 0
    w["G"] = G
    if gnss:
        add_gnss(w, seed, traj, times, frame0, perturb, gnss_lowspeed)
    if anchor:
        w["has_anchor"] = 1
        w["anchor_value"] = np.concatenate([Ps[0], quat_from_R(Rs[0])])
    w.set_prior(prior)
    return w


# ---------------------------------------------------------------- synthetic GNSS (config 5): measurement model in numpy, used only to make data
C_LIGHT, OMG_E, WGS_A, WGS_E2 = 2.99792458e8, 7.2921151467e-5, 6378137.0, 6.69437999014e-3
IONO = np.array([0.1118e-07, 0.2235e-07, -0.4172e-06, 0.6557e-06, 0.1249e+06, -0.4424e+06, 0.1507e+07, -0.2621e+06])   # m2dgrp.yaml gnss_iono_default_parameters


def geo2ecef(lat_deg, lon_deg, alt):
    lat, lon = np.radians(lat_deg), np.radians(lon_deg)
    N = WGS_A / np.sqrt(1 - WGS_E2 * np.sin(lat) ** 2)
    return np.array([(N + alt) * np.cos(lat) * np.cos(lon), (N + alt) * np.cos(lat) * np.sin(lon), (N * (1 - WGS_E2) + alt) * np.sin(lat)])


def R_ecef_enu(lat_deg, lon_deg):
    lat, lon = np.radians(lat_deg), np.radians(lon_deg)
    return np.array([[-np.sin(lon), -np.sin(lat) * np.cos(lon), np.cos(lat) * np.cos(lon)], [np.cos(lon), -np.sin(lat) * np.sin(lon), np.cos(lat) * np.sin(lon)],
                     [0, np.cos(lat), np.sin(lat)]])


def add_gnss(w, seed, traj, times, frame0, perturb, lowspeed, sats_per_sys=3, lat=31.03, lon=121.44, alt=20.0, yaw=0.3):
    """4 constellations x sats_per_sys satellites above 30 deg elevation, one pseudorange + Doppler pair per satellite and frame, observation epochs
    a few ms off the frame stamps (so the interpolation ratio of estimator.cpp:3187-3198 is exercised).  Delays (iono / tropo) are left out of
    the synthetic pseudoranges: they only shift the residuals by a few metres, which the clock states absorb."""
    W = w["W"]
    rng = np.random.default_rng(seed * 6007 + 11)      # constellation: fixed per seed
    orng = np.random.default_rng(seed * 6011 + 17 + frame0)
    anc = geo2ecef(lat, lon, alt)
    Re = R_ecef_enu(lat, lon)
    Ry = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1.0]])
    R_el = Re @ Ry
    sats = []
    for sys in range(4):
        for _ in range(sats_per_sys):
            az, el, rg = rng.uniform(0, 2 * np.pi), rng.uniform(np.radians(32), np.radians(80)), rng.uniform(2.0e7, 2.5e7)
            d_enu = np.array([np.sin(az) * np.cos(el), np.cos(az) * np.cos(el), np.sin(el)])
            pos = anc + Re @ (d_enu * rg)
            vel = Re @ (np.cross(d_enu, rng.normal(0, 1, 3)) * 1.5e3)
            sats.append((sys, pos, vel, rng.uniform(-2e-4, 2e-4), rng.uniform(-1e-11, 1e-11), rng.uniform(-8e-9, 8e-9)))
    dt_true = np.array([150.0, 180.0, 120.0, 200.0])   # receiver clock bias per constellation [m]
    ddt_true = 2.0                                       # drift [m/s]
    frames, lowers, syss, ratios, data = [], [], [], [], []
    wl = C_LIGHT / 1575.42e6
    for i in range(W + 1):
        for (sys, pos, vel, svdt, svddt, tgd) in sats:
            ts = times[i] + orng.uniform(-0.02, 0.02)
            if times[i] > ts:
                lower = 0 if i == 0 else i - 1
            else:
                lower = W - 1 if i == W else i
            ratio = (times[lower + 1] - ts) / (times[lower + 1] - times[lower])
            p, R, v, _, _ = traj.at(min(max(ts, traj.t[0]), traj.t[-1]))
            sp = pos + vel * (ts - times[0])
            P_e, V_e = R_el @ p + anc, R_el @ v
            los = sp - P_e
            rg = np.linalg.norm(los)
            unit = los / rg
            clk = dt_true[sys] + ddt_true * (ts - times[0] + frame0 / 15.0)
            psr = rg + OMG_E * (sp[0] * P_e[1] - sp[1] * P_e[0]) / C_LIGHT + clk - svdt * C_LIGHT + tgd * C_LIGHT + orng.normal(0, 0.5)
            dop_est = (vel - V_e) @ unit + OMG_E / C_LIGHT * (vel[0] * P_e[1] + sp[0] * V_e[1] - vel[1] * P_e[0] - sp[1] * V_e[0]) + ddt_true - svddt * C_LIGHT
            dopp = -(dop_est + orng.normal(0, 0.05)) / wl
            frames.append(i); lowers.append(lower); syss.append(sys); ratios.append(ratio)
            data.append([*sp, *vel, svdt, svddt, tgd, 2.0, 2.0, psr, dopp, wl, 345600.0 + ts, 0.0])
    w["gnss_enabled"], w["gnss_lowspeed"] = 1, lowspeed
    w["gnss_frame"], w["gnss_lower"], w["gnss_sys"], w["gnss_ratio"], w["gnss_data"] = frames, lowers, syss, np.array(ratios), np.array(data)
    w["gnss_iono"], w["gnss_headers"], w["gnss_ddt_weight"] = IONO.copy(), np.array(times), 10.0
    prng = np.random.default_rng(seed * 6029 + 5 + frame0)
    clk0 = np.array([[dt_true[q] + ddt_true * (times[i] - times[0] + frame0 / 15.0) for q in range(4)] for i in range(W + 1)])
    w["para_rcv_dt"] = clk0 + (prng.normal(0, 3.0, clk0.shape) if perturb else 0.0)
    w["para_rcv_ddt"] = np.full(W + 1, ddt_true) + (prng.normal(0, 0.2, W + 1) if perturb else 0.0)
    w["para_yaw_enu_local"] = np.array([yaw])
    w["para_anc_ecef"] = anc + (prng.normal(0, 0.5, 3) if perturb else 0.0)
    return w
