"""Seeded synthetic RGB-D + IMU + wheel stream of one ground-vehicle sequence: stationary lead-in (exercises the stationary
initialisation, estimator.cpp:1604-1650), then forward motion with a yaw-rate sinusoid in front of the two-plane room of synth.render_room
(near wall inside depth_threshold, far wall beyond it).  Camera 30 Hz (the back end takes every 2nd frame when multiple_thread=1), IMU 200 Hz,
wheel odometer 50 Hz.  Body frame = camera frame (x right, y down, z forward; body_T_cam0 = I as in config/realsense/m2dgrp.yaml:75-82);
the wheel frame is taken parallel to the body frame with a small lever arm.  Pure numpy."""
import numpy as np

import synth

G_NORM = 9.805
R0 = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])  # body axes in the z-up world at zero yaw: z_b -> x_w, x_b -> -y_w, y_b -> -z_w
TIO = np.array([0.02, 0.10, -0.05])
RIO = np.eye(3)


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


class Stream:
    def __init__(self, seed, t_still=1.5, t_move=3.0, v_max=0.4, imu_hz=200.0, wheel_hz=50.0, cam_hz=30.0, noise=True, near_z=2.5, far_z=6.0, yaw0=0.0, yaw_turn=0.0, split_x=0.6, turn_delay=0.0, slow_tail=0.0, v_tail=0.15, v_start=0.0):
        rng = np.random.default_rng(2000 + seed)
        self.seed, self.near_z, self.far_z, self.split_x = seed, near_z, far_z, split_x
        T = t_still + t_move
        h = 1e-4
        t = np.arange(0, T + 0.2, h)
        s = np.clip((t - t_still) / 0.8, 0, 1)
        speed = v_start + (v_max - v_start) * (3 * s ** 2 - 2 * s ** 3)             # smooth ramp (v_start > 0 with t_still = 0: the recording begins in motion)
        if slow_tail > 0.0:                                                         # ... and down to v_tail over the last slow_tail seconds (crawling, not stopping)
            e = np.clip((t - (T - slow_tail)) / (0.6 * slow_tail), 0, 1)
            speed = speed - (v_max - v_tail) * (3 * e ** 2 - 2 * e ** 3) * (t > t_still + 0.8)
        amp, f, ph = rng.uniform(0.10, 0.25), rng.uniform(0.8, 1.6), rng.uniform(0, 6.28)
        yaw_rate = np.where(t > t_still, amp * np.sin(f * (t - t_still) + ph) * (3 * s ** 2 - 2 * s ** 3), 0.0)
        if yaw_turn != 0.0:   # plus a smooth turn by yaw_turn [rad] spread over the motion phase
            t_on, t_len = t_still + turn_delay, max(t_move - turn_delay, 1e-9)
            u = np.clip((t - t_on) / t_len, 0, 1)
            yaw_rate = yaw_rate + yaw_turn * 6 * u * (1 - u) / t_len * (t > t_on)
        psi = yaw0 + np.concatenate([[0], np.cumsum(0.5 * (yaw_rate[1:] + yaw_rate[:-1]) * h)])
        vw = np.stack([speed * np.cos(psi), speed * np.sin(psi), np.zeros_like(t)], 1)
        pw = np.concatenate([np.zeros((1, 3)), np.cumsum(0.5 * (vw[1:] + vw[:-1]) * h, 0)])
        aw = np.gradient(vw, h, axis=0)
        self._t, self._psi, self._pw, self._vw, self._aw, self._wz = t, psi, pw, vw, aw, yaw_rate
        self.T = T
        self.ba = rng.normal(0, 0.02, 3) if noise else np.zeros(3)
        self.bg = rng.normal(0, 0.002, 3) if noise else np.zeros(3)
        # IMU (acc = R^T (a_w + g), gyr = R^T w_w), m2dgrp.yaml noise levels
        ti = np.arange(0, T + 0.1, 1.0 / imu_hz)
        self.imu_t = ti
        acc, gyr = np.zeros((len(ti), 3)), np.zeros((len(ti), 3))
        for k, tk in enumerate(ti):
            R = self.R_wb(tk)
            acc[k] = R.T @ (self._at(self._aw, tk) + np.array([0, 0, G_NORM])) + self.ba
            gyr[k] = R.T @ np.array([0, 0, self._at(self._wz, tk)]) + self.bg
        if noise:
            acc += rng.normal(0, 1.2374e-2, acc.shape)
            gyr += rng.normal(0, 3.0033e-3, gyr.shape)
        self.imu_acc, self.imu_gyr = acc, gyr
        # wheel odometer in the wheel frame: v_o = R_io^T (R^T v_w + w_b x t_io)
        tw = np.arange(0, T + 0.1, 1.0 / wheel_hz)
        self.wheel_t = tw
        vel, wg = np.zeros((len(tw), 3)), np.zeros((len(tw), 3))
        for k, tk in enumerate(tw):
            R = self.R_wb(tk)
            wb = R.T @ np.array([0, 0, self._at(self._wz, tk)])
            vel[k] = RIO.T @ (R.T @ self._at(self._vw, tk) + np.cross(wb, TIO))
            wg[k] = RIO.T @ wb
        if noise:
            moving = (np.linalg.norm(vel, axis=1) > 1e-9)[:, None]
            vel += rng.normal(0, 0.01, vel.shape) * moving     # an odometer at rest reports exact zeros
            wg += rng.normal(0, 0.004, wg.shape) * moving
        self.wheel_vel, self.wheel_gyr = vel, wg
        self.cam_t = np.arange(0.05, T, 1.0 / cam_hz)
        self._tex = None

    def _at(self, arr, tk):
        i = min(int(round(tk / 1e-4)), len(self._t) - 1)
        return arr[i]

    def R_wb(self, tk):
        return rot_z(self._at(self._psi, tk)) @ R0

    def p_wb(self, tk):
        return self._at(self._pw, tk)

    def image(self, k):
        """(gray u8, depth u16 mm) of camera frame k"""
        if self._tex is None:
            self._tex = synth.make_texture(1000 + self.seed)
        tk = self.cam_t[k]
        R_rc = R0.T @ self.R_wb(tk)          # camera -> render world (x right, y down, z forward)
        t_rc = R0.T @ self.p_wb(tk)
        return synth.render_room(self._tex, R_rc, t_rc, near_z=self.near_z, far_z=self.far_z, split_x=self.split_x)

    def feed(self, est, k, t_prev):
        """push IMU / wheel samples with t_prev < t <= cam_t[k] + one sample of slack, as a ROS callback order would"""
        t1 = self.cam_t[k] + 0.03
        for i in np.nonzero((self.imu_t > t_prev) & (self.imu_t <= t1))[0]:
            est.inputIMU(float(self.imu_t[i]), self.imu_acc[i], self.imu_gyr[i])
        for i in np.nonzero((self.wheel_t > t_prev) & (self.wheel_t <= t1))[0]:
            est.inputWheel(float(self.wheel_t[i]), self.wheel_vel[i], self.wheel_gyr[i])
        return t1

    # ---- feature frames without images (host-logic tests): project a seeded landmark cloud standing on the two walls
    def _landmarks(self, n=400):
        rng = np.random.default_rng(3000 + self.seed)
        x = rng.uniform(-4.0, 6.0, n)          # render-world x (right)
        y = rng.uniform(-2.0, 2.0, n)          # down
        z = np.where(x < self.split_x, self.near_z, self.far_z)
        return np.stack([x, y, z], 1) @ R0.T   # to the z-up world (render = R0^T world)

    def feature_frame(self, k, pixel_noise=0.2):
        """{id: (x_n, y_n, 1, u, v, vx, vy, depth)} of camera frame k, the layout FeatureTracker::trackImage returns (feature_tracker.cpp:344-368)"""
        if not hasattr(self, "_lm"):
            self._lm = self._landmarks()
            self._pn = np.random.default_rng(4000 + self.seed).normal(0, 1.0, (len(self.cam_t), len(self._lm), 2))
        out = {}
        cur = self._project(k, pixel_noise)
        prev = self._project(k - 1, pixel_noise) if k > 0 else {}
        dt = self.cam_t[k] - self.cam_t[k - 1] if k > 0 else 1.0
        for i, (xn, yn, u, v, d) in cur.items():
            if i in prev:
                vx, vy = (xn - prev[i][0]) / dt, (yn - prev[i][1]) / dt
            else:
                vx = vy = 0.0
            out[i] = np.array([xn, yn, 1.0, u, v, vx, vy, d])
        return out

    def _project(self, k, pixel_noise):
        tk = self.cam_t[k]
        R, p = self.R_wb(tk), self.p_wb(tk)
        Xc = (self._lm - p) @ R
        # occlusion as in render_room: a far-wall point is seen only where its ray passes the near plane beyond the near wall's edge
        Lr, Cr = self._lm @ R0, p @ R0                      # render-world coordinates (render = R0^T world)
        far = Lr[:, 2] > self.near_z + 1e-9
        with np.errstate(all="ignore"):
            xn = Cr[0] + (Lr[:, 0] - Cr[0]) * (self.near_z - Cr[2]) / (Lr[:, 2] - Cr[2])
        hidden = far & (xn < self.split_x)
        out = {}
        for i in range(len(Xc)):
            if Xc[i, 2] < 0.3 or hidden[i]:
                continue
            u = synth.FX * Xc[i, 0] / Xc[i, 2] + synth.CX + pixel_noise * self._pn[k, i, 0]
            v = synth.FY * Xc[i, 1] / Xc[i, 2] + synth.CY + pixel_noise * self._pn[k, i, 1]
            if 2 <= u < synth.W - 2 and 2 <= v < synth.H - 2:
                d = float(np.rint(Xc[i, 2] * 1000.0)) / 1000.0
                out[i] = ((u - synth.CX) / synth.FX, (v - synth.CY) / synth.FY, u, v, d)
        return out

    # ---- GNSS: raw measurements of a receiver riding on the body origin (the measurement model of synth_window.add_gnss)
    def gnss_setup(self, sats_per_sys=3, n_low=2, lat=31.03, lon=121.44, alt=20.0, alpha=0.3, time_diff=18.0, orbits=None):
        """constellation (4 systems x sats_per_sys above 32 deg, plus n_low GPS satellites at 8..22 deg that the elevation gate of
        Estimator::processGNSS has to drop once gnss_ready), receiver clock and the ENU <- world yaw alpha.
        orbits: an object with eph2pos(t, eph) / geph2pos(t, geph) (position, clock bias).  Then every satellite flies a broadcast orbit (Kepler elements
        for GPS / Galileo / BeiDou, a PZ-90 state vector for GLONASS), gnss_epoch() returns RAW observations (gf_gnss_raw_obs) and self._gnss["ephems"]
        lists the ephemerides to hand to inputEphem."""
        import synth_window as SW
        rng = np.random.default_rng(7000 + self.seed)
        anc, Re = SW.geo2ecef(lat, lon, alt), SW.R_ecef_enu(lat, lon)
        sats = []
        if orbits is not None:
            toe, t_mid = time_diff - 900.0, time_diff + 3.0
            A_sys = {0: 26560e3, 1: 25510e3, 2: 29600e3, 3: 27906e3}
            for sys in range(4):
                for q in range(sats_per_sys + (n_low if sys == 0 else 0)):
                    low = q >= sats_per_sys
                    for _ in range(20000):   # draw orbits until the satellite stands in the wanted elevation band over the anchor
                        e = dict(sat=100 * sys + q + 1, sys=sys if sys != 1 else 0, prn=10 + q, toe=toe, toc=toe, toe_tow=345600.0 - 900.0, A=A_sys[sys], e=rng.uniform(0.001, 0.01),
                                 i0=1.131 if sys == 1 else 0.96, OMG0=rng.uniform(0, 2 * np.pi), omg=rng.uniform(0, 2 * np.pi), M0=rng.uniform(0, 2 * np.pi), delta_n=rng.uniform(3e-9, 5e-9),
                                 OMG_dot=-8e-9, i_dot=1e-10, cuc=rng.normal(0, 1e-6), cus=rng.normal(0, 1e-6), crc=rng.normal(0, 100.0), crs=rng.normal(0, 100.0), cic=rng.normal(0, 1e-7),
                                 cis=rng.normal(0, 1e-7), af0=rng.uniform(-2e-4, 2e-4), af1=rng.uniform(-1e-11, 1e-11), af2=0.0, tgd0=rng.uniform(-8e-9, 8e-9), ura=3.0)   # ura 2 would zero the Galileo weights (gnss_psr_dopp_factor.cpp:36: ura - 2)
                        d = Re.T @ (orbits.eph2pos(t_mid, e)[0] - anc)
                        el = np.degrees(np.arcsin(d[2] / np.linalg.norm(d)))
                        if (8 < el < 22) if low else (32 < el < 80):
                            break
                    else:
                        raise RuntimeError("no visible orbit found")
                    if sys == 1:   # GLONASS: the same orbit as a PZ-90 state vector at toe (velocity by a difference quotient in the rotating frame)
                        e["sys"] = 0
                        p0, p1 = orbits.eph2pos(toe, e)[0], orbits.eph2pos(toe + 1e-3, e)[0]
                        e = dict(sat=e["sat"], toe=toe, pos=p0, vel=(p1 - p0) / 1e-3, acc=rng.normal(0, 1e-6, 3), tau_n=rng.uniform(-2e-4, 2e-4), gamma=rng.uniform(-1e-11, 1e-11))
                    sats.append(dict(sat=100 * sys + q + 1, sys=sys, eph=e))
            R0w = self.R_wb(self.cam_t[0])
            theta0 = float(np.arctan2(R0w[1, 0], R0w[0, 0]))
            self._gnss = dict(anc=anc, Re=Re, R_ew=Re @ rot_z(alpha), sats=sats, dt=np.array([150.0, 180.0, 120.0, 200.0]), ddt=2.0, time_diff=time_diff,
                              yaw_enu_local=alpha + theta0, noise=np.random.default_rng(7100 + self.seed), orbits=orbits, ephems=[sv["eph"] for sv in sats])
            return self._gnss
        for sys in range(4):
            for q in range(sats_per_sys + (n_low if sys == 0 else 0)):
                low = q >= sats_per_sys
                az, rg = rng.uniform(0, 2 * np.pi), rng.uniform(2.0e7, 2.5e7)
                el = rng.uniform(np.radians(8), np.radians(22)) if low else rng.uniform(np.radians(32), np.radians(80))
                d_enu = np.array([np.sin(az) * np.cos(el), np.cos(az) * np.cos(el), np.sin(el)])
                sats.append(dict(sat=100 * sys + q + 1, sys=sys, pos=anc + Re @ (d_enu * rg), vel=Re @ (np.cross(d_enu, rng.normal(0, 1, 3)) * 1.5e3),
                                 svdt=rng.uniform(-2e-4, 2e-4), svddt=rng.uniform(-1e-11, 1e-11), tgd=rng.uniform(-8e-9, 8e-9)))
        R0w = self.R_wb(self.cam_t[0])
        theta0 = float(np.arctan2(R0w[1, 0], R0w[0, 0]))          # the estimator's local frame is the world turned by -theta0 about z (initFirstIMUPose zeroes the yaw)
        self._gnss = dict(anc=anc, Re=Re, R_ew=Re @ rot_z(alpha), sats=sats, dt=np.array([150.0, 180.0, 120.0, 200.0]), ddt=2.0, time_diff=time_diff,
                          yaw_enu_local=alpha + theta0, noise=np.random.default_rng(7100 + self.seed))
        return self._gnss

    def gnss_epoch(self, t_local, flaky_sat=None):
        """(gps time, list of gf_gnss_obs-like dicts) of the epoch received at local time t_local; flaky_sat: this satellite reports psr_std 5 m"""
        import synth_window as SW
        G = self._gnss
        rng = G["noise"]
        p, v = G["anc"] + G["R_ew"] @ self.p_wb(t_local), G["R_ew"] @ self._at(self._vw, t_local)
        wl = SW.C_LIGHT / 1575.42e6
        out = []
        if G.get("orbits") is not None:   # raw observations of satellites on broadcast orbits
            O, T = G["orbits"], t_local + G["time_diff"]
            freq = {0: 1575.42e6, 1: 1602.0e6, 2: 1575.42e6, 3: 1561.098e6}
            for sv in G["sats"]:
                f = O.geph2pos if sv["sys"] == 1 else O.eph2pos
                tof = 0.075
                for _ in range(3):   # transmission time from the light time
                    sp, svdt = f(T - tof, sv["eph"])
                    tof = np.linalg.norm(sp - p) / SW.C_LIGHT
                sp2, svdt2 = f(T - tof + 1e-3, sv["eph"])
                svel, svddt = (sp2 - sp) / 1e-3, (svdt2 - svdt) / 1e-3
                los = sp - p
                rg = np.linalg.norm(los)
                unit = los / rg
                clk = G["dt"][sv["sys"]] + G["ddt"] * t_local
                tgd = sv["eph"].get("tgd0", 0.0)
                psr = rg + SW.OMG_E * (sp[0] * p[1] - sp[1] * p[0]) / SW.C_LIGHT + clk - svdt * SW.C_LIGHT + tgd * SW.C_LIGHT + rng.normal(0, 0.5)
                dop = (svel - v) @ unit + SW.OMG_E / SW.C_LIGHT * (svel[0] * p[1] + sp[0] * v[1] - svel[1] * p[0] - sp[1] * v[0]) + G["ddt"] - svddt * SW.C_LIGHT
                bad = flaky_sat is not None and sv["sat"] == flaky_sat
                out.append(dict(sat=sv["sat"], sys=sv["sys"], time=T, psr=float(psr), dopp=float(-(dop + rng.normal(0, 0.05)) * freq[sv["sys"]] / SW.C_LIGHT),
                                psr_std=5.0 if bad else 0.6, dopp_std=0.3, freq=freq[sv["sys"]], tow=345600.0 + t_local))
            return T, out
        for sv in G["sats"]:
            sp = sv["pos"] + sv["vel"] * t_local
            los = sp - p
            rg = np.linalg.norm(los)
            unit = los / rg
            clk = G["dt"][sv["sys"]] + G["ddt"] * t_local
            psr = rg + SW.OMG_E * (sp[0] * p[1] - sp[1] * p[0]) / SW.C_LIGHT + clk - sv["svdt"] * SW.C_LIGHT + sv["tgd"] * SW.C_LIGHT + rng.normal(0, 0.5)
            dop = (sv["vel"] - v) @ unit + SW.OMG_E / SW.C_LIGHT * (sv["vel"][0] * p[1] + sp[0] * v[1] - sv["vel"][1] * p[0] - sp[1] * v[0]) + G["ddt"] - sv["svddt"] * SW.C_LIGHT
            bad = flaky_sat is not None and sv["sat"] == flaky_sat
            out.append(dict(sat=sv["sat"], sys=sv["sys"], time=t_local + G["time_diff"], psr=float(psr), dopp=float(-(dop + rng.normal(0, 0.05)) / wl),
                            psr_std=5.0 if bad else 0.6, dopp_std=0.3, wavelength=wl, sv_pos=sp.copy(), sv_vel=sv["vel"].copy(), svdt=sv["svdt"], svddt=sv["svddt"],
                            tgd=sv["tgd"], pr_uura=2.0, dp_uura=2.0, tow=345600.0 + t_local))
        return t_local + G["time_diff"], out

    def gnss_alignment(self, t_oldest, anchor_error=(0.4, -0.3, 0.2), yaw_error=0.002, clock_error=2.0):
        """a GNSSVIAlign result of realistic quality for a window whose oldest frame is at t_oldest: anchor, yaw, clock biases of that frame, drift"""
        G = self._gnss
        return G["anc"] + np.array(anchor_error), G["yaw_enu_local"] + yaw_error, G["dt"] + G["ddt"] * t_oldest + clock_error, G["ddt"] + 0.1

    # ---- the stream as files: what tools/gf_replay reads (host/replay_node.h) plus a configuration in the reference's own YAML dialect
    def export(self, out_dir, n_frames=None, **cfg):
        """writes imu.csv, wheel.csv, image0.csv, image1.csv, frames/*.pgm, config.yaml and cam.yaml; cfg overrides YAML keys"""
        import os
        import gfamd
        os.makedirs(os.path.join(out_dir, "frames"), exist_ok=True)
        n = len(self.cam_t) if n_frames is None else min(n_frames, len(self.cam_t))
        t_end = self.cam_t[n - 1] + 0.05
        with open(os.path.join(out_dir, "imu.csv"), "w") as f:
            f.write("# t,ax,ay,az,gx,gy,gz\n")
            for t, a, g in zip(self.imu_t, self.imu_acc, self.imu_gyr):
                if t <= t_end:
                    f.write(",".join(repr(float(v)) for v in (t, *a, *g)) + "\n")
        with open(os.path.join(out_dir, "wheel.csv"), "w") as f:
            f.write("# t,vx,vy,vz,wx,wy,wz\n")
            for t, v, g in zip(self.wheel_t, self.wheel_vel, self.wheel_gyr):
                if t <= t_end:
                    f.write(",".join(repr(float(x)) for x in (t, *v, *g)) + "\n")
        with open(os.path.join(out_dir, "image0.csv"), "w") as f0, open(os.path.join(out_dir, "image1.csv"), "w") as f1:
            for k in range(n):
                img, dep = self.image(k)
                gfamd.write_pgm(os.path.join(out_dir, "frames", "%06d_gray.pgm" % k), img)
                gfamd.write_pgm(os.path.join(out_dir, "frames", "%06d_depth.pgm" % k), dep)
                f0.write("%r,frames/%06d_gray.pgm\n" % (float(self.cam_t[k]), k))
                f1.write("%r,frames/%06d_depth.pgm\n" % (float(self.cam_t[k]), k))
        keys = dict(imu=1, wheel=1, depth=1, gnss_enable=0, w_replace=0, wdetect=1, stationary_detect=1, use_motion=0, use_line=0, use_mcc=0, plane=0, use_yolo=0,
                    num_of_cam=1, equalize=0, depth_threshold=3, output_path='"%s"' % out_dir, cam0_calib='"cam.yaml"', cam1_calib='"cam.yaml"',
                    image_width=synth.W, image_height=synth.H, estimate_extrinsic=0, extrinsic_type=0, estimate_wheel_extrinsic=1, extrinsic_type_wheel=0,
                    multiple_thread=1, max_cnt=150, min_dist=30, freq=10, F_threshold=1.0, show_track=0, flow_back=1, max_solver_time=0.04, max_num_iterations=8,
                    keyframe_parallax=10.0, acc_n=1.2374091609523514e-02, gyr_n=3.0032654435730201e-03, acc_w=1.9218003442176448e-04, gyr_w=5.4692100664858005e-05,
                    g_norm=G_NORM, wheel_gyro_noise_sigma=0.004, wheel_velocity_noise_sigma=0.01, estimate_wheel_intrinsic=0, sx=1.0, sy=1.0, sw=1.0,
                    estimate_td=0, td=0.0, estimate_td_wheel=0, td_wheel=0.0, gnss_elevation_thres=30, gnss_psr_std_thres=2.0, gnss_dopp_std_thres=2.0,
                    gnss_track_num_thres=5, gnss_ddt_sigma=0.1, gnss_local_online_sync=0, gnss_local_time_diff=18.0)
        keys.update(cfg)
        T_io = np.eye(4)
        T_io[:3, :3], T_io[:3, 3] = RIO, TIO

        def mat(name, M):
            rows = ",\n          ".join(", ".join(repr(float(v)) for v in r) for r in M)
            return "%s: !!opencv-matrix\n   rows: %d\n   cols: %d\n   dt: d\n   data: [ %s ]\n" % (name, M.shape[0], M.shape[1], rows)

        with open(os.path.join(out_dir, "config.yaml"), "w") as f:
            f.write("%YAML:1.0\n\n# synthetic sequence, seed " + str(self.seed) + " (ground-fusion_amd/synth_stream.py)\n")
            for k, v in keys.items():
                f.write("%s: %s\n" % (k, v))
            f.write("\n" + mat("body_T_cam0", np.eye(4)) + "\n" + mat("body_T_cam1", np.eye(4)) + "\n" + mat("body_T_wheel", T_io))
            import synth_window as SW
            f.write("\n" + mat("gnss_iono_default_parameters", SW.IONO.reshape(1, 8)))
        if getattr(self, "_gnss", None) is not None and keys.get("gnss_enable"):
            # one epoch per back-end frame (every second camera frame) a few ms off the frame stamp, and an alignment on offer at every such frame
            orng = np.random.default_rng(99)
            W = int(cfg.get("window_size", 10))
            with open(os.path.join(out_dir, "gnss.csv"), "w") as fg, open(os.path.join(out_dir, "gnss_align.csv"), "w") as fa:
                fg.write("# t_msg,sat,sys,time,psr,dopp,psr_std,dopp_std,wavelength,sx,sy,sz,vx,vy,vz,svdt,svddt,tgd,pr_uura,dp_uura,tow\n")
                fa.write("# t,ax,ay,az,yaw,dt0,dt1,dt2,dt3,ddt\n")
                for k in range(0, n, 2):
                    tk = float(self.cam_t[k])
                    tg, epoch = self.gnss_epoch(tk + orng.uniform(-0.02, 0.02), flaky_sat=2 if (k // 2) % 6 == 5 else None)
                    for o in epoch:
                        fg.write(",".join(repr(float(v)) for v in (tg, o["sat"], o["sys"], o["time"], o["psr"], o["dopp"], o["psr_std"], o["dopp_std"], o["wavelength"], *o["sv_pos"],
                                                                   *o["sv_vel"], o["svdt"], o["svddt"], o["tgd"], o["pr_uura"], o["dp_uura"], o["tow"])) + "\n")
                    anc, yaw, dt4, ddt = self.gnss_alignment(tk - W / 15.0)
                    fa.write(",".join(repr(float(v)) for v in (tk - 0.03, *anc, yaw, *dt4, ddt)) + "\n")
        with open(os.path.join(out_dir, "cam.yaml"), "w") as f:
            f.write("%%YAML:1.0\n---\nmodel_type: PINHOLE\ncamera_name: camera\nimage_width: %d\nimage_height: %d\ndistortion_parameters:\n   k1: 0.0\n   k2: 0.0\n"
                    "   p1: 0.0\n   p2: 0.0\nprojection_parameters:\n   fx: %r\n   fy: %r\n   cx: %r\n   cy: %r\n" % (synth.W, synth.H, synth.FX, synth.FY, synth.CX, synth.CY))
        return n
