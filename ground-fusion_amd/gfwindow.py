"""Flat sliding-window description shared by the C-ABI (gf_ba_window, include/groundfusion_hip.h) and, with the same
layout, by the test oracle.  Mirrors what Estimator::vector2double + the factor constructors hand to Ceres
(estimator.cpp:2276-2353, :2890-3297).  Pure numpy/ctypes plumbing."""
import ctypes as C
import numpy as np

POSE, SPEEDBIAS, EX_POSE, EX_WHEEL, SX, SY, SW, TD, TD_WHEEL, FEATURE, RCV_DT, RCV_DDT, YAW, ANC = range(14)


def bid(kind, idx=0):
    return kind * 4096 + idx


def gsize(kind):
    return 7 if kind in (POSE, EX_POSE, EX_WHEEL) else 9 if kind == SPEEDBIAS else 3 if kind == ANC else 1


def lsize(kind):
    g = gsize(kind)
    return 6 if g == 7 else g


PD = C.POINTER(C.c_double)
PI = C.POINTER(C.c_int)


class WindowC(C.Structure):
    _fields_ = [("W", C.c_int), ("n_feature", C.c_int), ("n_visual", C.c_int), ("n_imu", C.c_int), ("n_wheel", C.c_int),
                ("fix_ex_pose", C.c_int), ("fix_ex_wheel", C.c_int), ("fix_ix", C.c_int), ("fix_td", C.c_int), ("fix_td_wheel", C.c_int), ("fix_poses", C.c_int),
                ("G", C.c_double * 3), ("vis_sqrt_info", C.c_double),
                ("para_Pose", PD), ("para_SpeedBias", PD), ("para_Ex_Pose", PD), ("para_Ex_Pose_wheel", PD), ("para_Ix", PD), ("para_Td", PD),
                ("para_Td_wheel", PD), ("para_Feature", PD),
                ("feature_fixed", C.POINTER(C.c_ubyte)),
                ("vis_feature", PI), ("vis_i", PI), ("vis_j", PI),
                ("vis_pts_i", PD), ("vis_pts_j", PD), ("vis_vel_i", PD), ("vis_vel_j", PD), ("vis_td_i", PD), ("vis_td_j", PD),
                ("imu_i", PI), ("imu_sum_dt", PD), ("imu_delta_p", PD), ("imu_delta_q", PD), ("imu_delta_v", PD), ("imu_lin_ba", PD), ("imu_lin_bg", PD),
                ("imu_jacobian", PD), ("imu_covariance", PD),
                ("wh_i", PI), ("wh_sum_dt", PD), ("wh_delta_p", PD), ("wh_delta_q", PD), ("wh_jacobian", PD), ("wh_covariance", PD), ("wh_lin", PD),
                ("wh_lin_vel", PD), ("wh_lin_gyr", PD), ("wh_vel_1", PD), ("wh_gyr_1", PD),
                ("prior_n", C.c_int), ("prior_nblocks", C.c_int), ("prior_block_id", PI), ("prior_J", PD), ("prior_r", PD), ("prior_x0", PD),
                ("gnss_enabled", C.c_int), ("gnss_lowspeed", C.c_int), ("n_gnss", C.c_int), ("has_anchor", C.c_int),
                ("para_rcv_dt", PD), ("para_rcv_ddt", PD), ("para_yaw_enu_local", PD), ("para_anc_ecef", PD),
                ("gnss_ddt_weight", C.c_double), ("anchor_value", C.c_double * 7),
                ("gnss_iono", PD), ("gnss_frame", PI), ("gnss_lower", PI), ("gnss_sys", PI), ("gnss_ratio", PD), ("gnss_data", PD), ("gnss_headers", PD),
                ("ex_pose_mask", C.c_int), ("ex_wheel_mask", C.c_int)]


class SummaryC(C.Structure):
    _fields_ = [("iterations", C.c_int), ("successful_steps", C.c_int), ("termination", C.c_int), ("initial_cost", C.c_double),
                ("final_cost", C.c_double), ("radius", C.c_double)]


_F64 = ["para_Pose", "para_SpeedBias", "para_Ex_Pose", "para_Ex_Pose_wheel", "para_Ix", "para_Td", "para_Td_wheel", "para_Feature",
        "vis_pts_i", "vis_pts_j", "vis_vel_i", "vis_vel_j", "vis_td_i", "vis_td_j", "imu_sum_dt", "imu_delta_p", "imu_delta_q", "imu_delta_v",
        "imu_lin_ba", "imu_lin_bg", "imu_jacobian", "imu_covariance", "wh_sum_dt", "wh_delta_p", "wh_delta_q", "wh_jacobian", "wh_covariance",
        "wh_lin", "wh_lin_vel", "wh_lin_gyr", "wh_vel_1", "wh_gyr_1", "prior_J", "prior_r", "prior_x0",
        "para_rcv_dt", "para_rcv_ddt", "para_yaw_enu_local", "para_anc_ecef", "gnss_iono", "gnss_ratio", "gnss_data", "gnss_headers"]
_I32 = ["vis_feature", "vis_i", "vis_j", "imu_i", "wh_i", "prior_block_id", "gnss_frame", "gnss_lower", "gnss_sys"]
_SCALARS = ["W", "n_feature", "n_visual", "n_imu", "n_wheel", "fix_ex_pose", "fix_ex_wheel", "fix_ix", "fix_td", "fix_td_wheel", "fix_poses",
            "vis_sqrt_info", "prior_n", "prior_nblocks", "gnss_enabled", "gnss_lowspeed", "n_gnss", "has_anchor", "gnss_ddt_weight", "ex_pose_mask", "ex_wheel_mask"]
STATE_KEYS = ["para_Pose", "para_SpeedBias", "para_Ex_Pose", "para_Ex_Pose_wheel", "para_Ix", "para_Td", "para_Td_wheel", "para_Feature",
              "para_rcv_dt", "para_rcv_ddt", "para_yaw_enu_local", "para_anc_ecef"]


class Window(dict):
    """dict of numpy arrays / scalars with the field names of gf_ba_window."""

    def copy(self):
        w = Window()
        for k, v in self.items():
            w[k] = v.copy() if isinstance(v, np.ndarray) else v
        return w

    def finalize(self):
        for k in _F64:
            self[k] = np.ascontiguousarray(self.get(k, np.zeros(0)), np.float64).reshape(-1)
        for k in _I32:
            self[k] = np.ascontiguousarray(self.get(k, np.zeros(0)), np.int32).reshape(-1)
        self["feature_fixed"] = np.ascontiguousarray(self.get("feature_fixed", np.zeros(0)), np.uint8).reshape(-1)
        self["G"] = np.ascontiguousarray(self.get("G", [0, 0, 9.805]), np.float64)
        self["n_feature"] = len(self["para_Feature"])
        self["n_visual"] = len(self["vis_feature"])
        self["n_imu"] = len(self["imu_i"])
        self["n_wheel"] = len(self["wh_i"])
        self["prior_n"] = len(self["prior_r"])
        self["prior_nblocks"] = len(self["prior_block_id"])
        self["n_gnss"] = len(self["gnss_frame"])
        self["anchor_value"] = np.ascontiguousarray(self.get("anchor_value", np.zeros(7)), np.float64)
        for k in _SCALARS:
            self.setdefault(k, 0)
        return self

    def to_c(self):
        """returns WindowC whose pointers reference this dict's arrays (state arrays are updated in place by solve)."""
        self.finalize()
        c = WindowC()
        for k in _SCALARS:
            setattr(c, k, self[k])
        for i in range(3):
            c.G[i] = self["G"][i]
        for i in range(7):
            c.anchor_value[i] = self["anchor_value"][i]
        for k in _F64:
            a = self[k]
            setattr(c, k, a.ctypes.data_as(PD) if a.size else None)
        for k in _I32:
            a = self[k]
            setattr(c, k, a.ctypes.data_as(PI) if a.size else None)
        a = self["feature_fixed"]
        c.feature_fixed = a.ctypes.data_as(C.POINTER(C.c_ubyte)) if a.size else None
        return c

    def set_prior(self, prior):
        if prior is None:
            for k in ("prior_block_id", "prior_J", "prior_r", "prior_x0"):
                self[k] = np.zeros(0)
        else:
            self["prior_block_id"], self["prior_J"], self["prior_r"], self["prior_x0"] = prior["block_id"], prior["J"], prior["r"], prior["x0"]
        return self.finalize()
