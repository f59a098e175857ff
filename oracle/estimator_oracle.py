"""TEST INFRASTRUCTURE — CPU restatement (numpy) of the bookkeeping half of Ground-Fusion's back end: Estimator::processMeasurements /
processIMU / processWheel / processImage / optimization (problem construction) / slideWindow and FeatureManager
(vins_estimator/src/estimator/estimator.cpp, feature_manager.cpp; line numbers cited per method).  The numerics it delegates to are the C
oracle's (oracle_py.ba_solve / ba_marginalize / *_preintegrate).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this file.  PARITY UNPINNED: the reference ships no golden vectors for this path and cannot be built here; this file
is written independently of ground-fusion_amd/csrc/gf_estimator.hip, from the reference's control flow, and the two are compared."""
import math
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _HERE)
sys.path.insert(0, os.path.join(_HERE, "..", "ground-fusion_amd"))
import oracle_py as O  # noqa: E402
import gfwindow as gw  # noqa: E402
import init_oracle as IO  # noqa: E402

INITIAL, NON_LINEAR = 0, 1
MARGIN_OLD, MARGIN_SECOND_NEW = 0, 1


def default_cfg():
    """config/realsense/m2dgrp.yaml"""
    return dict(window_size=10, max_features=512, max_visual=4096, use_imu=1, use_wheel=1, depth=1, estimate_extrinsic=0, estimate_wheel_extrinsic=1,
                estimate_wheel_intrinsic=0, estimate_td=0, estimate_td_wheel=0, use_mcc=0, wdetect=1, stationary_detect=1, only_initial_with_wheel=0,
                multiple_thread=1, num_iterations=8, acc_n=1.2374091609523514e-02, gyr_n=3.0032654435730201e-03, acc_w=1.9218003442176448e-04,
                gyr_w=5.4692100664858005e-05, g_norm=9.805, wheel_vel_n=0.01, wheel_gyr_n=0.004, min_parallax_px=10.0, depth_threshold=3.0, init_depth=5.0,
                focal_length=600.0, td=0.0, td_wheel=0.0, sx=1.0, sy=1.0, sw=1.0, tic=np.zeros(3), ric=np.eye(3),
                tio=np.array([0.0497956, 1.06332, -0.037465]),
                rio=np.array([[0.352551, -0.935764, -0.00734672], [0.0145238, 0.0133214, -0.999806], [0.93568, 0.352375, 0.0182873]]),
                gnss_enable=0, gnss_track_num_thres=5, gnss_elevation_thres=30.0, gnss_psr_std_thres=2.0, gnss_dopp_std_thres=2.0, gnss_ddt_sigma=0.1,
                gnss_local_time_diff=18.0, extrinsic_type=0, extrinsic_type_wheel=0,
                gnss_iono=np.array([0.1118e-07, 0.2235e-07, -0.4172e-06, 0.6557e-06, 0.1249e+06, -0.4424e+06, 0.1507e+07, -0.2621e+06]))


def subset_mask(extrinsic_type):
    """YAML extrinsic_type[_wheel] -> constancy bits of PoseSubsetParameterization (parameters.cpp:394-420 / :280-306 select the enum, estimator.cpp:2969-2985 /
    :3010-3026 the index sets; an out-of-range value leaves the zero-initialised enum, which is ADJUST_*_TRANSLATION)"""
    sets = {0: (), 2: (0, 1, 2), 3: (2,), 4: (2, 3, 4, 5)}
    return sum(1 << i for i in sets.get(int(extrinsic_type), (3, 4, 5)))


# ---------------------------------------------------------------- gnss_comm helpers the estimator itself calls (ecef2geo, ecef2rotation, sat_azel; RTKLIB lineage)
def ecef2geo(p):
    """latitude [deg], longitude [deg], height [m] (gnss_comm gnss_utility.cpp ecef2geo: Bowring's closed form)"""
    x, y, z = (float(v) for v in p)
    if x == 0 and y == 0:
        return np.zeros(3)
    a, e2 = 6378137.0, 6.69437999014e-3
    a2 = a * a
    b2 = a2 * (1 - e2)
    b = math.sqrt(b2)
    ep2 = (a2 - b2) / b2
    rho = math.sqrt(x * x + y * y)
    s1, s2 = z * a, rho * b
    h = math.sqrt(s1 * s1 + s2 * s2)
    st, ct = s1 / h, s2 / h
    s1 = z + ep2 * b * st ** 3
    s2 = rho - a * e2 * ct ** 3
    h = math.sqrt(s1 * s1 + s2 * s2)
    sin_lat, cos_lat = s1 / h, s2 / h
    N = a2 / math.sqrt(a2 * cos_lat * cos_lat + b2 * sin_lat * sin_lat)
    return np.array([math.degrees(math.atan(s1 / s2)), math.degrees(math.atan2(y, x)), rho / cos_lat - N])


def ecef2rotation(p):
    """R_ecef_enu at p"""
    lla = ecef2geo(p)
    lat, lon = math.radians(lla[0]), math.radians(lla[1])
    sl, cl, so, co = math.sin(lat), math.cos(lat), math.sin(lon), math.cos(lon)
    return np.array([[-so, -sl * co, cl * co], [co, -sl * so, cl * so], [0.0, cl, sl]])


C_LIGHT, OMG_E = 2.99792458e8, 7.2921151467e-5


def sat_azel(rcv, sat):
    dl = np.asarray(sat, float) - np.asarray(rcv, float)
    dl = dl / np.linalg.norm(dl)
    enu = ecef2rotation(rcv).T @ dl
    az = 0.0 if math.hypot(dl[0], dl[1]) < 1e-12 else math.atan2(enu[0], enu[1])
    if az < 0:
        az += 2 * math.pi
    return az, math.asin(enu[2])


def trop_delay(lla, el):
    """Saastamoinen, standard atmosphere, relative humidity 0.7 (gnss_comm calculate_trop_delay; RTKLIB tropmodel)"""
    if lla[2] < -100.0 or 1e4 < lla[2] or el <= 0:
        return 0.0
    hgt = 0.0 if lla[2] < 0.0 else lla[2]
    pres = 1013.25 * (1.0 - 2.2557e-5 * hgt) ** 5.2568
    temp = 15.0 - 6.5e-3 * hgt + 273.16
    e = 6.108 * 0.7 * math.exp((17.15 * temp - 4684.0) / (temp - 38.45))
    z = math.pi / 2.0 - el
    trph = 0.0022768 * pres / (1.0 - 0.00266 * math.cos(2.0 * math.radians(lla[0])) - 0.00028 * hgt / 1e3) / math.cos(z)
    return trph + 0.002277 * (1255.0 / temp + 0.05) * e / math.cos(z)


def ion_delay(tow, ion_in, lla, az, el):
    """Klobuchar (gnss_comm calculate_ion_delay; RTKLIB ionmodel)"""
    ion_default = [0.1118e-07, -0.7451e-08, -0.5961e-07, 0.1192e-06, 0.1167e+06, -0.2294e+06, -0.1311e+06, 0.1049e+07]
    if lla[2] < -1e3 or el <= 0:
        return 0.0
    ion = list(ion_in) if float(np.dot(ion_in, ion_in)) > 0.0 else ion_default
    psi = 0.0137 / (el / math.pi + 0.11) - 0.022
    phi = lla[0] / 180.0 + psi * math.cos(az)
    phi = 0.416 if phi > 0.416 else (-0.416 if phi < -0.416 else phi)
    lam = lla[1] / 180.0 + psi * math.sin(az) / math.cos(phi * math.pi)
    phi += 0.064 * math.cos((lam - 1.617) * math.pi)
    tt = 43200.0 * lam + tow
    tt -= math.floor(tt / 86400.0) * 86400.0
    f = 1.0 + 16.0 * (0.53 - el / math.pi) ** 3
    amp = ion[0] + phi * (ion[1] + phi * (ion[2] + phi * ion[3]))
    per = ion[4] + phi * (ion[5] + phi * (ion[6] + phi * ion[7]))
    amp = 0.0 if amp < 0.0 else amp
    per = 72000.0 if per < 72000.0 else per
    x = 2.0 * math.pi * (tt - 50400.0) / per
    return C_LIGHT * f * (5e-9 + amp * (1.0 + x * x * (-0.5 + x * x / 24.0)) if abs(x) < 1.57 else 5e-9)


def psr_res(xyzt, meas, iono):
    """gnss_comm psr_res restated with the measurement model of GnssPsrDoppFactor::Evaluate (gnss_psr_dopp_factor.cpp:76-97): residuals and rows
    [-unit(rcv -> sat), 1 on the clock of the satellite's system]"""
    rcv = np.asarray(xyzt[0:3], float)
    res, J = [], []
    for o in meas:
        sv = np.asarray(o["sv_pos"], float)
        ion = tro = 0.0
        if np.linalg.norm(rcv) > 0:
            lla = ecef2geo(rcv)
            az, el = sat_azel(rcv, sv)
            tro, ion = trop_delay(lla, el), ion_delay(o["tow"], iono, lla, az, el)
        r2s = sv - rcv
        rg = np.linalg.norm(r2s)
        est = rg + OMG_E * (sv[0] * rcv[1] - sv[1] * rcv[0]) / C_LIGHT + xyzt[3 + o["sys"]] - o["svdt"] * C_LIGHT + ion + tro + o["tgd"] * C_LIGHT
        res.append(est - o["psr"])
        row = np.zeros(7)
        row[0:3] = -r2s / rg
        row[3 + o["sys"]] = 1.0
        J.append(row)
    return np.array(res), np.array(J).reshape(-1, 7)


def dopp_res(vel_ddt, rcv, meas):
    """gnss_comm dopp_res with the Doppler model of the same factor (:99-101)"""
    vel, rcv = np.asarray(vel_ddt[0:3], float), np.asarray(rcv, float)
    res, J = [], []
    for o in meas:
        sv, svv = np.asarray(o["sv_pos"], float), np.asarray(o["sv_vel"], float)
        r2s = sv - rcv
        unit = r2s / np.linalg.norm(r2s)
        sag = OMG_E / C_LIGHT * (svv[0] * rcv[1] + sv[0] * vel[1] - svv[1] * rcv[0] - sv[1] * vel[0])
        est = (svv - vel) @ unit + vel_ddt[3] + sag - o["svddt"] * C_LIGHT
        res.append(est + o["dopp"] * o["wavelength"])
        J.append(np.array([-unit[0], -unit[1], -unit[2], 1.0]))
    return np.array(res), np.array(J).reshape(-1, 4)


# ---- broadcast ephemerides -> satellite state (gnss_comm eph2pos / geph2pos / eph2svdt / eph2vel; RTKLIB ephemeris.c lineage: IS-GPS-200 Kepler model,
# BeiDou GEO frame, GLONASS ICD Runge-Kutta at 60 s steps, velocities and clock drift by the 1 ms difference quotient).  Ephemerides are dicts with the
# field names of gf_gnss_ephem / gf_gnss_glo_ephem.
MU = {0: 3.9860050e14, 2: 3.986004418e14, 3: 3.986004418e14}
OMGE = {0: 7.2921151467e-5, 2: 7.2921151467e-5, 3: 7.292115e-5}
MU_GLO, J2_GLO, OMGE_GLO, RE_GLO, TSTEP = 3.9860044e14, 1.0826257e-3, 7.292115e-5, 6378136.0, 60.0
SIN_5, COS_5 = -0.0871557427476582, 0.9961946980917456


def eph2svdt(t, e):
    tk = t - e["toc"]
    for _ in range(2):
        tk -= e["af0"] + e["af1"] * tk + e["af2"] * tk * tk
    return e["af0"] + e["af1"] * tk + e["af2"] * tk * tk


def eph2pos(t, e):
    """(position, clock bias incl. the relativistic term)"""
    mu, omge = MU[e["sys"]], OMGE[e["sys"]]
    tk = t - e["toe"]
    M = e["M0"] + (math.sqrt(mu / e["A"] ** 3) + e["delta_n"]) * tk
    E, Ek, n = M, 0.0, 0
    while abs(E - Ek) > 1e-13 and n < 30:
        Ek = E
        E -= (E - e["e"] * math.sin(E) - M) / (1.0 - e["e"] * math.cos(E))
        n += 1
    sinE, cosE = math.sin(E), math.cos(E)
    u = math.atan2(math.sqrt(1.0 - e["e"] ** 2) * sinE, cosE - e["e"]) + e["omg"]
    r = e["A"] * (1.0 - e["e"] * cosE)
    i = e["i0"] + e["i_dot"] * tk
    s2, c2 = math.sin(2.0 * u), math.cos(2.0 * u)
    u += e["cus"] * s2 + e["cuc"] * c2
    r += e["crs"] * s2 + e["crc"] * c2
    i += e["cis"] * s2 + e["cic"] * c2
    x, y, cosi = r * math.cos(u), r * math.sin(u), math.cos(i)
    if e["sys"] == 3 and (e["prn"] <= 5 or e["prn"] >= 59):
        O = e["OMG0"] + e["OMG_dot"] * tk - omge * e["toe_tow"]
        sO, cO = math.sin(O), math.cos(O)
        xg, yg, zg = x * cO - y * cosi * sO, x * sO + y * cosi * cO, y * math.sin(i)
        so, co = math.sin(omge * tk), math.cos(omge * tk)
        rs = np.array([xg * co + yg * so * COS_5 + zg * so * SIN_5, -xg * so + yg * co * COS_5 + zg * co * SIN_5, -yg * SIN_5 + zg * COS_5])
    else:
        O = e["OMG0"] + (e["OMG_dot"] - omge) * tk - omge * e["toe_tow"]
        sO, cO = math.sin(O), math.cos(O)
        rs = np.array([x * cO - y * cosi * sO, x * sO + y * cosi * cO, y * math.sin(i)])
    tk = t - e["toc"]
    return rs, e["af0"] + e["af1"] * tk + e["af2"] * tk * tk - 2.0 * math.sqrt(mu * e["A"]) * e["e"] * sinE / C_LIGHT ** 2


def _glo_deq(x, acc):
    r2 = float(x[0:3] @ x[0:3])
    r3, omg2 = r2 * math.sqrt(r2), OMGE_GLO ** 2
    a = 1.5 * J2_GLO * MU_GLO * RE_GLO ** 2 / r2 / r3
    b = 5.0 * x[2] * x[2] / r2
    c = -MU_GLO / r3 - a * (1.0 - b)
    return np.array([x[3], x[4], x[5], (c + omg2) * x[0] + 2.0 * OMGE_GLO * x[4] + acc[0], (c + omg2) * x[1] - 2.0 * OMGE_GLO * x[3] + acc[1], (c - 2.0 * a) * x[2] + acc[2]])


def geph2svdt(t, g):
    tk = t - g["toe"]
    for _ in range(2):
        tk -= -g["tau_n"] + g["gamma"] * tk
    return -g["tau_n"] + g["gamma"] * tk


def geph2pos(t, g):
    tk = t - g["toe"]
    dts = -g["tau_n"] + g["gamma"] * tk
    x, acc = np.array([*g["pos"], *g["vel"]], float), np.asarray(g["acc"], float)
    tt = -TSTEP if tk < 0.0 else TSTEP
    while abs(tk) > 1e-9:
        if abs(tk) < TSTEP:
            tt = tk
        k1 = _glo_deq(x, acc)
        k2 = _glo_deq(x + k1 * tt / 2.0, acc)
        k3 = _glo_deq(x + k2 * tt / 2.0, acc)
        k4 = _glo_deq(x + k3 * tt, acc)
        x = x + (k1 + 2.0 * k2 + 2.0 * k3 + k4) * tt / 6.0
        tk -= tt
    return x[0:3].copy(), dts


def sat_state(raw, eph=None, geph=None):
    """GnssPsrDoppFactor's constructor (gnss_psr_dopp_factor.cpp:3-47): a raw L1 observation + its ephemeris -> the fields of gf_gnss_obs"""
    o = dict(sat=raw["sat"], sys=raw["sys"], time=raw["time"], psr=raw["psr"], dopp=raw["dopp"], psr_std=raw["psr_std"], dopp_std=raw["dopp_std"],
             wavelength=C_LIGHT / raw["freq"], tow=raw["tow"])
    sv_tx, tt = raw["time"] - raw["psr"] / C_LIGHT, 1e-3
    if geph is not None:
        sv_tx -= geph2svdt(sv_tx, geph)
        (p, d1), (p2, d2) = geph2pos(sv_tx, geph), geph2pos(sv_tx + tt, geph)
        o.update(tgd=0.0, pr_uura=2.0 * (raw["psr_std"] / 0.16), dp_uura=2.0 * (raw["dopp_std"] / 0.256))
    else:
        sv_tx -= eph2svdt(sv_tx, eph)
        (p, d1), (p2, d2) = eph2pos(sv_tx, eph), eph2pos(sv_tx + tt, eph)
        k = eph["ura"] - 2.0 if eph["sys"] == 2 else eph["ura"] - 1.0
        o.update(tgd=eph["tgd0"], pr_uura=k * (raw["psr_std"] / 0.16), dp_uura=k * (raw["dopp_std"] / 0.256))
    o.update(sv_pos=p, sv_vel=(p2 - p) / tt, svdt=d1, svddt=(d2 - d1) / tt)
    return o


def sat_elevation(rcv, sat):
    dl = np.asarray(sat, float) - np.asarray(rcv, float)
    dl = dl / np.linalg.norm(dl)
    return math.asin((ecef2rotation(rcv).T @ dl)[2])


# ---------------------------------------------------------------- small rotation helpers (utility/utility.h)
def R2ypr(R):
    n, o, a = R[:, 0], R[:, 1], R[:, 2]
    y = math.atan2(n[1], n[0])
    p = math.atan2(-n[2], n[0] * math.cos(y) + n[1] * math.sin(y))
    r = math.atan2(a[0] * math.sin(y) - a[1] * math.cos(y), -o[0] * math.sin(y) + o[1] * math.cos(y))
    return np.array([y, p, r]) / math.pi * 180.0


def ypr2R(ypr):
    y, p, r = np.asarray(ypr, float) / 180.0 * math.pi
    Rz = np.array([[math.cos(y), -math.sin(y), 0], [math.sin(y), math.cos(y), 0], [0, 0, 1]])
    Ry = np.array([[math.cos(p), 0, math.sin(p)], [0, 1, 0], [-math.sin(p), 0, math.cos(p)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(r), -math.sin(r)], [0, math.sin(r), math.cos(r)]])
    return Rz @ Ry @ Rx


def quat_to_R(w, x, y, z):
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R_to_quat(R):
    """Eigen::Quaterniond(Matrix3d) -> (w, x, y, z)"""
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        t = math.sqrt(t + 1.0)
        w = 0.5 * t
        t = 0.5 / t
        return np.array([w, (R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t])
    i = 0
    if R[1, 1] > R[0, 0]:
        i = 1
    if R[2, 2] > R[i, i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    t = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
    q = np.zeros(4)
    q[1 + i] = 0.5 * t
    t = 0.5 / t
    q[0] = (R[k, j] - R[j, k]) * t
    q[1 + j] = (R[j, i] + R[i, j]) * t
    q[1 + k] = (R[k, i] + R[i, k]) * t
    return q


def qmul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3], a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])


def ypr2R_deg_free(r):
    """Rx(r0) Ry(r1) Rz(r2), angles in radians: what Sophus::SO3(double rot_x, double rot_y, double rot_z) of the non-templated Sophus builds"""
    cx, sx, cy, sy, cz, sz = math.cos(r[0]), math.sin(r[0]), math.cos(r[1]), math.sin(r[1]), math.cos(r[2]), math.sin(r[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rx @ Ry @ Rz


def deltaQ_R(theta):  # Utility::deltaQ(theta).toRotationMatrix()
    q = np.array([1.0, theta[0] / 2, theta[1] / 2, theta[2] / 2])
    q = q / np.linalg.norm(q)
    return quat_to_R(*q)


def g2R(g):  # utility.cpp:12-22
    v0 = g / np.linalg.norm(g)
    v1 = np.array([0.0, 0.0, 1.0])
    c = float(v1 @ v0)
    if c < -1.0 + 1e-12:
        R0 = np.diag([1.0, -1.0, -1.0])
    else:
        axis = np.cross(v0, v1)
        s = math.sqrt((1.0 + c) * 2.0)
        R0 = quat_to_R(s * 0.5, *(axis / s))
    yaw = R2ypr(R0)[0]
    return ypr2R([-yaw, 0, 0]) @ R0


def smallest_right_singular_vector(rows):
    """Right singular vector of the smallest singular value of an (n x 4) system by one-sided (Hestenes) Jacobi rotations — the role of
    Eigen::JacobiSVD(...).matrixV().rightCols<1>() at FM:710.  Plain Python floats with a fixed left-to-right summation order, so the result does
    not depend on a BLAS/LAPACK build; numpy.linalg.svd agrees with it to the conditioning of the system (checked in tests)."""
    A = [[float(v) for v in r] for r in rows]
    n = len(A)
    V = [[1.0 if i == j else 0.0 for j in range(4)] for i in range(4)]
    for _ in range(60):
        off = 0.0
        for p in range(3):
            for q in range(p + 1, 4):
                a = b = g = 0.0
                for r in range(n):
                    x, y = A[r][p], A[r][q]
                    a += x * x
                    b += y * y
                    g += x * y
                if g == 0.0 or abs(g) <= 1e-300:
                    continue
                if abs(g) <= 1e-15 * math.sqrt(a * b):
                    continue
                off = max(off, abs(g) / math.sqrt(a * b))
                zeta = (b - a) / (2.0 * g)
                t = (1.0 if zeta >= 0 else -1.0) / (abs(zeta) + math.sqrt(1.0 + zeta * zeta))
                c = 1.0 / math.sqrt(1.0 + t * t)
                sn = c * t
                for r in range(n):
                    x, y = A[r][p], A[r][q]
                    A[r][p], A[r][q] = c * x - sn * y, sn * x + c * y
                for r in range(4):
                    x, y = V[r][p], V[r][q]
                    V[r][p], V[r][q] = c * x - sn * y, sn * x + c * y
        if off == 0.0:
            break
    best, bn = 0, -1.0
    for c in range(4):
        nn = 0.0
        for r in range(n):
            nn += A[r][c] * A[r][c]
        if bn < 0 or nn < bn:
            bn, best = nn, c
    return np.array([V[r][best] for r in range(4)])


# ---------------------------------------------------------------- FeatureManager
class FeaturePerFrame:
    def __init__(self, p8, td):  # feature_manager.h:33-45
        self.point = np.array(p8[0:3], float)
        self.uv = np.array(p8[3:5], float)
        self.velocity = np.array(p8[5:7], float)
        self.depth = float(p8[7])
        self.cur_td = td


class FeaturePerId:
    def __init__(self, fid, start):  # feature_manager.h:78-82
        self.feature_id, self.start_frame = fid, start
        self.feature_per_frame = []
        self.used_num, self.estimated_depth, self.estimate_flag, self.solve_flag = 0, -1.0, 0, 0

    def endFrame(self):
        return self.start_frame + len(self.feature_per_frame) - 1


class FeatureManager:
    def __init__(self, cfg):
        self.feature = []
        self.W = cfg["window_size"]
        self.FOCAL_LENGTH = cfg["focal_length"]
        self.MIN_PARALLAX = cfg["min_parallax_px"] / cfg["focal_length"]
        self.INIT_DEPTH = cfg["init_depth"]
        self.depth_threshold = cfg["depth_threshold"]
        self.last_track_num = self.new_feature_num = self.long_track_num = 0
        self.last_average_parallax = 0.0
        self.svd = cfg.get("svd", "jacobi")

    def used(self):
        return [f for f in self.feature if len(f.feature_per_frame) >= 4]

    def getFeatureCount(self):  # FM:43-55
        return len(self.used())

    def addFeatureCheckParallax(self, frame_count, image, td):  # FM:57-116; image: {id: 8-vector}, iterated in id order (std::map)
        self.last_track_num = self.new_feature_num = self.long_track_num = 0
        self.last_average_parallax = 0.0
        index = {f.feature_id: f for f in self.feature}
        for fid in sorted(image):
            fpf = FeaturePerFrame(image[fid], td)
            f = index.get(fid)
            if f is None:
                f = FeaturePerId(fid, frame_count)
                self.feature.append(f)
                index[fid] = f
                f.feature_per_frame.append(fpf)
                self.new_feature_num += 1
            else:
                f.feature_per_frame.append(fpf)
                self.last_track_num += 1
                if len(f.feature_per_frame) >= 4:
                    self.long_track_num += 1
        if frame_count < 2 or self.last_track_num < 20 or self.long_track_num < 40 or self.new_feature_num > 0.5 * self.last_track_num:
            return True
        psum, pnum = 0.0, 0
        for f in self.feature:
            if f.start_frame <= frame_count - 2 and f.endFrame() >= frame_count - 1:
                fi = f.feature_per_frame[frame_count - 2 - f.start_frame]
                fj = f.feature_per_frame[frame_count - 1 - f.start_frame]
                du = fi.point[0] / fi.point[2] - fj.point[0]  # FM:978-1010
                dv = fi.point[1] / fi.point[2] - fj.point[1]
                psum += max(0.0, math.sqrt(du * du + dv * dv))
                pnum += 1
        if pnum == 0:
            return True
        self.last_average_parallax = psum / pnum * self.FOCAL_LENGTH
        return psum / pnum >= self.MIN_PARALLAX

    def getCorrespondingWithDepth(self, l, r):  # FM:219-247
        out = []
        for f in self.feature:
            if f.start_frame <= l and f.endFrame() >= r:
                a, b = f.feature_per_frame[l - f.start_frame], f.feature_per_frame[r - f.start_frame]
                if a.depth < 0.1 or a.depth > 10 or b.depth < 0.1 or b.depth > 10:
                    continue
                out.append((a.point * a.depth, b.point * b.depth))
        return out

    def setDepth(self, x):  # FM:249-267
        for k, f in enumerate(self.used()):
            f.estimated_depth = 1.0 / x[k] if x[k] != 0 else math.copysign(math.inf, x[k])
            f.solve_flag = 2 if f.estimated_depth < 0 else 1

    def clearDepth(self):  # FM:280-284
        for f in self.feature:
            f.estimated_depth = -1.0

    def removeFailures(self):  # FM:269-278
        self.feature = [f for f in self.feature if f.solve_flag != 2]

    def getDepthVector(self):  # FM:286-302
        return np.array([1.0 / f.estimated_depth for f in self.used()])

    def triangulate(self, Ps, Rs, tic, ric):  # FM:669-724
        for f in self.feature:
            if f.estimated_depth > 0 or len(f.feature_per_frame) < 4:
                continue
            i = f.start_frame
            t0, R0 = Ps[i] + Rs[i] @ tic, Rs[i] @ ric
            rows = []
            for k, fr in enumerate(f.feature_per_frame):
                j = i + k
                t1, R1 = Ps[j] + Rs[j] @ tic, Rs[j] @ ric
                t = R0.T @ (t1 - t0)
                R = R0.T @ R1
                P = np.hstack([R.T, (-R.T @ t).reshape(3, 1)])
                v = fr.point / np.linalg.norm(fr.point)
                rows.append(v[0] * P[2] - v[2] * P[0])
                rows.append(v[1] * P[2] - v[2] * P[1])
            V = np.linalg.svd(np.array(rows), full_matrices=False)[2][-1] if self.svd == "lapack" else smallest_right_singular_vector(rows)
            f.estimated_depth = V[2] / V[3]
            f.estimate_flag = 2
            if f.estimated_depth < 0.1:
                f.estimated_depth, f.estimate_flag = self.INIT_DEPTH, 0

    def triangulateWithDepth(self, Ps, Rs, tic, ric):  # FM:726-799
        for f in self.feature:
            n = len(f.feature_per_frame)
            if n < 4 or f.estimated_depth > 0:
                continue
            s = f.start_frame
            tr, Rr = Ps[s] + Rs[s] @ tic, Rs[s] @ ric
            depths = []
            for i in range(n):
                fi = f.feature_per_frame[i]
                if fi.depth < 0.1 or fi.depth > self.depth_threshold:
                    continue
                t0, R0 = Ps[s + i] + Rs[s + i] @ tic, Rs[s + i] @ ric
                p0 = fi.point * fi.depth
                t2r, R2r = Rr.T @ (t0 - tr), Rr.T @ R0
                for j in range(n):
                    if i == j:
                        continue
                    t1, R1 = Ps[s + j] + Rs[s + j] @ tic, Rs[s + j] @ ric
                    t20, R20 = R0.T @ (t1 - t0), R0.T @ R1
                    pp = R20.T @ p0 - R20.T @ t20
                    res = f.feature_per_frame[j].point[:2] - pp[:2] / pp[2]
                    if math.sqrt(res[0] * res[0] + res[1] * res[1]) < 10.0 / 460:
                        depths.append((R2r @ p0 + t2r)[2])
            if not depths:
                continue
            acc = 0.0
            for d in depths:  # std::accumulate, left to right
                acc += d
            f.estimated_depth = acc / len(depths)
            f.estimate_flag = 1
            if f.estimated_depth < 0.1:
                f.estimated_depth, f.estimate_flag = self.INIT_DEPTH, 0

    def removeOutlier(self, ids):  # FM:801-816
        self.feature = [f for f in self.feature if f.feature_id not in ids]

    def removeBackShiftDepth(self, marg_R, marg_P, new_R, new_P):  # FM:818-856
        keep = []
        for f in self.feature:
            if f.start_frame != 0:
                f.start_frame -= 1
                keep.append(f)
                continue
            uv_i = f.feature_per_frame[0].point
            del f.feature_per_frame[0]
            if len(f.feature_per_frame) < 2:
                continue
            pts_j = new_R.T @ (marg_R @ (uv_i * f.estimated_depth) + marg_P - new_P)
            f.estimated_depth = pts_j[2] if pts_j[2] > 0 else self.INIT_DEPTH
            keep.append(f)
        self.feature = keep

    def removeBack(self):  # FM:858-874
        keep = []
        for f in self.feature:
            if f.start_frame != 0:
                f.start_frame -= 1
            else:
                del f.feature_per_frame[0]
                if not f.feature_per_frame:
                    continue
            keep.append(f)
        self.feature = keep

    def removeFront(self, frame_count):  # FM:914-934
        keep = []
        for f in self.feature:
            if f.start_frame == frame_count:
                f.start_frame -= 1
            else:
                j = self.W - 1 - f.start_frame
                if f.endFrame() >= frame_count - 1:
                    del f.feature_per_frame[j]
                    if not f.feature_per_frame:
                        continue
            keep.append(f)
        self.feature = keep


# ---------------------------------------------------------------- pre-integration holders
class ImuPre:
    def __init__(self, acc0, gyr0, ba, bg, noise):
        self.acc0, self.gyr0, self.ba, self.bg, self.noise = acc0.copy(), gyr0.copy(), ba.copy(), bg.copy(), noise
        self.dt, self.acc, self.gyr = [], [], []
        self._r = None

    def push_back(self, dt, a, g):
        self.dt.append(dt)
        self.acc.append(a.copy())
        self.gyr.append(g.copy())
        self._r = None

    def repropagate(self, ba, bg):
        self.ba, self.bg, self._r = ba.copy(), bg.copy(), None

    def r(self):
        if self._r is None:
            self._r = O.imu_preintegrate(np.array(self.dt), np.array(self.acc).reshape(-1, 3), np.array(self.gyr).reshape(-1, 3), self.acc0, self.gyr0, self.ba,
                                         self.bg, self.noise)
        return self._r


class WheelPre:
    def __init__(self, vel0, gyr0, sx, sy, sw, td, noise):
        self.vel0, self.gyr0, self.lin, self.noise = vel0.copy(), gyr0.copy(), np.array([sx, sy, sw, td], float), noise
        self.dt, self.vel, self.gyr = [], [], []
        self._r = None

    def push_back(self, dt, v, g):
        self.dt.append(dt)
        self.vel.append(v.copy())
        self.gyr.append(g.copy())
        self._r = None

    def r(self):
        if self._r is None:
            self._r = O.wheel_preintegrate(np.array(self.dt), np.array(self.vel).reshape(-1, 3), np.array(self.gyr).reshape(-1, 3), self.vel0, self.gyr0,
                                           self.lin, self.noise)
        return self._r


class ImageFrame:
    def __init__(self, pre, pre_w, points=None):  # initial/initial_alignment.h:25-40
        self.R, self.T, self.pre_integration, self.pre_integration_wheel = np.eye(3), np.zeros(3), pre, pre_w
        self.points, self.is_key_frame = dict(points or {}), False


# ---------------------------------------------------------------- Estimator
class Estimator:
    def __init__(self, cfg=None, tracker=None):
        c = self.cfg = dict(default_cfg(), **(cfg or {}))
        W = self.W = c["window_size"]
        self.f_manager = FeatureManager(c)
        self.tracker = tracker  # oracle_py.Tracker
        self.accBuf, self.gyrBuf, self.wheelVelBuf, self.wheelGyrBuf, self.featureBuf = [], [], [], [], []
        self.prevTime, self.curTime, self.prevTime_wheel, self.curTime_wheel = -1.0, 0.0, -1.0, 0.0
        self.inputImageCnt = 0
        self.Ps = [np.zeros(3) for _ in range(W + 1)]
        self.Vs = [np.zeros(3) for _ in range(W + 1)]
        self.Bas = [np.zeros(3) for _ in range(W + 1)]
        self.Bgs = [np.zeros(3) for _ in range(W + 1)]
        self.Rs = [np.eye(3) for _ in range(W + 1)]
        self.Headers = [0.0] * (W + 1)
        self.tic, self.ric = np.array(c["tic"], float), np.array(c["ric"], float).reshape(3, 3)
        self.tio, self.rio = np.array(c["tio"], float), np.array(c["rio"], float).reshape(3, 3)
        self.RIO = self.rio.copy()
        self.td, self.td_wheel, self.sx, self.sy, self.sw = c["td"], c["td_wheel"], c["sx"], c["sy"], c["sw"]
        self.g = np.array([0, 0, c["g_norm"]], float)
        self.frame_count, self.solver_flag, self.marginalization_flag = 0, INITIAL, MARGIN_OLD
        self.first_imu = self.first_wheel = self.initFirstPoseFlag = False
        self.acc_0, self.gyr_0, self.vel_0_wheel, self.gyr_0_wheel, self.latest_vel_wheel_0 = (np.zeros(3) for _ in range(5))
        self.latest_time = self.latest_time_wheel = 0.0
        self.latest_P, self.latest_V, self.latest_Ba, self.latest_Bg, self.latest_acc_0, self.latest_gyr_0 = (np.zeros(3) for _ in range(6))
        self.latest_P_wheel, self.latest_V_wheel, self.latest_gyr_wheel_0 = (np.zeros(3) for _ in range(3))
        self.latest_Q, self.latest_Q_wheel = np.eye(3), np.eye(3)
        self.latest_sx = self.latest_sy = self.latest_sw = 1.0
        self.pre_integrations = [None] * (W + 1)
        self.pre_integrations_wheel = [None] * (W + 1)
        self.tmp_pre_integration = self.tmp_wheel_pre_integration = None
        self.all_image_frame = {}  # header -> ImageFrame, iterated in sorted key order (std::map)
        self.initial_timestamp = 0.0
        self.wheelanomaly = self.visualstationary = self.wheelstationary = self.imustationary = self.systemstationary = False
        self.varstationary = self.preintegrationstationary = self.is_imu_excited = self.Bas_calibok = False
        self.trajectory = []
        self.dP_imu, self.dP_wheel = np.zeros(3), np.zeros(3)
        self.openExEstimation = self.openExWheelEstimation = self.openIxEstimation = 0
        self.prior = None
        self.imu_noise = np.array([c["acc_n"], c["gyr_n"], c["acc_w"], c["gyr_w"]])
        self.wheel_noise = np.array([c["wheel_vel_n"], c["wheel_gyr_n"]])
        self.predictPts, self.removeIndex = {}, set()
        self.last_summary = None
        self.n_optimizations = 0
        self.sum_of_back = self.sum_of_front = 0
        self.back_R0, self.back_P0 = np.eye(3), np.zeros(3)
        # GNSS (estimator.h:293-331)
        self.GNSSBuf, self.gnss_msg = [], []
        self.sat2ephem, self.sat2time_index = {}, {}
        self.gnss_meas_buf = [[] for _ in range(W + 1)]
        self.sat_track_status = {}
        self.gnss_ready, self.first_optimization, self.lowspeed = False, True, False
        self.diff_t_gnss_local = c["gnss_local_time_diff"]
        self.latest_gnss_iono_params = np.array(c["gnss_iono"], float)
        self.para_rcv_dt, self.para_rcv_ddt = np.zeros((W + 1, 4)), np.zeros(W + 1)
        self.yaw_enu_local, self.anc_ecef, self.R_ecef_enu = 0.0, np.zeros(3), np.eye(3)
        self.ecef_pos, self.enu_pos = np.zeros(3), np.zeros(3)
        self.alignment = None

    # ---- GNSS
    def inputGNSS(self, t, epoch):  # EST:397-404; epoch: list of dicts with the fields of gf_gnss_obs, or raw observations (key "freq", no "sv_pos")
        assert len(epoch) >= 1
        self.GNSSBuf.append((t, [dict(o) for o in epoch]))

    def inputEphem(self, eph):  # EST:1428-1437 (GLONASS ephemerides carry "tau_n")
        idx = self.sat2time_index.setdefault(eph["sat"], {})
        if eph["toe"] not in idx:
            self.sat2ephem.setdefault(eph["sat"], []).append(dict(eph))
            idx[eph["toe"]] = len(self.sat2ephem[eph["sat"]]) - 1

    def inputGNSSTimeDiff(self, t_diff):  # EST:1450-1453
        self.diff_t_gnss_local = t_diff

    def inputIonoParams(self, params):  # EST:1438-1448
        self.latest_gnss_iono_params = np.array(params, float)

    def setGNSSAlignment(self, anc_ecef, yaw_enu_local, rcv_dt, rcv_ddt):
        """what GNSSVIInitializer would return (coarse_localization + yaw_alignment + anchor_refinement, EST:1972-2013); not restated"""
        self.alignment = (np.array(anc_ecef, float), float(yaw_enu_local), np.array(rcv_dt, float), float(rcv_ddt))

    def getGNSSInterval(self, t0, t1):  # EST:476-510
        if not self.GNSSBuf:
            return False
        while self.GNSSBuf and self.GNSSBuf[0][1][0]["time"] < t1 + self.diff_t_gnss_local - 0.1:
            self.GNSSBuf.pop(0)
            if not self.GNSSBuf:
                return False
        self.gnss_msg = self.GNSSBuf.pop(0)[1]
        return True

    def processGNSS(self, gnss_meas):  # EST:1455-1535
        c, valid = self.cfg, []
        for obs in gnss_meas:
            if not 0 <= obs["sys"] <= 3:
                continue
            raw, eph = "sv_pos" not in obs, None
            if raw:   # :1467-1495
                if obs["sat"] not in self.sat2time_index:
                    continue
                ephem_time, ephem_index = 7200.0, None
                for toe in sorted(self.sat2time_index[obs["sat"]]):   # std::map order
                    if abs(toe - obs["time"]) < ephem_time:
                        ephem_time, ephem_index = abs(toe - obs["time"]), self.sat2time_index[obs["sat"]][toe]
                if ephem_time >= 7200.0:
                    continue
                eph = self.sat2ephem[obs["sat"]][ephem_index]
            if obs["psr_std"] > c["gnss_psr_std_thres"] or obs["dopp_std"] > c["gnss_dopp_std_thres"]:
                self.sat_track_status[obs["sat"]] = 0
                continue
            self.sat_track_status[obs["sat"]] = self.sat_track_status.get(obs["sat"], 0) + 1
            if self.sat_track_status[obs["sat"]] < c["gnss_track_num_thres"]:
                continue
            if self.gnss_ready:   # :1515-1526 (the satellite at the reception time)
                sat_ecef = obs["sv_pos"] if not raw else (geph2pos(obs["time"], eph)[0] if "tau_n" in eph else eph2pos(obs["time"], eph)[0])
                if sat_elevation(self.ecef_pos, sat_ecef) < math.radians(c["gnss_elevation_thres"]):
                    continue
            valid.append(obs if not raw else (sat_state(obs, geph=eph) if "tau_n" in eph else sat_state(obs, eph=eph)))
        self.gnss_meas_buf[self.frame_count] = valid

    def _avg_hor_vel(self):
        return np.linalg.norm(np.mean([np.abs(v[0:2]) for v in self.Vs], axis=0))

    def GNSSVIAlign(self):  # EST:1928-2043
        if not self.is_imu_excited and self.solver_flag == INITIAL:
            return False
        if self.gnss_ready:
            return True
        if self._avg_hor_vel() < 0.3:
            return False
        if self.alignment is not None:   # the caller's own initialiser result
            anc, yaw, dt4, ddt = self.alignment
            refined = rough = np.concatenate([anc, dt4])
            self.alignment = None
        else:
            out = self.gnssViInitialize()
            if out is None:
                return False
            refined, rough, yaw, ddt = out
        observed = [k for k in range(4) if rough[3 + k] != 0]
        for i in range(self.W + 1):   # :2015-2036
            self.para_rcv_ddt[i] = ddt
            for k in range(4):
                base = refined[3 + k] if rough[3 + k] != 0 else (refined[3 + observed[0]] if observed else 0.0)
                self.para_rcv_dt[i, k] = base + ddt * i
        self.anc_ecef, self.yaw_enu_local = np.array(refined[0:3], float), float(yaw)
        self.R_ecef_enu = ecef2rotation(self.anc_ecef)
        return True

    def gnssViInitialize(self):
        """GNSSVIInitializer (initial/gnss_vi_initializer.cpp): coarse_localization (= gnss_comm psr_pos on all measurements of the window),
        yaw_alignment, anchor_refinement; None where one of them fails"""
        W, iono = self.W, self.latest_gnss_iono_params
        accum = [o for b in self.gnss_meas_buf for o in b]
        if len(accum) < 4:
            return None
        solve = lambda G, b: -np.linalg.solve(G.T @ G, G.T @ b)
        xyzt = np.zeros(7)
        seen = {o["sys"] for o in accum}
        pins = [k for k in range(4) if k not in seen]
        dxn, it = 1.0, 0
        while it < 10 and dxn > 1e-4:   # psr_pos: Gauss-Newton from the Earth's centre, unobserved clocks pinned to zero
            b, G = psr_res(xyzt, accum, iono)
            for k in pins:
                row = np.zeros(7)
                row[3 + k] = 1.0
                G, b = np.vstack([G, row]), np.append(b, 0.0)
            dx = solve(G, b)
            xyzt = xyzt + dx
            dxn = float(np.linalg.norm(dx))
            it += 1
        if (it == 10 and dxn > 1e-4) or np.linalg.norm(xyzt[0:3]) == 0 or math.isnan(xyzt[0]):
            return None
        for k in range(4):
            if abs(xyzt[3 + k]) < 1:
                xyzt[3 + k] = 0.0
        rough = xyzt.copy()
        anchor = rough[0:3].copy()
        Ree = ecef2rotation(anchor)
        est_yaw, est_ddt, dxn, it = 0.0, 0.0, 1.0, 0
        while it < 10 and dxn > 1e-5:   # yaw_alignment
            G, b = [], []
            cy, sy = math.cos(est_yaw), math.sin(est_yaw)
            for i in range(W + 1):
                v = self.Vs[i]
                ve = Ree @ np.array([cy * v[0] - sy * v[1], sy * v[0] + cy * v[1], v[2]])
                r, Jd = dopp_res([*ve, est_ddt], anchor, self.gnss_meas_buf[i])
                dv = Ree @ np.array([-sy * v[0] - cy * v[1], cy * v[0] - sy * v[1], 0.0])
                for q in range(len(r)):
                    G.append([Jd[q, 0:3] @ dv, 1.0])
                    b.append(r[q])
            dx = solve(np.array(G), np.array(b))
            est_yaw += dx[0]
            est_ddt += dx[1]
            dxn = float(np.linalg.norm(dx))
            it += 1
        yaw = est_yaw
        if yaw > math.pi:
            yaw -= math.floor(est_yaw / (2.0 * math.pi) + 0.5) * (2.0 * math.pi)
        elif yaw < -math.pi:
            yaw -= math.ceil(est_yaw / (2.0 * math.pi) - 0.5) * (2.0 * math.pi)
        dt4 = rough[3:7].copy()
        cy, sy = math.cos(yaw), math.sin(yaw)
        dxn, it = 1.0, 0
        while it < 10 and dxn > 1e-5:   # anchor_refinement
            Gs, bs = [], []
            Ree = ecef2rotation(anchor)
            for i in range(W + 1):
                p = self.Ps[i]
                pe = Ree @ np.array([cy * p[0] - sy * p[1], sy * p[0] + cy * p[1], p[2]]) + anchor
                r, J = psr_res(np.concatenate([pe, dt4 + est_ddt * i]), self.gnss_meas_buf[i], iono)
                Gs.append(J)
                bs.append(r)
            for k in range(4):
                if rough[3 + k] == 0:
                    row = np.zeros((1, 7))
                    row[0, 3 + k] = 1.0
                    Gs.append(row)
                    bs.append(np.zeros(1))
            dx = solve(np.vstack(Gs), np.concatenate(bs))
            anchor = anchor + dx[0:3]
            dt4 = dt4 + dx[3:7]
            dxn = float(np.linalg.norm(dx))
            it += 1
        return np.concatenate([anchor, dt4]), rough, yaw, est_ddt

    def updateGNSSStatistics(self):  # EST:2045-2058
        c, s = math.cos(self.yaw_enu_local), math.sin(self.yaw_enu_local)
        p = self.Ps[self.W]
        self.enu_pos = np.array([c * p[0] - s * p[1], s * p[0] + c * p[1], p[2]])
        self.ecef_pos = self.anc_ecef + self.R_ecef_enu @ self.enu_pos

    def _after_optimization_gnss(self):  # EST:945-956, :1014-1025, :1113-1123
        if not self.cfg["gnss_enable"]:
            return
        if not self.gnss_ready:
            self.gnss_ready = self.GNSSVIAlign()
        if self.gnss_ready:
            self.updateGNSSStatistics()

    # ---- intake
    def inputIMU(self, t, acc, gyr):  # EST:330-346
        self.accBuf.append((t, np.array(acc, float)))
        self.gyrBuf.append((t, np.array(gyr, float)))
        self.fastPredictIMU(t, np.array(acc, float), np.array(gyr, float))   # what pubLatestOdometry publishes at IMU rate (EST:332-335)
        if self.cfg["multiple_thread"] and self.featureBuf:   # the waiting processThread ("wait for imu ...", EST:551-560), made deterministic
            self._drain()

    def inputWheel(self, t, vel, gyr):  # EST:347-360
        self.wheelVelBuf.append((t, np.array(vel, float)))
        self.wheelGyrBuf.append((t, np.array(gyr, float)))
        self.fastPredictWheel(t, np.array(vel, float), np.array(gyr, float))   # EST:363-366; shares latest_vel_wheel_0 with processWheel (quirk 15)
        if self.cfg["multiple_thread"] and self.featureBuf:   # "wait for wheel ...", EST:562-573
            self._drain()

    def inputFeature(self, t, image):  # EST:362-375
        self.featureBuf.append((t, image))
        if self.cfg["multiple_thread"]:
            self._drain()
        else:
            self.processMeasurements()

    def _drain(self):
        while self.processMeasurements():
            pass

    def inputImage(self, t, img, depth=None):  # EST:213-242
        ids, obs = self.tracker.track(t, img, depth)
        self.inputImageCnt += 1
        if self.cfg["multiple_thread"] and self.inputImageCnt % 2 != 0:
            return ids, obs
        self.inputFeature(t, {int(i): obs[k].copy() for k, i in enumerate(ids)})
        return ids, obs

    @staticmethod
    def _interval(a, b, t0, t1):  # getIMUInterval EST:406-439 / getWheelInterval :440-474
        av, bv = [], []
        if not a or not (t1 <= a[-1][0]):
            return av, bv
        while a and a[0][0] <= t0:
            a.pop(0)
            b.pop(0)
        while a and a[0][0] < t1:
            av.append(a.pop(0))
            bv.append(b.pop(0))
        if a:
            av.append(a[0])
            bv.append(b[0])
        return av, bv

    def processMeasurements(self):  # EST:526-709, one frame per call; True when a frame was taken
        if not self.featureBuf:
            return False
        t, image = self.featureBuf[0]
        self.curTime = t + self.td
        self.curTime_wheel = self.curTime - self.td_wheel
        if self.cfg["use_imu"] and not (self.accBuf and t + self.td <= self.accBuf[-1][0]):
            return False
        if self.cfg["use_wheel"] and not (self.wheelVelBuf and t + self.td - self.td_wheel <= self.wheelVelBuf[-1][0]):
            return False
        accV, gyrV = self._interval(self.accBuf, self.gyrBuf, self.prevTime, self.curTime) if self.cfg["use_imu"] else ([], [])
        self.featureBuf.pop(0)
        if self.cfg["gnss_enable"]:
            self.getGNSSInterval(self.prevTime, self.curTime)
        velV, wgyrV = self._interval(self.wheelVelBuf, self.wheelGyrBuf, self.prevTime_wheel, self.curTime_wheel) if self.cfg["use_wheel"] else ([], [])
        if self.cfg["use_imu"]:
            self.dP_imu = np.zeros(3)
            if not self.initFirstPoseFlag:
                self.initFirstIMUPose(accV)
            for i in range(len(accV)):
                dt = accV[i][0] - self.prevTime if i == 0 else (self.curTime - accV[i - 1][0] if i == len(accV) - 1 else accV[i][0] - accV[i - 1][0])
                self.processIMU(accV[i][0], dt, accV[i][1], gyrV[i][1])
        if self.cfg["use_wheel"]:
            self.dP_wheel = np.zeros(3)
            for i in range(len(velV)):
                dt = velV[i][0] - self.prevTime_wheel if i == 0 else (self.curTime_wheel - velV[i - 1][0] if i == len(velV) - 1 else velV[i][0] - velV[i - 1][0])
                self.processWheel(velV[i][0], dt, velV[i][1], wgyrV[i][1])
            if np.linalg.norm(self.dP_wheel - self.dP_imu) > 0.02 and self.cfg["wdetect"]:
                self.wheelanomaly = True
                self.n_wheel_anomaly = getattr(self, "n_wheel_anomaly", 0) + 1   # test bookkeeping
            self.wheelstationary = np.linalg.norm(self.dP_wheel) < 0.001
            self.preintegrationstationary = np.linalg.norm(self.dP_imu) < 0.001
        if self.cfg["gnss_enable"] and self.gnss_msg:
            self.processGNSS(self.gnss_msg)
        self.processImage(image, t)
        self.prevTime, self.prevTime_wheel = self.curTime, self.curTime_wheel
        if self.solver_flag == NON_LINEAR or self.is_imu_excited:   # pubOdometry, EST:679 -> visualization.cpp:287-357 (what vio.txt receives)
            W = self.W
            self.trajectory.append((t, np.array(self.Ps[W], float), np.array(self.Rs[W], float)))
        return True

    def initFirstIMUPose(self, accV):  # EST:710-731
        self.initFirstPoseFlag = True
        aver = np.zeros(3)
        for _, a in accV:
            aver = aver + a
        aver = aver / len(accV)
        R0 = g2R(aver)
        yaw = R2ypr(R0)[0]
        R0 = ypr2R([-yaw, 0, 0]) @ R0
        self.Rs[0] = R0 @ self.RIO

    def processIMU(self, t, dt, acc, gyr):  # EST:743-785
        if not self.first_imu:
            self.first_imu = True
            self.acc_0, self.gyr_0 = acc.copy(), gyr.copy()
        fc = self.frame_count
        if self.pre_integrations[fc] is None:
            self.pre_integrations[fc] = ImuPre(self.acc_0, self.gyr_0, self.Bas[fc], self.Bgs[fc], self.imu_noise)
        if fc != 0:
            self.pre_integrations[fc].push_back(dt, acc, gyr)
            self.tmp_pre_integration.push_back(dt, acc, gyr)
            un_acc_0 = self.Rs[fc] @ (self.acc_0 - self.Bas[fc]) - self.g
            un_acc_1 = self.Rs[fc] @ (acc - self.Bas[fc]) - self.g
            un_acc = 0.5 * (un_acc_0 + un_acc_1)
            self.dP_imu = self.dP_imu + dt * self.Vs[fc] + 0.5 * dt * dt * un_acc
        self.acc_0, self.gyr_0 = acc.copy(), gyr.copy()

    # ---- IMU- / wheel-rate propagation of the newest state for the odometry publishers (EST:4014-4028, :4079-4093, :4141-4198).  The members are
    # uninitialised in the reference until the first updateLatestStates (UB); zero / identity here.
    def fastPredictIMU(self, t, acc, gyr):  # EST:4014-4028
        dt = t - self.latest_time
        self.latest_time = t
        un_acc_0 = self.latest_Q @ (self.latest_acc_0 - self.latest_Ba) - self.g
        un_gyr = 0.5 * (self.latest_gyr_0 + gyr) - self.latest_Bg
        self.latest_Q = self.latest_Q @ deltaQ_R(un_gyr * dt)
        un_acc_1 = self.latest_Q @ (acc - self.latest_Ba) - self.g
        un_acc = 0.5 * (un_acc_0 + un_acc_1)
        self.latest_P = self.latest_P + dt * self.latest_V + 0.5 * dt * dt * un_acc
        self.latest_V = self.latest_V + dt * un_acc
        self.latest_acc_0, self.latest_gyr_0 = acc.copy(), gyr.copy()

    def fastPredictWheel(self, t, vel, gyr):  # EST:4079-4093 (un_gyr is built from the IMU's latest_gyr_0, as written there)
        dt = t - self.latest_time_wheel
        self.latest_time_wheel = t
        un_gyr = 0.5 * self.latest_sw * (self.latest_gyr_0 + gyr)
        un_vel_0 = self.latest_Q_wheel @ self.latest_vel_wheel_0
        sv = np.array([self.latest_sx, self.latest_sy, 1.0])
        self.latest_Q_wheel = self.latest_Q_wheel @ deltaQ_R(un_gyr * dt)
        self.latest_V_wheel = 0.5 * sv * (self.latest_Q_wheel @ vel + un_vel_0)
        self.latest_P_wheel = self.latest_P_wheel + dt * self.latest_V_wheel
        self.latest_vel_wheel_0, self.latest_gyr_wheel_0 = vel.copy(), gyr.copy()

    def updateLatestStates(self):  # EST:4141-4198
        fc = self.frame_count
        self.latest_time = self.Headers[fc] + self.td
        self.latest_P, self.latest_Q, self.latest_V = self.Ps[fc].copy(), self.Rs[fc].copy(), self.Vs[fc].copy()
        self.latest_Ba, self.latest_Bg = self.Bas[fc].copy(), self.Bgs[fc].copy()
        self.latest_acc_0, self.latest_gyr_0 = self.acc_0.copy(), self.gyr_0.copy()
        for (t, a), (_, w) in zip(list(self.accBuf), list(self.gyrBuf)):
            self.fastPredictIMU(t, a, w)
        self.latest_time_wheel = self.Headers[fc] + self.td - self.td_wheel
        self.latest_Q_wheel = self.Rs[fc] @ self.rio
        self.latest_P_wheel = self.Rs[fc] @ self.tio + self.Ps[fc]
        self.latest_sx, self.latest_sy, self.latest_sw = self.sx, self.sy, self.sw
        self.latest_vel_wheel_0, self.latest_gyr_wheel_0 = self.vel_0_wheel.copy(), self.gyr_0_wheel.copy()
        for (t, v), (_, w) in zip(list(self.wheelVelBuf), list(self.wheelGyrBuf)):
            self.fastPredictWheel(t, v, w)

    def processWheel(self, t, dt, vel, gyr):  # EST:786-842
        if not self.first_wheel:
            self.first_wheel = True
            self.vel_0_wheel, self.gyr_0_wheel = vel.copy(), gyr.copy()
        fc = self.frame_count
        if self.pre_integrations_wheel[fc] is None:
            self.pre_integrations_wheel[fc] = WheelPre(self.vel_0_wheel, self.gyr_0_wheel, self.sx, self.sy, self.sw, self.td_wheel, self.wheel_noise)
        if fc != 0:
            self.pre_integrations_wheel[fc].push_back(dt, vel, gyr)
            self.tmp_wheel_pre_integration.push_back(dt, vel, gyr)
            self.latest_time_wheel = t
            un_gyr = 0.5 * (self.gyr_0_wheel + gyr)
            un_vel_0 = self.Rs[fc] @ self.latest_vel_wheel_0
            if not self.systemstationary:
                self.Rs[fc] = self.Rs[fc] @ deltaQ_R(un_gyr * dt)
                self.Vs[fc] = 0.5 * (self.Rs[fc] @ vel + un_vel_0)
                self.Ps[fc] = self.Ps[fc] + dt * self.Vs[fc]
            else:
                self.Vs[fc] = np.zeros(3)
            self.latest_vel_wheel_0, self.latest_gyr_wheel_0 = vel.copy(), gyr.copy()
            V = self.Vs[fc]
            self.dP_wheel = self.dP_wheel + np.array([-dt * V[1], dt * V[0], -dt * V[2]])
        self.vel_0_wheel, self.gyr_0_wheel = vel.copy(), gyr.copy()

    # ---- votes
    def _frames(self):
        return [self.all_image_frame[k] for k in sorted(self.all_image_frame)]

    def _gvar(self):
        fr = self._frames()[1:]
        n = len(fr)
        with np.errstate(all="ignore"):
            gs = [f.pre_integration.r()["delta_v"] / f.pre_integration.r()["sum_dt"] for f in fr]
            s = np.zeros(3)
            for v in gs:
                s = s + v
            aver = s * 1.0 / n if n else np.full(3, np.nan)
            var = 0.0
            for v in gs:
                var += float((v - aver) @ (v - aver))
            var = math.sqrt(var / n) if n else math.nan
        return aver, var

    def checkimu(self):  # EST:2173-2216
        _, var = self._gvar()
        self.varstationary = bool(var < 0.1)

    def checkvisual(self):  # EST:2218-2274
        for i in range(self.W):
            corres = self.f_manager.getCorrespondingWithDepth(i, self.W)
            if len(corres) > 20:
                s = 0.0
                for a, b in corres:
                    d = a[:2] / a[2] - b[:2] / b[2]
                    s = s + math.sqrt(d[0] * d[0] + d[1] * d[1])
                if 1.0 * s / len(corres) * 460 < 0.5:
                    return True
                self.visualstationary = False
            self.visualstationary = False
        return False

    # ---- initialisation shortcuts
    def solveGyroscopeBias(self):  # initial_aligment.cpp:14-47
        fr = self._frames()
        A, b = np.zeros((3, 3)), np.zeros(3)
        for fi, fj in zip(fr[:-1], fr[1:]):
            p = fj.pre_integration.r()
            q_ij = R_to_quat(fi.R.T @ fj.R)
            tA = p["jacobian"].reshape(15, 15)[3:6, 12:15]
            dq = p["delta_q"]
            dq_inv = np.array([dq[0], -dq[1], -dq[2], -dq[3]]) / (dq @ dq)
            tb = 2 * qmul(dq_inv, q_ij)[1:4]
            A += tA.T @ tA
            b += tA.T @ tb
        delta_bg = np.linalg.solve(A, b)
        for i in range(self.W + 1):
            self.Bgs[i] = self.Bgs[i] + delta_bg
        for fj in fr[1:]:
            fj.pre_integration.repropagate(np.zeros(3), self.Bgs[0])

    def initialStructure(self):  # EST:1557-1682; the SfM branch in _initialStructureSfM
        aver_g, var = self._gvar()
        if not (var < 0.35):
            self.is_imu_excited = True
        G = np.array([0, 0, self.cfg["g_norm"]], float)
        if not self.Bas_calibok and self.systemstationary and self.solver_flag != NON_LINEAR:
            tmp = aver_g - g2R(aver_g).T @ G
            self.Bas = [tmp.copy() for _ in range(self.W + 1)]
            self.Bas_calibok = True
            self.solveGyroscopeBias()
            return True
        if not self.Bas_calibok and self.is_imu_excited:
            tmp = aver_g - g2R(aver_g).T @ G
            self.Bas = [tmp.copy() for _ in range(self.W + 1)]
            self.Bas_calibok = True
            self.solveGyroscopeBias()
            R0 = g2R(self.g)
            ypr = R2ypr(R0 @ self.Rs[0])
            R0 = ypr2R(-ypr) @ R0
            self.g = R0 @ self.g
            for i in range(self.frame_count + 1):
                self.Ps[i], self.Rs[i], self.Vs[i] = R0 @ self.Ps[i], R0 @ self.Rs[i], R0 @ self.Vs[i]
            return True
        return self._initialStructureSfM(aver_g)

    def relativePoseWithDepth(self):  # EST:2087-2124 with MotionEstimator::solveRelativeRT_PNP (initial/solve_5pts.cpp:244-277)
        W = self.W

        def rt_pnp(corres):
            X = [a for a, b in corres if a[2] > 0 and b[2] > 0]
            uv = [b[:2] / b[2] for a, b in corres if a[2] > 0 and b[2] > 0]
            r = IO.solve_pnp_ransac(np.array(X), np.array(uv))
            if r is None:            # cv::solvePnPRansac left rvec / tvec empty: the reference would read unset matrices; refused here
                raise RuntimeError("solvePnPRansac found no model")
            rv, tv = r[0], r[1]
            rota = ypr2R_deg_free(rv)   # Sophus::SO3(rot_x, rot_y, rot_z) of the non-templated Sophus: Rx(rot_x) Ry(rot_y) Rz(rot_z), NOT the Rodrigues vector
            return rota.T, -rota.T @ tv

        for i in range(W):
            corres = self.f_manager.getCorrespondingWithDepth(i, W)
            if len(corres) > 20:
                R, T = rt_pnp(corres)   # solveRelativeRT_PNP returns true unconditionally: the parallax test below it is never reached
                return R, T, i
        return None

    def _initialStructureSfM(self, aver_g):  # EST:1684-1847
        fc = self.frame_count
        sfm_f = []
        for f in self.f_manager.feature:
            sf = IO.SFMFeature(f.feature_id)
            for k, fpf in enumerate(f.feature_per_frame):
                sf.observation.append((f.start_frame + k, fpf.point[:2].copy()))
                sf.observation_depth.append((f.start_frame + k, fpf.depth))
            sfm_f.append(sf)
        rel = self.relativePoseWithDepth()
        if rel is None:
            return False
        relative_R, relative_T, l = rel
        r = IO.construct_with_depth(fc + 1, l, relative_R, relative_T, sfm_f)
        if r is None:
            self.marginalization_flag = MARGIN_OLD
            return False
        Q, T, tracked = r
        self.init_debug = dict(l=l, relative_R=relative_R, relative_T=relative_T, Q=np.array(Q), T=np.array(T), n_tracked=len(tracked))
        ric = self.ric
        i = 0
        for key in sorted(self.all_image_frame):   # solve pnp for all frame, :1749-1813
            fr = self.all_image_frame[key]
            if key == self.Headers[i]:
                fr.is_key_frame = True
                fr.R = quat_to_R(*Q[i]) @ ric.T
                fr.T = T[i].copy()
                i += 1
                continue
            if key > self.Headers[i]:
                i += 1
            qi = np.array([Q[i][0], -Q[i][1], -Q[i][2], -Q[i][3]]) / float(Q[i] @ Q[i])
            R_initial = quat_to_R(*qi)
            P_initial = -R_initial @ T[i]
            fr.is_key_frame = False
            X, uv = [], []
            for fid in sorted(fr.points):
                if fid in tracked:
                    X.append(tracked[fid]); uv.append(np.asarray(fr.points[fid][:2], float))
            if len(X) < 6:
                return False
            rt = IO.solve_pnp_iterative(np.array(X), np.array(uv), (IO.rodrigues_inv(R_initial), P_initial))
            if rt is None:
                return False
            R_pnp = IO.rodrigues(rt[0]).T
            fr.R = R_pnp @ ric.T
            fr.T = R_pnp @ (-rt[1])
        if self.visualInitialAlign():
            G = np.array([0, 0, self.cfg["g_norm"]], float)
            tmp = aver_g - g2R(aver_g).T @ G
            self.Bas = [tmp.copy() for _ in range(self.W + 1)]
            return True
        return False

    def visualInitialAlign(self):  # EST:1849-1926, VisualIMUAlignment initial_aligment.cpp:640-653
        c, fc = self.cfg, self.frame_count
        self.solveGyroscopeBias()
        if not c["depth"]:
            raise NotImplementedError("monocular alignment (LinearAlignment / LinearAlignmentWithWheel) is outside the RGB-D scope")
        r = IO.linear_alignment(self._frames(), self.tic, c["g_norm"], bool(c["use_wheel"]), self.rio, self.tio)
        if r is None:
            return False
        self.g, x = r
        self.init_debug.update(g_c0=self.g.copy(), x=x.copy())
        for i in range(fc + 1):
            fr = self.all_image_frame[self.Headers[i]]
            self.Ps[i], self.Rs[i] = fr.T.copy(), fr.R.copy()
            fr.is_key_frame = True
        s = float(x[-1])
        for i in range(self.W + 1):
            self.pre_integrations[i].repropagate(np.zeros(3), self.Bgs[i])
        P0 = self.Ps[0].copy()
        for i in range(fc, -1, -1):
            self.Ps[i] = s * self.Ps[i] - self.Rs[i] @ self.tic - (s * P0 - self.Rs[0] @ self.tic)
        kv = -1
        for fr in self._frames():
            if fr.is_key_frame:
                kv += 1
                self.Vs[kv] = fr.R @ x[3 * kv:3 * kv + 3]
        R0 = g2R(self.g)
        yaw = R2ypr(R0 @ self.Rs[0])[0]
        R0 = ypr2R(np.array([-yaw, 0, 0])) @ R0
        self.g = R0 @ self.g
        for i in range(fc + 1):
            self.Ps[i], self.Rs[i], self.Vs[i] = R0 @ self.Ps[i], R0 @ self.Rs[i], R0 @ self.Vs[i]
        self.f_manager.clearDepth()
        self.f_manager.triangulateWithDepth(self.Ps, self.Rs, self.tic, self.ric)
        self.f_manager.triangulate(self.Ps, self.Rs, self.tic, self.ric)
        return True

    # ---- processImage
    def processImage(self, image, header):  # EST:843-1163
        fc = self.frame_count
        self.marginalization_flag = MARGIN_OLD if self.f_manager.addFeatureCheckParallax(fc, image, self.td) else MARGIN_SECOND_NEW
        self.Headers[fc] = header
        if header not in self.all_image_frame:  # std::map::insert keeps an existing key
            self.all_image_frame[header] = ImageFrame(self.tmp_pre_integration, self.tmp_wheel_pre_integration, image)
        self.tmp_pre_integration = ImuPre(self.acc_0, self.gyr_0, self.Bas[fc], self.Bgs[fc], self.imu_noise)
        self.tmp_wheel_pre_integration = WheelPre(self.vel_0_wheel, self.gyr_0_wheel, self.sx, self.sy, self.sw, self.td_wheel, self.wheel_noise)
        self.checkimu()
        self.imustationary = self.varstationary and self.preintegrationstationary
        if self.checkvisual():
            self.visualstationary = True
        self.systemstationary = bool((self.imustationary and self.wheelstationary) or (self.visualstationary and self.wheelstationary)
                                     or (self.imustationary and self.visualstationary))
        self.predictPts, self.removeIndex = {}, set()
        if self.solver_flag == INITIAL:
            if fc == self.W:
                for i, f in enumerate(self._frames()):
                    if i <= self.W:
                        f.R, f.T = self.Rs[i].copy(), self.Ps[i].copy()
                result = False
                if header - self.initial_timestamp > 0.1:
                    result = self.initialStructure()
                    self.initial_timestamp = header
                if result:
                    self.solveGyroscopeBias()
                    for i in range(self.W + 1):
                        self.pre_integrations[i].repropagate(np.zeros(3), self.Bgs[i])
                    self.solver_flag = NON_LINEAR
                self.optimization()
                if result:
                    self._after_optimization_gnss()
                self.slideWindow()
                if not result:
                    self.updateLatestStates()   # EST:1030-1033: only on the branch without a successful initialisation
            if self.frame_count < self.W:
                self.frame_count += 1
                k = self.frame_count
                self.Ps[k], self.Vs[k], self.Rs[k], self.Bas[k], self.Bgs[k] = (x[k - 1].copy() for x in (self.Ps, self.Vs, self.Rs, self.Bas, self.Bgs))
        else:
            self.f_manager.triangulateWithDepth(self.Ps, self.Rs, self.tic, self.ric)
            self.f_manager.triangulate(self.Ps, self.Rs, self.tic, self.ric)
            removeIndex = set()
            if self.cfg["use_mcc"]:
                self.movingConsistencyCheckW(removeIndex)
                self.f_manager.removeOutlier(removeIndex)
            self.optimization()
            self._after_optimization_gnss()
            if not self.cfg["use_mcc"]:
                inner = set()
                self.movingConsistencyCheckW(inner)
                self.f_manager.removeOutlier(inner)
            if not self.cfg["multiple_thread"]:
                self.removeIndex = removeIndex
                self.predictPtsInNextFrame()
                if self.tracker is not None:
                    self.tracker.remove_outliers(sorted(removeIndex))
                    pid = sorted(self.predictPts)
                    self.tracker.set_prediction(pid, np.array([self.predictPts[i] for i in pid], float).reshape(-1, 3))
            self.slideWindow()
            self.f_manager.removeFailures()
            self.updateLatestStates()   # EST:1161

    # ---- optimisation
    def vector2double(self):  # EST:2276-2353
        W = self.W
        st = {"para_Pose": np.zeros((W + 1, 7)), "para_SpeedBias": np.zeros((W + 1, 9))}
        for i in range(W + 1):
            q = R_to_quat(self.Rs[i])
            st["para_Pose"][i] = [*self.Ps[i], q[1], q[2], q[3], q[0]]
            st["para_SpeedBias"][i] = [*self.Vs[i], *self.Bas[i], *self.Bgs[i]]
        q = R_to_quat(self.ric)
        st["para_Ex_Pose"] = np.array([*self.tic, q[1], q[2], q[3], q[0]])
        q = R_to_quat(self.rio)
        st["para_Ex_Pose_wheel"] = np.array([*self.tio, q[1], q[2], q[3], q[0]])
        st["para_Ix"] = np.array([self.sx, self.sy, self.sw], float)
        st["para_Td"], st["para_Td_wheel"] = np.array([self.td], float), np.array([self.td_wheel], float)
        with np.errstate(all="ignore"):
            st["para_Feature"] = self.f_manager.getDepthVector()
        if self.cfg["gnss_enable"]:   # para_rcv_dt / para_rcv_ddt ARE the state (estimator.h:305-308); yaw and anchor are copied when gnss_ready (:2347-2352)
            st["para_rcv_dt"], st["para_rcv_ddt"] = self.para_rcv_dt.reshape(-1).copy(), self.para_rcv_ddt.copy()
            st["para_yaw_enu_local"], st["para_anc_ecef"] = np.array([self.yaw_enu_local]), self.anc_ecef.copy()
        return st

    def double2vector(self, w):  # EST:2440-2569
        W = self.W
        R, P, V, Ba, Bg = O.double2vector(W, self.Rs[0], self.Ps[0], w["para_Pose"], w["para_SpeedBias"])
        for i in range(W + 1):
            self.Rs[i], self.Ps[i], self.Vs[i] = R[9 * i:9 * i + 9].reshape(3, 3).copy(), P[3 * i:3 * i + 3].copy(), V[3 * i:3 * i + 3].copy()
            self.Bas[i], self.Bgs[i] = Ba[3 * i:3 * i + 3].copy(), Bg[3 * i:3 * i + 3].copy()
        e = w["para_Ex_Pose"]
        self.tic, self.ric = e[0:3].copy(), quat_to_R(e[6], e[3], e[4], e[5])
        if self.cfg["use_wheel"]:
            e = w["para_Ex_Pose_wheel"]
            q = np.array([e[6], e[3], e[4], e[5]])
            q = q / np.linalg.norm(q)
            self.tio, self.rio = e[0:3].copy(), quat_to_R(*q)
            self.sx, self.sy, self.sw = (float(x) for x in w["para_Ix"])
            self.td_wheel = float(w["para_Td_wheel"][0])
        with np.errstate(all="ignore"):
            self.f_manager.setDepth(w["para_Feature"])
        self.td = float(w["para_Td"][0])
        if self.gnss_ready:  # :2562-2568 (the clock states live in para_rcv_dt / para_rcv_ddt themselves)
            self.para_rcv_dt, self.para_rcv_ddt = w["para_rcv_dt"].reshape(-1, 4).copy(), w["para_rcv_ddt"].copy()
            self.yaw_enu_local, self.anc_ecef = float(w["para_yaw_enu_local"][0]), w["para_anc_ecef"].copy()
            self.R_ecef_enu = ecef2rotation(self.anc_ecef)

    def build_window(self):  # EST:2890-3297
        c, W, fc = self.cfg, self.W, self.frame_count
        w = gw.Window()
        w.update(self.vector2double())
        w["W"], w["G"], w["vis_sqrt_info"] = W, self.g.copy(), c["focal_length"] / 1.5
        moving = np.linalg.norm(self.Vs[0]) > 0.2
        if (c["estimate_extrinsic"] and fc == W and moving) or self.openExEstimation:
            self.openExEstimation = 1
        else:
            w["fix_ex_pose"] = 1
        wheel_on = c["use_wheel"] and not c["only_initial_with_wheel"]
        if wheel_on:
            if (c["estimate_wheel_extrinsic"] and fc == W and moving) or self.openExWheelEstimation:
                self.openExWheelEstimation = 1
            else:
                w["fix_ex_wheel"] = 1
            if (c["estimate_wheel_intrinsic"] and fc == W and moving) or self.openIxEstimation:
                self.openIxEstimation = 1
            else:
                w["fix_ix"] = 1
        else:
            w["fix_ex_wheel"] = w["fix_ix"] = 1
        # PoseSubsetParameterization masks (EST:2969-2985, :3010-3026; YAML values parameters.cpp:394-420, :280-306)
        w["ex_pose_mask"] = subset_mask(c.get("extrinsic_type", 0)) if c["estimate_extrinsic"] else 0
        w["ex_wheel_mask"] = subset_mask(c.get("extrinsic_type_wheel", 0)) if (wheel_on and c["estimate_wheel_extrinsic"]) else 0
        still = np.linalg.norm(self.Vs[0]) < 0.2
        w["fix_td"] = 1 if (not c["estimate_td"] or still) else 0
        w["fix_td_wheel"] = 1 if (not c["estimate_td_wheel"] or still) else 0
        if self.gnss_ready:  # :2904-2941
            self.lowspeed = bool(self._avg_hor_vel() < 0.3)
        if self.first_optimization and c["gnss_enable"]:  # :2943-2951
            w["has_anchor"], w["anchor_value"] = 1, w["para_Pose"][0].copy()
            self.first_optimization = False
        if c["gnss_enable"]:
            gn = {k: [] for k in ("frame", "lower", "sys", "ratio", "data")}
            if self.gnss_ready:  # :3178-3210; the factors of frame 0 also feed the MARGIN_OLD marginalisation (:3398-3418), lowspeed or not
                for i in range(W + 1):
                    for o in self.gnss_meas_buf[i]:
                        obs_local_ts = o["time"] - self.diff_t_gnss_local
                        if self.Headers[i] > obs_local_ts:
                            lower = 0 if i == 0 else i - 1
                        else:
                            lower = W - 1 if i == W else i
                        lower_ts, upper_ts = self.Headers[lower], self.Headers[lower + 1]
                        gn["frame"].append(i)
                        gn["lower"].append(lower)
                        gn["sys"].append(o["sys"])
                        gn["ratio"].append((upper_ts - obs_local_ts) / (upper_ts - lower_ts))
                        gn["data"].append([*o["sv_pos"], *o["sv_vel"], o["svdt"], o["svddt"], o["tgd"], o["pr_uura"], o["dp_uura"], o["psr"], o["dopp"],
                                           o["wavelength"], o["tow"], 0.0])
            w["gnss_enabled"], w["gnss_lowspeed"] = int(self.gnss_ready), int(self.lowspeed)
            w["gnss_frame"], w["gnss_lower"], w["gnss_sys"] = (np.array(gn[k], np.int32) for k in ("frame", "lower", "sys"))
            w["gnss_ratio"], w["gnss_data"] = np.array(gn["ratio"], float), np.array(gn["data"], float).reshape(-1)
            w["gnss_iono"], w["gnss_headers"], w["gnss_ddt_weight"] = self.latest_gnss_iono_params.copy(), np.array(self.Headers, float), 1.0 / c["gnss_ddt_sigma"]
        imu = {k: [] for k in ("i", "sum_dt", "delta_p", "delta_q", "delta_v", "lin_ba", "lin_bg", "jacobian", "covariance")}
        for i in range(fc):
            p = self.pre_integrations[i + 1]
            r = p.r()
            if r["sum_dt"] > 10.0:
                continue
            imu["i"].append(i)
            imu["lin_ba"].append(p.ba)
            imu["lin_bg"].append(p.bg)
            for k in ("sum_dt", "delta_p", "delta_q", "delta_v", "jacobian", "covariance"):
                imu[k].append(r[k])
        wh = {k: [] for k in ("i", "sum_dt", "delta_p", "delta_q", "jacobian", "covariance", "lin", "lin_vel", "lin_gyr", "vel_1", "gyr_1")}
        if wheel_on:
            for i in range(fc):
                p = self.pre_integrations_wheel[i + 1]
                r = p.r()
                if r["sum_dt"] > 10.0 or (c["wdetect"] and self.wheelanomaly):
                    continue
                wh["i"].append(i)
                wh["lin"].append(p.lin)
                wh["lin_vel"].append(p.vel0)
                wh["lin_gyr"].append(p.gyr0)
                wh["vel_1"].append(p.vel[-1] if p.vel else p.vel0)
                wh["gyr_1"].append(p.gyr[-1] if p.gyr else p.gyr0)
                for k in ("sum_dt", "delta_p", "delta_q", "jacobian", "covariance"):
                    wh[k].append(r[k])
        if self.systemstationary and c["stationary_detect"]:
            w["para_SpeedBias"][:, 0:3] = 0
            w["fix_poses"] = 1
        vis = {k: [] for k in ("feature", "i", "j", "pts_i", "pts_j", "vel_i", "vel_j", "td_i", "td_j")}
        fixed = []
        for k, f in enumerate(self.f_manager.used()):
            fixed.append(1 if f.estimate_flag == 1 else 0)
            f0 = f.feature_per_frame[0]
            for d, fr in enumerate(f.feature_per_frame):
                if d == 0:
                    continue
                vis["feature"].append(k)
                vis["i"].append(f.start_frame)
                vis["j"].append(f.start_frame + d)
                vis["pts_i"].append(f0.point)
                vis["pts_j"].append(fr.point)
                vis["vel_i"].append(f0.velocity)
                vis["vel_j"].append(fr.velocity)
                vis["td_i"].append(f0.cur_td)
                vis["td_j"].append(fr.cur_td)
        flat = lambda a: np.array(a, float).reshape(-1) if len(a) else np.zeros(0)
        for k, v in imu.items():
            w["imu_" + k] = np.array(v, np.int32) if k == "i" else flat(v)
        for k, v in wh.items():
            w["wh_" + k] = np.array(v, np.int32) if k == "i" else flat(v)
        for k, v in vis.items():
            w["vis_" + k] = np.array(v, np.int32) if k in ("feature", "i", "j") else flat(v)
        w["feature_fixed"] = np.array(fixed, np.uint8)
        w.set_prior(self.prior)
        return w

    def optimization(self):  # EST:2890-3636
        w = self.build_window()
        self.last_window = w.copy()
        self.last_summary = O.ba_solve(w, self.cfg["num_iterations"])
        self.n_optimizations += 1
        if self.cfg["gnss_enable"]:  # :3322-3325
            while w["para_yaw_enu_local"][0] > math.pi:
                w["para_yaw_enu_local"][0] -= 2.0 * math.pi
            while w["para_yaw_enu_local"][0] < -math.pi:
                w["para_yaw_enu_local"][0] += 2.0 * math.pi
        self.double2vector(w)
        if self.frame_count < self.W:
            self.wheelanomaly = False
            return
        run = self.marginalization_flag == MARGIN_OLD
        if not run:
            run = self.prior is not None and gw.bid(gw.POSE, self.W - 1) in list(self.prior["block_id"])
        if run:
            w.update(self.vector2double())
            w.finalize()
            self.last_marg_window = w.copy()   # test bookkeeping
            self.prior = O.ba_marginalize(w, self.marginalization_flag)
        self.wheelanomaly = False

    # ---- window bookkeeping
    def slideWindow(self):  # EST:3638-3790
        W, fc = self.W, self.frame_count
        if self.marginalization_flag == MARGIN_OLD:
            t_0 = self.Headers[0]
            self.back_R0, self.back_P0 = self.Rs[0].copy(), self.Ps[0].copy()
            if fc != W:
                return
            for lst in (self.Headers, self.Rs, self.Ps, self.Vs, self.Bas, self.Bgs, self.pre_integrations):
                lst.append(lst.pop(0))  # the chain of swaps rotates the oldest entry to the back
            if self.cfg["use_wheel"]:
                self.pre_integrations_wheel.append(self.pre_integrations_wheel.pop(0))
            if self.cfg["gnss_enable"]:  # :3674-3681, :3700-3704: buffers swap along, the clock states are copied down (the newest keeps its value)
                self.gnss_meas_buf.append(self.gnss_meas_buf.pop(0))
                self.gnss_meas_buf[W] = []
                self.para_rcv_dt[0:W] = self.para_rcv_dt[1:W + 1].copy()
                self.para_rcv_ddt[0:W] = self.para_rcv_ddt[1:W + 1].copy()
            self.Headers[W] = self.Headers[W - 1]
            for lst in (self.Ps, self.Rs, self.Vs, self.Bas, self.Bgs):
                lst[W] = lst[W - 1].copy()
            self.pre_integrations[W] = ImuPre(self.acc_0, self.gyr_0, self.Bas[W], self.Bgs[W], self.imu_noise)
            if self.cfg["use_wheel"]:
                self.pre_integrations_wheel[W] = WheelPre(self.vel_0_wheel, self.gyr_0_wheel, self.sx, self.sy, self.sw, self.td_wheel, self.wheel_noise)
            if t_0 in self.all_image_frame:
                self.all_image_frame[t_0].pre_integration = self.all_image_frame[t_0].pre_integration_wheel = None
                for k in [k for k in self.all_image_frame if k < t_0]:
                    del self.all_image_frame[k]
            self.sum_of_back += 1  # slideWindowOld EST:3804-3837
            if self.solver_flag == NON_LINEAR:
                R0, R1 = self.back_R0 @ self.ric, self.Rs[0] @ self.ric
                P0, P1 = self.back_P0 + self.back_R0 @ self.tic, self.Ps[0] + self.Rs[0] @ self.tic
                self.f_manager.removeBackShiftDepth(R0, P0, R1, P1)
            else:
                self.f_manager.removeBack()
        else:
            if fc != W:
                return
            self.Headers[fc - 1], self.Ps[fc - 1], self.Rs[fc - 1] = self.Headers[fc], self.Ps[fc].copy(), self.Rs[fc].copy()
            src, dst = self.pre_integrations[fc], self.pre_integrations[fc - 1]
            if src is not None and dst is not None:  # (always true once IMU data arrived for the newest frame)
                for dt, a, g_ in zip(src.dt, src.acc, src.gyr):
                    dst.push_back(dt, a, g_)
            self.Vs[fc - 1], self.Bas[fc - 1], self.Bgs[fc - 1] = self.Vs[fc].copy(), self.Bas[fc].copy(), self.Bgs[fc].copy()
            self.pre_integrations[W] = ImuPre(self.acc_0, self.gyr_0, self.Bas[W], self.Bgs[W], self.imu_noise)
            if self.cfg["use_wheel"]:
                src, dst = self.pre_integrations_wheel[fc], self.pre_integrations_wheel[fc - 1]
                if src is not None and dst is not None:
                    for dt, v, g_ in zip(src.dt, src.vel, src.gyr):
                        dst.push_back(dt, v, g_)
                self.pre_integrations_wheel[W] = WheelPre(self.vel_0_wheel, self.gyr_0_wheel, self.sx, self.sy, self.sw, self.td_wheel, self.wheel_noise)
            self.gnss_meas_buf[fc - 1] = self.gnss_meas_buf[fc]  # :3761-3768
            self.gnss_meas_buf[fc] = []
            self.para_rcv_dt[fc - 1] = self.para_rcv_dt[fc].copy()
            self.para_rcv_ddt[fc - 1] = self.para_rcv_ddt[fc]
            self.sum_of_front += 1
            self.f_manager.removeFront(fc)

    def _reproj(self, i, j, depth, uvi, uvj):  # EST:3899-3919
        pts_w = self.Rs[i] @ (self.ric @ (depth * uvi) + self.tic) + self.Ps[i]
        pts_cj = self.ric.T @ (self.Rs[j].T @ (pts_w - self.Ps[j]) - self.tic)
        r = pts_cj[:2] / pts_cj[2] - uvj[:2]
        return math.sqrt(r[0] * r[0] + r[1] * r[1]), np.linalg.norm(pts_cj - uvj) / depth

    def movingConsistencyCheckW(self, removeIndex):  # EST:3955-3995
        for f in self.f_manager.feature:
            n = len(f.feature_per_frame)
            if not (n >= 2 and f.start_frame < self.W - 2):
                continue
            depth = f.estimated_depth
            if depth < 0:
                continue
            err = err3 = 0.0
            cnt = 0
            for d in range(1, n):
                e2, e3 = self._reproj(f.start_frame, f.start_frame + d, depth, f.feature_per_frame[0].point, f.feature_per_frame[d].point)
                err += e2
                err3 += e3
                cnt += 1
            if cnt > 0 and (self.cfg["focal_length"] * err / cnt > 10 or err3 / cnt > 2.0):
                removeIndex.add(f.feature_id)

    def predictPtsInNextFrame(self):  # EST:3862-3897
        fc = self.frame_count
        if fc < 2:
            return
        curT, prevT = np.eye(4), np.eye(4)
        curT[:3, :3], curT[:3, 3] = self.Rs[fc], self.Ps[fc]
        prevT[:3, :3], prevT[:3, 3] = self.Rs[fc - 1], self.Ps[fc - 1]
        nextT = curT @ (np.linalg.inv(prevT) @ curT)
        for f in self.f_manager.feature:
            if f.estimated_depth > 0:
                n = len(f.feature_per_frame)
                if n >= 2 and f.start_frame + n - 1 == fc:
                    pts_j = self.ric @ (f.estimated_depth * f.feature_per_frame[0].point) + self.tic
                    pts_w = self.Rs[f.start_frame] @ pts_j + self.Ps[f.start_frame]
                    pts_local = nextT[:3, :3].T @ (pts_w - nextT[:3, 3])
                    self.predictPts[f.feature_id] = self.ric.T @ (pts_local - self.tic)
