"""Broadcast ephemerides -> satellite state: gnss_comm's eph2pos / geph2pos / eph2svdt / eph2vel are not in the reference tree, so
both the library (gf_gnss_eph2pos, gf_gnss_obs_from_ephem; host C++) and the oracle (estimator_oracle.eph2pos ...) restate the published broadcast-orbit
algorithms.  PARITY UNPINNED against gnss_comm itself; pinned here against physics instead: the Kepler model must follow a two-body orbit integrated
independently, the GLONASS Runge-Kutta must agree with a tight-tolerance integration of the ICD's equations, and library and oracle must agree with
each other.  No GPU involved."""
import math
import os
import sys

import numpy as np
import pytest
from scipy.integrate import solve_ivp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gfamd  # noqa: E402
import estimator_oracle as EO  # noqa: E402


def kepler(sys_=0, prn=12, **kw):
    e = dict(sat=7, sys=sys_, prn=prn, toe=5000.0, toc=5000.0, toe_tow=345600.0, A=26560e3, e=0.012, i0=0.96, OMG0=1.1, omg=0.4, M0=0.7, delta_n=0.0, OMG_dot=0.0, i_dot=0.0,
             cuc=0.0, cus=0.0, crc=0.0, crs=0.0, cic=0.0, cis=0.0, af0=1e-4, af1=2e-11, af2=0.0, tgd0=5e-9, ura=2.0)
    e.update(kw)
    return e


def rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


@pytest.mark.parametrize("sys_", [0, 2, 3])
def test_kepler_model_follows_a_two_body_orbit(sys_):
    """without harmonic corrections and rates the broadcast model IS a two-body orbit seen from the rotating Earth: integrate r'' = -mu r / |r|^3 from its
    state at toe and compare half an hour later (inertial frame: undo the Earth rotation omge (tk + toe_tow) that the model folds into the node)"""
    e = kepler(sys_)
    mu, omge = EO.MU[sys_], EO.OMGE[sys_]
    eci = lambda t: rot_z(omge * (t - e["toe"] + e["toe_tow"])) @ EO.eph2pos(t, e)[0]
    h = 0.05
    r0, v0 = eci(e["toe"]), (eci(e["toe"] + h) - eci(e["toe"] - h)) / (2 * h)
    sol = solve_ivp(lambda t, y: np.concatenate([y[3:], -mu * y[:3] / np.linalg.norm(y[:3]) ** 3]), (0, 1800.0), np.concatenate([r0, v0]), rtol=1e-12, atol=1e-6)
    assert np.linalg.norm(sol.y[:3, -1] - eci(e["toe"] + 1800.0)) < 0.05     # metres after 7 000 km of flight (limited by the difference-quotient velocity)
    # energy and |r| = A (1 - e cos E) hold along the way
    for t in (e["toe"] - 3000.0, e["toe"] + 777.0, e["toe"] + 3600.0):
        r = np.linalg.norm(EO.eph2pos(t, e)[0])
        assert e["A"] * (1 - e["e"]) - 1 < r < e["A"] * (1 + e["e"]) + 1


def test_beidou_geo_frame_is_the_standard_model_tilted_by_five_degrees():
    """prn <= 5: position built in the inertial frame, tilted by -5 deg about x and turned by the Earth rotation since toe.  Undoing both must give the
    inertial-frame position of the standard formula (same elements, a satellite that is not GEO)"""
    geo, meo = kepler(3, prn=3, A=42164e3, i0=0.1), kepler(3, prn=20, A=42164e3, i0=0.1)
    omge = EO.OMGE[3]
    for tk in (0.0, 900.0, -1200.0):
        t = geo["toe"] + tk
        pg, dg = EO.eph2pos(t, geo)
        pm, dm = EO.eph2pos(t, meo)
        a = math.radians(-5.0)
        Rx = np.array([[1, 0, 0], [0, math.cos(a), math.sin(a)], [0, -math.sin(a), math.cos(a)]])
        inertial_geo = np.linalg.inv(Rx) @ rot_z(omge * tk) @ pg                 # back through Rz(-omge tk) Rx(-5 deg)
        inertial_meo = rot_z(omge * tk) @ pm                                     # the standard formula's node term carries -omge tk
        assert np.linalg.norm(inertial_geo - inertial_meo) < 1e-6 and dg == dm


def test_glonass_runge_kutta_against_a_tight_integration():
    g = dict(sat=105, toe=5000.0, pos=[1.2e7, 1.5e7, 1.6e7], vel=[-1500.0, 2500.0, -1200.0], acc=[1e-6, -2e-6, 0.5e-6], tau_n=2e-5, gamma=1e-12)
    for tk in (870.0, -655.0, 30.0):
        p, dts = EO.geph2pos(g["toe"] + tk, g)
        sol = solve_ivp(lambda t, y: EO._glo_deq(y, np.array(g["acc"])), (0, tk), np.array([*g["pos"], *g["vel"]], float), rtol=1e-13, atol=1e-6)
        assert np.linalg.norm(sol.y[:3, -1] - p) < 2e-3         # 60 s steps of the 4th-order scheme over 15 minutes: millimetres
        assert dts == -g["tau_n"] + g["gamma"] * tk
    assert abs(EO.geph2svdt(g["toe"] + 100.0, g) - (-g["tau_n"] + g["gamma"] * 100.0)) < 1e-15


def test_library_and_oracle_agree_and_the_constructor_logic_holds():
    """gf_gnss_eph2pos / gf_gnss_obs_from_ephem against the numpy restatement, Kepler with all correction terms, BeiDou GEO, GLONASS; then the pieces of
    GnssPsrDoppFactor's constructor (gnss_psr_dopp_factor.cpp:3-47): transmission time = reception time - psr / c - clock bias, velocity and clock drift as
    1 ms difference quotients, uura scalings per constellation"""
    full = dict(delta_n=4.5e-9, OMG_dot=-8.1e-9, i_dot=2e-10, cuc=1.2e-6, cus=-3e-6, crc=210.0, crs=-60.0, cic=9e-8, cis=-1.1e-7, af2=1e-20)
    ephs = [kepler(0, **full), kepler(2, **full, A=29600e3), kepler(3, prn=25, **full, A=27906e3), kepler(3, prn=4, **full, A=42164e3, i0=0.08)]
    g = dict(sat=105, toe=5000.0, pos=[1.2e7, 1.5e7, 1.6e7], vel=[-1500.0, 2500.0, -1200.0], acc=[1e-6, -2e-6, 0.5e-6], tau_n=2e-5, gamma=1e-12)
    for e in ephs:
        for t in (4100.0, 5000.0, 6234.5):
            p, d = gfamd.gnss_eph2pos(t, eph=e)
            po, do = EO.eph2pos(t, e)
            assert np.abs(p - po).max() < 1e-6 and abs(d - do) < 1e-15
    for t in (4300.0, 5000.5, 5800.0):
        p, d = gfamd.gnss_eph2pos(t, geph=g)
        po, do = EO.geph2pos(t, g)
        assert np.abs(p - po).max() < 1e-6 and abs(d - do) < 1e-15
    raw = dict(sat=7, sys=0, time=5600.0, psr=2.31e7, dopp=-1500.0, psr_std=0.64, dopp_std=0.512, freq=1575.42e6, tow=345600.0 + 600.0)
    for e in ephs + [None]:
        r = dict(raw, sys=e["sys"] if e else 1)
        a = gfamd.gnss_obs_from_ephem(r, eph=e, geph=None if e else g)
        b = EO.sat_state(r, eph=e, geph=None if e else g)
        for k in b:
            assert np.allclose(a[k], b[k], rtol=0, atol=1e-6 if k in ("sv_pos", "sv_vel") else 1e-15), k
        # the constructor's steps, spelled out
        tx = r["time"] - r["psr"] / EO.C_LIGHT
        tx -= EO.eph2svdt(tx, e) if e else EO.geph2svdt(tx, g)
        f = (lambda t: EO.eph2pos(t, e)) if e else (lambda t: EO.geph2pos(t, g))
        assert np.array_equal(b["sv_pos"], f(tx)[0]) and b["svdt"] == f(tx)[1]
        assert np.allclose(b["sv_vel"], (f(tx + 1e-3)[0] - f(tx)[0]) / 1e-3) and 2500 < np.linalg.norm(b["sv_vel"]) < 4000 if (e is None or e["A"] < 4e7) else True
        k = 2.0 if e is None else (e["ura"] - 2.0 if e["sys"] == 2 else e["ura"] - 1.0)
        assert b["pr_uura"] == pytest.approx(k * 0.64 / 0.16) and b["dp_uura"] == pytest.approx(k * 0.512 / 0.256) and b["tgd"] == (e["tgd0"] if e else 0.0)
    with pytest.raises(gfamd.GfError):
        gfamd.gnss_obs_from_ephem(raw, eph=ephs[0], geph=g)


def test_process_gnss_resolves_the_nearest_ephemeris():
    """estimator.cpp:1467-1495 in the fill phase (no GPU): an observation without an ephemeris is skipped, the ephemeris nearest in toe is used, one older than
    EPH_VALID_SECONDS is not; a repeated (satellite, toe) pair is ignored (inputEphem, :1428-1437)"""
    import synth_stream as SS
    kw = dict(gnss_enable=1, gnss_track_num_thres=1, gnss_local_time_diff=18.0, tio=SS.TIO, rio=SS.RIO)
    st = SS.Stream(4, t_still=0.15, t_move=1.6, v_max=0.8)
    G = st.gnss_setup(sats_per_sys=2, n_low=0, orbits=EO)
    est_o, est_p = EO.Estimator(dict(kw)), gfamd.SlidingWindowEstimator(gfamd.default_estimator_cfg(**kw))
    ephs = G["ephems"]
    for e in ephs[1:]:                       # satellite 1 has no ephemeris at first
        for est in (est_o, est_p):
            est.inputEphem(e)
            est.inputEphem(dict(e, af0=9.9) if "af0" in e else dict(e, tau_n=9.9))   # same (sat, toe): ignored
    stale = dict(ephs[1], toe=ephs[1]["toe"] - 8000.0, toc=ephs[1]["toc"] - 8000.0, M0=0.0)     # far older: never the nearest, and beyond 7200 s anyway
    for est in (est_o, est_p):
        est.inputEphem(stale)
    tp = -1.0
    for k in range(4):
        tk = float(st.cam_t[3 * k])
        if k == 2:
            for est in (est_o, est_p):
                est.inputEphem(ephs[0])
        for est in (est_o, est_p):
            tp1 = st.feed(est, 3 * k, tp)
            est.inputGNSS(*st.gnss_epoch(tk + 0.01))
            est.inputFeature(tk, st.feature_frame(3 * k))
        tp = tp1
        buf = est_p.debug("gnss_meas_buf")
        got, q = [], 0
        for _ in range(est_o.W + 1):
            n = int(buf[q])
            got.append([int(x) for x in buf[q + 1:q + 1 + n]])
            q += 1 + n
        assert got == [[o["sat"] for o in b] for b in est_o.gnss_meas_buf], k
    sizes = [len(b) for b in est_o.gnss_meas_buf]
    assert sizes[0] == sizes[1] == 7 and sizes[2] == sizes[3] == 8        # satellite 1 joins once its ephemeris is there
    o = [x for x in est_o.gnss_meas_buf[3] if x["sat"] == ephs[1]["sat"]][0]
    assert np.linalg.norm(o["sv_pos"] - EO.eph2pos(o["time"] - o["psr"] / EO.C_LIGHT, ephs[1])[0]) < 1.0   # resolved with the fresh ephemeris, not the stale one
