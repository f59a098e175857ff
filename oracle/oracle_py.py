"""TEST INFRASTRUCTURE — ctypes binding of the CPU oracle (oracle/libgf_oracle.so).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class TrackerCfg(C.Structure):
    _fields_ = [("max_cnt", C.c_int), ("min_dist", C.c_int), ("flow_back", C.c_int), ("depth_cam", C.c_int),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("k1", C.c_double), ("k2", C.c_double), ("p1", C.c_double), ("p2", C.c_double)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        p = os.path.join(_HERE, "libgf_oracle.so")
        if not os.path.exists(p):
            build()
        _LIB = C.CDLL(p)
        _LIB.gfo_tracker_create.restype = C.c_void_p
        _LIB.gfo_tracker_lk_iters.restype = C.c_longlong
    return _LIB


def use_native():
    """bench.py's cpu_baseline leg only: switch to the -O3 -march=native build, compiled on THIS machine by `make native`.
    Returns True when the native library is loaded, False (portable build kept) when it cannot be built here."""
    global _LIB
    p = os.path.join(_HERE, "libgf_oracle_native.so")
    try:
        subprocess.check_call(["make", "-s", "-C", _HERE, "native"], stderr=subprocess.DEVNULL)
        nat = C.CDLL(p)
    except (OSError, subprocess.CalledProcessError):
        return False
    nat.gfo_tracker_create.restype = C.c_void_p
    nat.gfo_tracker_lk_iters.restype = C.c_longlong
    _LIB = nat
    return True


def set_threads(n):
    """CPU-baseline variant (b): per-point parallel LK on n threads + 4 marginalisation threads (gf_oracle.h)"""
    lib().gfo_set_threads(int(n))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def default_cfg(max_cnt=150, min_dist=30, flow_back=1, depth_cam=1):
    return TrackerCfg(max_cnt, min_dist, flow_back, depth_cam, 603.95556640625, 603.1257934570312,
                      324.0858154296875, 232.72303771972656, 0.0, 0.0, 0.0, 0.0)


class Tracker:
    def __init__(self, cfg=None):
        self.cfg = cfg or default_cfg()
        self.h = C.c_void_p(lib().gfo_tracker_create(C.byref(self.cfg)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().gfo_tracker_destroy(self.h)
            self.h = None

    def track(self, t, img, depth=None, cap=4096):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        ids = np.zeros(cap, np.int32)
        obs = np.zeros((cap, 8), np.float64)
        if depth is not None:
            depth = np.ascontiguousarray(depth, np.uint16)
            dp, ds = _p(depth, C.c_uint16), depth.shape[1]
        else:
            dp, ds = None, 0
        n = lib().gfo_tracker_track(self.h, C.c_double(t), _p(img, C.c_uint8), w, h, w, dp, ds,
                                    _p(ids, C.c_int), _p(obs, C.c_double), cap)
        return ids[:n].copy(), obs[:n].copy()

    def set_prediction(self, ids, xyz):
        ids = np.ascontiguousarray(ids, np.int32)
        xyz = np.ascontiguousarray(xyz, np.float64)
        lib().gfo_tracker_set_prediction(self.h, _p(ids, C.c_int), _p(xyz, C.c_double), len(ids))

    def remove_outliers(self, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        lib().gfo_tracker_remove_outliers(self.h, _p(ids, C.c_int), len(ids))

    def state(self, cap=4096):
        ids = np.zeros(cap, np.int32)
        cnt = np.zeros(cap, np.int32)
        pts = np.zeros((cap, 2), np.float32)
        n = lib().gfo_tracker_state(self.h, _p(ids, C.c_int), _p(cnt, C.c_int), _p(pts, C.c_float), cap)
        return ids[:n].copy(), cnt[:n].copy(), pts[:n].copy()

    def lk_iters(self):
        return int(lib().gfo_tracker_lk_iters(self.h))


def set_lk_accum(mode):
    """0: int64 sums (parity mode); 1: float-lane accumulation of an x86 OpenCV build (sensitivity measurement only)"""
    lib().gfo_set_lk_accum(int(mode))


def pyr_down(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().gfo_pyr_down(_p(img, C.c_uint8), w, h, _p(out, C.c_uint8))
    return out


def scharr(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w, 2), np.int16)
    lib().gfo_scharr(_p(img, C.c_uint8), w, h, _p(out, C.c_int16))
    return out


def lk(prev, nxt, prev_pts, next_pts=None, max_level=3, max_count=30, eps=0.01):
    prev = np.ascontiguousarray(prev, np.uint8)
    nxt = np.ascontiguousarray(nxt, np.uint8)
    h, w = prev.shape
    pp = np.ascontiguousarray(prev_pts, np.float32)
    use_init = next_pts is not None
    npnts = np.ascontiguousarray(next_pts, np.float32).copy() if use_init else np.zeros_like(pp)
    st = np.zeros(len(pp), np.uint8)
    it = C.c_longlong(0)
    lib().gfo_lk(_p(prev, C.c_uint8), _p(nxt, C.c_uint8), w, h, _p(pp, C.c_float), _p(npnts, C.c_float), _p(st, C.c_uint8),
                 len(pp), max_level, max_count, C.c_double(eps), int(use_init), C.byref(it))
    return npnts, st, it.value


def fill_circle(img, cx, cy, r, color=0):
    h, w = img.shape
    lib().gfo_fill_circle(_p(img, C.c_uint8), w, h, cx, cy, r, color)
    return img


def min_eigen_val(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.float32)
    lib().gfo_min_eigen_val(_p(img, C.c_uint8), w, h, _p(out, C.c_float))
    return out


def good_features(img, max_corners, quality=0.01, min_dist=30.0, mask=None):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((max(max_corners, 1) if max_corners > 0 else w * h, 2), np.float32)
    mp = _p(np.ascontiguousarray(mask, np.uint8), C.c_uint8) if mask is not None else None
    n = lib().gfo_good_features(_p(img, C.c_uint8), w, h, _p(out, C.c_float), max_corners, C.c_double(quality),
                                C.c_double(min_dist), mp)
    return out[:n].copy()


# ------------------------------------------------------------------ back end
def _bind_backend():
    import gfwindow
    L = lib()
    L.gfo_ba_solve.argtypes = [C.POINTER(gfwindow.WindowC), C.c_int, C.POINTER(gfwindow.SummaryC)]
    return gfwindow


def ba_solve(win, max_iters=8):
    """in-place on win's state arrays; returns summary dict"""
    gw = _bind_backend()
    c = win.to_c()
    s = gw.SummaryC()
    rc = lib().gfo_ba_solve(C.byref(c), max_iters, C.byref(s))
    assert rc == 0
    return {k: getattr(s, k) for k, _ in gw.SummaryC._fields_}


def ba_marginalize(win, mode=0, cap_n=512):
    gw = _bind_backend()
    c = win.to_c()
    n, nb, m = C.c_int(0), C.c_int(0), C.c_int(0)
    bidv = np.zeros(256, np.int32)
    J = np.zeros(cap_n * cap_n)
    r = np.zeros(cap_n)
    x0 = np.zeros(cap_n * 2)
    rc = lib().gfo_ba_marginalize(C.byref(c), mode, cap_n, C.byref(n), C.byref(nb), _p(bidv, C.c_int), _p(J, C.c_double), _p(r, C.c_double),
                                  _p(x0, C.c_double), C.byref(m))
    if rc != 0:
        return None
    nn = n.value
    ids = bidv[:nb.value].copy()
    gs = sum(gw.gsize(int(i) // 4096) for i in ids)
    return {"block_id": ids, "J": J[:nn * nn].copy(), "r": r[:nn].copy(), "x0": x0[:gs].copy(), "m": m.value, "n": nn}


def ba_marg_system(win, mode=0, cap=1024):
    """assembled marginalisation system (A, b; dropped columns first) and the oracle's Schur complement (A_r, b_r)"""
    _bind_backend()
    c = win.to_c()
    A = np.zeros(cap * cap); b = np.zeros(cap); Ar = np.zeros(cap * cap); br = np.zeros(cap)
    pos, m, n = C.c_int(0), C.c_int(0), C.c_int(0)
    rc = lib().gfo_ba_marg_system(C.byref(c), mode, cap, _p(A, C.c_double), _p(b, C.c_double), _p(Ar, C.c_double), _p(br, C.c_double),
                                  C.byref(pos), C.byref(m), C.byref(n))
    assert rc == 0
    P, N = pos.value, n.value
    return {"A": A[:P * P].reshape(P, P).copy(), "b": b[:P].copy(), "Ar": Ar[:N * N].reshape(N, N).copy(), "br": br[:N].copy(), "m": m.value, "n": N}


def factor_eval(win, kind, k):
    _bind_backend()
    c = win.to_c()
    res = np.zeros(512)
    jac = np.zeros(512 * 64)
    nres, ncols = C.c_int(0), C.c_int(0)
    rc = lib().gfo_factor_eval(C.byref(c), kind, k, _p(res, C.c_double), _p(jac, C.c_double), C.byref(nres), C.byref(ncols))
    assert rc == 0
    return res[:nres.value].copy(), jac[:nres.value * ncols.value].reshape(nres.value, ncols.value).copy()


def imu_preintegrate(dt, acc, gyr, acc0, gyr0, ba, bg, noise):
    f = lambda a: np.ascontiguousarray(a, np.float64)
    dt, acc, gyr, acc0, gyr0, ba, bg, noise = map(f, (dt, acc, gyr, acc0, gyr0, ba, bg, noise))
    out = {"delta_p": np.zeros(3), "delta_q": np.zeros(4), "delta_v": np.zeros(3), "jacobian": np.zeros(225), "covariance": np.zeros(225)}
    sd = C.c_double(0)
    lib().gfo_imu_preintegrate(len(dt), _p(dt, C.c_double), _p(acc, C.c_double), _p(gyr, C.c_double), _p(acc0, C.c_double), _p(gyr0, C.c_double),
                               _p(ba, C.c_double), _p(bg, C.c_double), _p(noise, C.c_double), _p(out["delta_p"], C.c_double), _p(out["delta_q"], C.c_double),
                               _p(out["delta_v"], C.c_double), _p(out["jacobian"], C.c_double), _p(out["covariance"], C.c_double), C.byref(sd))
    out["sum_dt"] = sd.value
    return out


def wheel_preintegrate(dt, vel, gyr, vel0, gyr0, lin, noise):
    f = lambda a: np.ascontiguousarray(a, np.float64)
    dt, vel, gyr, vel0, gyr0, lin, noise = map(f, (dt, vel, gyr, vel0, gyr0, lin, noise))
    out = {"delta_p": np.zeros(3), "delta_q": np.zeros(4), "jacobian": np.zeros(18), "covariance": np.zeros(36)}
    sd = C.c_double(0)
    lib().gfo_wheel_preintegrate(len(dt), _p(dt, C.c_double), _p(vel, C.c_double), _p(gyr, C.c_double), _p(vel0, C.c_double), _p(gyr0, C.c_double),
                                 _p(lin, C.c_double), _p(noise, C.c_double), _p(out["delta_p"], C.c_double), _p(out["delta_q"], C.c_double),
                                 _p(out["jacobian"], C.c_double), _p(out["covariance"], C.c_double), C.byref(sd))
    out["sum_dt"] = sd.value
    return out


def sym_eig(A):
    A = np.ascontiguousarray(A, np.float64)
    n = A.shape[0]
    d = np.zeros(n)
    V = np.zeros((n, n))
    lib().gfo_sym_eig(n, _p(A, C.c_double), _p(d, C.c_double), _p(V, C.c_double))
    return d, V


def ba_linearize(win, cap=1024):
    _bind_backend()
    c = win.to_c()
    H = np.zeros(cap * cap)
    g = np.zeros(cap)
    cost = C.c_double(0)
    nf, ne = C.c_int(0), C.c_int(0)
    ids = np.zeros(cap, np.int32)
    rc = lib().gfo_ba_linearize(C.byref(c), cap, _p(H, C.c_double), _p(g, C.c_double), C.byref(cost), C.byref(nf), C.byref(ne), _p(ids, C.c_int))
    assert rc == 0
    n = nf.value + ne.value
    return {"H": H[:n * n].reshape(n, n).copy(), "g": g[:n].copy(), "cost": cost.value, "n_f": nf.value, "n_e": ne.value, "ids": ids[:n].copy()}


def double2vector(W, R0, P0, para_Pose, para_SpeedBias):
    f = lambda a: np.ascontiguousarray(a, np.float64).reshape(-1)
    R0, P0, pp, sb = map(f, (R0, P0, para_Pose, para_SpeedBias))
    out = [np.zeros(9 * (W + 1)), np.zeros(3 * (W + 1)), np.zeros(3 * (W + 1)), np.zeros(3 * (W + 1)), np.zeros(3 * (W + 1))]
    lib().gfo_double2vector(W, _p(R0, C.c_double), _p(P0, C.c_double), _p(pp, C.c_double), _p(sb, C.c_double), *[_p(o, C.c_double) for o in out])
    return out
