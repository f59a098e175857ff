"""ctypes binding of the C-ABI (include/groundfusion_hip.h) — the call surface tests and bench.py use.
Mirrors the reference's FeatureTracker interface names (trackImage / setPrediction / removeOutliers).
Raises if the HIP library is missing or no GPU is present: there is no CPU fallback in the product path."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GF_LIB_PATH") or os.path.join(HERE, "lib", "libgroundfusion_hip.so")   # override: profiling builds of the same sources
_LIB = None


class GfError(RuntimeError):
    pass


class TrackerCfg(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("batch", C.c_int), ("max_cnt", C.c_int), ("min_dist", C.c_int),
                ("flow_back", C.c_int), ("depth_cam", C.c_int),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("k1", C.c_double), ("k2", C.c_double), ("p1", C.c_double), ("p2", C.c_double)]


class FeatureObs(C.Structure):
    _fields_ = [("id", C.c_int), ("camera_id", C.c_int), ("v", C.c_double * 8)]


class TrackerStats(C.Structure):
    _fields_ = [("ms_pyramid", C.c_double), ("ms_lk", C.c_double), ("ms_detect", C.c_double), ("ms_total_gpu", C.c_double),
                ("ms_host_pre", C.c_double), ("ms_wait_lk", C.c_double), ("ms_host_mid", C.c_double), ("ms_wait_detect", C.c_double), ("ms_host_post", C.c_double),
                ("frames", C.c_longlong), ("lk_launches", C.c_longlong), ("lk_points", C.c_longlong),
                ("lk_level_passes", C.c_longlong), ("lk_iterations", C.c_longlong), ("tracked_features", C.c_longlong),
                ("output_features", C.c_longlong)]


OBS_DTYPE = np.dtype([("id", np.int32), ("camera_id", np.int32), ("v", np.float64, (8,))])
assert OBS_DTYPE.itemsize == C.sizeof(FeatureObs)

EXPORTS = ["gf_last_error", "gf_device_count", "gf_set_device", "gf_tracker_create", "gf_tracker_destroy", "gf_tracker_track",
           "gf_tracker_track_batch", "gf_tracker_track_batch_device", "gf_tracker_set_prediction", "gf_tracker_remove_outliers",
           "gf_tracker_get_state", "gf_tracker_set_profiling", "gf_tracker_get_stats", "gf_tracker_reset_stats", "gf_lk_track",
           "gf_good_features", "gf_min_eigen_val", "gf_pyramid_level"]


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise GfError("HIP extension %s is missing: run `python __graft_entry__.py` (build) first; there is no CPU fallback" % LIB_PATH)
        # torch bundles its own libamdhip64.so (SONAME libamdhip64.so.7, same as /opt/rocm's). Two HIP runtimes in
        # one process cannot both own the device, so let torch's copy load first; ours then binds to it by SONAME.
        if os.environ.get("GF_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        _LIB = C.CDLL(LIB_PATH)
        _LIB.gf_last_error.restype = C.c_char_p
    return _LIB


def _chk(rc):
    if rc != 0:
        raise GfError("gf status %d: %s" % (rc, lib().gf_last_error().decode()))


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def device_count():
    n = C.c_int(0)
    lib().gf_device_count(C.byref(n))
    return n.value


def default_cfg(width=640, height=480, batch=1, max_cnt=150, min_dist=30, flow_back=1, depth_cam=1):
    return TrackerCfg(width, height, batch, max_cnt, min_dist, flow_back, depth_cam, 603.95556640625, 603.1257934570312,
                      324.0858154296875, 232.72303771972656, 0.0, 0.0, 0.0, 0.0)


class FeatureTracker:
    """`batch` independent FeatureTracker instances (feature_tracker.h:43-99) advanced in lock-step on one GPU."""

    def __init__(self, cfg=None):
        self.cfg = cfg or default_cfg()
        self.h = C.c_void_p()
        _chk(lib().gf_tracker_create(C.byref(self.cfg), C.byref(self.h)))
        self.cap = 4 * ((self.cfg.max_cnt + 3) // 4)

    def close(self):
        if getattr(self, "h", None):
            lib().gf_tracker_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _unpack(self, out, n):
        res = []
        for b in range(self.cfg.batch):
            o = out[b, :n[b]]
            res.append((o["id"].copy(), o["v"].copy()))
        return res

    def trackImage(self, t, img, depth=None):
        """batch == 1 convenience: returns (ids, obs[n,8])."""
        return self.trackImageBatch([t], [img], None if depth is None else [depth])[0]

    def trackImageBatch(self, ts, imgs, depths=None):
        B = self.cfg.batch
        ts = np.ascontiguousarray(ts, np.float64)
        imgs = [np.ascontiguousarray(i, np.uint8) for i in imgs]
        gp = (C.POINTER(C.c_uint8) * B)(*[_p(i, C.c_uint8) for i in imgs])
        if depths is not None:
            depths = [np.ascontiguousarray(d, np.uint16) for d in depths]
            dp = (C.POINTER(C.c_uint16) * B)(*[_p(d, C.c_uint16) for d in depths])
        else:
            dp = None
        out = np.zeros((B, self.cap), OBS_DTYPE)
        n = np.zeros(B, np.int32)
        _chk(lib().gf_tracker_track_batch(self.h, _p(ts, C.c_double), gp, self.cfg.width, dp, self.cfg.width,
                                          out.ctypes.data_as(C.POINTER(FeatureObs)), self.cap, _p(n, C.c_int)))
        return self._unpack(out, n)

    def trackImageBatchDevice(self, ts, d_gray_ptr, d_depth_ptr=None, unpack=True, out=None, n_out=None):
        """d_*_ptr: integer device addresses (e.g. torch tensor .data_ptr()) of batch contiguous frames.  out / n_out: the caller's [batch][cap] OBS_DTYPE table
        and [batch] int32 counts to write into (e.g. one of a ring, while estimators still read the previous frames' tables); default: the tracker's own pair."""
        B = self.cfg.batch
        ts = np.ascontiguousarray(ts, np.float64)
        if out is None:
            if not hasattr(self, "_out"):
                self._out = np.zeros((B, self.cap), OBS_DTYPE)
                self._n = np.zeros(B, np.int32)
            out, n_out = self._out, self._n
        assert out.shape == (B, self.cap) and out.dtype == OBS_DTYPE and out.flags.c_contiguous and n_out.shape == (B,) and n_out.dtype == np.int32
        _chk(lib().gf_tracker_track_batch_device(self.h, _p(ts, C.c_double), C.c_void_p(d_gray_ptr),
                                                 C.c_void_p(d_depth_ptr) if d_depth_ptr else None,
                                                 out.ctypes.data_as(C.POINTER(FeatureObs)), self.cap, _p(n_out, C.c_int)))
        return self._unpack(out, n_out) if unpack else n_out

    def prefetchHost(self, gray_addr, depth_addr=None):
        """gf_tracker_prefetch_batch on a block of `batch` frames lying back to back in (page-locked) host memory: gray_addr / depth_addr = integer host addresses
        of batch x height x width u8 / u16 images; the copy runs on the tracker's copy stream while the frame in flight is processed"""
        B, n = self.cfg.batch, self.cfg.width * self.cfg.height
        gp = (C.POINTER(C.c_uint8) * B)(*[C.cast(gray_addr + b * n, C.POINTER(C.c_uint8)) for b in range(B)])
        dp = (C.POINTER(C.c_uint16) * B)(*[C.cast(depth_addr + 2 * b * n, C.POINTER(C.c_uint16)) for b in range(B)]) if depth_addr else None
        _chk(lib().gf_tracker_prefetch_batch(self.h, gp, self.cfg.width, dp, self.cfg.width))

    def trackPrefetched(self, ts, unpack=True):
        B = self.cfg.batch
        ts = np.ascontiguousarray(ts, np.float64)
        if not hasattr(self, "_out"):
            self._out = np.zeros((B, self.cap), OBS_DTYPE)
            self._n = np.zeros(B, np.int32)
        _chk(lib().gf_tracker_track_prefetched(self.h, _p(ts, C.c_double), self._out.ctypes.data_as(C.POINTER(FeatureObs)), self.cap, _p(self._n, C.c_int)))
        return self._unpack(self._out, self._n) if unpack else self._n

    def setPrediction(self, ids, xyz, seq=0):
        ids = np.ascontiguousarray(ids, np.int32)
        xyz = np.ascontiguousarray(xyz, np.float64)
        _chk(lib().gf_tracker_set_prediction(self.h, seq, _p(ids, C.c_int), _p(xyz, C.c_double), len(ids)))

    def removeOutliers(self, ids, seq=0):
        ids = np.ascontiguousarray(ids, np.int32)
        _chk(lib().gf_tracker_remove_outliers(self.h, seq, _p(ids, C.c_int), len(ids)))

    def state(self, seq=0):
        ids = np.zeros(self.cap, np.int32)
        cnt = np.zeros(self.cap, np.int32)
        pts = np.zeros((self.cap, 2), np.float32)
        n = C.c_int(0)
        _chk(lib().gf_tracker_get_state(self.h, seq, _p(ids, C.c_int), _p(cnt, C.c_int), _p(pts, C.c_float), self.cap, C.byref(n)))
        return ids[:n.value].copy(), cnt[:n.value].copy(), pts[:n.value].copy()

    def set_profiling(self, on=True):
        _chk(lib().gf_tracker_set_profiling(self.h, int(on)))

    def stats(self):
        s = TrackerStats()
        _chk(lib().gf_tracker_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in TrackerStats._fields_}

    def reset_stats(self):
        _chk(lib().gf_tracker_reset_stats(self.h))


def lk_track(prev, nxt, prev_pts, next_pts=None, max_level=3):
    prev = np.ascontiguousarray(prev, np.uint8)
    nxt = np.ascontiguousarray(nxt, np.uint8)
    h, w = prev.shape
    pp = np.ascontiguousarray(prev_pts, np.float32)
    use_init = next_pts is not None
    npn = np.ascontiguousarray(next_pts, np.float32).copy() if use_init else np.zeros_like(pp)
    st = np.zeros(len(pp), np.uint8)
    it = C.c_longlong(0)
    _chk(lib().gf_lk_track(_p(prev, C.c_uint8), _p(nxt, C.c_uint8), w, h, _p(pp, C.c_float), _p(npn, C.c_float), _p(st, C.c_uint8),
                           len(pp), max_level, int(use_init), C.byref(it)))
    return npn, st, it.value


def good_features(img, max_corners, min_dist=30, mask=None):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((max_corners, 2), np.float32)
    n = C.c_int(0)
    mp = _p(np.ascontiguousarray(mask, np.uint8), C.c_uint8) if mask is not None else None
    _chk(lib().gf_good_features(_p(img, C.c_uint8), w, h, mp, max_corners, min_dist, _p(out, C.c_float), C.byref(n)))
    return out[:n.value].copy()


def min_eigen_val(img):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.float32)
    _chk(lib().gf_min_eigen_val(_p(img, C.c_uint8), w, h, _p(out, C.c_float)))
    return out


def pyramid_level(img, level):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    lw, lh = w, h
    for _ in range(level):
        lw, lh = (lw + 1) // 2, (lh + 1) // 2
    out = np.zeros((lh, lw), np.uint8)
    der = np.zeros((lh, lw, 2), np.int16)
    _chk(lib().gf_pyramid_level(_p(img, C.c_uint8), w, h, level, _p(out, C.c_uint8), _p(der, C.c_int16)))
    return out, der


# ------------------------------------------------------------------ back end (Estimator::optimization)
import gfwindow as _gw  # noqa: E402


class BaCfg(C.Structure):
    _fields_ = [("window_size", C.c_int), ("max_features", C.c_int), ("max_visual", C.c_int), ("batch", C.c_int), ("max_gnss", C.c_int)]


class BaPriorC(C.Structure):
    _fields_ = [("cap_n", C.c_int), ("cap_blocks", C.c_int), ("n", C.c_int), ("nblocks", C.c_int), ("m", C.c_int), ("valid", C.c_int),
                ("block_id", C.POINTER(C.c_int)), ("J", C.POINTER(C.c_double)), ("r", C.POINTER(C.c_double)), ("x0", C.POINTER(C.c_double))]


class BaStats(C.Structure):
    _fields_ = [("ms_upload", C.c_double), ("ms_solve", C.c_double), ("ms_marginalize", C.c_double), ("ms_download", C.c_double), ("ms_jtj", C.c_double),
                ("solves", C.c_longlong), ("jtj_launches", C.c_longlong), ("jtj_flops", C.c_longlong),
                ("ms_step", C.c_double), ("step_launches", C.c_longlong), ("step_flops", C.c_longlong), ("jtj_alg_flops", C.c_longlong),
                ("ms_jtj_contract", C.c_double), ("jtj_contract_launches", C.c_longlong)]


EXPORTS += ["gf_ba_create", "gf_ba_destroy", "gf_ba_solve", "gf_ba_marginalize", "gf_ba_upload", "gf_ba_solve_resident", "gf_ba_download", "gf_ba_get_stats",
            "gf_ba_reset_stats", "gf_ba_linearize", "gf_ba_solve_resident_async", "gf_ba_wait", "gf_ba_debug_stamps", "gf_imu_preintegrate", "gf_imu_preintegrate_state", "gf_wheel_preintegrate", "gf_ba_double2vector",
            "gf_preint_create", "gf_preint_destroy", "gf_imu_preintegrate_batch", "gf_preint_stats",
            "gf_featsweep_create", "gf_featsweep_destroy", "gf_triangulate_with_depth_batch", "gf_moving_consistency_batch", "gf_featsweep_stats"]


class Estimator:
    """Batched window solver: the Ceres problem of Estimator::optimization() (estimator.cpp:2890-3631) on the GPU."""

    def __init__(self, window_size=10, max_features=150, max_visual=1500, batch=1, max_gnss=0):
        self.cfg = BaCfg(window_size, max_features, max_visual, batch, max_gnss)
        self.h = C.c_void_p()
        _chk(lib().gf_ba_create(C.byref(self.cfg), C.byref(self.h)))
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            lib().gf_ba_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _carr(self, wins):
        arr = (_gw.WindowC * len(wins))()
        for i, w in enumerate(wins):
            arr[i] = w.to_c()
        return arr

    def solve(self, wins, max_iters=8):
        """in place on the windows' state arrays; returns list of summary dicts"""
        arr = self._carr(wins)
        sums = (_gw.SummaryC * len(wins))()
        _chk(lib().gf_ba_solve(self.h, arr, len(wins), max_iters, sums))
        return [{k: getattr(s, k) for k, _ in _gw.SummaryC._fields_} for s in sums]

    def upload(self, wins):
        arr = self._carr(wins)
        self._keep = (wins, arr)
        _chk(lib().gf_ba_upload(self.h, arr, len(wins)))

    def solve_resident(self, max_iters=8, marginalize_mode=-1, reset=True):
        _chk(lib().gf_ba_solve_resident(self.h, max_iters, marginalize_mode, int(reset)))

    def export_newest_poses(self, d_ptr, count):
        """newest pose of every resident window into a device array [count][7] (d_ptr: integer device address)"""
        _chk(lib().gf_ba_export_newest_poses(self.h, C.c_void_p(d_ptr), count))

    def solve_resident_async(self, max_iters=8, marginalize_mode=-1, reset=True):
        _chk(lib().gf_ba_solve_resident_async(self.h, max_iters, marginalize_mode, int(reset)))

    def wait(self):
        _chk(lib().gf_ba_wait(self.h))

    def download(self, wins=None, with_priors=False, cap_n=256):
        wins = wins if wins is not None else self._keep[0]
        arr = self._carr(wins)
        sums = (_gw.SummaryC * len(wins))()
        priors, pc = None, None
        if with_priors:
            pc = (BaPriorC * len(wins))()
            priors = []
            for i in range(len(wins)):
                p = {"block_id": np.zeros(64, np.int32), "J": np.zeros(cap_n * cap_n), "r": np.zeros(cap_n), "x0": np.zeros(cap_n * 2)}
                priors.append(p)
                pc[i].cap_n, pc[i].cap_blocks = cap_n, 64
                pc[i].block_id, pc[i].J, pc[i].r, pc[i].x0 = _p(p["block_id"], C.c_int), _p(p["J"], C.c_double), _p(p["r"], C.c_double), _p(p["x0"], C.c_double)
        _chk(lib().gf_ba_download(self.h, arr, len(wins), sums, pc))
        out = [{k: getattr(s, k) for k, _ in _gw.SummaryC._fields_} for s in sums]
        if with_priors:
            res = []
            for i, p in enumerate(priors):
                if not pc[i].valid:
                    res.append(None)
                    continue
                n, nb = pc[i].n, pc[i].nblocks
                ids = p["block_id"][:nb].copy()
                gs = sum(_gw.gsize(int(q) // 4096) for q in ids)
                res.append({"block_id": ids, "J": p["J"][:n * n].copy(), "r": p["r"][:n].copy(), "x0": p["x0"][:gs].copy(), "m": pc[i].m, "n": n})
            return out, res
        return out

    def marginalize(self, wins, mode=0, cap_n=256):
        self.upload(wins)
        self.solve_resident(0, mode, True)
        return self.download(wins, True, cap_n)[1]

    def linearize(self, win, cap=1024):
        c = win.to_c()
        H = np.zeros(cap * cap)
        g = np.zeros(cap)
        cost = C.c_double(0)
        nf, ne = C.c_int(0), C.c_int(0)
        ids = np.zeros(cap, np.int32)
        _chk(lib().gf_ba_linearize(self.h, C.byref(c), cap, _p(H, C.c_double), _p(g, C.c_double), C.byref(cost), C.byref(nf), C.byref(ne), _p(ids, C.c_int)))
        n = nf.value + ne.value
        return {"H": H[:n * n].reshape(n, n).copy(), "g": g[:n].copy(), "cost": cost.value, "n_f": nf.value, "n_e": ne.value, "ids": ids[:n].copy()}

    def stats(self):
        s = BaStats()
        _chk(lib().gf_ba_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in BaStats._fields_}

    def reset_stats(self):
        _chk(lib().gf_ba_reset_stats(self.h))

    def set_split_jtj(self, on):
        """the visual sweep as two kernels: block rows through HBM + a contraction-only MFMA kernel (north_star's formulation; same bits as the fused kernel)"""
        _chk(lib().gf_ba_set_split_jtj(self.h, int(bool(on))))


def imu_preintegrate(dt, acc, gyr, acc0, gyr0, ba, bg, noise):
    f = lambda a: np.ascontiguousarray(a, np.float64)
    dt, acc, gyr, acc0, gyr0, ba, bg, noise = map(f, (dt, acc, gyr, acc0, gyr0, ba, bg, noise))
    out = {"delta_p": np.zeros(3), "delta_q": np.zeros(4), "delta_v": np.zeros(3), "jacobian": np.zeros(225), "covariance": np.zeros(225)}
    sd = C.c_double(0)
    _chk(lib().gf_imu_preintegrate(len(dt), _p(dt, C.c_double), _p(acc, C.c_double), _p(gyr, C.c_double), _p(acc0, C.c_double), _p(gyr0, C.c_double),
                                   _p(ba, C.c_double), _p(bg, C.c_double), _p(noise, C.c_double), _p(out["delta_p"], C.c_double), _p(out["delta_q"], C.c_double),
                                   _p(out["delta_v"], C.c_double), _p(out["jacobian"], C.c_double), _p(out["covariance"], C.c_double), C.byref(sd)))
    out["sum_dt"] = sd.value
    return out


def imu_preintegrate_state(dt, acc, gyr, acc0, gyr0, ba, bg):
    """delta_p / delta_q / delta_v / sum_dt alone (gf_imu_preintegrate_state): the bits gf_imu_preintegrate returns for them"""
    f = lambda a: np.ascontiguousarray(a, np.float64)
    dt, acc, gyr, acc0, gyr0, ba, bg = map(f, (dt, acc, gyr, acc0, gyr0, ba, bg))
    out = {"delta_p": np.zeros(3), "delta_q": np.zeros(4), "delta_v": np.zeros(3)}
    sd = C.c_double(0)
    _chk(lib().gf_imu_preintegrate_state(len(dt), _p(dt, C.c_double), _p(acc, C.c_double), _p(gyr, C.c_double), _p(acc0, C.c_double), _p(gyr0, C.c_double),
                                         _p(ba, C.c_double), _p(bg, C.c_double), _p(out["delta_p"], C.c_double), _p(out["delta_q"], C.c_double), _p(out["delta_v"], C.c_double), C.byref(sd)))
    out["sum_dt"] = sd.value
    return out


class PreintBatch:
    """gf_preint_*: IMU pre-integration of many intervals in one launch; no CPU fallback"""

    def __init__(self):
        self.h = C.c_void_p()
        _chk(lib().gf_preint_create(C.byref(self.h)))

    def close(self):
        if self.h:
            lib().gf_preint_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, first, dt, acc, gyr, acc0, gyr0, ba, bg, noise):
        f = lambda a: np.ascontiguousarray(a, np.float64)
        first = np.ascontiguousarray(first, np.int32)
        n = len(first) - 1
        dt, acc, gyr, acc0, gyr0, ba, bg, noise = map(f, (dt, acc, gyr, acc0, gyr0, ba, bg, noise))
        out = {"delta_p": np.zeros((n, 3)), "delta_q": np.zeros((n, 4)), "delta_v": np.zeros((n, 3)), "jacobian": np.zeros((n, 225)), "covariance": np.zeros((n, 225)),
               "sum_dt": np.zeros(n)}
        _chk(lib().gf_imu_preintegrate_batch(self.h, n, _p(first, C.c_int), _p(dt, C.c_double), _p(acc, C.c_double), _p(gyr, C.c_double), _p(acc0, C.c_double),
                                             _p(gyr0, C.c_double), _p(ba, C.c_double), _p(bg, C.c_double), _p(noise, C.c_double), _p(out["delta_p"], C.c_double),
                                             _p(out["delta_q"], C.c_double), _p(out["delta_v"], C.c_double), _p(out["jacobian"], C.c_double),
                                             _p(out["covariance"], C.c_double), _p(out["sum_dt"], C.c_double)))
        return out

    def stats(self):
        a, b, ms = C.c_longlong(0), C.c_longlong(0), C.c_double(0)
        _chk(lib().gf_preint_stats(self.h, C.byref(a), C.byref(b), C.byref(ms)))
        return dict(launches=a.value, intervals=b.value, kernel_ms=ms.value)


class FeatureSweeps:
    """gf_featsweep_*: triangulateWithDepth / movingConsistencyCheckW of many windows in one launch each; no CPU fallback.
    windows: list of dicts with Rs ((W+1) x 3 x 3), Ps ((W+1) x 3), tic, ric, start_frame [F], obs (list of [n_f x 4] arrays), estimated_depth [F], estimate_flag [F]."""

    def __init__(self):
        self.h = C.c_void_p()
        _chk(lib().gf_featsweep_create(C.byref(self.h)))

    def close(self):
        if self.h:
            lib().gf_featsweep_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _pack(windows):
        W = len(windows[0]["Ps"]) - 1
        f64 = lambda a: np.ascontiguousarray(a, np.float64)
        Rs, Ps = f64([w["Rs"] for w in windows]), f64([w["Ps"] for w in windows])
        tic, ric = f64([w["tic"] for w in windows]), f64([w["ric"] for w in windows])
        ff = np.concatenate([[0], np.cumsum([len(w["start_frame"]) for w in windows])]).astype(np.int32)
        sf = np.ascontiguousarray(np.concatenate([w["start_frame"] for w in windows]), np.int32)
        nobs = [len(o) for w in windows for o in w["obs"]]
        fo = np.concatenate([[0], np.cumsum(nobs)]).astype(np.int32)
        obs = f64(np.concatenate([np.asarray(o, float).reshape(-1, 4) for w in windows for o in w["obs"]]))
        dep = f64(np.concatenate([w["estimated_depth"] for w in windows]))
        return W, Rs, Ps, tic, ric, ff, sf, fo, obs, dep

    def triangulate_with_depth(self, windows, depth_threshold, init_depth):
        W, Rs, Ps, tic, ric, ff, sf, fo, obs, dep = self._pack(windows)
        flag = np.ascontiguousarray(np.concatenate([w["estimate_flag"] for w in windows]), np.int32)
        d = C.c_double
        _chk(lib().gf_triangulate_with_depth_batch(self.h, len(windows), W, _p(Rs, d), _p(Ps, d), _p(tic, d), _p(ric, d), _p(ff, C.c_int), _p(sf, C.c_int), _p(fo, C.c_int),
                                                   _p(obs, d), d(depth_threshold), d(init_depth), _p(dep, d), _p(flag, C.c_int)))
        return [(dep[ff[b]:ff[b + 1]].copy(), flag[ff[b]:ff[b + 1]].copy()) for b in range(len(windows))]

    def moving_consistency(self, windows, focal_length):
        W, Rs, Ps, tic, ric, ff, sf, fo, obs, dep = self._pack(windows)
        rem = np.zeros(len(sf), np.int32)
        d = C.c_double
        _chk(lib().gf_moving_consistency_batch(self.h, len(windows), W, _p(Rs, d), _p(Ps, d), _p(tic, d), _p(ric, d), _p(ff, C.c_int), _p(sf, C.c_int), _p(fo, C.c_int),
                                               _p(obs, d), _p(dep, d), d(focal_length), _p(rem, C.c_int)))
        return [rem[ff[b]:ff[b + 1]].copy() for b in range(len(windows))]

    def stats(self):
        a, b, ms = C.c_longlong(0), C.c_longlong(0), C.c_double(0)
        _chk(lib().gf_featsweep_stats(self.h, C.byref(a), C.byref(b), C.byref(ms)))
        return dict(launches=a.value, features=b.value, kernel_ms=ms.value)


def wheel_preintegrate(dt, vel, gyr, vel0, gyr0, lin, noise):
    f = lambda a: np.ascontiguousarray(a, np.float64)
    dt, vel, gyr, vel0, gyr0, lin, noise = map(f, (dt, vel, gyr, vel0, gyr0, lin, noise))
    out = {"delta_p": np.zeros(3), "delta_q": np.zeros(4), "jacobian": np.zeros(18), "covariance": np.zeros(36)}
    sd = C.c_double(0)
    _chk(lib().gf_wheel_preintegrate(len(dt), _p(dt, C.c_double), _p(vel, C.c_double), _p(gyr, C.c_double), _p(vel0, C.c_double), _p(gyr0, C.c_double),
                                     _p(lin, C.c_double), _p(noise, C.c_double), _p(out["delta_p"], C.c_double), _p(out["delta_q"], C.c_double),
                                     _p(out["jacobian"], C.c_double), _p(out["covariance"], C.c_double), C.byref(sd)))
    out["sum_dt"] = sd.value
    return out


def double2vector(W, R0, P0, para_Pose, para_SpeedBias):
    """Estimator::double2vector pose part (host code, no GPU needed)"""
    f = lambda a: np.ascontiguousarray(a, np.float64).reshape(-1)
    R0, P0, pp, sb = map(f, (R0, P0, para_Pose, para_SpeedBias))
    out = [np.zeros(9 * (W + 1)), np.zeros(3 * (W + 1)), np.zeros(3 * (W + 1)), np.zeros(3 * (W + 1)), np.zeros(3 * (W + 1))]
    _chk(lib().gf_ba_double2vector(W, _p(R0, C.c_double), _p(P0, C.c_double), _p(pp, C.c_double), _p(sb, C.c_double), *[_p(o, C.c_double) for o in out]))
    return out


# ------------------------------------------------------------------ Estimator::processImage surface (gf_estimator_*)
class EstimatorCfg(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("window_size", "max_features", "max_visual", "use_imu", "use_wheel", "depth", "estimate_extrinsic",
                                       "estimate_wheel_extrinsic", "estimate_wheel_intrinsic", "estimate_td", "estimate_td_wheel", "use_mcc", "wdetect",
                                       "stationary_detect", "only_initial_with_wheel", "multiple_thread", "num_iterations", "with_tracker")] + \
               [(k, C.c_double) for k in ("acc_n", "gyr_n", "acc_w", "gyr_w", "g_norm", "wheel_vel_n", "wheel_gyr_n", "min_parallax_px", "depth_threshold",
                                          "init_depth", "focal_length", "td", "td_wheel", "sx", "sy", "sw")] + \
               [("tic", C.c_double * 3), ("ric", C.c_double * 9), ("tio", C.c_double * 3), ("rio", C.c_double * 9), ("tracker", TrackerCfg)] + \
               [(k, C.c_int) for k in ("gnss_enable", "gnss_track_num_thres", "max_gnss_per_frame")] + \
               [(k, C.c_double) for k in ("gnss_elevation_thres", "gnss_psr_std_thres", "gnss_dopp_std_thres", "gnss_ddt_sigma", "gnss_local_time_diff")] + \
               [("gnss_iono", C.c_double * 8), ("max_solver_time", C.c_double), ("extrinsic_type", C.c_int), ("extrinsic_type_wheel", C.c_int)]


class GnssEphem(C.Structure):
    """gf_gnss_ephem"""
    _fields_ = [("sat", C.c_int), ("sys", C.c_int), ("prn", C.c_int)] + [(k, C.c_double) for k in (
        "toe", "toc", "toe_tow", "A", "e", "i0", "OMG0", "omg", "M0", "delta_n", "OMG_dot", "i_dot", "cuc", "cus", "crc", "crs", "cic", "cis", "af0", "af1", "af2", "tgd0", "ura")]


class GnssGloEphem(C.Structure):
    """gf_gnss_glo_ephem"""
    _fields_ = [("sat", C.c_int), ("toe", C.c_double), ("pos", C.c_double * 3), ("vel", C.c_double * 3), ("acc", C.c_double * 3), ("tau_n", C.c_double), ("gamma", C.c_double)]


class GnssRawObs(C.Structure):
    """gf_gnss_raw_obs"""
    _fields_ = [("sat", C.c_int), ("sys", C.c_int)] + [(k, C.c_double) for k in ("time", "psr", "dopp", "psr_std", "dopp_std", "freq", "tow")]


def _fill(cs, d):
    for name, ct in cs._fields_:
        if name in ("pos", "vel", "acc", "sv_pos", "sv_vel"):
            for j in range(3):
                getattr(cs, name)[j] = float(d[name][j])
        else:
            setattr(cs, name, d[name])
    return cs


def gnss_obs_from_ephem(raw, eph=None, geph=None):
    """gf_gnss_obs_from_ephem: dict with the fields of gf_gnss_obs"""
    out = GnssObs()
    r = _fill(GnssRawObs(), raw)
    e = C.byref(_fill(GnssEphem(), eph)) if eph is not None else None
    g = C.byref(_fill(GnssGloEphem(), geph)) if geph is not None else None
    _chk(lib().gf_gnss_obs_from_ephem(C.byref(r), e, g, C.byref(out)))
    return {name: (np.array(getattr(out, name)[:]) if name in ("sv_pos", "sv_vel") else getattr(out, name)) for name, _ in GnssObs._fields_}


def gnss_eph2pos(t, eph=None, geph=None):
    pos, dts = np.zeros(3), C.c_double(0)
    e = C.byref(_fill(GnssEphem(), eph)) if eph is not None else None
    g = C.byref(_fill(GnssGloEphem(), geph)) if geph is not None else None
    _chk(lib().gf_gnss_eph2pos(C.c_double(t), e, g, _p(pos, C.c_double), C.byref(dts)))
    return pos, dts.value


class GnssObs(C.Structure):
    """gf_gnss_obs: one L1 observation of an epoch with the satellite state its ephemeris gives (include/groundfusion_hip.h)"""
    _fields_ = [("sat", C.c_int), ("sys", C.c_int)] + [(k, C.c_double) for k in ("time", "psr", "dopp", "psr_std", "dopp_std", "wavelength")] + \
               [("sv_pos", C.c_double * 3), ("sv_vel", C.c_double * 3)] + [(k, C.c_double) for k in ("svdt", "svddt", "tgd", "pr_uura", "dp_uura", "tow")]


def default_estimator_cfg(**kw):
    c = EstimatorCfg()
    _chk(lib().gf_estimator_default_cfg(C.byref(c)))
    for k, v in kw.items():
        if k in ("tic", "ric", "tio", "rio", "gnss_iono"):
            a = np.asarray(v, np.float64).reshape(-1)
            for i in range(len(a)):
                getattr(c, k)[i] = a[i]
        else:
            setattr(c, k, v)
    return c


class SlidingWindowEstimator:
    """Estimator::inputIMU / inputWheel / inputFeature / inputImage / processImage (estimator.h:104-116) on the C-ABI."""

    def __init__(self, cfg=None):
        self.cfg = cfg or default_estimator_cfg()
        self.h = C.c_void_p()
        _chk(lib().gf_estimator_create(C.byref(self.cfg), C.byref(self.h)))
        self.W = self.cfg.window_size

    def close(self):
        if getattr(self, "h", None):
            lib().gf_estimator_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def inputIMU(self, t, acc, gyr):
        a, g = np.ascontiguousarray(acc, np.float64), np.ascontiguousarray(gyr, np.float64)
        _chk(lib().gf_estimator_input_imu(self.h, C.c_double(t), _p(a, C.c_double), _p(g, C.c_double)))

    def inputWheel(self, t, vel, gyr):
        v, g = np.ascontiguousarray(vel, np.float64), np.ascontiguousarray(gyr, np.float64)
        _chk(lib().gf_estimator_input_wheel(self.h, C.c_double(t), _p(v, C.c_double), _p(g, C.c_double)))

    def inputFeature(self, t, image):
        """image: {feature_id: 8-vector} (the map trackImage returns)"""
        ids = sorted(image)
        obs = (FeatureObs * max(len(ids), 1))()
        for k, i in enumerate(ids):
            obs[k].id, obs[k].camera_id = int(i), 0
            v = np.asarray(image[i], np.float64).reshape(-1)
            for j in range(8):
                obs[k].v[j] = v[j]
        _chk(lib().gf_estimator_input_feature(self.h, C.c_double(t), obs, len(ids)))

    def inputGNSS(self, t, epoch):
        """epoch: list of dicts with the fields of gf_gnss_obs (Estimator::inputGNSS, estimator.cpp:397), or of gf_gnss_raw_obs (no "sv_pos")"""
        if epoch and "sv_pos" not in epoch[0]:
            return self.inputGNSSRaw(t, epoch)
        obs = (GnssObs * max(len(epoch), 1))()
        for k, o in enumerate(epoch):
            for name, _ in GnssObs._fields_:
                if name in ("sv_pos", "sv_vel"):
                    for j in range(3):
                        getattr(obs[k], name)[j] = float(o[name][j])
                else:
                    setattr(obs[k], name, o[name])
        _chk(lib().gf_estimator_input_gnss(self.h, C.c_double(t), obs, len(epoch)))

    def inputGNSSRaw(self, t, epoch):
        """epoch: list of dicts with the fields of gf_gnss_raw_obs; ephemerides through inputEphem"""
        obs = (GnssRawObs * max(len(epoch), 1))()
        for k, o in enumerate(epoch):
            _fill(obs[k], o)
        _chk(lib().gf_estimator_input_gnss_raw(self.h, C.c_double(t), obs, len(epoch)))

    def inputEphem(self, eph):
        """a dict with the fields of gf_gnss_ephem, or of gf_gnss_glo_ephem (recognised by its key "tau_n")"""
        if "tau_n" in eph:
            _chk(lib().gf_estimator_input_glo_ephem(self.h, C.byref(_fill(GnssGloEphem(), eph))))
        else:
            _chk(lib().gf_estimator_input_ephem(self.h, C.byref(_fill(GnssEphem(), eph))))

    def inputGNSSTimeDiff(self, t_diff):
        _chk(lib().gf_estimator_input_gnss_time_diff(self.h, C.c_double(t_diff)))

    def inputIonoParams(self, params):
        a = np.ascontiguousarray(params, np.float64)
        assert a.size == 8
        _chk(lib().gf_estimator_input_iono_params(self.h, _p(a, C.c_double)))

    def setGNSSAlignment(self, anc_ecef, yaw_enu_local, rcv_dt, rcv_ddt):
        a, d = np.ascontiguousarray(anc_ecef, np.float64), np.ascontiguousarray(rcv_dt, np.float64)
        assert a.size == 3 and d.size == 4
        _chk(lib().gf_estimator_set_gnss_alignment(self.h, _p(a, C.c_double), C.c_double(yaw_enu_local), _p(d, C.c_double), C.c_double(rcv_ddt)))

    def latest(self):
        """latest_time / latest_P / latest_Q / latest_V and the wheel counterparts (estimator.h:239-242, :354-356)"""
        a, b = np.zeros(16), np.zeros(16)
        _chk(lib().gf_estimator_get_latest(self.h, _p(a, C.c_double), _p(b, C.c_double)))
        return dict(time=a[0], P=a[1:4].copy(), Q=a[4:13].reshape(3, 3).copy(), V=a[13:16].copy(),
                    time_wheel=b[0], P_wheel=b[1:4].copy(), Q_wheel=b[4:13].reshape(3, 3).copy(), V_wheel=b[13:16].copy())

    def gnss_state(self):
        N = self.W + 1
        info, dt, ddt, yaw, anc, ecef, enu = np.zeros(8, np.int32), np.zeros((N, 4)), np.zeros(N), C.c_double(0), np.zeros(3), np.zeros(3), np.zeros(3)
        _chk(lib().gf_estimator_get_gnss_state(self.h, _p(info, C.c_int), _p(dt, C.c_double), _p(ddt, C.c_double), C.byref(yaw), _p(anc, C.c_double),
                                               _p(ecef, C.c_double), _p(enu, C.c_double)))
        return dict(gnss_ready=int(info[0]), lowspeed=int(info[1]), n_newest=int(info[2]), first_optimization=int(info[3]), queued=int(info[4]),
                    rcv_dt=dt, rcv_ddt=ddt, yaw_enu_local=yaw.value, anc_ecef=anc, ecef_pos=ecef, enu_pos=enu)

    def inputImage(self, t, img, depth=None):
        img = np.ascontiguousarray(img, np.uint8)
        cap = self.cfg.tracker.max_cnt + 8
        out = (FeatureObs * cap)()
        n = C.c_int(0)
        if depth is not None:
            depth = np.ascontiguousarray(depth, np.uint16)
            dp, ds = _p(depth, C.c_uint16), depth.shape[1]
        else:
            dp, ds = None, 0
        _chk(lib().gf_estimator_input_image(self.h, C.c_double(t), _p(img, C.c_uint8), img.shape[1], dp, ds, out, cap, C.byref(n)))
        return {out[k].id: np.array(out[k].v[:]) for k in range(n.value)}

    def state(self):
        N = self.W + 1
        Ps, Rs, Vs, Bas, Bgs, H = np.zeros((N, 3)), np.zeros((N, 3, 3)), np.zeros((N, 3)), np.zeros((N, 3)), np.zeros((N, 3)), np.zeros(N)
        info = np.zeros(16, np.int32)
        extr = np.zeros(32)
        _chk(lib().gf_estimator_get_state(self.h, _p(Ps, C.c_double), _p(Rs, C.c_double), _p(Vs, C.c_double), _p(Bas, C.c_double), _p(Bgs, C.c_double),
                                          _p(H, C.c_double), _p(info, C.c_int), _p(extr, C.c_double)))
        keys = ("frame_count", "solver_flag", "marginalization_flag", "n_features", "prior_valid", "prior_n", "systemstationary", "iterations",
                "successful_steps", "n_optimizations", "openExWheelEstimation", "last_track_num", "long_track_num", "new_feature_num", "sum_of_back",
                "sum_of_front")
        d = dict(zip(keys, (int(x) for x in info)))
        d.update(Ps=Ps, Rs=Rs, Vs=Vs, Bas=Bas, Bgs=Bgs, Headers=H, tic=extr[0:3].copy(), ric=extr[3:12].reshape(3, 3).copy(), tio=extr[12:15].copy(),
                 rio=extr[15:24].reshape(3, 3).copy(), sx=extr[24], sy=extr[25], sw=extr[26], td=extr[27], td_wheel=extr[28], initial_cost=extr[29],
                 final_cost=extr[30], last_average_parallax=extr[31])
        return d

    def flags(self):
        """(frame_count, solver_flag, marginalization_flag) without the window's states"""
        info = np.zeros(16, np.int32)
        _chk(lib().gf_estimator_get_state(self.h, None, None, None, None, None, None, _p(info, C.c_int), None))
        return int(info[0]), int(info[1]), int(info[2])

    def set_state(self, frame_count, solver_flag, Ps=None, Rs=None, Vs=None, Bas=None, Bgs=None):
        f = lambda a: None if a is None else _p(np.ascontiguousarray(a, np.float64), C.c_double)
        keep = [np.ascontiguousarray(a, np.float64) if a is not None else None for a in (Ps, Rs, Vs, Bas, Bgs)]
        _chk(lib().gf_estimator_set_state(self.h, frame_count, solver_flag, *[None if a is None else _p(a, C.c_double) for a in keep]))

    def features(self, cap=4096):
        ids, sf, nobs, ef, sflag = (np.zeros(cap, np.int32) for _ in range(5))
        dep = np.zeros(cap)
        n = C.c_int(0)
        _chk(lib().gf_estimator_get_features(self.h, cap, _p(ids, C.c_int), _p(sf, C.c_int), _p(nobs, C.c_int), _p(dep, C.c_double), _p(ef, C.c_int),
                                             _p(sflag, C.c_int), C.byref(n)))
        k = n.value
        return dict(id=ids[:k].copy(), start_frame=sf[:k].copy(), n_obs=nobs[:k].copy(), estimated_depth=dep[:k].copy(), estimate_flag=ef[:k].copy(),
                    solve_flag=sflag[:k].copy())

    def feedback(self, cap=4096):
        pid, rid = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        xyz = np.zeros((cap, 3))
        npred, nrem = C.c_int(0), C.c_int(0)
        _chk(lib().gf_estimator_get_feedback(self.h, cap, _p(pid, C.c_int), _p(xyz, C.c_double), C.byref(npred), _p(rid, C.c_int), C.byref(nrem)))
        return pid[:npred.value].copy(), xyz[:npred.value].copy(), rid[:nrem.value].copy()

    def prior(self, cap_n=512):
        n, nb = C.c_int(0), C.c_int(0)
        ids = np.zeros(256, np.int32)
        J, r = np.zeros(cap_n * cap_n), np.zeros(cap_n)
        _chk(lib().gf_estimator_get_prior(self.h, cap_n, 256, C.byref(n), C.byref(nb), _p(ids, C.c_int), _p(J, C.c_double), _p(r, C.c_double)))
        return dict(n=n.value, block_id=ids[:nb.value].copy(), J=J[:n.value ** 2].reshape(n.value, n.value).copy(), r=r[:n.value].copy())

    def debug(self, op, args=(), cap=8192):
        a = np.ascontiguousarray(args, np.float64).reshape(-1)
        out = np.zeros(cap)
        n = C.c_int(0)
        _chk(lib().gf_estimator_debug(self.h, op.encode(), _p(a, C.c_double) if a.size else None, int(a.size), _p(out, C.c_double), cap, C.byref(n)))
        return out[:n.value].copy()


# ------------------------------------------------------------------ ROS-free I/O (gf_io.hip): config files, trajectory output, raw frames
def estimator_cfg_from_yaml(config_file):
    """readParameters(config_file) (parameters.cpp:138-558) + the cam0_calib file -> EstimatorCfg with the tracker part filled."""
    c = EstimatorCfg()
    _chk(lib().gf_estimator_cfg_from_yaml(os.fsencode(config_file), C.byref(c)))
    return c


def tum_append(path, t, P, R):
    """one `t x y z qx qy qz qw` line as pubOdometry writes it (visualization.cpp:346-357)"""
    P, R = np.ascontiguousarray(P, np.float64), np.ascontiguousarray(R, np.float64)
    _chk(lib().gf_tum_append(os.fsencode(path), C.c_double(t), _p(P, C.c_double), _p(R, C.c_double)))


def read_pgm(path):
    """binary PGM -> u8 (maxval <= 255) or u16 array [h, w]"""
    w, h, mv = C.c_int(0), C.c_int(0), C.c_int(0)
    fn = lib().gf_pgm_read
    fn.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_size_t]
    _chk(fn(os.fsencode(path), C.byref(w), C.byref(h), C.byref(mv), None, 0))
    img = np.zeros((h.value, w.value), np.uint16 if mv.value > 255 else np.uint8)
    _chk(fn(os.fsencode(path), C.byref(w), C.byref(h), C.byref(mv), img.ctypes.data_as(C.c_void_p), img.nbytes))
    return img


def write_pgm(path, img):
    """the inverse of read_pgm (numpy only; used by the dataset exporter and tests)"""
    img = np.ascontiguousarray(img)
    assert img.dtype in (np.uint8, np.uint16) and img.ndim == 2
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n%d\n" % (img.shape[1], img.shape[0], 255 if img.dtype == np.uint8 else 65535))
        f.write(img.tobytes() if img.dtype == np.uint8 else img.astype(">u2").tobytes())


# ------------------------------------------------------------------ many sequences on one batched solver (gf_estimator_group_*)
class EstimatorGroup:
    """n Estimators sharing one batched back-end handle; members are SlidingWindowEstimator views (IMU / wheel input, state queries)."""

    def __init__(self, cfg, n, device_preint=None, device_sweeps=None):
        self.cfg, self.n = cfg, n
        self.g = C.c_void_p()
        _chk(lib().gf_estimator_group_create(C.byref(cfg), n, C.byref(self.g)))
        if device_preint is not None:   # one pre-integration launch per step instead of the members' host loops
            _chk(lib().gf_estimator_group_set_device_preint(self.g, int(bool(device_preint))))
        if device_sweeps is not None:   # triangulateWithDepth / movingConsistencyCheckW of all members as one launch each per step
            _chk(lib().gf_estimator_group_set_device_sweeps(self.g, int(bool(device_sweeps))))
        self.members = []
        for i in range(n):
            m = SlidingWindowEstimator.__new__(SlidingWindowEstimator)
            m.cfg, m.W, m.h = cfg, cfg.window_size, C.c_void_p()
            _chk(lib().gf_estimator_group_member(self.g, i, C.byref(m.h)))
            m.close = lambda: None          # owned by the group
            self.members.append(m)

    def close(self):
        if getattr(self, "g", None):
            for m in self.members:
                m.h = None
            lib().gf_estimator_group_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def inputFeatures(self, seqs, ts, images):
        """one frame {feature_id: 8-vector} per listed sequence; all of them are processed concurrently, their solves as one batch"""
        tot = sum(len(im) for im in images)
        obs = (FeatureObs * max(tot, 1))()
        k = 0
        for im in images:
            for i in sorted(im):
                obs[k].id, obs[k].camera_id = int(i), 0
                v = np.asarray(im[i], np.float64).reshape(-1)
                for j in range(8):
                    obs[k].v[j] = v[j]
                k += 1
        sq = np.ascontiguousarray(seqs, np.int32)
        tt = np.ascontiguousarray(ts, np.float64)
        no = np.ascontiguousarray([len(im) for im in images], np.int32)
        _chk(lib().gf_estimator_group_input_features(self.g, len(sq), _p(sq, C.c_int), _p(tt, C.c_double), obs, _p(no, C.c_int)))

    def submitFeatures(self, seqs, ts, obs, n_obs, stride=-1):
        """inputFeature's first half (estimator.cpp:447-459): hand the frames to the group's workers and return.  obs: OBS_DTYPE array, either the frames back to
        back (stride < 0) or a tracker's padded table [len(seqs)][stride]; the arrays are kept alive until wait()."""
        sq = np.ascontiguousarray(seqs, np.int32)
        tt = np.ascontiguousarray(ts, np.float64)
        no = np.ascontiguousarray(n_obs, np.int32)
        _chk(lib().gf_estimator_group_submit_features(self.g, len(sq), _p(sq, C.c_int), _p(tt, C.c_double), C.c_void_p(obs.ctypes.data), _p(no, C.c_int), C.c_longlong(stride)))
        self._flight = (sq, tt, no, obs)

    def wait(self):
        try:
            _chk(lib().gf_estimator_group_wait(self.g))
        finally:
            self._flight = None

    def stats(self):
        b, w, l = C.c_longlong(0), C.c_longlong(0), C.c_longlong(0)
        _chk(lib().gf_estimator_group_stats(self.g, C.byref(b), C.byref(w), C.byref(l)))
        return {"batches": b.value, "windows": w.value, "largest_batch": l.value}


# ------------------------------------------------------------------ ROS bag files (gf_bag_*, gf_ros_decode_*)
class Bag:
    """rosbag play without ROS: connections, messages of chosen topics in play order, and the three message types the node subscribes to."""

    def __init__(self, path):
        self.h = C.c_void_p()
        _chk(lib().gf_bag_open(str(path).encode(), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None):
            lib().gf_bag_close(self.h)
            self.h = None

    __del__ = close

    def connections(self):
        out = []
        for i in range(lib().gf_bag_connection_count(self.h)):
            cid, topic, typ = C.c_int(), C.create_string_buffer(256), C.create_string_buffer(256)
            _chk(lib().gf_bag_connection(self.h, i, C.byref(cid), topic, 256, typ, 256))
            out.append((cid.value, topic.value.decode(), typ.value.decode()))
        return out

    def select(self, topics=()):
        arr = (C.c_char_p * max(len(topics), 1))(*[t.encode() for t in topics])
        n = C.c_longlong()
        _chk(lib().gf_bag_select(self.h, arr, len(topics), C.byref(n)))
        return n.value

    def message(self, i):
        """(connection id, record time, payload bytes)"""
        cid, t, data, ln = C.c_int(), C.c_double(), C.POINTER(C.c_ubyte)(), C.c_size_t()
        _chk(lib().gf_bag_message(self.h, C.c_longlong(i), C.byref(cid), C.byref(t), C.byref(data), C.byref(ln)))
        return cid.value, t.value, C.string_at(data, ln.value)


def ros_decode_imu(payload):
    t, acc, gyr = C.c_double(), np.zeros(3), np.zeros(3)
    _chk(lib().gf_ros_decode_imu(payload, C.c_size_t(len(payload)), C.byref(t), _p(acc, C.c_double), _p(gyr, C.c_double)))
    return t.value, acc, gyr


def ros_decode_odometry(payload):
    t, lin, ang, pos = C.c_double(), np.zeros(3), np.zeros(3), np.zeros(3)
    _chk(lib().gf_ros_decode_odometry(payload, C.c_size_t(len(payload)), C.byref(t), _p(lin, C.c_double), _p(ang, C.c_double), _p(pos, C.c_double)))
    return t.value, lin, ang, pos


def ros_decode_image(payload, depth=False):
    t, w, h = C.c_double(), C.c_int(), C.c_int()
    _chk(lib().gf_ros_decode_image(payload, C.c_size_t(len(payload)), int(depth), C.byref(t), C.byref(w), C.byref(h), None, C.c_size_t(0)))
    out = np.zeros((h.value, w.value), np.uint16 if depth else np.uint8)
    _chk(lib().gf_ros_decode_image(payload, C.c_size_t(len(payload)), int(depth), C.byref(t), C.byref(w), C.byref(h), out.ctypes.data_as(C.c_void_p), C.c_size_t(out.nbytes)))
    return t.value, out


# ------------------------------------------------------------------ multi-GPU exchange without torch (gf_comm_*, gf_pose_gather)
def comm_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 makes it, the others receive it out of band)"""
    buf = (C.c_ubyte * 128)()
    _chk(lib().gf_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """gf_comm: an RCCL communicator of this process's rank on `device`"""

    def __init__(self, unique_id, world, rank, device=0):
        self.h = C.c_void_p()
        _chk(lib().gf_comm_create((C.c_ubyte * 128).from_buffer_copy(unique_id), world, rank, device, C.byref(self.h)))
        self.world, self.rank = world, rank

    def info(self):
        w, r, d, comm, stream = C.c_int(), C.c_int(), C.c_int(), C.c_void_p(), C.c_void_p()
        _chk(lib().gf_comm_info(self.h, C.byref(w), C.byref(r), C.byref(d), C.byref(comm), C.byref(stream)))
        return dict(world=w.value, rank=r.value, device=d.value, nccl_comm=comm.value, stream=stream.value)

    def allgather(self, x):
        x = np.ascontiguousarray(x, np.float64).reshape(-1)
        out = np.zeros((self.world, x.size))
        _chk(lib().gf_comm_allgather_f64(self.h, _p(x, C.c_double), x.size, _p(out, C.c_double)))
        return out

    def close(self):
        if getattr(self, "h", None):
            lib().gf_comm_destroy(self.h)
            self.h = None


def pose_gather(est, comm, count, d_out_ptr):
    """gf_pose_gather: est's newest poses -> ncclAllGather on comm's communicator and stream -> device array [world][count][7] at d_out_ptr"""
    i = comm.info()
    _chk(lib().gf_pose_gather(est.h, C.c_void_p(i["nccl_comm"]), C.c_void_p(i["stream"]), count, C.c_void_p(d_out_ptr)))


def numa_node_of_device(device=0):
    node, buf = C.c_int(), C.create_string_buffer(512)
    _chk(lib().gf_numa_node_of_device(device, C.byref(node), buf, 512))
    return node.value, buf.value.decode()
