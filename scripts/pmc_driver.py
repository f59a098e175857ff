"""torch-free driver for PMC collection (rocprofv3 --pmc segfaults with the torch-bundled HIP runtime in the process).
  python scripts/pmc_driver.py tracker [B]   B sequences x 4 frames through the host-image entry point (default 256: the bench's batch)
  python scripts/pmc_driver.py backend [B]   B resident windows: 2 x (8-iteration solve + MARGIN_OLD); backend_split: the same with the visual sweep as two kernels
  python scripts/pmc_driver.py calib         the three FETCH_SIZE calibration kernels (known byte counts)"""
import os, sys, ctypes as C
os.environ["GF_NO_TORCH_PRELOAD"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ground-fusion_amd"))
import numpy as np, gfamd
what = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if what == "tracker":
    import synth
    N = 4
    base = [synth.tracker_sequence(1000 + b, N) for b in range(8)]
    depth = np.full(base[0][0].shape, 1800, np.uint16)
    trk = gfamd.FeatureTracker(gfamd.default_cfg(batch=B))
    for k in range(N):
        trk.trackImageBatch([k / 15.0] * B, [base[b % 8][k] for b in range(B)], [depth] * B)
    st = trk.stats()
    print("PMCINFO tracker B %d lk_launches %d lk_points %d alg_bytes_per_launch %.1f" % (B, st["lk_launches"], st["lk_points"], (484.0 * 5 * st["lk_level_passes"] + 484.0 * st["lk_iterations"]) / max(st["lk_launches"], 1)))
    trk.close()
elif what in ("backend", "backend_split"):
    import synth_window as SW
    est = gfamd.Estimator(batch=B)
    if what == "backend_split":     # north_star's formulation: sweep -> block rows in HBM -> contraction-only kernel (gf_ba_set_split_jtj)
        est.set_split_jtj(True)
    base = [SW.make_window(1000 + b, gfamd) for b in range(8)]
    est.upload([base[b % 8] for b in range(B)])
    for it in range(2):
        est.solve_resident(8, 0, True)
    print("PMCINFO backend B %d" % B)
    est.close()
else:
    fn = gfamd.lib().gf_calib_fetch
    fn.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    for mode in (0, 1, 2):
        rb, ln, ms = C.c_double(0), C.c_double(0), C.c_double(0)
        gfamd._chk(fn(mode, 2 << 30, C.byref(rb), C.byref(ln), C.byref(ms)))
        print("PMCINFO calib mode %d requested_bytes %.0f lines64 %.0f ms %.3f" % (mode, rb.value, ln.value, ms.value))
