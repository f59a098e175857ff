"""profiles/pmc_summary.json from the PMC passes of scripts/pmc_collect.sh (gpurun_out/<tag>_pmc_*.csv / .info): what bench.py quotes.
usage: python scripts/pmc_summarize.py <tag>"""
import csv, json, os, re, shutil, sys
tag = sys.argv[1]
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out")
def load(name):
    out = {}
    with open(os.path.join(G, "%s_pmc_%s.csv" % (tag, name))) as f:
        for r in csv.DictReader(f):
            out.setdefault(r["kernel"], {})[r["counter"]] = (float(r["mean_per_launch"]), int(r["launches"]), float(r["mean_duration_ns"]))
    return out
def info(name):
    return open(os.path.join(G, "%s_pmc_%s.info" % (tag, name))).read()
cal, tf, tw, tsq, bsq, bsq2 = load("calib_fetch"), load("tracker_fetch"), load("tracker_write"), load("tracker_sq"), load("backend_sq"), load("backend_sq2")
try:
    bf, bw = load("backend_fetch"), load("backend_write")
except OSError:
    bf, bw = {}, {}
ci = {int(m.group(1)): (float(m.group(2)), float(m.group(3))) for m in re.finditer(r"calib mode (\d) requested_bytes (\d+) lines64 (\d+)", info("calib_fetch"))}
kb = 1024.0
stream = cal["calib_stream_kernel"]["FETCH_SIZE"][0] * kb / ci[0][0]
# the two tile launches are the same kernel: dispatch order = mode 1, mode 2 (identical values, see the note)
tile = cal["calib_tile_kernel"]["FETCH_SIZE"][0] * kb
m = re.search(r"tracker B (\d+) lk_launches (\d+) lk_points (\d+) alg_bytes_per_launch ([\d.]+)", info("tracker_fetch"))
B, lk_launches, lk_points, alg = int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4))
ppl = lk_points / lk_launches
S = {"_source": "scripts/pmc_collect.sh %s on one MI355X (separate rocprofv3 --pmc passes with --kernel-trace only, torch-free driver scripts/pmc_driver.py, %d sequences / windows); "
                "files profiles/%s_pmc_*.csv" % (tag, B, tag),
     "fetch_size_calibration": {
         "streaming_16B_per_lane": {"FETCH_SIZE_bytes_over_requested_bytes": stream},
         "lk_tile_32x32_two_16B_lanes_per_row": {"requested_bytes": ci[1][0], "FETCH_SIZE_bytes": tile, "row_segments": ci[1][1],
                                                 "FETCH_SIZE_bytes_per_row_segment": tile / ci[1][1]},
         "reading": "FETCH_SIZE counts 64 B per 128-byte-line request in every pattern: half the bytes of a 16-B/lane stream (as MI355X_MICROARCH.md says), and 64 B per 32-byte "
                    "row segment of an LK tile refill whether the segment sits in one 64-byte line or straddles two (same count in both calibration launches).  "
                    "HBM-side bytes of a launch therefore lie between FETCH_SIZE (64-B sectors) and 2 x FETCH_SIZE (whole 128-B lines); both are quoted."}}
def traffic(k):
    f, w = tf[k]["FETCH_SIZE"][0] * kb, tw[k]["WRITE_SIZE"][0] * kb
    return {"FETCH_SIZE_bytes": f, "WRITE_SIZE_bytes": w, "hbm_bytes_lower": f + w, "hbm_bytes_upper": 2 * f + w, "launches": tf[k]["FETCH_SIZE"][1], "duration_us_profiled": tf[k]["FETCH_SIZE"][2] / 1e3}
lk = traffic("gf::lk_track_kernel")
lk.update({"points_per_launch": ppl, "algorithmic_bytes_per_launch": alg, "hbm_bytes_per_point": lk["hbm_bytes_upper"] / ppl, "hbm_bytes_per_point_lower": lk["hbm_bytes_lower"] / ppl,
           "over_fetch_vs_algorithmic": [lk["hbm_bytes_lower"] / alg, lk["hbm_bytes_upper"] / alg],
           "note": "PMC at %d sequences (%d points per launch), profiles/%s_pmc_tracker_fetch.csv / _write.csv: traffic = 2 x FETCH_SIZE + WRITE_SIZE (upper bound: whole 128-B lines; "
                   "lower bound FETCH_SIZE + WRITE_SIZE = %.0f B per point), scaled to this launch by points" % (B, ppl, tag, lk["hbm_bytes_lower"] / ppl)})
S["lk_track_kernel"] = lk
for k in ("gf::detect_strip_kernel<30>", "gf::pyr_head_kernel", "gf::pyr_level0_vec16_kernel", "gf::pyr_down_pad4_kernel", "gf::pyr_down_tail_kernel", "gf::select_topk_kernel", "gf::select_corners_kernel"):
    if k in tf:
        S[k.split("::")[1].split("<")[0]] = traffic(k)
def sq(tab, k):
    d = {c: v[0] for c, v in tab[k].items()}
    d["duration_us_profiled"] = list(tab[k].values())[0][2] / 1e3
    return d
def util(d):   # MFMA busy cycles over SIMD-cycles of the launch (1024 SIMDs at 2.4 GHz)
    return d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (d["duration_us_profiled"] * 1e-6 * 2.4e9 * 1024)
def find(tab, *prefixes):   # kernel names carry their template arguments ("gfb::ba_step<false; 8>"): match by prefix, first hit
    for p in prefixes:
        for k in tab:
            if k.startswith(p):
                return k
    return None
for k, name in ((find(bsq, "gfb::ba_linearize_visual_win<false; 12; 0>", "gfb::ba_linearize_visual_win<false; 12>"), "ba_linearize_visual_win"), (find(bsq, "gfb::ba_step<false"), "ba_step"),
                (find(bsq, "gfb::ba_linearize_misc_win"), "ba_linearize_misc_win"), (find(bsq, "gfb::ba_marg_finish<false>"), "ba_marg_finish")):
    if k in bsq:
        d = sq(bsq, k)
        d.update({c: v[0] for c, v in bsq2.get(k, {}).items()})
        d["mfma_utilisation"] = util(d)
        d["wave_cycle_split"] = {"parked": d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], "issue_stalled": d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], "issuing": d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"]}
        if k in bf and k in bw:   # HBM-side traffic of the launch (256 windows): FETCH_SIZE counts 64 B per request (see the calibration), so reads lie between 1x and 2x
            f, wv = bf[k]["FETCH_SIZE"][0] * kb, bw[k]["WRITE_SIZE"][0] * kb
            d.update({"FETCH_SIZE_bytes": f, "WRITE_SIZE_bytes": wv, "hbm_bytes_lower": f + wv, "hbm_bytes_upper": 2 * f + wv,
                      "hbm_GBps_lower": (f + wv) / (d["duration_us_profiled"] * 1e-6) / 1e9, "hbm_GBps_upper": (2 * f + wv) / (d["duration_us_profiled"] * 1e-6) / 1e9})
        d["note"] = "counter-derived MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (profiled kernel duration x 2.4 GHz x 1024 SIMDs), profiles/%s_pmc_backend_sq.csv" % tag
        S[name] = d
try:   # the split formulation's two kernels (scripts/pmc_collect.sh: backend_split passes)
    ssq = load("backend_split_sq")
    try:
        sf, sw = load("backend_split_fetch"), load("backend_split_write")
    except OSError:
        sf, sw = {}, {}
    for k, name in ((find(ssq, "gfb::ba_linearize_visual_win<false; 12; 2>"), "ba_linearize_visual_win_contract"), (find(ssq, "gfb::ba_linearize_visual_win<false; 12; 1>"), "ba_linearize_visual_win_sweep")):
        if k:
            d = sq(ssq, k)
            d["mfma_utilisation"] = util(d)
            d["wave_cycle_split"] = {"parked": d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], "issue_stalled": d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"], "issuing": d["SQ_ACTIVE_INST_ANY"] / d["SQ_WAVE_CYCLES"]}
            if k in sf and k in sw:
                f, wv = sf[k]["FETCH_SIZE"][0] * kb, sw[k]["WRITE_SIZE"][0] * kb
                d.update({"FETCH_SIZE_bytes": f, "WRITE_SIZE_bytes": wv, "hbm_GBps_lower": (f + wv) / (d["duration_us_profiled"] * 1e-6) / 1e9, "hbm_GBps_upper": (2 * f + wv) / (d["duration_us_profiled"] * 1e-6) / 1e9})
            d["note"] = "split formulation (gf_ba_set_split_jtj), profiles/%s_pmc_backend_split_sq.csv" % tag
            S[name] = d
            print(name, "mfma util %.3f" % d["mfma_utilisation"], d["wave_cycle_split"], "dur us", d["duration_us_profiled"])
    for n in ("backend_split_sq",) + (("backend_split_fetch", "backend_split_write") if sf else ()):
        for ext in ("csv", "info"):
            shutil.copy(os.path.join(G, "%s_pmc_%s.%s" % (tag, n, ext)), os.path.join(R, "profiles", "%s_pmc_%s.%s" % (tag, n, ext)))
except OSError:
    pass
for k, name in (("gf::lk_track_kernel", "lk_track_kernel_sq"), ("gf::detect_strip_kernel<30>", "detect_strip_kernel_sq")):
    if k in tsq:
        S[name] = sq(tsq, k)
# which device code the counters describe: bench.py quotes them only while the library's sources still hash to this (same function as bench.kernel_source_sha16)
import hashlib
_h = hashlib.sha256()
_d = os.path.join(R, "ground-fusion_amd", "csrc")
for _fn in sorted(os.listdir(_d)):
    if _fn.endswith((".hip", ".hpp")):
        _h.update(_fn.encode()); _h.update(open(os.path.join(_d, _fn), "rb").read())
S["kernel_source_sha16"] = _h.hexdigest()[:16]
json.dump(S, open(os.path.join(R, "profiles", "pmc_summary.json"), "w"), indent=1)
for n in ("calib_fetch", "tracker_fetch", "tracker_write", "tracker_sq", "backend_sq", "backend_sq2") + (("backend_fetch", "backend_write") if bf else ()):
    for ext in ("csv", "info"):
        shutil.copy(os.path.join(G, "%s_pmc_%s.%s" % (tag, n, ext)), os.path.join(R, "profiles", "%s_pmc_%s.%s" % (tag, n, ext)))
print(json.dumps({k: S[k] for k in ("fetch_size_calibration",)}, indent=1))
print("lk", {k: v for k, v in lk.items() if k != "note"})
for n in ("ba_linearize_visual_win", "ba_step", "ba_linearize_misc_win", "ba_marg_finish"):
    print(n, "mfma util %.3f" % S[n]["mfma_utilisation"], S[n]["wave_cycle_split"], "dur us", S[n]["duration_us_profiled"])
