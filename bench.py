#!/usr/bin/env python3
"""bench.py — Ground-Fusion hot path on MI355X (contract: see the task brief / DESIGN.md "Measurement").

A step = one pass of the hot path over one batch of synthetic input resident in HBM: every one of the
`--batch` independent 640x480 sequences owned by this GPU advances by one frame through
FeatureTracker::trackImage (HIP pyramid + Scharr + LK fwd/rev + Shi-Tomasi top-up) [+ the sliding-window
solve once the back end is enabled].  One process per GPU; sequences are sharded across ranks with no
data-path collective (weak scaling); the only collectives are the timing/throughput reductions.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ground-fusion_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

W, H = 640, 480
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def make_frames(n_frames, batch, seed0, device):
    """Synthetic sequences generated on the GPU (plumbing): band-limited texture, similarity warp per frame.
    Returns uint8 [n_frames, batch, H, W] and uint16-as-int16 depth [batch, H, W]."""
    import synth
    g = torch.Generator(device="cpu")
    n_tex = min(batch, 8)
    tex = torch.stack([torch.from_numpy(synth.make_texture(seed0 + i)) for i in range(n_tex)]).to(device)  # [n_tex,1024,1024]
    ys, xs = torch.meshgrid(torch.arange(H, device=device, dtype=torch.float32), torch.arange(W, device=device, dtype=torch.float32), indexing="ij")
    frames = torch.empty((n_frames, batch, H, W), dtype=torch.uint8, device=device)
    rng = np.random.default_rng(seed0)
    for b in range(batch):
        vx, vy = rng.uniform(-6, 6), rng.uniform(-4, 4)
        wz = rng.uniform(-0.004, 0.004)
        ox, oy = rng.uniform(150, 300), rng.uniform(150, 300)
        t = tex[b % n_tex]
        for k in range(n_frames):
            ang, sc = wz * k, 1.0 + 0.002 * k
            c, s = np.cos(ang), np.sin(ang)
            xc, yc = xs - W / 2, ys - H / 2
            u = sc * (c * xc - s * yc) + W / 2 + vx * k + 0.37 * np.sin(0.9 * k) + ox
            v = sc * (s * xc + c * yc) + H / 2 + vy * k + oy
            u = torch.remainder(u, 1022.0); v = torch.remainder(v, 1022.0)
            x0 = u.floor().long(); y0 = v.floor().long()
            a = u - x0; bb = v - y0
            img = (1 - a) * (1 - bb) * t[y0, x0] + a * (1 - bb) * t[y0, x0 + 1] + (1 - a) * bb * t[y0 + 1, x0] + a * bb * t[y0 + 1, x0 + 1]
            frames[k, b] = img.round().clamp(0, 255).to(torch.uint8)
    depth = torch.full((batch, H, W), 1800, dtype=torch.int16, device=device)
    return frames, depth


def cpu_baseline(frames_host, dt, threads):
    """Oracle (CPU restatement, kind 'port') on a bounded sample of the same workload: `threads` sequences in parallel,
    each single-threaded like the reference's FeatureTracker; returns tracked-features/s (same definition as the GPU)."""
    import threading
    import oracle_py
    oracle_py.lib()
    n_frames, nseq = frames_host.shape[0], frames_host.shape[1]
    depth = np.full((H, W), 1800, np.uint16)
    counts = [0] * nseq

    def run(b):
        tr = oracle_py.Tracker(oracle_py.default_cfg())
        prev = set()
        for k in range(n_frames):
            ids, _ = tr.track(dt * k, frames_host[k, b], depth)
            if k > 0:
                counts[b] += len(prev & set(ids.tolist()))
            prev = set(ids.tolist())

    t0 = time.perf_counter()
    ths = [threading.Thread(target=run, args=(b,)) for b in range(nseq)]
    for i in range(0, nseq, threads):
        for th in ths[i:i + threads]:
            th.start()
        for th in ths[i:i + threads]:
            th.join()
    el = time.perf_counter() - t0
    return sum(counts) / el, el


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="independent sequences per GPU")
    ap.add_argument("--max-cnt", type=int, default=150)
    ap.add_argument("--min-dist", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)

    import gfamd
    gfamd._chk(gfamd.lib().gf_set_device(local_rank))
    B, K, Wm = args.batch, args.steps, args.warmup
    dt = 1.0 / 15.0
    n_frames = Wm + K + 1
    frames, depth = make_frames(n_frames, B, 1000 + 100 * rank, dev)
    torch.cuda.synchronize()
    frame_bytes = B * H * W

    trk = gfamd.FeatureTracker(gfamd.default_cfg(batch=B, max_cnt=args.max_cnt, min_dist=args.min_dist))
    trk.set_profiling(True)
    step = [0]

    def do_step():
        k = step[0]
        trk.trackImageBatchDevice([dt * k] * B, frames.data_ptr() + k * frame_bytes, depth.data_ptr(), unpack=False)
        step[0] += 1

    for _ in range(Wm + 1):  # frame 0 only detects; it is part of the warm-up
        do_step()
    trk.reset_stats()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        do_step()
    barrier()
    el = time.perf_counter() - t0
    st = trk.stats()

    tot = torch.tensor([el, float(st["tracked_features"]), float(st["output_features"])], dtype=torch.float64, device=dev)
    if dist is not None:
        mx = tot.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone(); dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        el_max, tracked, outf = mx[0].item(), sm[1].item(), sm[2].item()
    else:
        el_max, tracked, outf = el, st["tracked_features"], st["output_features"]

    if rank == 0:
        # roofline of the dominant kernel (lk_track_kernel): algorithmic bytes per SURVEY.md §8(d):
        #   484*(1+4) B per (point, level pass) [u8 window + s16x2 derivative window] + 484 B per iteration [moving window]
        launches = max(st["lk_launches"], 1)
        alg_bytes = 484.0 * 5.0 * st["lk_level_passes"] + 484.0 * st["lk_iterations"]
        lk_ms = st["ms_lk"] / launches
        achieved = alg_bytes / launches / (lk_ms * 1e-3) / 1e9 if lk_ms > 0 else 0.0
        res = {
            "metric": "sliding-window solves/sec + tracked-features/sec, 640x480x10-frame window",
            "value": tracked / el_max, "unit": "tracked-features/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": 1e3 * el_max / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/s16 fixed-point + f32 (LK), f32/f64 (Shi-Tomasi)", "data": "synthetic",
            "config": {"workload": "configs[1] front end: %d independent 640x480 RGBD sequences per GPU, max_cnt %d, min_dist %d, LK 21x21 3 levels + flow-back, Shi-Tomasi top-up"
                                   % (B, args.max_cnt, args.min_dist), "sequences_per_gpu": B, "window": 10, "features": args.max_cnt},
            "frames_per_s": B * world * K / el_max, "output_features_per_s": outf / el_max, "solves_per_s": None,
            "gpu_ms_per_step": {"pyramid": st["ms_pyramid"] / K, "lk": st["ms_lk"] / K, "detect": st["ms_detect"] / K, "total": st["ms_total_gpu"] / K},
            "host_ms_per_step": {k: st[k] / K for k in ("ms_host_pre", "ms_wait_lk", "ms_host_mid", "ms_wait_detect", "ms_host_post")},
            "roofline": {"kernel": "lk_track_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "launch_ms": lk_ms, "algorithmic_bytes_per_launch": alg_bytes / launches,
                         "points_per_launch": st["lk_points"] / launches, "iterations_per_launch": st["lk_iterations"] / launches},
        }
        if not args.no_cpu_baseline:
            nseq = min(8, B)
            cores = min(os.cpu_count() or 1, nseq)
            fh = frames[: min(n_frames, 12), :nseq].cpu().numpy()
            v, cel = cpu_baseline(fh, dt, cores)
            res["cpu_baseline"] = {"value": v, "unit": "tracked-features/s", "cores": cores, "kind": "port",
                                   "sample": "%d sequences x %d frames of the same synthetic streams through the CPU oracle tracker, %d threads (one per sequence), %.1f s" % (nseq, fh.shape[0], cores, cel)}
        print(json.dumps(res))
    trk.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
