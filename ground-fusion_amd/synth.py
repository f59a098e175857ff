"""Seeded synthetic 640x480 RGB-D streams for parity tests and bench.py.

Scene: band-limited random texture (Gaussian-filtered uniform noise, sigma 2 px, contrast-stretched to
[20,235] so the reference's `> 250` brightness test, feature_tracker.cpp:160-167, never fires by accident)
mapped on a two-plane room (near wall inside depth_threshold 3 m, far wall beyond it), rendered by ray
casting through the pinhole camera of config/realsense/wt_cam.yaml.  Pure numpy; no GPU, no reference code.
"""
import numpy as np

FX, FY, CX, CY = 603.95556640625, 603.1257934570312, 324.0858154296875, 232.72303771972656
W, H = 640, 480


def _gauss_blur(a, sigma):
    r = int(3 * sigma + 0.5)
    x = np.arange(-r, r + 1)
    k = np.exp(-0.5 * (x / sigma) ** 2)
    k /= k.sum()
    a = np.apply_along_axis(lambda v: np.convolve(np.pad(v, r, mode="wrap"), k, mode="valid"), 0, a)
    a = np.apply_along_axis(lambda v: np.convolve(np.pad(v, r, mode="wrap"), k, mode="valid"), 1, a)
    return a


def make_texture(seed, size=1024, sigma=2.0):
    rng = np.random.default_rng(seed)
    a = _gauss_blur(rng.random((size, size)), sigma)
    a = (a - a.min()) / (a.max() - a.min())
    # sharpen contrast a little so Shi-Tomasi finds well-spread corners
    a = np.clip((a - 0.5) * 2.2 + 0.5, 0, 1)
    return (20.0 + 215.0 * a).astype(np.float32)


def _bilinear(tex, u, v):
    s = tex.shape[0]
    u = np.mod(u, s - 1.0)
    v = np.mod(v, s - 1.0)
    x0 = np.floor(u).astype(np.int64)
    y0 = np.floor(v).astype(np.int64)
    a = (u - x0).astype(np.float32)
    b = (v - y0).astype(np.float32)
    return ((1 - a) * (1 - b) * tex[y0, x0] + a * (1 - b) * tex[y0, x0 + 1]
            + (1 - a) * b * tex[y0 + 1, x0] + a * b * tex[y0 + 1, x0 + 1])


def warp_frame(tex, dx, dy, angle=0.0, scale=1.0, w=W, h=H, ox=200.0, oy=200.0):
    """Frame k of a simple similarity-warp sequence (tracker-only tests)."""
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    c, s = np.cos(angle), np.sin(angle)
    xc, yc = xs - w / 2, ys - h / 2
    u = scale * (c * xc - s * yc) + w / 2 + dx + ox
    v = scale * (s * xc + c * yc) + h / 2 + dy + oy
    img = _bilinear(tex, u, v)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def render_room(tex, R_wc, t_wc, w=W, h=H, near_z=2.0, far_z=5.0, split_x=0.6, px_per_m=260.0):
    """Ray-cast the two-plane room. R_wc,t_wc: camera-to-world. World: x right, y down, z forward.
    Returns (gray u8 HxW, depth u16 mm HxW)."""
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    rays_c = np.stack([(xs - CX) / FX, (ys - CY) / FY, np.ones_like(xs)], -1)
    rays_w = rays_c @ R_wc.T
    o = t_wc
    lam_n = (near_z - o[2]) / rays_w[..., 2]
    xn = o[0] + lam_n * rays_w[..., 0]
    lam_f = (far_z - o[2]) / rays_w[..., 2]
    use_near = xn < split_x
    lam = np.where(use_near, lam_n, lam_f)
    P = o[None, None, :] + lam[..., None] * rays_w
    u = P[..., 0] * px_per_m * np.where(use_near, 1.0, 0.55) + 3000.0 + np.where(use_near, 0.0, 517.0)
    v = P[..., 1] * px_per_m * np.where(use_near, 1.0, 0.55) + 3000.0
    img = _bilinear(tex, u, v)
    depth_m = lam  # z in camera frame = lam because ray_c z == 1 (before rotation) -> depth along optical axis
    depth = np.clip(np.rint(depth_m * 1000.0), 0, 65535).astype(np.uint16)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8), depth


def tracker_sequence(seed, n_frames, w=W, h=H):
    """Deterministic warp sequence with sub-pixel motion: list of u8 frames."""
    tex = make_texture(seed)
    rng = np.random.default_rng(seed + 7)
    vx, vy = rng.uniform(-6, 6), rng.uniform(-4, 4)
    wz = rng.uniform(-0.004, 0.004)
    frames = []
    for k in range(n_frames):
        frames.append(warp_frame(tex, vx * k + 0.37 * np.sin(0.9 * k), vy * k, wz * k, 1.0 + 0.002 * k, w, h))
    return frames
