"""Seeded synthetic marker-array frames — integer-only, backend independent.

The reference ships no sample video (`marker_detection.py:479` points at a file that is not in the
repo), so every parity test and the benchmark run on frames made here.  The recipe is integer
arithmetic plus one exactly-rounded `sqrt` of an integer, so NumPy (host, tests, CPU baseline) and
torch (device, benchmark batches) produce identical bytes for the same `(spec, seed, frame)`.

Geometry is kept in 1/16-pixel fixed point:
  pixel (x, y) has centre (16x, 16y); a dot with centre c16 and diameter D16 covers the pixel by
  cov16 = clamp(D16/2 + 8 - floor(|p - c16|), 0, 16)          (a 1-px anti-aliased edge)
  value = bg - ((bg - fg) * cov16 + 8) >> 4 + noise           (clipped to 0..255)
Noise is a 4-byte Irwin-Hall sum from a counter hash of (seed, frame, pixel).  Frame 0 carries no
position / diameter jitter (it defines the reference IDs); noise applies to every frame.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np

_M32 = 0xFFFFFFFF


def _hash32(x):
    """lowbias32 integer hash on int64 arrays/tensors holding values in [0, 2^32)."""
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


@dataclass(frozen=True)
class FrameSpec:
    width: int
    height: int
    centers16: np.ndarray          # [M, 2] int64 nominal (x16, y16)
    diameter16: int                # nominal dot diameter, 1/16 px
    bg: int = 190
    fg: int = 40
    jitter16: int = 48             # +-3 px centre jitter (uniform), frames > 0
    djitter16: int = 32            # +-2 px diameter jitter (uniform), frames > 0
    noise_sigma: float = 2.0       # grey levels; 0 disables
    grid: Optional[Tuple[int, int, int, int, int]] = None   # (nx, ny, x0_16, y0_16, pitch16)
    name: str = "custom"

    @property
    def n_markers(self) -> int:
        return int(self.centers16.shape[0])


def grid_spec(width, height, n, pitch, diameter, name="grid", **kw) -> FrameSpec:
    """n x n dots centred in the frame (pitch / diameter in px)."""
    x0 = (width - (n - 1) * pitch) * 8        # = 16 * (W - (n-1) pitch) / 2
    y0 = (height - (n - 1) * pitch) * 8
    jj, ii = np.meshgrid(np.arange(n), np.arange(n))          # ii = row (y), jj = col (x)
    c = np.stack([x0 + jj.ravel() * pitch * 16, y0 + ii.ravel() * pitch * 16], axis=1)
    return FrameSpec(width, height, c.astype(np.int64), int(diameter * 16),
                     grid=(n, n, int(x0), int(y0), int(pitch * 16)), name=name, **kw)


def config1() -> FrameSpec:
    """BASELINE config 1: 640x480, 7x7, pitch 60, diameter 20 (small branch)."""
    return grid_spec(640, 480, 7, 60, 20, name="640x480_7x7")


def config2() -> FrameSpec:
    """BASELINE configs 2-4: 1280x1024, 13x13, pitch 72, diameter 40 (large branch)."""
    return grid_spec(1280, 1024, 13, 72, 40, name="1280x1024_13x13")


def config5() -> FrameSpec:
    """BASELINE config 5: 1920x1200, 21x21, pitch 56, diameter 26 +-2."""
    return grid_spec(1920, 1200, 21, 56, 26, name="1920x1200_21x21")


# 65-dot ring layout of the physical sensor (nominal XY in mm), `ForceDistribution.py:29-95`
# condensed to (ring radius, count, first angle in degrees); the four outer dots sit on the axes.
_RINGS = ((0.0, 1, 0.0), (3.49, 6, 30.0), (6.92, 12, 0.0), (10.23, 18, 10.0), (13.37, 24, 0.0),
          (16.29, 4, 0.0))


def ring65_spec(width=480, height=450, px_per_mm=13.0, diameter=20, **kw) -> FrameSpec:
    pts = []
    for radius, count, a0 in _RINGS:
        for k in range(count):
            a = np.deg2rad(a0 + 360.0 * k / count)
            pts.append((width / 2 + radius * px_per_mm * np.cos(a),
                        height / 2 + radius * px_per_mm * np.sin(a)))
    c16 = np.round(np.asarray(pts) * 16).astype(np.int64)
    kw.setdefault("jitter16", 32)
    return FrameSpec(width, height, c16, int(diameter * 16), name="ring65", **kw)


# --------------------------------------------------------------------------------------------
def _dot_params(spec: FrameSpec, seed: int, frames, xp):
    """Jittered (cx16, cy16, r16) per frame and dot -> int64 [F, M] each."""
    M = spec.n_markers
    if xp is np:
        f = np.asarray(frames, dtype=np.int64).reshape(-1, 1)
        k = np.arange(M, dtype=np.int64).reshape(1, -1)
        c = np.asarray(spec.centers16, dtype=np.int64)
        cx0, cy0 = c[:, 0].reshape(1, -1), c[:, 1].reshape(1, -1)
    else:
        f = frames.reshape(-1, 1)
        k = xp.arange(M, dtype=xp.int64, device=f.device).reshape(1, -1)
        c = xp.as_tensor(np.ascontiguousarray(spec.centers16), dtype=xp.int64, device=f.device)
        cx0, cy0 = c[:, 0].reshape(1, -1), c[:, 1].reshape(1, -1)
    base = _hash32((f * 0x9E3779B1 + int(seed) * 0x85EBCA77 + 0x1234567) & _M32)
    hx = _hash32(base + k * 3 + 0)
    hy = _hash32(base + k * 3 + 1)
    hd = _hash32(base + k * 3 + 2)
    live = (f > 0)
    jx = (hx % (2 * spec.jitter16 + 1)) - spec.jitter16
    jy = (hy % (2 * spec.jitter16 + 1)) - spec.jitter16
    jd = (hd % (2 * spec.djitter16 + 1)) - spec.djitter16
    cx = cx0 + jx * live
    cy = cy0 + jy * live
    d16 = spec.diameter16 + jd * live
    return cx, cy, d16 // 2


def dot_truth(spec: FrameSpec, seed: int, frames) -> np.ndarray:
    """Ground-truth (cx, cy, diameter) in px, float64 [F, M, 3] — for sanity checks only."""
    cx, cy, r = _dot_params(spec, seed, np.asarray(frames, dtype=np.int64), np)
    return np.stack([cx / 16.0, cy / 16.0, r / 8.0], axis=-1)


def _noise(spec: FrameSpec, seed: int, f, pix, xp):
    if spec.noise_sigma <= 0:
        return 0
    K = int(round(spec.noise_sigma / 147.7994 * 65536))
    h = _hash32(_hash32((f * 0x27D4EB2F + int(seed) * 0x165667B1 + 0x7F4A7C15) & _M32) + pix)
    s = (h & 0xFF) + ((h >> 8) & 0xFF) + ((h >> 16) & 0xFF) + ((h >> 24) & 0xFF) - 510
    return (s * K + 32768) >> 16


def _isqrt_floor(d2, xp):
    if xp is np:
        return np.floor(np.sqrt(d2.astype(np.float64))).astype(np.int64)
    return xp.floor(xp.sqrt(d2.to(xp.float64))).to(xp.int64)


def _render(spec: FrameSpec, seed: int, frames, xp, device=None):
    H, W = spec.height, spec.width
    if xp is np:
        f = np.asarray(frames, dtype=np.int64).reshape(-1)
        ys = np.arange(H, dtype=np.int64).reshape(1, H, 1)
        xs = np.arange(W, dtype=np.int64).reshape(1, 1, W)
    else:
        f = xp.as_tensor(frames, dtype=xp.int64, device=device).reshape(-1)
        ys = xp.arange(H, dtype=xp.int64, device=device).reshape(1, H, 1)
        xs = xp.arange(W, dtype=xp.int64, device=device).reshape(1, 1, W)
    F = f.shape[0]
    cx, cy, r16 = _dot_params(spec, seed, f, xp)           # [F, M]
    if spec.grid is not None:
        nx, ny, x0, y0, p16 = spec.grid
        # every dot (radius + jitter + edge) stays inside its pitch cell -> one candidate per px
        assert spec.diameter16 // 2 + spec.djitter16 // 2 + spec.jitter16 + 16 < p16 // 2
        j = ((xs * 16 - x0 + p16 // 2) // p16)
        i = ((ys * 16 - y0 + p16 // 2) // p16)
        j = j.clip(0, nx - 1) if xp is np else j.clamp(0, nx - 1)
        i = i.clip(0, ny - 1) if xp is np else i.clamp(0, ny - 1)
        idx = (i * nx + j)                                  # [1, H, W]
        if xp is np:
            idx = np.broadcast_to(idx, (F, H, W)).reshape(F, -1)
            gx = np.take_along_axis(cx, idx, 1).reshape(F, H, W)
            gy = np.take_along_axis(cy, idx, 1).reshape(F, H, W)
            gr = np.take_along_axis(r16, idx, 1).reshape(F, H, W)
        else:
            idx = idx.expand(F, H, W).reshape(F, -1)
            gx = xp.gather(cx, 1, idx).reshape(F, H, W)
            gy = xp.gather(cy, 1, idx).reshape(F, H, W)
            gr = xp.gather(r16, 1, idx).reshape(F, H, W)
        dx = xs * 16 - gx
        dy = ys * 16 - gy
        cov = gr + 8 - _isqrt_floor(dx * dx + dy * dy, xp)
        cov = cov.clip(0, 16) if xp is np else cov.clamp(0, 16)
    else:
        cov = np.zeros((F, H, W), dtype=np.int64) if xp is np else \
            xp.zeros((F, H, W), dtype=xp.int64, device=device)
        for m in range(spec.n_markers):
            dx = xs * 16 - cx[:, m].reshape(F, 1, 1)
            dy = ys * 16 - cy[:, m].reshape(F, 1, 1)
            c = r16[:, m].reshape(F, 1, 1) + 8 - _isqrt_floor(dx * dx + dy * dy, xp)
            c = c.clip(0, 16) if xp is np else c.clamp(0, 16)
            cov = np.maximum(cov, c) if xp is np else xp.maximum(cov, c)
    v = spec.bg - (((spec.bg - spec.fg) * cov + 8) >> 4)
    pix = ys * W + xs
    v = v + _noise(spec, seed, f.reshape(F, 1, 1), pix, xp)
    v = v.clip(0, 255) if xp is np else v.clamp(0, 255)
    return v.astype(np.uint8) if xp is np else v.to(xp.uint8)


def make_frames(spec: FrameSpec, frames, seed: int = 0, channels: int = 1) -> np.ndarray:
    """NumPy frames: uint8 [F, H, W] (channels=1) or [F, H, W, 3] with B=G=R (channels=3)."""
    g = _render(spec, seed, frames, np)
    if channels == 3:
        g = np.repeat(g[..., None], 3, axis=-1)
    return g


def make_frames_torch(spec: FrameSpec, frames, seed: int = 0, channels: int = 1, device="cuda",
                      chunk: int = 32):
    """Same bytes as `make_frames`, rendered on `device` in chunks (benchmark batches)."""
    import torch
    frames = list(frames)
    out = torch.empty((len(frames), spec.height, spec.width) + ((3,) if channels == 3 else ()),
                      dtype=torch.uint8, device=device)
    for s in range(0, len(frames), chunk):
        g = _render(spec, seed, frames[s:s + chunk], torch, device=device)
        if channels == 3:
            out[s:s + chunk] = g[..., None].expand(*g.shape, 3)
        else:
            out[s:s + chunk] = g
    return out


def default_camera(spec: FrameSpec):
    """Synthetic camera of SURVEY.md §8d config 2 scaled to the frame (all float32)."""
    K = np.array([[1400.0, 0, spec.width / 2.0], [0, 1400.0, spec.height / 2.0], [0, 0, 1]],
                 dtype=np.float32)
    dist = np.zeros(5, dtype=np.float32)
    R = np.eye(3, dtype=np.float32)
    T = np.array([0.0, 0.0, 30.0], dtype=np.float32)
    return K, dist, R, T


def jittered_copies_torch(frame, n: int, seed: int = 0, shift: int = 3, sigma: float = 2.0, device="cuda", chunk: int = 256,
                          start: int = 0, total=None):
    """Frames `start .. start + n - 1` of a sequence of `total` (default n) frames made from ONE real frame (uint8 [H, W] or
    [H, W, 3], e.g. the reference's img/raw_markers.png): frame i is the original moved by a seeded integer (dx, dy) in
    [-shift, shift]^2 (edge pixels repeated) plus seeded Gaussian noise of `sigma` grey levels, rounded and clipped; frame
    0 is the original itself (it defines the IDs).  Made on `device` (the shifts of the whole sequence come from a CPU
    generator, the noise from the device's, seeded per block): the real-layout workload of bench.py and tests - whoever
    needs the bytes on the host copies them back.  Returns (frames, shifts [n, 2])."""
    import torch
    base = torch.as_tensor(np.ascontiguousarray(frame), device=device)
    H, W = base.shape[:2]
    total = n if total is None else int(total)
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    d = torch.randint(-shift, shift + 1, (total, 2), generator=g)
    d[0] = 0
    d = d[start:start + n]
    ys = (torch.arange(H)[None, :] - d[:, 1:2]).clamp_(0, H - 1).to(device)          # source row of every output row
    xs = (torch.arange(W)[None, :] - d[:, 0:1]).clamp_(0, W - 1).to(device)
    gd = torch.Generator(device=device)
    gd.manual_seed(int(seed) + 1 + 7919 * int(start))
    out = torch.empty((n,) + tuple(base.shape), dtype=torch.uint8, device=device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        f = base[ys[s:e, :, None], xs[s:e, None, :]].to(torch.float32)             # [c, H, W(, 3)]
        if sigma > 0:
            noise = torch.randn(f.shape, generator=gd, device=device) * sigma
            if s == 0 and start == 0:
                noise[0] = 0
            f = f + noise
        out[s:e] = f.round_().clamp_(0, 255).to(torch.uint8)
    return out, d.numpy()
