"""Deterministic synthetic frames for parity tests and bench.py (SURVEY 8d / BASELINE.md section 4).

cur = low-pass filtered noise texture scaled to the full pixel range (+ small noise);
ref = cur displaced by a known global full-pel motion (dx, dy) plus independent small noise, so a
motion search has something real to find.  Planes are returned PADDED (edge replicated, like the
reference's extendPicBorder, pixel.cpp:1044-1058) with `margin` pixels on every side.
"""
import numpy as np


def _box3(a):
    for _ in range(3):
        a = (a + np.roll(a, 1, 0) + np.roll(a, 1, 1) + np.roll(np.roll(a, 1, 0), 1, 1)) * 0.25
    return a


def frame_pair(width, height, depth, seed, margin=96, max_shift=24, noise=2.0):
    """Returns (cur_padded, ref_padded, stride, (dx, dy)); origin of the picture is [margin, margin]."""
    rng = np.random.default_rng(0x5EED0000 + seed)
    pm = (1 << depth) - 1
    dt = np.uint8 if depth == 8 else np.uint16
    ext = max_shift + 8
    big = rng.random((height + 2 * ext, width + 2 * ext))
    big = _box3(_box3(big))
    big = (big - big.min()) / (big.max() - big.min() + 1e-12)
    # add coarse structure so large PUs have gradients too
    yy, xx = np.mgrid[0:big.shape[0], 0:big.shape[1]]
    big = 0.75 * big + 0.25 * (0.5 + 0.5 * np.sin(xx / 37.0 + seed) * np.cos(yy / 53.0))
    dx, dy = (int(v) for v in rng.integers(-max_shift, max_shift + 1, 2))
    cur = big[ext:ext + height, ext:ext + width]
    ref = big[ext + dy:ext + dy + height, ext + dx:ext + dx + width]   # ref(x, y) = cur(x + dx, y + dy)
    scale = pm * (1 << 0)
    cur = np.clip(cur * scale + rng.normal(0, noise * (1 << (depth - 8)), cur.shape), 0, pm).astype(dt)
    ref = np.clip(ref * scale + rng.normal(0, noise * (1 << (depth - 8)), ref.shape), 0, pm).astype(dt)
    curp = np.pad(cur, margin, mode="edge")
    refp = np.pad(ref, margin, mode="edge")
    return np.ascontiguousarray(curp), np.ascontiguousarray(refp), width + 2 * margin, (dx, dy)
