"""Deterministic synthetic frames for parity tests and bench.py (SURVEY 8d / BASELINE.md section 4).

cur = low-pass filtered noise texture scaled to the full pixel range (+ small noise);
ref = cur displaced by a known global full-pel motion plus independent small noise, so a motion search has
something real to find: the block at (x, y) of cur sits at (x + dx, y + dy) in ref, i.e. the true MV is (dx, dy).  Planes are returned PADDED (edge replicated, like the
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
    bh, bw = height + 2 * ext, width + 2 * ext
    # multi-octave value noise: structure at 64/32/16/8-pixel scales (so every PU size has gradients and the SAD
    # surface has a basin around the true motion) plus a little fine texture
    big = np.zeros((bh, bw))
    for cell, amp in ((64, 0.40), (32, 0.25), (16, 0.18), (8, 0.12)):
        g = rng.random((bh // cell + 3, bw // cell + 3))
        up = np.kron(g, np.ones((cell, cell)))
        for _ in range(2):                      # two box passes of the cell size ~ quadratic B-spline smoothing
            c = np.cumsum(np.cumsum(np.pad(up, ((cell, 0), (cell, 0))), 0), 1)
            up = (c[cell:, cell:] - c[:-cell, cell:] - c[cell:, :-cell] + c[:-cell, :-cell]) / (cell * cell)
        big += amp * up[cell:cell + bh, cell:cell + bw]
    big += 0.05 * _box3(rng.random((bh, bw)))
    big = (big - big.min()) / (big.max() - big.min() + 1e-12)
    dx, dy = (int(v) for v in rng.integers(-max_shift, max_shift + 1, 2))
    cur = big[ext:ext + height, ext:ext + width]
    ref = big[ext - dy:ext - dy + height, ext - dx:ext - dx + width]   # ref(x + dx, y + dy) = cur(x, y): true MV = (dx, dy)
    scale = pm * (1 << 0)
    cur = np.clip(cur * scale + rng.normal(0, noise * (1 << (depth - 8)), cur.shape), 0, pm).astype(dt)
    ref = np.clip(ref * scale + rng.normal(0, noise * (1 << (depth - 8)), ref.shape), 0, pm).astype(dt)
    curp = np.pad(cur, margin, mode="edge")
    refp = np.pad(ref, margin, mode="edge")
    return np.ascontiguousarray(curp), np.ascontiguousarray(refp), width + 2 * margin, (dx, dy)
