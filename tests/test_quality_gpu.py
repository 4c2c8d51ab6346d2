"""x265hip_ssim_frame through the C ABI against the oracle (pinned to the reference encoder's own per-frame SSIM, see
test_quality_oracle_vs_ref.py) and against the committed encoder fixtures.  Float arithmetic in the reference's order of operations: tolerance 0."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from depths import DEPTHS, GOLDEN_DEPTHS

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, HERE)
import x265hip  # noqa: E402,F401  (makes the package importable as x265hip_pkg)
from oracle_py import Oracle  # noqa: E402
from test_quality_oracle_vs_ref import golden_pictures, ssim_oracle  # noqa: E402

pytestmark = pytest.mark.gpu


def hip_ssim(api, rec, src, ctu, stride_pad=(0, 0), time_it=False):
    t = api.torch
    H, W = rec.shape
    dt = np.uint8 if api.depth == 8 else np.uint16
    s1, s2 = W + stride_pad[0], W + stride_pad[1]
    a = np.zeros((H, s1), dt); a[:, :W] = rec
    b = np.zeros((H, s2), dt); b[:, :W] = src
    d_a, d_b = api.to_device(a.reshape(-1)), api.to_device(b.reshape(-1))
    nrows = (H + ctu - 1) // ctu
    api.lib.x265hip_ssim_workspace.restype = C.c_size_t
    ws = t.zeros(max(1, api.lib.x265hip_ssim_workspace(W, H) // 4), dtype=t.float32, device="cuda")
    d_rs = t.full((nrows,), -1.0, dtype=t.float32, device="cuda"); d_rc = t.full((nrows,), -1, dtype=t.int32, device="cuda")
    d_fr = t.full((2,), -1.0, dtype=t.float64, device="cuda")
    P = lambda x: C.c_void_p(x.data_ptr())
    call = lambda: api.lib.x265hip_ssim_frame(api.stream(), P(d_a), C.c_ssize_t(s1), P(d_b), C.c_ssize_t(s2), W, H, ctu, P(ws), P(d_rs), P(d_rc), P(d_fr))
    api.h.check(call())
    t.cuda.synchronize()
    if time_it:
        e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call()
        e1.record(); t.cuda.synchronize()
        print("ssim_frame %d bit %dx%d: %.4f ms (3 launches)" % (api.depth, W, H, e0.elapsed_time(e1) / 50))
    fr = d_fr.cpu().numpy()
    return d_rs.cpu().numpy(), d_rc.cpu().numpy().view(np.uint32), float(fr[0]), int(fr[1])


def picture_pair(depth, W, H, seed, kind):
    rng = np.random.default_rng(seed)
    pm = (1 << depth) - 1
    if kind == "random":
        return rng.integers(0, pm + 1, (H, W)), rng.integers(0, pm + 1, (H, W))
    if kind == "extreme":                                       # largest sums: every pixel at the maximum against zero / against itself
        a = np.full((H, W), pm); b = a.copy(); b[:, W // 2:] = 0
        return a, b
    yy, xx = np.mgrid[0:H, 0:W]
    src = (np.sin(xx / 9.0) + np.cos(yy / 7.0) + 2) / 4 * pm * 0.8 + rng.normal(0, 3 << (depth - 8), (H, W))
    src = np.clip(np.rint(src), 0, pm).astype(np.int64)
    step = 12 << (depth - 8)
    return np.clip((src + step // 2) // step * step + rng.integers(-2, 3, (H, W)), 0, pm), src          # a coarsely quantised reconstruction


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("W,H,ctu,kind", [(1920, 1080, 64, "coded"), (136, 72, 64, "coded"), (200, 152, 32, "random"), (136, 104, 16, "coded"), (64, 64, 64, "extreme"),
                                          (10, 10, 16, "random"), (4096, 40, 32, "random"), (333, 259, 64, "coded")])
def test_ssim_frame_matches_oracle(depth, W, H, ctu, kind):
    from x265hip_pkg.frame import FrameApi
    api, ora = FrameApi(depth), Oracle(depth)
    rec, src = picture_pair(depth, W, H, W * 7 + H, kind)
    ers, erc, etot, ecnt = ssim_oracle(ora, rec, src, ctu)
    rs, rc, tot, cnt = hip_ssim(api, rec, src, ctu, stride_pad=(5, 32), time_it=(W == 1920))
    assert np.array_equal(rc, erc) and cnt == ecnt
    assert np.array_equal(rs.view(np.uint32), ers.view(np.uint32)), "CTU row sums differ (bitwise float compare): %s vs %s" % (rs, ers)
    assert tot == etot


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_ssim_frame_matches_the_encoders_reported_ssim(depth):
    """committed fixtures: pictures the reference encoder reconstructed, with the SSIM it reported"""
    from x265hip_pkg.frame import FrameApi
    api = FrameApi(depth)
    ctu, pics = golden_pictures(depth)
    for pic in pics:
        rs, rc, tot, cnt = hip_ssim(api, pic["rec"][0], pic["src"][0], ctu)
        assert tot / cnt == pic["ssim"]
