"""x265hip_deblock_frame through the C ABI against the oracle (pinned bit-exactly to the reference's Deblock::deblockCTU on its own CUData objects, see
test_deblock_oracle_vs_ref.py): the three planes and the boundary strength of every edge segment, random coded pictures up to 1080p."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
sys.path.insert(0, HERE)
import x265hip  # noqa: E402,F401  (makes the package importable as x265hip_pkg)
from oracle_py import Oracle  # noqa: E402
from deblock_util import slice_first_row, I8, U8, coded_picture, descriptor, run_oracle  # noqa: E402
from test_deblock_oracle_vs_ref import CASES  # noqa: E402

pytestmark = pytest.mark.gpu


def hip_deblock(api, pic, pad=(0, 0), time_it=False):
    t = api.torch
    W, H = pic["W"], pic["H"]
    sY, sC = W + pad[0], W // 2 + pad[1]
    host = []
    for c, p in enumerate(pic["planes"]):
        buf = np.zeros((p.shape[0], sC if c else sY), p.dtype); buf[:, :p.shape[1]] = p
        host.append(buf)
    dev = [api.to_device(b.reshape(-1)) for b in host]
    arrs = {k: api.to_device(np.ascontiguousarray(pic[k]).reshape(-1)) for k in U8 + I8 + ("mv0", "mv1")}
    d = descriptor(pic, lambda k: arrs[k].data_ptr())
    sfr = slice_first_row(pic)
    if sfr is not None:
        d_sfr = api.to_device(sfr)
        d.sliceFirstRow = d_sfr.data_ptr()
    bs = t.full((2 * (H // 4) * (W // 4),), 7, dtype=t.uint8, device="cuda")
    P = lambda x: C.c_void_p(x.data_ptr())
    call = lambda planes, b: api.lib.x265hip_deblock_frame(api.stream(), C.byref(d), P(planes[0]), C.c_ssize_t(sY), P(planes[1]), P(planes[2]), C.c_ssize_t(sC), b)
    api.h.check(call(dev, P(bs)))
    t.cuda.synchronize()
    out = [x.cpu().numpy().reshape(h.shape)[:, :p.shape[1]] for x, h, p in zip(dev, host, pic["planes"])]
    if time_it:
        scratch = [api.to_device(b.reshape(-1)) for b in host]
        e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        call(scratch, None); t.cuda.synchronize()
        e0.record()
        for _ in range(50):
            call(scratch, None)
        e1.record(); t.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        nbytes = sum(b.nbytes for b in pic["planes"])
        print("deblock_frame %d bit %dx%d: %.4f ms (2 launches), %.0f GB/s algorithmic (planes read + written once per pass)" % (api.depth, W, H, ms, 4 * nbytes / ms / 1e6))
    return out, bs.cpu().numpy().reshape(2, H // 4, W // 4)


@pytest.mark.parametrize("depth,W,H,ctu,seed,slice_p,bypass", CASES + [(8, 1920, 1080, 64, 11, False, False), (10, 1920, 1080, 32, 12, True, True)])
def test_deblock_frame_matches_oracle(depth, W, H, ctu, seed, slice_p, bypass):
    from x265hip_pkg.frame import FrameApi
    api, ora = FrameApi(depth), Oracle(depth)
    H -= H % 8
    pic = coded_picture(depth, W, H, ctu, seed, slice_p, bypass)
    exp, ebs = run_oracle(ora, pic, want_bs=True)
    got, bs = hip_deblock(api, pic, pad=(12, 6), time_it=(W == 1920))
    assert np.array_equal(bs[0][:, ::2], ebs[0][:, ::2]) and np.array_equal(bs[1][::2], ebs[1][::2]), "boundary strengths differ"
    assert not bs[0][:, 1::2].any() and not bs[1][1::2].any()
    for c in range(3):
        bad = np.argwhere(got[c] != exp[c])
        assert bad.size == 0, "plane %d: %d samples differ, first at (y, x) %s: hip %d oracle %d" % (c, len(bad), bad[0], got[c][tuple(bad[0])], exp[c][tuple(bad[0])])


@pytest.mark.parametrize("depth,W,H,ctu,seed,slice_p,bypass,rows", [(8, 136, 200, 64, 11, True, False, (2,)), (8, 200, 152, 32, 12, False, False, (1, 3)), (10, 96, 112, 16, 13, False, True, (2, 3, 6)),
                                                                     (10, 1920, 1080, 64, 14, True, False, (4, 8, 13))])
def test_deblock_frame_with_slices_matches_oracle(depth, W, H, ctu, seed, slice_p, bypass, rows):
    """--slices (x265hip_deblock_pic::sliceFirstRow): the top edge of a slice's first CTU row is left alone; the oracle's form is pinned to the reference's Deblock on CUData
    objects initialised with the same slice flags (test_deblock_oracle_vs_ref.py)"""
    from x265hip_pkg.frame import FrameApi
    api, ora = FrameApi(depth), Oracle(depth)
    H -= H % 8
    pic = coded_picture(depth, W, H, ctu, seed, slice_p, bypass)
    one = run_oracle(ora, pic)
    pic["slice_rows"] = rows
    exp, ebs = run_oracle(ora, pic, want_bs=True)
    assert any(not np.array_equal(a, b) for a, b in zip(one, exp))
    got, bs = hip_deblock(api, pic, pad=(4, 10))
    for r in rows:
        assert not ebs[1][r * ctu // 4].any()
    for c in range(3):
        bad = np.argwhere(got[c] != exp[c])
        assert bad.size == 0, "plane %d: %d samples differ, first at (y, x) %s" % (c, len(bad), bad[0])


def test_deblock_frame_refuses_bad_descriptions():
    from x265hip_pkg.frame import FrameApi
    api = FrameApi(8)
    pic = coded_picture(8, 64, 64, 64, 1)
    t = api.torch
    buf = t.zeros(64 * 64, dtype=t.uint8, device="cuda")
    d = descriptor(pic, lambda k: buf.data_ptr())
    P = C.c_void_p(buf.data_ptr())
    for field, value in (("width", 60), ("ctuSize", 48), ("log2CUSize", None), ("height", 0)):
        old = getattr(d, field); setattr(d, field, value)
        assert api.lib.x265hip_deblock_frame(api.stream(), C.byref(d), P, C.c_ssize_t(64), P, P, C.c_ssize_t(32), None) != 0
        setattr(d, field, old)
