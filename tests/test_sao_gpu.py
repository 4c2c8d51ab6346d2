"""SAO statistics slots (saoCuStatsBO / E0..E3) through the table, host pointers, against the oracle (pinned to the reference's
primitives by test_sao_oracle_vs_ref.py): class sums, counts and the sign buffers left behind."""
import numpy as np
import pytest

from depths import DEPTHS

import x265hip
from backends import Oracle
from sao_util import cases, run_hip, run_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", DEPTHS)
def test_sao_stats_slots_match_oracle(depth):
    lib, ora = x265hip.HipLib(depth), Oracle(depth)
    for i, c in enumerate(cases(depth, 900 + depth, n=100)):
        a, b = run_hip(lib, c), run_oracle(ora, c)
        for x, y, what in zip(a, b, ("stats", "count", "upBuff1", "upBufft")):
            assert np.array_equal(x, y), "case %d type %d endX %d endY %d: %s" % (i, c[0], c[5], c[6], what)


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,ctu,nd,po", [((200, 136), 64, 0, 0), ((192, 128), 64, 0, 0), ((72, 40), 32, 0, 0), ((130, 70), 16, 0, 0), ((200, 136), 64, 1, 0),
                                            ((64, 64), 64, 0, 0), ((1920, 1080), 64, 0, 0), ((100, 68), 32, 0, 2), ((960, 540), 32, 1, 2), ((36, 20), 8, 0, 2),
                                            ((200, 136), 64, 2, 0), ((1920, 1080), 64, 2, 0), ((130, 70), 16, 2, 0), ((960, 540), 32, 2, 2), ((100, 68), 32, 2, 2), ((64, 64), 64, 2, 0)])
def test_sao_frame_stats_match_oracle(depth, size, ctu, nd, po):
    """x265hip_sao_stats_frame (all CTUs of a picture in one launch) against the oracle's calcSaoStatsCTU (pinned to the reference's SAO class)"""
    import ctypes as C
    from x265hip_pkg.frame import FrameApi
    from test_sao_oracle_vs_ref import sao_frame_oracle, sao_frame_oracle_pre, sao_frame_pair
    api = FrameApi(depth)
    t = api.torch
    W, H = size
    fenc, rec = sao_frame_pair(depth, W, H, 170 + depth + W)
    # nd = 2: SAO::calcSaoStatsCu_BeforeDblk's border statistics (rec = the picture before deblocking)
    exp = sao_frame_oracle_pre(Oracle(depth), fenc, rec, ctu, po) if nd == 2 else sao_frame_oracle(Oracle(depth), fenc, rec, ctu, nd, po)
    d_f, d_r = api.to_device(fenc.reshape(-1)), api.to_device(rec.reshape(-1))
    d_out = t.full((exp.size,), -7, dtype=t.int32, device="cuda")
    P = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    api.h.check(api.lib.x265hip_sao_stats_frame(api.stream(), P(d_f), P(d_r), C.c_ssize_t(W), W, H, ctu, nd, po, P(d_out)))
    t.cuda.synchronize()
    got = d_out.cpu().numpy().reshape(exp.shape)
    bad = np.argwhere(got != exp)
    assert bad.size == 0, "first mismatch (ctu, which, type, class) %s: hip %d oracle %d" % (bad[0], got[tuple(bad[0])], exp[tuple(bad[0])])


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,ctu,nd,po,rows", [((200, 200), 64, 0, 0, (2,)), ((192, 128), 32, 0, 0, (1, 3)), ((136, 120), 16, 1, 0, (2, 3, 6)), ((100, 132), 32, 1, 2, (1, 2, 4)),
                                                 ((1920, 1080), 64, 0, 0, (4, 8, 13)), ((960, 540), 32, 0, 2, (4, 8, 13))])
def test_sao_frame_stats_with_slices_match_oracle(depth, size, ctu, nd, po, rows):
    """x265hip_sao_stats_frame_slices: CTU rows that begin a slice (no row above them; the row before counts down to its bottom line) -- the oracle's form is pinned to the
    reference's SAO class on CTUs carrying the same slice flags (test_sao_oracle_vs_ref.py)"""
    import ctypes as C
    from x265hip_pkg.frame import FrameApi
    from test_sao_oracle_vs_ref import sao_frame_oracle, sao_frame_pair, slice_first_row
    api = FrameApi(depth)
    t = api.torch
    W, H = size
    fenc, rec = sao_frame_pair(depth, W, H, 270 + depth + W)
    exp = sao_frame_oracle(Oracle(depth), fenc, rec, ctu, nd, po, rows)
    assert not np.array_equal(exp, sao_frame_oracle(Oracle(depth), fenc, rec, ctu, nd, po))
    d_f, d_r, d_s = api.to_device(fenc.reshape(-1)), api.to_device(rec.reshape(-1)), api.to_device(slice_first_row(H, ctu, rows))
    d_out = t.full((exp.size,), -7, dtype=t.int32, device="cuda")
    P = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    api.h.check(api.lib.x265hip_sao_stats_frame_slices(api.stream(), P(d_f), P(d_r), C.c_ssize_t(W), W, H, ctu, nd, po, P(d_out), P(d_s)))
    t.cuda.synchronize()
    got = d_out.cpu().numpy().reshape(exp.shape)
    bad = np.argwhere(got != exp)
    assert bad.size == 0, "first mismatch (ctu, which, type, class) %s: hip %d oracle %d" % (bad[0], got[tuple(bad[0])], exp[tuple(bad[0])])


@pytest.mark.parametrize("depth", DEPTHS)
def test_plane_ssd_matches_oracle(depth):
    """x265hip_plane_ssd = Encoder::computeSSD (the PSNR numerator): exact 64-bit sums, incl. all-extreme planes"""
    import ctypes as C
    from x265hip_pkg.frame import FrameApi
    api, ora = FrameApi(depth), Oracle(depth)
    t = api.torch
    rng = np.random.default_rng(depth)
    pm = (1 << depth) - 1
    dt = np.uint8 if depth == 8 else np.uint16
    ora.lib.xo_plane_ssd.restype = C.c_uint64
    for (W, H, stride, kind) in ((1920, 1080, 1920, 0), (37, 19, 40, 0), (3840, 2160, 3904, 1), (64, 64, 64, 1), (1, 1, 4, 0)):
        a = rng.integers(0, pm + 1, stride * H).astype(dt) if kind == 0 else np.zeros(stride * H, dt)
        b = rng.integers(0, pm + 1, stride * H).astype(dt) if kind == 0 else np.full(stride * H, pm, dt)
        exp = int(ora.lib.xo_plane_ssd(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_ssize_t(stride), W, H))
        assert exp == int(((a.reshape(H, stride)[:, :W].astype(np.int64) - b.reshape(H, stride)[:, :W].astype(np.int64)) ** 2).sum())
        d_a, d_b = api.to_device(a), api.to_device(b)
        d_o = t.full((1,), -1, dtype=t.int64, device="cuda")
        api.h.check(api.lib.x265hip_plane_ssd(api.stream(), C.c_void_p(d_a.data_ptr()), C.c_void_p(d_b.data_ptr()), C.c_ssize_t(stride), W, H, C.c_void_p(d_o.data_ptr())))
        t.cuda.synchronize()
        assert int(d_o.cpu().numpy().view(np.uint64)[0]) == exp, (W, H)


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,ctu", [((200, 136), 64), ((72, 40), 32), ((130, 70), 16), ((64, 64), 64), ((1920, 1080), 64), ((3, 5), 16)])
def test_sao_apply_frame_matches_oracle(depth, size, ctu):
    """x265hip_sao_apply_frame against the oracle (pinned to the reference's in-place SAO::generateLumaOffsets sequence)"""
    import ctypes as C
    import time
    from x265hip_pkg.frame import FrameApi
    from test_sao_oracle_vs_ref import sao_apply_oracle, sao_frame_pair, sao_params
    api = FrameApi(depth)
    t = api.torch
    W, H = size
    _, rec = sao_frame_pair(depth, max(W, 8), max(H, 8), 55 + depth + W)
    rec = np.ascontiguousarray(rec[:H, :W])
    n = ((W + ctu - 1) // ctu) * ((H + ctu - 1) // ctu)
    prm = sao_params(np.random.default_rng(3 * W + depth), n, depth)
    exp = sao_apply_oracle(Oracle(depth), rec, ctu, prm)
    d_in, d_prm = api.to_device(rec.reshape(-1)), api.to_device(prm.reshape(-1))
    d_out = t.zeros_like(d_in)
    P = lambda x: C.c_void_p(x.data_ptr())  # noqa: E731
    api.h.check(api.lib.x265hip_sao_apply_frame(api.stream(), P(d_in), P(d_out), C.c_ssize_t(W), W, H, ctu, P(d_prm)))
    t.cuda.synchronize()
    got = d_out.cpu().numpy().view(rec.dtype).reshape(H, W)
    bad = np.argwhere(got != exp)
    assert bad.size == 0, "first differing pixel %s: hip %d oracle %d" % (bad[0], got[tuple(bad[0])], exp[tuple(bad[0])])
    if W >= 1920:
        e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            api.lib.x265hip_sao_apply_frame(api.stream(), P(d_in), P(d_out), C.c_ssize_t(W), W, H, ctu, P(d_prm))
        e1.record(); t.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("sao_apply %d bit %dx%d: %.4f ms, %.0f GB/s algorithmic (plane read + written)" % (depth, W, H, ms, 2 * W * H * rec.itemsize / ms / 1e6))


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,ctu", [((200, 136), 64), ((72, 40), 32), ((136, 72), 16), ((1920, 1080), 64)])
def test_sao_apply_chroma_planes_match_oracle(depth, size, ctu):
    """Cb / Cr of a 4:2:0 picture = x265hip_sao_apply_frame on the chroma plane with its dimensions and CTU size (oracle pinned to generateChromaOffsets)"""
    import ctypes as C
    from x265hip_pkg.frame import FrameApi
    from test_sao_oracle_vs_ref import sao_apply_oracle, sao_frame_pair, sao_params
    api, ora = FrameApi(depth), Oracle(depth)
    t = api.torch
    W, H = size[0] // 2, size[1] // 2
    n = ((size[0] + ctu - 1) // ctu) * ((size[1] + ctu - 1) // ctu)
    P = lambda x: C.c_void_p(x.data_ptr())
    for c in (1, 2):
        _, rec = sao_frame_pair(depth, W, H, 50 + c + depth + W)
        prm = sao_params(np.random.default_rng(W + depth + c), n, depth)
        exp = sao_apply_oracle(ora, rec, ctu // 2, prm)
        d_in = api.to_device(np.ascontiguousarray(rec).reshape(-1)); d_out = t.zeros_like(d_in); d_prm = api.to_device(prm.astype(np.int32).reshape(-1))
        api.h.check(api.lib.x265hip_sao_apply_frame(api.stream(), P(d_in), P(d_out), C.c_ssize_t(W), W, H, ctu // 2, P(d_prm)))
        got = d_out.cpu().numpy().reshape(H, W)
        assert np.array_equal(got, exp)
