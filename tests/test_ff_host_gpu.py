"""x265hip_ff_picture (include/x265hip_ctx.h: the host-level in-loop filter producer integration/filter_adapter.cpp binds) against the oracle: a random coded picture goes in as
HOST arrays with padded row pitches, the deblocked planes must equal xo_deblock_frame's and the SAO statistics of every CTU of the three planes xo_sao_stats_frame's on the
deblocked picture (both pinned to the reference's Deblock / SAO classes: test_deblock_oracle_vs_ref.py, test_sao_oracle_vs_ref.py)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
sys.path.insert(0, HERE)
import x265hip  # noqa: E402
from oracle_py import Oracle  # noqa: E402
from deblock_util import I8, U8, DeblockPic, coded_picture, descriptor, run_oracle, slice_first_row  # noqa: E402

pytestmark = pytest.mark.gpu


class FfDesc(C.Structure):
    _fields_ = [("pic", DeblockPic), ("recon", C.c_void_p * 3), ("fenc", C.c_void_p * 3), ("deblock", C.c_int), ("saoStats", C.c_int), ("saoNonDeblocked", C.c_int),
                ("stats", C.c_void_p * 3), ("ctuRowFirst", C.c_int), ("ctuRowCount", C.c_int)]


@pytest.mark.parametrize("depth,W,H,ctu,slice_p,bypass,deblock,sao,nd", [(8, 256, 192, 64, False, False, 1, 3, 0), (10, 200, 120, 64, True, True, 1, 3, 0), (8, 192, 128, 32, False, False, 1, 1, 1),
                                                                       (8, 128, 128, 16, True, False, 0, 3, 0), (10, 1920, 1080, 64, False, False, 1, 3, 0), (8, 256, 192, 64, False, False, 1, 0, 0)])
def test_ff_picture_matches_oracle(depth, W, H, ctu, slice_p, bypass, deblock, sao, nd):
    H -= H % 8
    ora = Oracle(depth)
    lib = x265hip.HipLib(depth, fill_table=False).lib
    lib.x265hip_last_error.restype = C.c_char_p
    pic = coded_picture(depth, W, H, ctu, 77 + W + depth, slice_p, bypass)
    rng = np.random.default_rng(5 + W)
    dt = pic["planes"][0].dtype
    sY, sC = W + 40, W // 2 + 24                                                        # padded pitches, as PicYuv has
    src = [np.clip(p.astype(np.int32) + rng.integers(-6, 7, p.shape), 0, (1 << depth) - 1).astype(dt) for p in pic["planes"]]      # a "source" the picture is a coded version of
    def padded(p, s):
        b = np.full((p.shape[0], s), 3, dt); b[:, :p.shape[1]] = p
        return b
    recon = [padded(p, sC if c else sY) for c, p in enumerate(pic["planes"])]
    fenc = [padded(p, sC if c else sY) for c, p in enumerate(src)]
    before = [r.copy() for r in recon]
    ctx, ff = C.c_void_p(), C.c_void_p()
    assert lib.x265hip_ctx_create(0, C.byref(ctx)) == 0, lib.x265hip_last_error()
    assert lib.x265hip_ff_create(ctx, W, H, ctu, C.c_ssize_t(sY), C.c_ssize_t(sC), C.byref(ff)) == 0, lib.x265hip_last_error()
    keep = {k: np.ascontiguousarray(pic[k]) for k in U8 + I8 + ("mv0", "mv1")}
    nctu = ((W + ctu - 1) // ctu) * ((H + ctu - 1) // ctu)
    stats = [np.full(nctu * 320, -9, np.int32) for _ in range(3)]
    d = FfDesc()
    d.pic = descriptor(pic, lambda k: keep[k].ctypes.data)
    d.recon[:] = [r.ctypes.data for r in recon]; d.fenc[:] = [f.ctypes.data for f in fenc]
    d.deblock, d.saoStats, d.saoNonDeblocked = deblock, sao, nd
    d.stats[:] = [s.ctypes.data for s in stats]
    for rep in range(2):                                                                # twice: the producer keeps no state between pictures
        for r, b in zip(recon, before):
            r[:] = b
        assert lib.x265hip_ff_picture(ff, C.byref(d)) == 0, lib.x265hip_last_error()
        exp = run_oracle(ora, pic) if deblock else [p.copy() for p in pic["planes"]]
        for c in range(3):
            w = W if c == 0 else W // 2
            assert np.array_equal(recon[c][:, :w], exp[c]), "plane %d after deblocking" % c
            assert np.array_equal(recon[c][:, w:], before[c][:, w:]), "the padding of plane %d was touched" % c
            if (c == 0 and sao & 1) or (c > 0 and sao & 2):
                h, cs = (H, ctu) if c == 0 else (H // 2, ctu // 2)
                want = np.zeros(nctu * 320, np.int32)
                P = lambda a: C.c_void_p(a.ctypes.data)
                f_c, r_c = np.ascontiguousarray(src[c]), np.ascontiguousarray(exp[c])
                ora.lib.xo_sao_stats_frame(P(f_c), P(r_c), C.c_ssize_t(w), w, h, cs, nd, 0 if c == 0 else 2, P(want))
                assert np.array_equal(stats[c], want), "SAO statistics of plane %d" % c
            else:
                assert (stats[c] == -9).all()
    lib.x265hip_ff_destroy(ff); lib.x265hip_ctx_destroy(ctx)


@pytest.mark.parametrize("depth,W,H,ctu,cut,sao,nd,csp,slices", [(8, 256, 320, 64, [1], 3, 0, 1, ()), (10, 320, 328, 64, [2, 1], 3, 0, 1, ()), (8, 200, 168, 32, [3, 1], 3, 1, 1, ()), (8, 128, 136, 16, [4, 1, 2], 1, 0, 1, ()),
                                                          (10, 1920, 1080, 64, [4], 3, 0, 1, ()), (8, 256, 256, 64, [2], 0, 0, 1, ()),
                                                          (8, 256, 320, 64, [2, 1], 3, 0, 2, ()), (10, 200, 168, 32, [5], 3, 0, 2, ()), (8, 256, 192, 64, [1, 2], 3, 1, 3, ()), (10, 136, 128, 16, [8], 3, 0, 3, ()),
                                                          (8, 256, 320, 64, [2, 1, 2], 3, 0, 1, (2,)), (10, 192, 328, 32, [3, 2, 2, 4], 3, 0, 2, (3, 7))])       # --slices: a band stays inside its slice

def test_ff_picture_in_bands_of_ctu_rows(depth, W, H, ctu, cut, sao, nd, csp, slices):
    """desc.ctuRowFirst / ctuRowCount (FrameFilter::processRow's order under frame threads): TWO pictures go through one producer band by band, their bands interleaved -- each
    band reads the CU arrays of its rows and the row above, the planes from 8 lines above it, and writes those lines and its rows' statistics back.  After the last band each
    picture equals the whole-picture result of the oracle; the statistics of a band are taken before the rows below it are deblocked (xo_sao_stats_rows on the oracle's band state).
    csp 2 / 3: 4:2:2 / 4:4:4 pictures (desc.pic.chromaFormat: chroma planes half as wide and as high as luma / full size, chroma CTUs ctu / 2 x ctu / ctu x ctu); [n] as the cut =
    the whole picture in one call."""
    H -= H % 8
    ora = Oracle(depth)
    ora.lib.xo_deblock_rows.restype = None
    ora.lib.xo_sao_stats_rows_wh.restype = None
    hs, vs = (0 if csp == 3 else 1), (1 if csp == 1 else 0)
    lib = x265hip.HipLib(depth, fill_table=False).lib
    lib.x265hip_last_error.restype = C.c_char_p
    sY, sC = W + 40, (W >> hs) + 24
    nx, ny = (W + ctu - 1) // ctu, (H + ctu - 1) // ctu
    nctu = nx * ny
    ctx, ff = C.c_void_p(), C.c_void_p()
    assert lib.x265hip_ctx_create(0, C.byref(ctx)) == 0, lib.x265hip_last_error()
    assert lib.x265hip_ff_create(ctx, W, H, ctu, C.c_ssize_t(sY), C.c_ssize_t(sC), C.byref(ff)) == 0, lib.x265hip_last_error()
    P = lambda a: C.c_void_p(a.ctypes.data)
    pics = []
    for k in range(2):
        pic = coded_picture(depth, W, H, ctu, 900 + 13 * k + W + depth, bool(k), False, csp=csp)
        if slices:
            pic["slice_rows"] = slices
        sfr = slice_first_row(pic)
        rng = np.random.default_rng(50 + k + W)
        dt = pic["planes"][0].dtype
        src = [np.clip(p.astype(np.int32) + rng.integers(-6, 7, p.shape), 0, (1 << depth) - 1).astype(dt) for p in pic["planes"]]
        def padded(p, s_):
            b = np.full((p.shape[0], s_), 3, dt); b[:, :p.shape[1]] = p
            return b
        recon = [padded(p, sC if c else sY) for c, p in enumerate(pic["planes"])]
        fenc = [padded(p, sC if c else sY) for c, p in enumerate(src)]
        keep = {n: np.ascontiguousarray(pic[n]) for n in U8 + I8 + ("mv0", "mv1")}
        stats = [np.full(nctu * 320, -9, np.int32) for _ in range(3)]
        d = FfDesc()
        d.pic = descriptor(pic, lambda n, keep=keep: keep[n].ctypes.data)
        if sfr is not None:
            d.pic.sliceFirstRow = sfr.ctypes.data
        d.recon[:] = [r.ctypes.data for r in recon]; d.fenc[:] = [f.ctypes.data for f in fenc]
        d.deblock, d.saoStats, d.saoNonDeblocked = 1, sao, nd
        d.stats[:] = [s_.ctypes.data for s_ in stats]
        # the oracle's band state of the same picture (contiguous planes)
        oplanes = [np.ascontiguousarray(p.copy()) for p in pic["planes"]]
        od = descriptor(pic, lambda n, keep=keep: keep[n].ctypes.data)
        if sfr is not None:
            od.sliceFirstRow = sfr.ctypes.data
        pics.append(dict(pic=pic, sfr=sfr, src=src, recon=recon, fenc=fenc, keep=keep, stats=stats, d=d, oplanes=oplanes, od=od, want=[np.full(nctu * 320, -9, np.int32) for _ in range(3)]))
    bands, r, i = [], 0, 0
    while r < ny:
        h = min(cut[i % len(cut)], ny - r); bands.append((r, r + h)); r += h; i += 1
    # picture 0 runs one band ahead of picture 1
    order = []
    for b in range(len(bands) + 1):
        if b < len(bands): order.append((0, bands[b]))
        if b > 0: order.append((1, bands[b - 1]))
    for k, (r0, r1) in order:
        S = pics[k]
        S["d"].ctuRowFirst, S["d"].ctuRowCount = (r0, r1 - r0) if (r0 > 0 or r1 < ny) else (0, 0)
        assert lib.x265hip_ff_picture(ff, C.byref(S["d"])) == 0, lib.x265hip_last_error()
        ora.lib.xo_deblock_rows(C.byref(S["od"]), P(S["oplanes"][0]), C.c_ssize_t(W), P(S["oplanes"][1]), P(S["oplanes"][2]), C.c_ssize_t(W >> hs), None, r0, r1)
        for c in range(3):
            w = W if c == 0 else W >> hs
            assert np.array_equal(S["recon"][c][:, :w], S["oplanes"][c]), "picture %d plane %d after the band of rows %d..%d" % (k, c, r0, r1 - 1)
            assert (S["recon"][c][:, w:] == 3).all(), "the padding of plane %d was touched" % c
            if (c == 0 and sao & 1) or (c > 0 and sao & 2):
                h, cw_, ch_ = (H, ctu, ctu) if c == 0 else (H >> vs, ctu >> hs, ctu >> vs)
                ora.lib.xo_sao_stats_rows_wh(P(np.ascontiguousarray(S["src"][c])), P(S["oplanes"][c]), C.c_ssize_t(w), w, h, cw_, ch_, nd, 0 if c == 0 else 2, P(S["want"][c]), P(S["sfr"]) if S["sfr"] is not None else None, r0, r1)
            assert np.array_equal(S["stats"][c], S["want"][c]), "picture %d: statistics of plane %d after rows %d..%d (entries outside the bands so far must be untouched)" % (k, c, r0, r1 - 1)
    for S in pics:
        whole = run_oracle(ora, S["pic"])
        for c in range(3):
            assert np.array_equal(S["oplanes"][c], whole[c])
    # argument errors: a band beyond the picture, a band without a count
    bad = pics[0]["d"]
    bad.ctuRowFirst, bad.ctuRowCount = ny - 1, 2
    assert lib.x265hip_ff_picture(ff, C.byref(bad)) == -3 and b"CTU rows" in lib.x265hip_last_error()
    bad.ctuRowFirst, bad.ctuRowCount = 1, 0
    assert lib.x265hip_ff_picture(ff, C.byref(bad)) == -3
    lib.x265hip_ff_destroy(ff); lib.x265hip_ctx_destroy(ctx)
