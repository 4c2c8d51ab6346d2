"""The band form of the in-loop filter path (bands of CTU rows, FrameFilter::processRow's order, framefilter.cpp:576-676) against the whole-picture form, on the oracle:
oracle/x265_oracle.c's xo_deblock_rows / xo_sao_stats_rows over successive bands must give what xo_deblock_frame / xo_sao_stats_frame_slices give on the whole picture -- and the
whole-picture functions are pinned to the reference's Deblock / SAO classes (tests/test_deblock_oracle_vs_ref.py, tests/test_sao_oracle_vs_ref.py).  The statistics of a band are
taken BEFORE the rows below it are deblocked, as the reference takes them."""
import ctypes as C

import numpy as np
import pytest

from depths import GOLDEN_DEPTHS
from backends import Oracle
from deblock_util import U8, I8, coded_picture, descriptor, run_oracle, slice_first_row


def _bands(n_rows, cut):
    """cut: list of band heights repeated until the rows are used up"""
    out, r, i = [], 0, 0
    while r < n_rows:
        h = min(cut[i % len(cut)], n_rows - r)
        out.append((r, r + h)); r += h; i += 1
    return out


def _stats(ora, fenc, rec, W, H, ctu, non_deblocked, plane_offset, sfr, rows=None):
    nx, ny = (W + ctu - 1) // ctu, (H + ctu - 1) // ctu
    out = np.full(nx * ny * 320, -7, np.int32)
    P = lambda x: C.c_void_p(x.ctypes.data) if x is not None else None
    r0, r1 = rows if rows else (0, ny)
    ora.lib.xo_sao_stats_rows(P(fenc), P(rec), C.c_ssize_t(W), W, H, ctu, non_deblocked, plane_offset, P(out), P(sfr), r0, r1)
    return out.reshape(nx * ny, 320)


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
@pytest.mark.parametrize("W,H,ctu,cut,slices", [(192, 256, 64, [1], ()), (256, 320, 64, [2, 1], ()), (200, 168, 32, [3, 1, 2], ()), (128, 136, 16, [4, 1], ()), (192, 328, 64, [2], (2, 4)),
                                                (320, 192, 64, [1, 2], ())])
def test_bands_of_ctu_rows_give_the_whole_pictures_deblocking_and_statistics(depth, W, H, ctu, cut, slices):
    ora = Oracle(depth)
    ora.lib.xo_deblock_rows.restype = None
    ora.lib.xo_sao_stats_rows.restype = None
    pic = coded_picture(depth, W, H, ctu, seed=1000 + W + H + ctu + depth, slice_p=bool((W + H) & 8), bypass=(ctu == 32))
    if slices:
        pic["slice_rows"] = slices
    rng = np.random.default_rng(W * 7 + H)
    dt = np.uint8 if depth == 8 else np.uint16
    fenc = [np.clip(p.astype(np.int64) + rng.integers(-6, 7, p.shape), 0, (1 << depth) - 1).astype(dt) for p in pic["planes"]]
    sfr = slice_first_row(pic)
    # whole picture: deblock everything, then every CTU's statistics
    whole = run_oracle(ora, pic)
    ny = (H + ctu - 1) // ctu
    want = [_stats(ora, fenc[p], whole[p], W >> (p > 0), H >> (p > 0), ctu >> (p > 0), 0, 2 if p else 0, sfr) for p in range(3)]
    # bands: deblock a band (its top edge changes the row above), take ITS statistics with the rows below still as reconstructed
    planes = [np.ascontiguousarray(p.copy()) for p in pic["planes"]]
    keep = {k: np.ascontiguousarray(pic[k]) for k in U8 + I8 + ("mv0", "mv1")}
    d = descriptor(pic, lambda k: keep[k].ctypes.data)
    if sfr is not None:
        d.sliceFirstRow = sfr.ctypes.data
    P = lambda x: C.c_void_p(x.ctypes.data)
    got = [np.full_like(w, -7) for w in want]
    nx = (W + ctu - 1) // ctu
    for r0, r1 in _bands(ny, cut):
        ora.lib.xo_deblock_rows(C.byref(d), P(planes[0]), C.c_ssize_t(W), P(planes[1]), P(planes[2]), C.c_ssize_t(W // 2), None, r0, r1)
        for p in range(3):
            s = _stats(ora, fenc[p], planes[p], W >> (p > 0), H >> (p > 0), ctu >> (p > 0), 0, 2 if p else 0, sfr, rows=(r0, r1))
            assert (s[:r0 * nx] == -7).all() and (s[r1 * nx:] == -7).all(), "statistics outside the band were written"
            got[p][r0 * nx:r1 * nx] = s[r0 * nx:r1 * nx]
    for p in range(3):
        assert np.array_equal(planes[p], whole[p]), "plane %d: bands != whole picture" % p
        assert np.array_equal(got[p], want[p]), "plane %d: statistics of the bands != statistics of the whole deblocked picture" % p


def test_a_band_does_not_touch_rows_outside_its_reach():
    """a band writes its own rows and the last 3 luma (1 chroma) lines of the row above -- what x265hip_ff_picture moves for a band"""
    ora = Oracle(8)
    ora.lib.xo_deblock_rows.restype = None
    pic = coded_picture(8, 256, 256, 64, seed=5)
    planes = [np.ascontiguousarray(p.copy()) for p in pic["planes"]]
    keep = {k: np.ascontiguousarray(pic[k]) for k in U8 + I8 + ("mv0", "mv1")}
    d = descriptor(pic, lambda k: keep[k].ctypes.data)
    P = lambda x: C.c_void_p(x.ctypes.data)
    ora.lib.xo_deblock_rows(C.byref(d), P(planes[0]), C.c_ssize_t(256), P(planes[1]), P(planes[2]), C.c_ssize_t(128), None, 1, 3)
    for p, (lines_above, ctu) in enumerate(((3, 64), (1, 32), (1, 32))):
        changed = np.nonzero((planes[p] != pic["planes"][p]).any(axis=1))[0]
        assert changed.size and changed.min() >= ctu - lines_above and changed.max() < 3 * ctu
