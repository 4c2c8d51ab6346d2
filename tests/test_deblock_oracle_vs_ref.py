"""Deblocking: the per-unit restatement xo_deblock_frame against the reference's Deblock::deblockCTU run on its own CUData objects
(oracle/_ref/x265deblock_*, oracle/ref_deblock.cpp) -- random coded pictures: every CU size / partition shape / transform depth, intra / inter / skip,
P and B slices with repeated reference pictures, QP extremes, PPS offsets, lossless CUs, pictures that are no CTU multiple.  Bit-exact, all three planes."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
from oracle_py import Oracle  # noqa: E402
from deblock_util import coded_picture, dbk_bin, run_oracle, run_reference  # noqa: E402

CASES = [(8, 64, 64, 64, 1, False, False), (8, 136, 72, 64, 2, True, False), (8, 200, 152, 32, 3, False, True), (8, 96, 80, 16, 4, False, False),
         (10, 136, 104, 64, 5, False, False), (10, 64, 64, 32, 6, True, True), (8, 320, 192, 64, 7, False, False), (10, 264, 136, 16, 8, False, True)]


@pytest.mark.parametrize("depth,W,H,ctu,seed,slice_p,bypass", CASES)
def test_deblock_frame_matches_reference(depth, W, H, ctu, seed, slice_p, bypass):
    if not os.path.exists(dbk_bin(depth)):
        pytest.skip("oracle/_ref/x265deblock_%d not built (needs /root/reference at build time)" % depth)
    pic = coded_picture(depth, W, H, ctu, seed, slice_p, bypass)
    ref = run_reference(pic)
    got = run_oracle(Oracle(depth), pic)
    changed = sum(int((r != p).sum()) for r, p in zip(ref, pic["planes"]))
    assert changed > W * H // 50, "the picture hardly exercises the filter (%d samples changed)" % changed
    for c in range(3):
        bad = np.argwhere(got[c] != ref[c])
        assert bad.size == 0, "plane %d: %d samples differ, first at (y, x) %s: oracle %d reference %d" % (c, len(bad), bad[0], got[c][tuple(bad[0])], ref[c][tuple(bad[0])])


@pytest.mark.parametrize("depth,W,H,ctu,seed,slice_p,bypass,rows", [(8, 136, 200, 64, 11, True, False, (2,)), (8, 200, 152, 32, 12, False, False, (1, 3)), (10, 96, 112, 16, 13, False, True, (2, 3, 6)),
                                                                     (10, 320, 192, 64, 14, True, False, (1, 2))])
def test_deblock_frame_with_slices_matches_reference(depth, W, H, ctu, seed, slice_p, bypass, rows):
    """--slices: the CTUs of a slice's first row have no CTU above (CUData::initCTU, cudata.cpp:323): the row's top edge is left alone, luma and chroma"""
    if not os.path.exists(dbk_bin(depth)):
        pytest.skip("oracle/_ref/x265deblock_%d not built (needs /root/reference at build time)" % depth)
    pic = coded_picture(depth, W, H, ctu, seed, slice_p, bypass)
    one = run_reference(pic)
    pic["slice_rows"] = rows
    ref = run_reference(pic)
    assert any(not np.array_equal(a, b) for a, b in zip(one, ref)), "the slice boundaries changed nothing"
    got = run_oracle(Oracle(depth), pic)
    for c in range(3):
        bad = np.argwhere(got[c] != ref[c])
        assert bad.size == 0, "plane %d: %d samples differ, first at (y, x) %s" % (c, len(bad), bad[0])


@pytest.mark.parametrize("depth,W,H,ctu,seed,slice_p,bypass,csp", [(8, 136, 72, 64, 21, True, False, 2), (8, 200, 152, 32, 22, False, True, 2), (10, 96, 80, 16, 23, False, False, 2),
                                                                   (8, 136, 104, 64, 24, False, False, 3), (10, 64, 64, 32, 25, True, True, 3), (8, 264, 136, 16, 26, False, False, 3)])
def test_deblock_frame_in_other_chroma_formats_matches_reference(depth, W, H, ctu, seed, slice_p, bypass, csp):
    """4:2:2 (csp 2) and 4:4:4 (csp 3): the chroma edges lie on the 8-sample CHROMA grid in each direction, a chroma segment takes the strength of the first luma unit it covers,
    and the chroma QP is clipped instead of mapped (deblock.cpp:104-113, 417-497) -- the reference's Deblock on a PicYuv / CUData of that format"""
    if not os.path.exists(dbk_bin(depth)):
        pytest.skip("oracle/_ref/x265deblock_%d not built (needs /root/reference at build time)" % depth)
    pic = coded_picture(depth, W, H, ctu, seed, slice_p, bypass, csp=csp)
    assert pic["planes"][1].shape == (H >> (csp == 1), W >> (csp != 3))
    ref = run_reference(pic)
    got = run_oracle(Oracle(depth), pic)
    as420 = run_oracle(Oracle(depth), dict(pic, csp=1, planes=[pic["planes"][0], pic["planes"][1][: H // 2, : W // 2].copy(), pic["planes"][2][: H // 2, : W // 2].copy()]))
    assert not np.array_equal(as420[1], got[1][: H // 2, : W // 2]), "the format changed nothing in the chroma planes"
    for c in range(3):
        bad = np.argwhere(got[c] != ref[c])
        assert bad.size == 0, "plane %d: %d samples differ, first at (y, x) %s: oracle %d reference %d" % (c, len(bad), bad[0], got[c][tuple(bad[0])], ref[c][tuple(bad[0])])
