"""SAO statistics: the oracle restatement against the reference's own primitives (oracle/_ref, op sao_stats)."""
import os

import numpy as np
import pytest

from depths import DEPTHS

from backends import Oracle, Ref, ref_available
from sao_util import cases, run_oracle, run_ref


@pytest.mark.parametrize("depth", DEPTHS)
def test_sao_stats_match_reference(depth):
    if not ref_available(depth):
        pytest.skip("no reference binary")
    ref, ora = Ref(depth), Oracle(depth)
    try:
        for i, c in enumerate(cases(depth, 600 + depth, n=120)):
            a, b = run_ref(ref, c), run_oracle(ora, c)
            for x, y, what in zip(a, b, ("stats", "count", "upBuff1", "upBufft")):
                assert np.array_equal(x, y), "case %d type %d endX %d endY %d: %s" % (i, c[0], c[5], c[6], what)
    finally:
        ref.close()


def sao_frame_reference(depth, fenc, rec, ctu, non_deblock=0, chroma=None, slice_rows=(), csp=1):
    """chroma = [(fencCb, recCb), (fencCr, recCr)] of a 4:2:0 picture -> array [planes, ctus, 2, 5, 32]; luma only -> [ctus, 2, 5, 32]"""
    import os, subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    H, W = fenc.shape
    with tempfile.TemporaryDirectory() as td:
        inp, out = os.path.join(td, "in.raw"), os.path.join(td, "out.bin")
        parts = [fenc.reshape(-1), rec.reshape(-1)] + ([a.reshape(-1) for pr in chroma for a in pr] if chroma else [])
        np.concatenate(parts).tofile(inp)
        r = subprocess.run([os.path.join(root, "oracle", "_ref", "x265sao_%d" % depth), str(W), str(H), str(ctu), inp, out, str(non_deblock), "3" if chroma else "1"],
                           capture_output=True, text=True, env=dict(os.environ, X265REF_SLICE_ROWS=",".join(str(r) for r in slice_rows), X265REF_CSP=str(csp)))
        assert r.returncode == 0, r.stderr[-1000:]
        o = np.fromfile(out, np.int32)
        return o.reshape(3, -1, 2, 5, 32) if chroma else o.reshape(-1, 2, 5, 32)


def slice_first_row(H, ctu, rows):
    n = (H + ctu - 1) // ctu
    a = np.zeros(n + 1, np.uint8)
    for r in rows:
        if 0 < r < n:
            a[r] = 1
    return a


def sao_frame_oracle(ora, fenc, rec, ctu, non_deblock=0, plane_offset=0, slice_rows=()):
    import ctypes as C
    H, W = fenc.shape
    n = ((W + ctu - 1) // ctu) * ((H + ctu - 1) // ctu)
    out = np.zeros((n, 2, 5, 32), np.int32)
    P = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    f, r = np.ascontiguousarray(fenc), np.ascontiguousarray(rec)
    if slice_rows:
        sfr = slice_first_row(H, ctu, slice_rows)
        ora.lib.xo_sao_stats_frame_slices(P(f), P(r), C.c_ssize_t(W), W, H, ctu, non_deblock, plane_offset, P(out), P(sfr))
    else:
        ora.lib.xo_sao_stats_frame(P(f), P(r), C.c_ssize_t(W), W, H, ctu, non_deblock, plane_offset, P(out))
    return out


def sao_frame_pair(depth, W, H, seed):
    rng = np.random.default_rng(seed)
    pm = (1 << depth) - 1
    dt = np.uint8 if depth == 8 else np.uint16
    base = rng.integers(0, pm + 1, (H // 4 + 2, W // 4 + 2))
    fenc = np.kron(base, np.ones((4, 4), np.int64))[:H, :W] + rng.integers(-3, 4, (H, W))
    rec = fenc + rng.integers(-5, 6, (H, W)) * (rng.random((H, W)) < 0.7)
    return np.clip(fenc, 0, pm).astype(dt), np.clip(rec, 0, pm).astype(dt)


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,ctu,nd", [((200, 136), 64, 0), ((192, 128), 64, 0), ((72, 40), 32, 0), ((130, 70), 16, 0), ((200, 136), 64, 1), ((64, 64), 64, 0)])
def test_sao_frame_stats_match_reference(depth, size, ctu, nd):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "x265sao_%d" % depth)):
        pytest.skip("no reference SAO binary")
    fenc, rec = sao_frame_pair(depth, size[0], size[1], 70 + depth + size[0])
    a = sao_frame_reference(depth, fenc, rec, ctu, nd)
    b = sao_frame_oracle(Oracle(depth), fenc, rec, ctu, nd)
    assert a.shape == b.shape
    for addr in range(a.shape[0]):
        for t in range(5):
            assert np.array_equal(a[addr, :, t], b[addr, :, t]), "CTU %d type %d" % (addr, t)


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,ctu,nd", [((200, 136), 64, 0), ((192, 128), 64, 1), ((72, 40), 32, 0), ((136, 72), 16, 0)])
def test_sao_frame_stats_chroma_match_reference(depth, size, ctu, nd):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "x265sao_%d" % depth)):
        pytest.skip("no reference SAO binary")
    W, H = size
    y = sao_frame_pair(depth, W, H, 7 + depth + W)
    cb, cr = sao_frame_pair(depth, W // 2, H // 2, 8 + depth + W), sao_frame_pair(depth, W // 2, H // 2, 9 + depth + W)
    a = sao_frame_reference(depth, y[0], y[1], ctu, nd, chroma=[cb, cr])
    ora = Oracle(depth)
    exp = [sao_frame_oracle(ora, y[0], y[1], ctu, nd, 0), sao_frame_oracle(ora, cb[0], cb[1], ctu // 2, nd, 2), sao_frame_oracle(ora, cr[0], cr[1], ctu // 2, nd, 2)]
    for plane in range(3):
        assert np.array_equal(a[plane], exp[plane]), "plane %d" % plane


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,ctu,nd,rows", [((200, 200), 64, 0, (2,)), ((192, 128), 32, 0, (1, 3)), ((136, 120), 16, 1, (2, 3, 6)), ((200, 264), 64, 1, (1, 2, 4))])
def test_sao_frame_stats_with_slices_match_reference(depth, size, ctu, nd, rows):
    """--slices: no row above the first CTU row of a slice, the last one counts down to its bottom line (m_bFirstRowInSlice / m_bLastRowInSlice, sao.cpp:744-746, 763-766);
    luma and the two chroma planes"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "x265sao_%d" % depth)):
        pytest.skip("no reference SAO binary")
    W, H = size
    y = sao_frame_pair(depth, W, H, 17 + depth + W)
    cb, cr = sao_frame_pair(depth, W // 2, H // 2, 18 + depth + W), sao_frame_pair(depth, W // 2, H // 2, 19 + depth + W)
    one = sao_frame_reference(depth, y[0], y[1], ctu, nd, chroma=[cb, cr])
    a = sao_frame_reference(depth, y[0], y[1], ctu, nd, chroma=[cb, cr], slice_rows=rows)
    assert not np.array_equal(one, a), "the slice boundaries changed nothing"
    ora = Oracle(depth)
    exp = [sao_frame_oracle(ora, y[0], y[1], ctu, nd, 0, rows), sao_frame_oracle(ora, cb[0], cb[1], ctu // 2, nd, 2, rows), sao_frame_oracle(ora, cr[0], cr[1], ctu // 2, nd, 2, rows)]
    for plane in range(3):
        assert np.array_equal(a[plane], exp[plane]), "plane %d" % plane


def sao_params(rng, n_ctu, depth):
    """per CTU { typeIdx (-1 off, 0..3 EO, 4 BO), bandPos, offset[4] }: EO offsets with the standard's signs, BO offsets any sign"""
    p = np.zeros((n_ctu, 6), np.int32)
    for a in range(n_ctu):
        t = int(rng.integers(-1, 5))
        p[a, 0] = t
        if t == 4:
            p[a, 1] = rng.integers(0, 32); p[a, 2:] = rng.integers(-7, 8, 4)
        elif t >= 0:
            p[a, 2:] = (rng.integers(0, 8), rng.integers(0, 8), -rng.integers(0, 8), -rng.integers(0, 8))
    return p


def sao_apply_reference(depth, fenc, rec, ctu, params):
    import os, subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    H, W = rec.shape
    with tempfile.TemporaryDirectory() as td:
        inp, out, prm = os.path.join(td, "in.raw"), os.path.join(td, "out.bin"), os.path.join(td, "p.bin")
        np.concatenate([fenc.reshape(-1), rec.reshape(-1)]).tofile(inp); params.astype(np.int32).tofile(prm)
        r = subprocess.run([os.path.join(root, "oracle", "_ref", "x265sao_%d" % depth), str(W), str(H), str(ctu), inp, out, "0", "1", prm], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1000:]
        return np.fromfile(out, np.int32).reshape(H, W)


def sao_apply_oracle(ora, rec, ctu, params):
    import ctypes as C
    H, W = rec.shape
    src = np.ascontiguousarray(rec); dst = np.zeros_like(src); prm = np.ascontiguousarray(params.astype(np.int32))
    P = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    ora.lib.xo_sao_apply_frame(P(src), P(dst), C.c_ssize_t(W), W, H, ctu, P(prm))
    return dst


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,ctu", [((200, 136), 64), ((192, 128), 64), ((72, 40), 32), ((130, 70), 16), ((64, 64), 64), ((256, 64), 32)])
def test_sao_apply_frame_matches_reference(depth, size, ctu):
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "x265sao_%d" % depth)):
        pytest.skip("no reference SAO binary")
    W, H = size
    fenc, rec = sao_frame_pair(depth, W, H, 31 + depth + W)
    n = ((W + ctu - 1) // ctu) * ((H + ctu - 1) // ctu)
    prm = sao_params(np.random.default_rng(W + depth), n, depth)
    a = sao_apply_reference(depth, fenc, rec, ctu, prm)
    b = sao_apply_oracle(Oracle(depth), rec, ctu, prm)
    bad = np.argwhere(a != b.astype(np.int32))
    assert bad.size == 0, "first differing pixel (y, x) %s: reference %d oracle %d, CTU params %s" % (bad[0], a[tuple(bad[0])], b[tuple(bad[0])], prm[(bad[0][0] // ctu) * ((W + ctu - 1) // ctu) + bad[0][1] // ctu])
    assert (a != rec.astype(np.int32)).sum() > 0


def sao_apply_reference_420(depth, planes, ctu, params3):
    """planes: [(fenc, rec)] * 3 (Y, Cb, Cr); params3: [3, nCtu, 6] -> the three planes after SAO::generateLumaOffsets / generateChromaOffsets"""
    import os, subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    H, W = planes[0][1].shape
    with tempfile.TemporaryDirectory() as td:
        inp, out, prm = os.path.join(td, "in.raw"), os.path.join(td, "out.bin"), os.path.join(td, "p.bin")
        np.concatenate([a.reshape(-1) for pair in planes for a in pair]).tofile(inp); params3.astype(np.int32).tofile(prm)
        r = subprocess.run([os.path.join(root, "oracle", "_ref", "x265sao_%d" % depth), str(W), str(H), str(ctu), inp, out, "0", "3", prm], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1000:]
        d = np.fromfile(out, np.int32)
    return [d[:W * H].reshape(H, W), d[W * H:W * H + W * H // 4].reshape(H // 2, W // 2), d[W * H + W * H // 4:].reshape(H // 2, W // 2)]


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,ctu", [((200, 136), 64), ((192, 128), 64), ((72, 40), 32), ((136, 72), 16)])
def test_sao_apply_chroma_matches_reference(depth, size, ctu):
    """Cb / Cr through SAO::generateChromaOffsets (sao.cpp:626-730) = the plane-level restatement called with the chroma plane's dimensions and CTU size"""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "x265sao_%d" % depth)):
        pytest.skip("no reference SAO binary")
    W, H = size
    planes = [sao_frame_pair(depth, W, H, 41 + depth + W), sao_frame_pair(depth, W // 2, H // 2, 42 + depth + W), sao_frame_pair(depth, W // 2, H // 2, 43 + depth + W)]
    n = ((W + ctu - 1) // ctu) * ((H + ctu - 1) // ctu)
    rng = np.random.default_rng(W * 3 + depth)
    prm = np.stack([sao_params(rng, n, depth) for _ in range(3)])
    # Cb and Cr share one type per CTU (sao_type_idx_chroma; generateChromaOffsets applies Cr with Cb's type, sao.cpp:721): own band position and offsets only
    prm[2, :, 0] = prm[1, :, 0]
    for a in range(n):
        t = prm[1, a, 0]
        if t == 4:
            prm[2, a, 1] = rng.integers(0, 32); prm[2, a, 2:] = rng.integers(-7, 8, 4)
        elif t >= 0:
            prm[2, a, 2:] = (rng.integers(0, 8), rng.integers(0, 8), -rng.integers(0, 8), -rng.integers(0, 8))
    ref = sao_apply_reference_420(depth, planes, ctu, prm)
    ora = Oracle(depth)
    for c in range(3):
        got = sao_apply_oracle(ora, planes[c][1], ctu if c == 0 else ctu // 2, prm[c])
        bad = np.argwhere(ref[c] != got.astype(np.int32))
        assert bad.size == 0, "plane %d first differing pixel (y, x) %s: reference %d oracle %d" % (c, bad[0], ref[c][tuple(bad[0])], got[tuple(bad[0])])
        assert (ref[c] != planes[c][1].astype(np.int32)).sum() > 0


def sao_frame_oracle_pre(ora, fenc, rec, ctu, plane_offset=0):
    import ctypes as C
    H, W = fenc.shape
    n = ((W + ctu - 1) // ctu) * ((H + ctu - 1) // ctu)
    out = np.zeros((n, 2, 5, 32), np.int32)
    P = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    f, r = np.ascontiguousarray(fenc), np.ascontiguousarray(rec)
    ora.lib.xo_sao_stats_frame_predeblock(P(f), P(r), C.c_ssize_t(W), W, H, ctu, plane_offset, P(out))
    return out


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,ctu", [((200, 136), 64), ((192, 128), 64), ((72, 40), 32), ((136, 72), 16), ((64, 64), 64)])
def test_sao_predeblock_stats_match_reference(depth, size, ctu):
    """SAO::calcSaoStatsCu_BeforeDblk (sao.cpp:908-1207): the border statistics on the not yet deblocked picture, luma and 4:2:0 chroma, against the
    reference's SAO class (oracle/ref_sao.cpp, mode 2)"""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "x265sao_%d" % depth)):
        pytest.skip("no reference SAO binary")
    W, H = size
    y = sao_frame_pair(depth, W, H, 17 + depth + W)
    cb, cr = sao_frame_pair(depth, W // 2, H // 2, 18 + depth + W), sao_frame_pair(depth, W // 2, H // 2, 19 + depth + W)
    a = sao_frame_reference(depth, y[0], y[1], ctu, 2, chroma=[cb, cr])
    ora = Oracle(depth)
    exp = [sao_frame_oracle_pre(ora, y[0], y[1], ctu, 0), sao_frame_oracle_pre(ora, cb[0], cb[1], ctu // 2, 2), sao_frame_oracle_pre(ora, cr[0], cr[1], ctu // 2, 2)]
    for plane in range(3):
        for addr in range(a.shape[1]):
            for t in range(5):
                assert np.array_equal(a[plane, addr, :, t], exp[plane][addr, :, t]), "plane %d CTU %d type %d" % (plane, addr, t)
    assert a[0, :, 1].sum() > 0 or (W <= ctu and H <= ctu)            # a picture of one CTU has no border left out


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,ctu,nd,csp", [((200, 136), 64, 0, 2), ((192, 128), 64, 1, 2), ((72, 40), 32, 0, 2), ((136, 72), 16, 0, 2), ((200, 136), 64, 0, 3), ((72, 40), 32, 1, 3), ((136, 72), 16, 0, 3)])
def test_sao_frame_stats_in_other_chroma_formats_match_reference(depth, size, ctu, nd, csp):
    """4:2:2 (csp 2): the chroma planes are half as wide and as high as luma, their CTUs ctu / 2 x ctu; 4:4:4 (csp 3): full size.  SAO::calcSaoStatsCTU shifts the picture and
    CTU sizes by the format's shifts and keeps plane_offset 2 for every chroma plane (sao.cpp:748-756, 773): the oracle's plane-level function with the CTU's width and height
    given apart (xo_sao_stats_rows_wh) against the reference's SAO on a PicYuv of that format"""
    import ctypes as C
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "oracle", "_ref", "x265sao_%d" % depth)):
        pytest.skip("no reference SAO binary")
    W, H = size
    hs, vs = (0 if csp == 3 else 1), 0
    y = sao_frame_pair(depth, W, H, 7 + depth + W)
    cb, cr = sao_frame_pair(depth, W >> hs, H >> vs, 8 + depth + W), sao_frame_pair(depth, W >> hs, H >> vs, 9 + depth + W)
    a = sao_frame_reference(depth, y[0], y[1], ctu, nd, chroma=[cb, cr], csp=csp)
    ora = Oracle(depth)
    ora.lib.xo_sao_stats_rows_wh.restype = None
    P = lambda x: C.c_void_p(x.ctypes.data)  # noqa: E731
    assert np.array_equal(a[0], sao_frame_oracle(ora, y[0], y[1], ctu, nd, 0)), "luma"
    for plane, (f, r) in ((1, cb), (2, cr)):
        cw, ch = ctu >> hs, ctu >> vs
        n = (((W >> hs) + cw - 1) // cw) * (((H >> vs) + ch - 1) // ch)
        out = np.zeros((n, 2, 5, 32), np.int32)
        f, r = np.ascontiguousarray(f), np.ascontiguousarray(r)
        ora.lib.xo_sao_stats_rows_wh(P(f), P(r), C.c_ssize_t(W >> hs), W >> hs, H >> vs, cw, ch, nd, 2, P(out), None, 0, ((H >> vs) + ch - 1) // ch)
        assert out.shape == a[plane].shape and np.array_equal(a[plane], out), "plane %d" % plane
