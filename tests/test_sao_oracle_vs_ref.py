"""SAO statistics: the oracle restatement against the reference's own primitives (oracle/_ref, op sao_stats)."""
import numpy as np
import pytest

from backends import Oracle, Ref, ref_available
from sao_util import cases, run_oracle, run_ref


@pytest.mark.parametrize("depth", [8, 10])
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


def sao_frame_reference(depth, fenc, rec, ctu, non_deblock=0, chroma=None):
    """chroma = [(fencCb, recCb), (fencCr, recCr)] of a 4:2:0 picture -> array [planes, ctus, 2, 5, 32]; luma only -> [ctus, 2, 5, 32]"""
    import os, subprocess, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    H, W = fenc.shape
    with tempfile.TemporaryDirectory() as td:
        inp, out = os.path.join(td, "in.raw"), os.path.join(td, "out.bin")
        parts = [fenc.reshape(-1), rec.reshape(-1)] + ([a.reshape(-1) for pr in chroma for a in pr] if chroma else [])
        np.concatenate(parts).tofile(inp)
        r = subprocess.run([os.path.join(root, "oracle", "_ref", "x265sao_%d" % depth), str(W), str(H), str(ctu), inp, out, str(non_deblock), "3" if chroma else "1"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1000:]
        o = np.fromfile(out, np.int32)
        return o.reshape(3, -1, 2, 5, 32) if chroma else o.reshape(-1, 2, 5, 32)


def sao_frame_oracle(ora, fenc, rec, ctu, non_deblock=0, plane_offset=0):
    import ctypes as C
    H, W = fenc.shape
    n = ((W + ctu - 1) // ctu) * ((H + ctu - 1) // ctu)
    out = np.zeros((n, 2, 5, 32), np.int32)
    P = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
    f, r = np.ascontiguousarray(fenc), np.ascontiguousarray(rec)
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


@pytest.mark.parametrize("depth", [8, 10])
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


@pytest.mark.parametrize("depth", [8, 10])
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
