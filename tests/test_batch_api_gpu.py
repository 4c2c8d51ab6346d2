"""The batched DEVICE entry points of include/x265hip.h with many items per launch (the table slots only ever
launch one item): pixel compare, block ops, transforms, quant family, interpolation, intra -- against the oracle.
Also argument validation / empty batches."""
import ctypes as C

import numpy as np
import pytest

from depths import DEPTHS

import x265hip  # noqa: F401
from x265hip_pkg.frame import FrameApi
from backends import Oracle
from cases import pix_buf, short_buf

pytestmark = pytest.mark.gpu
_IP = C.c_ssize_t


def _dp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.fixture(scope="module", params=[8, 10])
def env(request):
    depth = request.param
    api = FrameApi(depth)
    return depth, api, Oracle(depth), np.random.default_rng(500 + depth)


def test_pixelcmp_batch_many(env):
    depth, api, ora, rng = env
    T = api.torch
    A = pix_buf(rng, depth, 300 * 200, "rand"); B = pix_buf(rng, depth, 340 * 200, "rand")
    dA, dB = api.to_device(A), api.to_device(B)
    n = 257                                                       # ragged: not a multiple of the 4 items per workgroup
    for op, name, (w, h) in [(0, "sad", (16, 16)), (0, "sad", (12, 16)), (1, "satd", (32, 8)), (1, "satd", (4, 16)),
                             (1, "satd", (64, 64)), (2, "sa8d", (32, 32)), (3, "sse_pp", (16, 16)), (4, "psy_cost_pp", (8, 8))]:
        offA = rng.integers(0, 300 * 100, n).astype(np.int32); offB = rng.integers(0, 340 * 100, n).astype(np.int32)
        d_oa, d_ob = api.to_device(offA), api.to_device(offB)
        out = T.zeros(n, dtype=T.int64, device="cuda")
        api.h.check(api.lib.x265hip_pixelcmp_batch(api.stream(), op, w, h, _dp(dA), _IP(300), _dp(d_oa), _dp(dB), _IP(340), _dp(d_ob), n, _dp(out)))
        T.cuda.synchronize()
        raw = out.cpu().numpy()
        got = raw if op == 3 else raw.view(np.int32)[:n]
        for i in range(0, n, 7):
            args = (w, h, A, 300, int(offA[i]), B, 340, int(offB[i])) if name in ("sad", "satd") else (w, A, 300, int(offA[i]), B, 340, int(offB[i]))
            assert int(got[i]) == int(getattr(ora, name)(*args)), "%s %dx%d item %d" % (name, w, h, i)


def test_transform_and_quant_batch_many(env):
    depth, api, ora, rng = env
    T = api.torch
    pm = (1 << depth) - 1
    n = 101
    for N in (4, 8, 16, 32):
        src = short_buf(rng, n * N * N, "rand", -pm, pm)
        d_src = api.to_device(src)
        d_coef = T.zeros(n * N * N, dtype=T.int16, device="cuda")
        api.h.check(api.lib.x265hip_transform_batch(api.stream(), 0, N, _dp(d_src), _IP(N), None, _dp(d_coef), _IP(N), None, n))
        qc = np.full(N * N, 16384, np.int32); d_qc = api.to_device(qc)
        qbits = 14 + 4 + (15 - depth - int(np.log2(N)))
        d_q = T.zeros(n * N * N, dtype=T.int16, device="cuda"); d_du = T.zeros(n * N * N, dtype=T.int32, device="cuda")
        d_ns = T.full((n,), 77, dtype=T.int32, device="cuda")
        api.h.check(api.lib.x265hip_quant_batch(api.stream(), _dp(d_coef), _dp(d_qc), _dp(d_du), _dp(d_q), qbits, 85 << (qbits - 9), N * N, n, _dp(d_ns)))
        d_deq = T.zeros(n * N * N, dtype=T.int16, device="cuda")
        shift = 20 - 14 - (15 - depth - int(np.log2(N)))
        api.h.check(api.lib.x265hip_dequant_normal_batch(api.stream(), _dp(d_q), _dp(d_deq), n * N * N, 64 << 4, shift))
        d_rec = T.zeros(n * N * N, dtype=T.int16, device="cuda")
        api.h.check(api.lib.x265hip_transform_batch(api.stream(), 1, N, _dp(d_deq), _IP(N), None, _dp(d_rec), _IP(N), None, n))
        T.cuda.synchronize()
        coef, q, du, ns, deq, rec = (t.cpu().numpy() for t in (d_coef, d_q, d_du, d_ns, d_deq, d_rec))
        for i in range(0, n, 5):
            s = slice(i * N * N, (i + 1) * N * N)
            e_coef = ora.dct(N, src[s], N)
            assert np.array_equal(coef[s], e_coef)
            e_ns, e_q, e_du = ora.quant(e_coef, qc, qbits, 85 << (qbits - 9), N * N)
            assert int(ns[i]) == e_ns and np.array_equal(q[s], e_q) and np.array_equal(du[s], e_du)
            e_deq = ora.dequant_normal(e_q, N * N, 64 << 4, shift)
            assert np.array_equal(deq[s], e_deq)
            assert np.array_equal(rec[s], ora.idct(N, e_deq, np.zeros(N * N, np.int16), N))


def test_interp_batch_many(env):
    depth, api, ora, rng = env
    T = api.torch
    src = pix_buf(rng, depth, 256 * 160, "rand"); d_src = api.to_device(src)
    n = 90
    for op, kind, taps, (w, h) in [(0, "hpp", 8, (16, 16)), (2, "vpp", 8, (8, 32)), (6, "hvpp", 8, (32, 32)), (1, "hps", 4, (8, 8)),
                                   (3, "vps", 4, (4, 8)), (7, "p2s", 8, (64, 16))]:
        offs = (rng.integers(8, 100, n) * 256 + rng.integers(8, 150, n)).astype(np.int32)
        nidx = 4 if taps == 8 else 8
        idx = rng.integers(1, nidx, n); idy = rng.integers(1, nidx, n); ext = rng.integers(0, 2, n) if kind == "hps" else np.zeros(n, int)
        cidx = (idx | (idy << 8) | (ext << 16)).astype(np.int32)
        rows = h + 7
        doff = (np.arange(n) * (w * rows)).astype(np.int32)
        short_out = kind in ("hps", "vps", "p2s")
        d_dst = T.zeros(n * w * rows, dtype=T.int16 if (short_out or depth > 8) else T.uint8, device="cuda")
        d_offs, d_doff, d_cidx = api.to_device(offs), api.to_device(doff), api.to_device(cidx)      # keep the tensors alive
        api.h.check(api.lib.x265hip_interp_batch(api.stream(), op, taps, w, h, _dp(d_src), _IP(256), _dp(d_offs), _dp(d_dst), _IP(w),
                                                 _dp(d_doff), _dp(d_cidx), n))
        T.cuda.synchronize()
        got = d_dst.cpu().numpy()
        if not short_out:
            got = got.view(src.dtype)
        for i in range(0, n, 6):
            orows = h + (taps - 1 if (kind == "hps" and ext[i]) else 0)
            dst0 = np.zeros(w * orows, np.int16 if short_out else src.dtype)
            exp = ora.interp(kind, taps, w, h, src, 256, int(offs[i]), dst0, w, int(idx[i]), int(ext[i]) if kind == "hps" else int(idy[i]))
            assert np.array_equal(got[doff[i]:doff[i] + w * orows], exp), "%s %dx%d item %d" % (kind, w, h, i)


def test_intra_batch_many(env):
    depth, api, ora, rng = env
    T = api.torch
    n = 64
    for N in (4, 8, 16, 32):
        L = 4 * N + 1
        nb = pix_buf(rng, depth, n * L, "rand"); d_nb = api.to_device(nb)
        d_f = T.zeros(n * L, dtype=d_nb.dtype, device="cuda")
        api.h.check(api.lib.x265hip_intra_filter_batch(api.stream(), N, _dp(d_nb), None, _dp(d_f), None, n))
        mf = (rng.integers(0, 35, n) | (rng.integers(0, 2, n) << 8)).astype(np.int32)
        d_dst = T.zeros(n * N * N, dtype=d_nb.dtype, device="cuda")
        d_mf = api.to_device(mf)
        api.h.check(api.lib.x265hip_intra_pred_batch(api.stream(), N, _dp(d_nb), None, _dp(d_dst), _IP(N), None, _dp(d_mf), n))
        d_all = T.zeros(n * 33 * N * N, dtype=d_nb.dtype, device="cuda")
        api.h.check(api.lib.x265hip_intra_allangs_batch(api.stream(), N, _dp(d_nb), None, _dp(d_f), None, _dp(d_all), 1, n))
        T.cuda.synchronize()
        flt = d_f.cpu().numpy().view(nb.dtype); dst = d_dst.cpu().numpy().view(nb.dtype); alla = d_all.cpu().numpy().view(nb.dtype)
        for i in range(0, n, 3):
            s = nb[i * L:(i + 1) * L]
            e_f = ora.intra_filter(N, s, np.zeros(L, nb.dtype))
            assert np.array_equal(flt[i * L:(i + 1) * L], e_f)
            mode, bf = int(mf[i] & 0xFF), int(mf[i] >> 8)
            assert np.array_equal(dst[i * N * N:(i + 1) * N * N], ora.intra_pred(N, s, np.zeros(N * N, nb.dtype), N, mode, bf))
            assert np.array_equal(alla[i * 33 * N * N:(i + 1) * 33 * N * N], ora.intra_allangs(N, s, e_f, 1))


def test_argument_validation_and_empty_batches(env):
    depth, api, ora, rng = env
    lib = api.lib
    z = None
    assert lib.x265hip_pixelcmp_batch(api.stream(), 0, 16, 16, z, _IP(0), z, z, _IP(0), z, 0, z) == 0          # n == 0: nothing to do
    assert lib.x265hip_pixelcmp_batch(api.stream(), 99, 16, 16, z, _IP(0), z, z, _IP(0), z, 1, z) == -3
    assert lib.x265hip_pixelcmp_batch(api.stream(), 2, 16, 8, z, _IP(0), z, z, _IP(0), z, 1, z) == -3         # sa8d needs squares
    assert lib.x265hip_transform_batch(api.stream(), 0, 12, z, _IP(0), z, z, _IP(0), z, 1) == -3
    assert lib.x265hip_transform_batch(api.stream(), 2, 8, z, _IP(0), z, z, _IP(0), z, 1) == -3                # DST is 4x4 only
    assert lib.x265hip_interp_batch(api.stream(), 0, 6, 8, 8, z, _IP(0), z, z, _IP(0), z, z, 1) == -3
    assert lib.x265hip_me_batch(api.stream(), 16, 16, z, _IP(0), z, _IP(0), z, 0, z, 0, 16, 1, 2, z, z, z, C.c_int64(0)) == 0
    assert lib.x265hip_me_batch(api.stream(), 16, 16, z, _IP(0), z, _IP(0), z, 5, z, 0, 16, 1, 2, z, z, z, C.c_int64(0)) == -3
    assert b"me_batch" in lib.x265hip_last_error()
    assert lib.x265hip_abi_check(C.c_size_t(18240), depth) == 0 and lib.x265hip_abi_check(C.c_size_t(18240), 12) == -1


@pytest.mark.parametrize("depth", DEPTHS)
def test_intra_cost_batch_matches_oracle(depth):
    """x265hip_intra_cost_batch over many CUs of one plane (offset addressing, 64x64 workspace path)."""
    from x265hip_pkg.frame import FrameApi
    from backends import Oracle
    api, ora = FrameApi(depth), Oracle(depth)
    T = api.torch
    rng = np.random.default_rng(5 + depth)
    dt = np.uint8 if depth == 8 else np.uint16
    W = H = 256
    plane = rng.integers(0, 1 << depth, W * H).astype(dt)
    d_plane = api.to_device(plane)
    for lg in (2, 3, 4, 5, 6):
        n, size = 24, 1 << lg
        offs = np.array([int(rng.integers(0, H - size)) * W + int(rng.integers(0, W - size)) for _ in range(n)], np.int32)
        pitch = 4 * size + 1 + 3
        nb_ref = rng.integers(0, 1 << depth, n * pitch).astype(dt)
        nb_flt = rng.integers(0, 1 << depth, n * pitch).astype(dt)
        d_cost = T.zeros(n * 35, dtype=T.int32, device="cuda")
        api.intra_cost_batch(lg, d_plane, W, api.to_device(offs), api.to_device(nb_ref), api.to_device(nb_flt), pitch, n, d_cost)
        T.cuda.synchronize()
        got = d_cost.cpu().numpy().reshape(n, 35)
        for i in range(n):
            exp = ora.intra_costs(size, plane, W, int(offs[i]), nb_ref[i * pitch:(i + 1) * pitch], nb_flt[i * pitch:(i + 1) * pitch])
            assert np.array_equal(got[i], exp), "size %d CU %d" % (size, i)


def test_dct16_pairs_on_the_matrix_cores(env):
    """16x16 forward and inverse transforms run two TUs per 32x32x32 MFMA product (block-diagonal operands, csrc/xh_dct32.h): odd TU counts (a wavefront with one TU),
    full-range int16 inputs, strided sources / destinations with odd offsets."""
    depth, api, ora, rng = env
    T = api.torch
    N = 16
    for n in (1, 2, 67):
        stride = 40
        off = (np.arange(n, dtype=np.int32) * (N * stride + 8) + (np.arange(n, dtype=np.int32) % 4)).astype(np.int32)
        total = int(off[-1]) + N * stride + 8
        src = rng.integers(-32768, 32768, total).astype(np.int16)
        src[:N * stride] = -32768
        d_src = api.to_device(src); d_off = api.to_device(off)
        d_coef = T.full((n * N * N,), 77, dtype=T.int16, device="cuda")
        api.h.check(api.lib.x265hip_transform_batch(api.stream(), 0, N, _dp(d_src), _IP(stride), _dp(d_off), _dp(d_coef), _IP(N), None, n))
        coef_in = rng.integers(-32768, 32768, n * N * N).astype(np.int16)
        coef_in[:N * N] = 32767
        d_cin = api.to_device(coef_in)
        d_rec = T.full((total,), 1234, dtype=T.int16, device="cuda")
        api.h.check(api.lib.x265hip_transform_batch(api.stream(), 1, N, _dp(d_cin), _IP(N), None, _dp(d_rec), _IP(stride), _dp(d_off), n))
        T.cuda.synchronize()
        coef, rec = d_coef.cpu().numpy(), d_rec.cpu().numpy()
        touched = np.zeros(total, bool)
        for i in range(n):
            blk = np.lib.stride_tricks.as_strided(src[int(off[i]):], shape=(N, N), strides=(stride * 2, 2))
            assert np.array_equal(coef[i * N * N:(i + 1) * N * N], ora.dct(N, np.ascontiguousarray(blk).reshape(-1), N)), "forward TU %d of %d" % (i, n)
            exp = ora.idct(N, coef_in[i * N * N:(i + 1) * N * N], np.zeros(N * N, np.int16), N).reshape(N, N)
            got = np.lib.stride_tricks.as_strided(rec[int(off[i]):], shape=(N, N), strides=(stride * 2, 2))
            assert np.array_equal(got, exp), "inverse TU %d of %d" % (i, n)
            for y in range(N):
                touched[int(off[i]) + y * stride:int(off[i]) + y * stride + N] = True
        assert np.all(rec[~touched] == 1234), "the inverse kernel wrote outside its blocks"


def test_idct32_on_the_matrix_cores_is_exact_for_every_int16(env):
    """The 32x32 inverse transform runs as two int8 MFMA products per stage (csrc/xh_dct32.h: three exact byte planes of an int16 operand): full-range coefficients incl.
    -32768 / 32767, sparse blocks, a strided destination with odd offsets -- all equal to the oracle (partialButterflyInverse32, dct.cpp:242-416, both clip16 points)."""
    depth, api, ora, rng = env
    T = api.torch
    n, N = 67, 32
    coef = rng.integers(-32768, 32768, n * N * N).astype(np.int16)
    coef[:N * N] = -32768; coef[N * N:2 * N * N] = 32767                                  # saturating blocks
    coef[2 * N * N:3 * N * N] = 0; coef[2 * N * N] = 32767                                 # DC only
    coef[3 * N * N:4 * N * N] = np.where(rng.random(N * N) < 0.05, coef[3 * N * N:4 * N * N], 0)
    stride = 48
    d_off = (np.arange(n, dtype=np.int32) * (N * stride + 8) + (np.arange(n, dtype=np.int32) % 4)).astype(np.int32)       # odd element offsets: the unaligned store path
    total = int(d_off[-1]) + N * stride + 8
    d_coef = api.to_device(coef); dd_off = api.to_device(d_off)
    d_rec = T.full((total,), 1234, dtype=T.int16, device="cuda")
    api.h.check(api.lib.x265hip_transform_batch(api.stream(), 1, N, _dp(d_coef), _IP(N), None, _dp(d_rec), _IP(stride), _dp(dd_off), n))
    T.cuda.synchronize()
    rec = d_rec.cpu().numpy()
    for i in range(n):
        exp = ora.idct(N, coef[i * N * N:(i + 1) * N * N], np.zeros(N * N, np.int16), N).reshape(N, N)
        got = np.lib.stride_tricks.as_strided(rec[int(d_off[i]):], shape=(N, N), strides=(stride * 2, 2))
        assert np.array_equal(got, exp), "TU %d" % i
    touched = np.zeros(total, bool)
    for i in range(n):
        for y in range(N):
            touched[int(d_off[i]) + y * stride:int(d_off[i]) + y * stride + N] = True
    assert np.all(rec[~touched] == 1234), "the kernel wrote outside its blocks"
