"""Table slots next to the lookahead / SEA paths, called with HOST pointers the way the reference calls them:
propagateCost, fix8Pack / fix8Unpack (cuTree; pixel.cpp:906-948) and integral_init{4..32}{h,v} (SEA; framefilter.cpp:38-139),
against the oracle restatements (themselves pinned to the reference: test_lookahead_oracle_vs_ref.py, test_me_oracle_vs_ref.py)."""
import ctypes as C

import numpy as np
import pytest

from depths import DEPTHS

import x265hip
from backends import Oracle

pytestmark = pytest.mark.gpu
P = C.c_void_p


def ptr(a, off=0):
    return C.c_void_p(a.ctypes.data + off * a.itemsize)


@pytest.mark.parametrize("depth", DEPTHS)
def test_cutree_row_slots(depth):
    lib, ora = x265hip.HipLib(depth), Oracle(depth)
    rng = np.random.default_rng(depth)
    fn = lib.scalar("propagateCost", None, (P, P, P, P, P, P, C.c_int))
    for n in (1, 7, 120, 481):
        pin = rng.integers(0, 65536, n).astype(np.uint16)
        intra = rng.integers(1, 40000, n).astype(np.int32)
        inter = (rng.integers(0, 16384, n) | (rng.integers(0, 4, n) << 14)).astype(np.uint16)
        invq = rng.integers(100, 400, n).astype(np.int32)
        for fps in (256.0, 197.3, 61.0):
            f = np.array([fps], np.float64)
            got = np.full(n, -1, np.int32); exp = np.full(n, -1, np.int32)
            fn(ptr(got), ptr(pin), ptr(intra), ptr(inter), ptr(invq), ptr(f), n)
            ora.me_lib.xo_cu_propagate_cost.argtypes = [P, P, P, P, P, C.c_double, C.c_int]
            ora.me_lib.xo_cu_propagate_cost(ptr(exp), ptr(pin), ptr(intra), ptr(inter), ptr(invq), fps, n)
            assert np.array_equal(got, exp), "propagateCost n=%d fps=%g" % (n, fps)
    pack = lib.scalar("fix8Pack", None, (P, P, C.c_int)); unpack = lib.scalar("fix8Unpack", None, (P, P, C.c_int))
    src = np.concatenate([rng.uniform(-12, 12, 300), [0.0, -0.00390625, 0.00390625, 51.99, -51.99]])
    q = np.zeros(src.size, np.uint16); pack(ptr(q), ptr(src), src.size)
    assert np.array_equal(q, (src * 256.0).astype(np.int64).astype(np.int16).view(np.uint16))       # (uint16_t)(int16_t)(x * 256.0): truncation toward zero
    back = np.zeros(src.size, np.float64); unpack(ptr(back), ptr(q), src.size)
    assert np.array_equal(back, q.view(np.int16).astype(np.float64) / 256.0)


@pytest.mark.parametrize("depth", DEPTHS)
def test_integral_row_slots_build_the_oracle_planes(depth):
    """drive the 12 slots the way FrameFilter::processPostRow does (framefilter.cpp:757-833) on a small padded picture"""
    lib, ora = x265hip.HipLib(depth), Oracle(depth)
    rng = np.random.default_rng(40 + depth)
    W, H, pad = 64, 64, 20
    stride, rows = W + 2 * pad, H + 2 * pad
    pm = (1 << depth) - 1
    pic = rng.integers(0, pm + 1, stride * rows).astype(np.uint8 if depth == 8 else np.uint16)
    org = pad * stride + pad
    exp = ora.sea_integral_planes(pic, stride, org, H, pad, pad)
    sizes = [4, 8, 12, 16, 24, 32]
    hfn = [lib.scalar("integral_inith", None, (P, P, C.c_ssize_t), extra=i) for i in range(6)]
    vfn = [lib.scalar("integral_initv", None, (P, C.c_ssize_t), extra=i) for i in range(6)]
    BW = [32, 32, 32, 24, 16, 16, 16, 12, 8, 8, 4, 4]; BH = [32, 24, 8, 32, 16, 12, 4, 16, 32, 8, 16, 4]
    for k in (0, 3, 5, 7, 9, 11):                    # one plane per width (and several heights)
        plane = np.full(stride * rows, 0xdeadbeef, np.uint32)
        plane[:stride] = 0
        for y in range(-pad, H + pad - 1):
            sum_off = org + (y + 1) * stride - pad
            hfn[sizes.index(BW[k])](ptr(plane, sum_off), ptr(pic, org + y * stride - pad), stride)
            if y >= BH[k] - pad:
                vfn[sizes.index(BH[k])](ptr(plane, sum_off - BH[k] * stride), stride)
        a = plane.reshape(rows, stride)[1:rows - BH[k], :stride - BW[k]]
        b = exp[k].reshape(rows, stride)[1:rows - BH[k], :stride - BW[k]]
        assert np.array_equal(a, b), "integral plane %d through the slots" % k
