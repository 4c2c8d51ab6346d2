"""Pins the restated inter-TU pipeline (xo_tq_tu: MC -> residual -> DCT -> quant [-> dequant -> IDCT -> recon -> SSE])
against the same chain composed, in the order quant.cpp:397-480 / 543-605 prescribe, from the REAL reference
primitives (oracle/_ref)."""
import numpy as np
import pytest

from depths import DEPTHS

import x265hip  # noqa: F401
from x265hip_pkg.synth import frame_pair
from backends import Oracle, Ref, ref_available

QS = [26214, 23302, 20560, 18396, 16384, 14564]
IQS = [40, 45, 51, 57, 64, 72]


def ref_chain(ref, depth, log2n, cur, stride, off, rf, roff, mv, qp, add_num):
    n = 1 << log2n
    dt = cur.dtype
    so = roff + (mv[0] >> 2) + (mv[1] >> 2) * stride
    xf, yf = mv[0] & 3, mv[1] & 3
    pred = np.zeros(n * n, dt)
    if xf == 0 and yf == 0:
        pred = ref.copy_pp(n, n, pred, n, rf[so:], stride)
    elif yf == 0:
        pred = ref.interp("hpp", 8, n, n, rf, stride, so, pred, n, xf)
    elif xf == 0:
        pred = ref.interp("vpp", 8, n, n, rf, stride, so, pred, n, yf)
    else:
        pred = ref.interp("hvpp", 8, n, n, rf, stride, so, pred, n, xf, yf)
    resi = ref.sub_ps(n, np.zeros(n * n, np.int16), n, cur[off:], pred, stride, n)
    coef = ref.dct(n, resi, n)
    per, rem = qp // 6, qp % 6
    tshift = 15 - depth - log2n
    qbits = 14 + per + tshift
    ns, q, du = ref.quant(coef, np.full(n * n, QS[rem], np.int32), qbits, add_num << (qbits - 9), n * n)
    if ns == 0:
        res2 = np.zeros(n * n, np.int16)
    else:
        deq = ref.dequant_normal(q, n * n, IQS[rem] << per, 20 - 14 - tshift)
        if ns == 1 and q[0] != 0:
            s2 = 12 - (depth - 8) - 3
            dc = ((((int(deq[0]) + 1) >> 1) * 8) + (1 << (s2 - 1))) >> s2
            res2 = np.full(n * n, np.int16(dc), np.int16)
        else:
            res2 = ref.idct(n, deq, np.zeros(n * n, np.int16), n)
    recon = ref.add_ps(n, np.zeros(n * n, dt), n, pred, res2, n, n)
    sse = ref.sse_pp(n, cur, stride, off, recon, n, 0)
    return ns, q, du, recon, sse


@pytest.mark.parametrize("depth", DEPTHS)
def test_tq_pipeline_matches_reference_primitives(depth):
    if not ref_available(depth):
        pytest.skip("no reference binary")
    rng = np.random.default_rng(5 + depth)
    ref, ora = Ref(depth), Oracle(depth)
    try:
        W, H, margin = 128, 96, 32
        cur, rf, stride, (dx, dy) = frame_pair(W, H, depth, 3, margin=margin, max_shift=6)
        cur, rf = cur.reshape(-1), rf.reshape(-1)
        n = 0
        for log2n in (2, 3, 4, 5):
            N = 1 << log2n
            for _ in range(12):
                px = int(rng.integers(0, (W - N) // 4 + 1)) * 4; py = int(rng.integers(0, (H - N) // 4 + 1)) * 4
                off = (margin + py) * stride + margin + px
                mv = (4 * dx + int(rng.integers(-6, 7)), 4 * dy + int(rng.integers(-6, 7))) if rng.random() < 0.8 else (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
                qp = int(rng.choice([4, 17, 22, 28, 37, 45, 51])); add = int(rng.choice([85, 171]))
                a = ref_chain(ref, depth, log2n, cur, stride, off, rf, off, mv, qp, add)
                b = ora.tq_tu(log2n, cur, stride, off, rf, stride, off, mv, qp, add, want_recon=True)
                assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), "coeff mismatch N=%d mv=%s qp=%d" % (N, mv, qp)
                assert np.array_equal(a[3], b[3]) and a[4] == b[4], "recon mismatch N=%d mv=%s qp=%d numSig=%d" % (N, mv, qp, a[0])
                n += 1
        assert n == 48
    finally:
        ref.close()
