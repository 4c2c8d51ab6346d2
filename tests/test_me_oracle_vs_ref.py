"""Pins the restated motion-search driver (oracle/x265_oracle_me.c) against the REAL reference
MotionEstimate::motionEstimate + BitCost (encoder/motion.cpp, bitcost.cpp, compiled into oracle/_ref)."""
import numpy as np
import pytest

from depths import DEPTHS

import x265hip  # noqa: F401  (makes the package importable as x265hip_pkg)
from x265hip_pkg.synth import frame_pair
from backends import Oracle, Ref, ref_available

PUS = [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 16), (16, 32), (64, 32), (32, 64), (16, 12), (12, 16),
       (16, 4), (4, 16), (32, 24), (24, 32), (32, 8), (8, 32), (64, 48), (48, 64), (64, 16), (16, 64), (8, 4), (4, 8)]


@pytest.mark.parametrize("depth", DEPTHS)
def test_lambda_and_mvcost_tables(depth):
    if not ref_available(depth):
        pytest.skip("no reference binary")
    ref, ora = Ref(depth), Oracle(depth)
    try:
        assert np.array_equal(ref.lambda_tab(), ora.lambda_tab())
        for qp in (0, 12, 22, 28, 37, 51, 69):
            assert np.array_equal(ref.mvcost_row(qp, 4000), ora.mvcost_row(qp, 4000)), "qp %d" % qp
    finally:
        ref.close()


# SEA on these PU shapes reads fenc rows / columns OUTSIDE the PU for its DC terms (motion.cpp:1467-1468: deltaX = w, deltaY = h for
# sizes <= 8, then sad_x4 at fenc + deltaX / + deltaY * FENC_STRIDE :1504-1510): the reference's result depends on what earlier PUs
# left in MotionEstimate::fencPUYuv, i.e. it is not a function of the inputs.  They are excluded from parity (and not offloaded).
SEA_UNDEFINED = {(8, 4), (4, 8), (8, 32), (32, 8)}


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("method", [0, 1, 2, 3, 4, 5])      # DIA, HEX, UMH, STAR, SEA, FULL
def test_motion_estimate_matches_reference(depth, method):
    if not ref_available(depth):
        pytest.skip("no reference binary")
    rng = np.random.default_rng(1000 * depth + method)
    ref, ora = Ref(depth), Oracle(depth)
    n = 0
    try:
        for seed in range(3):
            W, H, margin = 256, 192, 96
            cur, rf, stride, (dx, dy) = frame_pair(W, H, depth, seed, margin=margin, max_shift=12 if seed else 30)
            cur, rf = cur.reshape(-1), rf.reshape(-1)
            integral = None
            for (w, h) in PUS:
                if method == 4 and (w, h) in SEA_UNDEFINED:
                    continue
                reps = 1 if method == 5 else 6 if method == 2 else 3
                if method == 4 and integral is None:
                    # SEA: the 12 integral planes of this reference picture (the synthetic planes are H + 2*margin rows, margin on every side)
                    integral = ora.sea_integral_planes(rf, stride, margin * stride + margin, H, margin, margin)
                for _ in range(reps):
                    px = int(rng.integers(0, (W - w) // 4 + 1)) * 4
                    py = int(rng.integers(0, (H - h) // 4 + 1)) * 4
                    off = (margin + py) * stride + margin + px
                    merange = int(rng.choice([8, 16, 57])) if method not in (4, 5) else int(rng.choice([6, 12])) if method == 4 else 6
                    qp = int(rng.choice([22, 28, 37]))
                    subme = int(rng.integers(0, 8))
                    # MVP near the true motion most of the time, sometimes far/zero
                    r = rng.random()
                    if r < 0.6:
                        qmvp = (4 * dx + int(rng.integers(-9, 10)), 4 * dy + int(rng.integers(-9, 10)))
                    elif r < 0.8:
                        qmvp = (0, 0)
                    else:
                        qmvp = (int(rng.integers(-120, 121)), int(rng.integers(-120, 121)))
                    # search window around the MVP, clipped to the padded picture (search.cpp setSearchRange)
                    lim = margin - 12
                    mvp_f = (qmvp[0] >> 2, qmvp[1] >> 2)
                    bounds = [max(mvp_f[0] - merange, -px - lim), max(mvp_f[1] - merange, -py - lim),
                              min(mvp_f[0] + merange, W - px - w + lim), min(mvp_f[1] + merange, H - py - h + lim)]
                    nc = int(rng.integers(0, 4))
                    mvc = [int(v) for v in rng.integers(-80, 81, 2 * nc)]
                    a = ref.me(w, h, cur, stride, off, rf, stride, off, bounds, qmvp, mvc, merange, method, subme, qp,
                               sea=(margin * stride + margin, H, margin, margin) if method == 4 else None)
                    row = ora.mvcost_row(qp, 1 << 13)
                    b = ora.me(w, h, cur, stride, off, rf, stride, off, bounds, qmvp, mvc, merange, method, subme, row, integral=integral)
                    assert a == b, "PU %dx%d method %d subme %d qp %d mvp %s bounds %s: ref %s oracle %s" % (
                        w, h, method, subme, qp, qmvp, bounds, a, b)
                    n += 1
    finally:
        ref.close()
    assert n >= 70


@pytest.mark.parametrize("depth", DEPTHS)
def test_library_cost_row_matches_oracle_and_reference(depth):
    """x265hip_mvcost_row is host code of the PRODUCT (no GPU needed): it must reproduce BitCost::setQP."""
    from x265hip_pkg.frame import mvcost_row
    ora = Oracle(depth)
    ref = Ref(depth) if ref_available(depth) else None
    try:
        for qp in (0, 7, 22, 28, 37, 51):
            row = mvcost_row(depth, qp, 3000)
            assert np.array_equal(row, ora.mvcost_row(qp, 3000))
            if ref:
                assert np.array_equal(row, ref.mvcost_row(qp, 3000))
    finally:
        if ref:
            ref.close()
