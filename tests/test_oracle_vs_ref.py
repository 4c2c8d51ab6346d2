"""Pins oracle/x265_oracle.c against the REAL reference C primitives (oracle/_ref, built from
/root/reference by oracle/Makefile).  Skipped where the reference binary is absent."""
import numpy as np
import pytest

from depths import DEPTHS, GOLDEN_DEPTHS

from backends import Oracle, Ref, ref_available
from cases import FAMILIES, run_case, same


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_oracle_matches_reference(depth, family):
    if not ref_available(depth):
        pytest.skip("oracle/_ref/x265ref_%d not built (no /root/reference here)" % depth)
    rng = np.random.default_rng(0xC0FFEE + depth)
    ref, ora = Ref(depth), Oracle(depth)
    n = 0
    try:
        for label, method, args in FAMILIES[family](depth, rng):
            a = run_case(ref, method, args)
            b = run_case(ora, method, args)
            assert same(a, b), "%s (depth %d): oracle != reference" % (label, depth)
            n += 1
    finally:
        ref.close()
    assert n > 20


@pytest.mark.parametrize("depth", DEPTHS)
def test_dct_matrices_are_the_reference_tables(depth):
    if not ref_available(depth):
        pytest.skip("no reference binary")
    ref, ora = Ref(depth), Oracle(depth)
    try:
        for n in (4, 8, 16, 32):
            assert np.array_equal(ref.dct_matrix(n), ora.dct_matrix(n))
    finally:
        ref.close()


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_known_answers_from_survey(depth):
    """values captured from the real reference build during the survey."""
    ora = Oracle(depth)
    pm = (1 << depth) - 1
    dt = np.uint8 if depth == 8 else np.uint16
    i = np.arange(64 * 64)
    a = ((7 * i + 3) & pm).astype(dt)
    b = ((13 * i + 5) & pm).astype(dt)
    r = (((37 * i) % 511) - 255).astype(np.int16)
    exp = {8: (22800, 37184, 99328, (-59, 449, -159)), 10: (83780, 101856, 453184, (-15, 112, -40))}[depth]
    assert ora.sad(16, 16, a, 64, 0, b, 64, 0) == exp[0]
    assert ora.satd(16, 16, a, 64, 0, b, 64, 0) == exp[1]
    assert ora.sa8d(32, a, 64, 0, b, 64, 0) == exp[2]
    c = ora.dct(32, r, 32)  # the survey driver used a dense 32-wide residual
    assert (int(c[0]), int(c[1]), int(c[32])) == exp[3]
