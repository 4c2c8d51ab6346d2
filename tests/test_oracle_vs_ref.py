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


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_vector_builds_of_the_reference_table_return_what_the_O2_build_returns(depth):
    """bench.py's cpu_baseline also times the reference's C table at -O3 -march=x86-64-v3 / -v4 (oracle/Makefile: the stand-in for the asm table, which cannot be
    assembled here).  A faster build that computed something else would be no baseline: every case family through the widest build this host runs against the -O2 build."""
    from refproc import widest_variant
    var = widest_variant(depth)
    if not ref_available(depth) or var is None:
        pytest.skip("no -O3 vector build of the reference table for this host")
    rng = np.random.default_rng(0xBEEF + depth)
    a, b = Ref(depth), Ref(depth, var)
    n = 0
    try:
        for family in sorted(FAMILIES):
            for label, method, args in FAMILIES[family](depth, rng):
                if n % 3 == 0:          # a third of the cases: the families are large and this is the same source twice
                    assert same(run_case(a, method, args), run_case(b, method, args)), "%s (depth %d): -O3 %s build != -O2 build" % (label, depth, var)
                n += 1
    finally:
        a.close()
        b.close()
    assert n > 200
