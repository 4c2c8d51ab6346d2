"""GPU parity: the HIP library, called through the drop-in table (C ABI), against the oracle on
seeded harness-style inputs, and against the golden vectors captured from the real reference.
Bit-exact (integer path): any difference is a failure."""
import numpy as np
import pytest

from depths import DEPTHS, GOLDEN_DEPTHS

from backends import Hip, Oracle
from cases import FAMILIES, run_case, same
from golden_io import load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_hip_matches_oracle(depth, family):
    rng = np.random.default_rng(0xBADC0DE + depth)
    hip, ora = Hip(depth), Oracle(depth)
    n = 0
    for label, method, args in FAMILIES[family](depth, rng):
        a = run_case(ora, method, args)
        b = run_case(hip, method, args)
        assert same(a, b), "%s (depth %d): HIP != oracle" % (label, depth)
        n += 1
    assert n > 20


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_hip_reproduces_golden(depth, family):
    hip = Hip(depth)
    for label, method, args, outs in load(family, depth):
        assert same(run_case(hip, method, args), outs), "%s (depth %d): HIP != reference golden" % (label, depth)
