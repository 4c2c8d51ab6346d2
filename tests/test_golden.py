"""Golden vectors captured from the real reference (tests/golden/make_golden.py) replayed through
the oracle (CPU) -- the same fixtures are replayed through the HIP library in test_hip_parity.py."""
import pytest

from depths import GOLDEN_DEPTHS

from backends import Oracle
from cases import FAMILIES, run_case, same
from golden_io import load


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
@pytest.mark.parametrize("family", sorted(FAMILIES))
def test_oracle_reproduces_golden(depth, family):
    ora = Oracle(depth)
    n = 0
    for label, method, args, outs in load(family, depth):
        assert same(run_case(ora, method, args), outs), "%s (depth %d)" % (label, depth)
        n += 1
    assert n >= 50
