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
