"""SAO statistics slots (saoCuStatsBO / E0..E3) through the table, host pointers, against the oracle (pinned to the reference's
primitives by test_sao_oracle_vs_ref.py): class sums, counts and the sign buffers left behind."""
import numpy as np
import pytest

import x265hip
from backends import Oracle
from sao_util import cases, run_hip, run_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", [8, 10])
def test_sao_stats_slots_match_oracle(depth):
    lib, ora = x265hip.HipLib(depth), Oracle(depth)
    for i, c in enumerate(cases(depth, 900 + depth, n=100)):
        a, b = run_hip(lib, c), run_oracle(ora, c)
        for x, y, what in zip(a, b, ("stats", "count", "upBuff1", "upBufft")):
            assert np.array_equal(x, y), "case %d type %d endX %d endY %d: %s" % (i, c[0], c[5], c[6], what)
