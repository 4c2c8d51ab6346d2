"""The bit and cost arithmetic of the per-PU choice among references (x265hip_inter_merge_batch / xo_inter_merge): BitCost::bitcost and
RDCost::getCost of the oracle and of the library's host helpers against the reference's own classes (oracle/_ref, op bits_cost)."""
import numpy as np
import pytest

from depths import DEPTHS

import x265hip  # noqa: F401
from x265hip_pkg.frame import mvbits_row, rd_lambda
from backends import Oracle
from refproc import RefProc, ref_available


@pytest.mark.parametrize("depth", DEPTHS)
def test_bit_sizes_and_rd_lambda_match_the_reference(depth):
    ora = Oracle(depth)
    half = 1 << 12
    row = ora.mvbits_row(half)
    assert np.array_equal(row, mvbits_row(depth, half))                      # library host helper == oracle, bit for bit (float)
    for qp in (0, 12, 22, 28, 37, 51):
        assert ora.rd_lambda(qp) == rd_lambda(depth, qp)
    if not ref_available(depth):
        pytest.skip("reference binary not built here")
    rng = np.random.default_rng(depth)
    r = RefProc(depth)
    try:
        for qp in (12, 22, 28, 37, 51):
            n = 200
            mv = rng.integers(-700, 701, (n, 2)); mvp = rng.integers(-300, 301, (n, 2)); bits = rng.integers(0, 200, n)
            ints = [qp, n] + np.concatenate([mv, mvp, bits[:, None]], axis=1).reshape(-1).tolist()
            out = r.call("bits_cost", ints)
            bc, gc = np.frombuffer(out[0], np.uint32), np.frombuffer(out[1], np.uint32)
            lam = ora.rd_lambda(qp)
            for i in range(n):
                mine = int(np.float32(np.float32(row[half + mv[i, 0] - mvp[i, 0]] + row[half + mv[i, 1] - mvp[i, 1]]) + np.float32(0.5)))
                assert mine == int(bc[i]), "bitcost(%s, %s)" % (mv[i], mvp[i])
                assert (int(bits[i]) * lam + 128) >> 8 == int(gc[i])
    finally:
        r.close()
