"""SURVEY 8(f1) at the data level, CPU side: the oracle's motionEstimate restatement replays the calls a reference encode with --threaded-me
made (Search::puMotionEstimation's window, predictor and candidate lists; fixtures from oracle/ref_tme.cpp) and must return the reference's MV and cost."""
import numpy as np
import pytest

from depths import GOLDEN_DEPTHS

from backends import Oracle
from tme_util import TmeFixture


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_oracle_replays_the_threaded_me_calls_of_a_reference_encode(depth):
    fx, ora = TmeFixture(depth), Oracle(depth)
    assert len(fx) > 1500
    c = fx.col
    assert not c["chromaSatd"].any() and not c["vertRestriction"].any() and not c["srcPlane"].any() and (c["maxSlices"] == 1).all()
    rows = {}
    step = 1
    n = 0
    for i in range(0, len(fx), step):
        qp = int(c["qp"][i])
        if qp not in rows:
            rows[qp] = ora.mvcost_row(qp, 1 << 14)
        pl = fx.planes[int(c["plane"][i])]
        w, h, nc = int(c["w"][i]), int(c["h"][i]), int(c["numCand"][i])
        got = ora.me(w, h, fx.block(i), w, 0, pl["px"], pl["stride"], pl["origin"] + int(c["blockOffset"][i]),
                     [int(c["mnx"][i]), int(c["mny"][i]), int(c["mxx"][i]), int(c["mxy"][i])], (int(c["qmvpx"][i]), int(c["qmvpy"][i])),
                     [int(v) for v in fx.mvc[i, :2 * nc]], int(c["merange"][i]), int(c["method"][i]), int(c["subme"][i]), rows[qp])
        assert got == (int(c["outx"][i]), int(c["outy"][i]), int(c["cost"][i])), "call %d (%dx%d, %d candidates): oracle %s reference %s" % (
            i, w, h, nc, got, (int(c["outx"][i]), int(c["outy"][i]), int(c["cost"][i])))
        n += 1
    assert n == len(fx)


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_oracle_replays_the_chroma_satd_searches_of_a_reference_encode(depth):
    """Search::predInterSearch's call form (the Yuv overload of setSourcePU with bChroma, search.cpp:2582): at subme 3 / 4 every sub-pel cost carries
    the SATD of the Cb and Cr predictions (motion.cpp:1805-1865).  Fixtures: a regular (not threaded-me) encode, P and B pictures, up to 12 candidates."""
    from tme_util import MecFixture
    fx, ora = MecFixture(depth), Oracle(depth)
    c = fx.col
    assert len(fx) > 800 and c["chromaSatd"].all() and (c["subme"] >= 3).all()
    rows = {}
    for i in range(len(fx)):
        qp = int(c["qp"][i])
        if qp not in rows:
            rows[qp] = ora.mvcost_row(qp, 1 << 14)
        pl, cb, cr = fx.planes[int(c["plane"][i])], fx.planes[int(c["cbPlane"][i])], fx.planes[int(c["crPlane"][i])]
        w, h, nc, cw = int(c["w"][i]), int(c["h"][i]), int(c["numCand"][i]), int(c["cw"][i])
        y, u, v = fx.blocks(i)
        got = ora.me_chroma(w, h, y, w, 0, pl["px"], pl["stride"], pl["origin"] + int(c["blockOffset"][i]),
                            [int(c["mnx"][i]), int(c["mny"][i]), int(c["mxx"][i]), int(c["mxy"][i])], (int(c["qmvpx"][i]), int(c["qmvpy"][i])),
                            [int(x) for x in fx.mvc[i, :2 * nc]], int(c["merange"][i]), int(c["method"][i]), int(c["subme"][i]), rows[qp],
                            (u, v), cw, 0, (cb["px"], cr["px"]), cb["stride"], cb["origin"] + int(c["chromaOffset"][i]))
        exp = (int(c["outx"][i]), int(c["outy"][i]), int(c["cost"][i]))
        assert got == exp, "call %d (%dx%d, subme %d): oracle %s reference %s" % (i, w, h, int(c["subme"][i]), got, exp)


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_oracle_replays_the_diamond_searches_of_a_reference_encode(depth):
    """MotionEstimate::diamondSearch (motion.cpp:631-773), the predictor stage of ThreadedME (search.cpp:355-363): every recorded call, including the
    second loop's positions that COST_MV_X4 offsets twice (xo_diamond_search's header)."""
    from tme_util import DiaFixture
    fx, ora = DiaFixture(depth), Oracle(depth)
    c = fx.col
    assert len(fx) >= 300
    rows, moved, far = {}, 0, 0
    for i in range(len(fx)):
        qp = int(c["qp"][i])
        if qp not in rows:
            rows[qp] = ora.mvcost_row(qp, 1 << 14)
        pl = fx.planes[int(c["plane"][i])]
        w, h = int(c["w"][i]), int(c["h"][i])
        got = ora.diamond(w, h, fx.block(i), w, 0, pl["px"], pl["stride"], pl["origin"] + int(c["blockOffset"][i]),
                          [int(c["mnx"][i]), int(c["mny"][i]), int(c["mxx"][i]), int(c["mxy"][i])], (int(c["mvpx"][i]), int(c["mvpy"][i])), rows[qp])
        exp = (int(c["outx"][i]), int(c["outy"][i]), int(c["cost"][i]))
        assert got == exp, "call %d (%dx%d): oracle %s reference %s" % (i, w, h, got, exp)
        moved += exp[:2] != (0, 0); far += max(abs(exp[0]), abs(exp[1])) > 8
    assert moved > 100 and far > 20          # the second loop (distances 8..64 around a moved centre) is exercised


def test_oracle_replays_the_get_pmv_calls_of_reference_encodes():
    """CUData::getPMV (cudata.cpp:1806-1990): 24,000 distinct recorded calls (threaded-me and regular encodes, P and B pictures, up to three references,
    temporal candidates) -- AMVP candidates and the motion-candidate list must be the reference's."""
    import os
    from tme_util import GOLD
    rows = np.load(os.path.join(GOLD, "amvp.npz"))["calls"]
    ora = Oracle(8)
    assert len(rows) >= 20000
    scaled = 0
    for r in rows:
        amvp, mvc = ora.get_pmv(r[38:92], r[0], r[1], r[2], r[3], r[6:38], r[92], r[93])
        nm = int(r[98])
        assert np.array_equal(amvp, r[94:98]) and len(mvc) == 2 * nm and np.array_equal(mvc, r[99:99 + 2 * nm]), "getPMV: list %d ref %d: oracle %s %s reference %s %s" % (
            r[0], r[1], amvp, mvc, r[94:98], r[99:99 + 2 * nm])
        scaled += r[92] != r[93]
    assert scaled > 1000


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_oracle_replays_select_check_update_mvp(depth):
    """Search::selectMVP, checkBestMVP, updateMVP (search.cpp:2347-2382, 4947-4967) on the records of a --threaded-me encode"""
    from tme_util import MvpSelFixture, u32, lam64
    fx, ora = MvpSelFixture(depth), Oracle(depth)
    assert len(fx.select) >= 2000
    for i, r in enumerate(fx.select):
        pl = fx.planes[int(r[0])]
        idx, _ = ora.select_mvp(int(r[1]), int(r[2]), fx.block(i), pl["px"], pl["stride"], pl["origin"] + int(r[3]), r[4:8], r[8:12])
        assert idx == int(r[13]), "selectMVP call %d (%dx%d): oracle %d reference %d" % (i, r[1], r[2], idx, r[13])
    for r in fx.check:
        got = ora.check_best_mvp(lam64(r[9], r[10]), r[0:4], (r[4], r[5]), int(r[6]), u32(r[7]), u32(r[8]))
        assert got == (int(r[11]), u32(r[12]), u32(r[13]))
    for r in fx.update:
        got = ora.update_mvp(lam64(r[8], r[9]), (r[0], r[1]), (r[2], r[3]), (r[4], r[5]), u32(r[6]), u32(r[7]))
        assert got == (u32(r[10]), u32(r[11]))


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_whole_pu_motion_estimation_calls_replay_to_the_references_medata(depth):
    """SURVEY 8(f1), one level up from the single functions: whole Search::puMotionEstimation calls of ThreadedME's PU stage (search.cpp:226-556; 2Nx2N, 2NxN and
    Nx2N partitions, P and B pictures, two references per list) -- the neighbour records of the CTU's table, AMVP, the choice of the predictor, the lookahead's MV as
    candidate and second search, both searches, the bit / cost bookkeeping, the bidirectional candidate -- composed from the oracle's pieces by tests/tme_pu.py must
    leave the MEData record the reference left (threadedme.h:122-130), including what the second partition of a CU inherits from the first."""
    import tme_pu
    planes, calls = tme_pu.load_fixture(depth)
    be = tme_pu.OracleBackend(Oracle(depth), depth)
    assert len(calls) > 800
    n, kinds = 0, set()
    for ci, call in enumerate(calls):
        c = tme_pu.decode(call)
        outs = tme_pu.replay(c, be, planes, be.dt)
        for pi, o in enumerate(outs):
            e = tme_pu.expected(c, pi)
            assert tme_pu.same(o, e), "call %d partition %d (part %d, %s): glue %s reference %s" % (ci, pi, c["part"], list(c["geo"][pi]), o, e)
            kinds.add((e["ref"][0] >= 0, e["ref"][1] >= 0)); n += 1
    assert n > 1300 and kinds == {(True, False), (False, True), (True, True)}
