"""The restated lookahead frame-cost path (oracle/x265_oracle_la.c) against the REAL reference classes
(oracle/_ref/x265la_*: Lowres::init, LookaheadTLD::lowresIntraEstimate, CostEstimateGroup::estimateFrameCost)."""
import ctypes as C

import numpy as np
import pytest

from depths import DEPTHS

from backends import Oracle
from lookahead_util import (Geometry, la_available, lowerres_planes_oracle, lowres_planes_oracle, oracle_frame_cost, oracle_intra, oracle_propagate, run_reference, synth_clip)

# (p0, b, p1, keep): P estimates, B estimates, and one B estimate that reuses the list-0 search a P estimate cached
TRIPLES = [(0, 1, 1, 0), (0, 2, 2, 0), (0, 2, 3, 1), (0, 1, 2, 0), (1, 2, 3, 0), (0, 2, 3, 1), (0, 3, 3, 0), (0, 1, 3, 0)]


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,aq,shift", [((192, 144), 0, (3, 2)), ((208, 120), 1, (-6, 4)), ((64, 48), 1, (1, 0)), ((32, 16), 0, (1, 1)), ((16, 48), 1, (0, 1))])
def test_lookahead_cost_matches_reference(depth, size, aq, shift):
    if not la_available(depth):
        pytest.skip("no reference lookahead binary")
    W, H = size
    frames = synth_clip(W, H, 4, depth, seed=100 + depth + W, shift=shift)
    hdr, ref_frames, ref_triples = run_reference(depth, frames, TRIPLES, aq)
    ora = Oracle(depth)
    g = Geometry(W, H)
    assert (g.stride, g.lw, g.lh, g.wcu, g.hcu) == (hdr["stride"], hdr["lw"], hdr["lh"], hdr["wcu"], hdr["hcu"])
    planes, intra = [], []
    for f, fr in enumerate(frames):
        pl = lowres_planes_oracle(ora, fr, g)
        assert np.array_equal(pl.astype(np.int32), ref_frames[f]["planes"]), "lowres planes of frame %d" % f
        inv_q = ref_frames[f]["invQ"] if aq else None
        it = oracle_intra(ora, pl, g, inv_q)
        for k in ("intraCost", "intraMode", "lowresCosts", "rowSatds"):
            assert np.array_equal(it[k], ref_frames[f][k]), "intra %s of frame %d" % (k, f)
        planes.append(pl); intra.append(it)
    cache = {}                                                  # (b, list, distance) -> (mvs, mvCosts), the reference's per-frame MV caches
    for t, rt in zip(TRIPLES, ref_triples):
        p0, b, p1, keep = t
        if not keep:
            cache = {k: v for k, v in cache.items() if k[0] != b}
        do = tuple(int(v) for v in rt["doSearch"])
        assert do == (int((b, 0, b - p0) not in cache), int(p1 > b and (b, 1, p1 - b) not in cache))
        st = {}
        if not do[0]:
            st["mvs0"], st["mvc0"] = (a.copy() for a in cache[(b, 0, b - p0)])
        if p1 > b and not do[1]:
            st["mvs1"], st["mvc1"] = (a.copy() for a in cache[(b, 1, p1 - b)])
        inv_q = ref_frames[b]["invQ"] if aq else None
        o = oracle_frame_cost(ora, planes[b], planes[p0], planes[p1] if p1 > b else None, g, intra[b]["intraCost"], inv_q, st, do)
        for k in ("mvs0", "mvc0", "lowresCosts", "rowSatds") + (("mvs1", "mvc1") if p1 > b else ()):
            assert np.array_equal(o[k], rt[k]), "%s of estimate %s" % (k, t)
        norm = o["costEst"] * 100 // (130 + 0) if p1 > b else o["costEst"]          # slicetype.cpp:4456-4457, bFrameBias 0
        assert (norm, o["costEstAq"]) == (rt["costEstNorm"], rt["costEstAq"]), "totals of estimate %s" % (t,)
        if p1 == b:                                                              # intraMbs[b - p0] only counts in P estimates (:4619-4620)
            assert o["intraMbs"] == rt["intraMbs"], "intraMbs of estimate %s" % (t,)
        cache[(b, 0, b - p0)] = (o["mvs0"], o["mvc0"])
        if p1 > b:
            cache[(b, 1, p1 - b)] = (o["mvs1"], o["mvc1"])


# ("prop", p0, b, p1, referenced, seed): the estimate, then one cuTree propagation step with pre-filled propagateCost arrays
PROPS = [("prop", 0, 1, 1, 1, 1), ("prop", 0, 2, 3, 1, 2), ("prop", 1, 2, 3, 0, 3), ("prop", 0, 3, 3, 0, 4), ("prop", 0, 1, 3, 1, 5)]


def _keep(a, dt):
    a = np.ascontiguousarray(a, dt); _keep.live.append(a); return C.c_void_p(a.ctypes.data)
_keep.live = []
_P32 = lambda a: _keep(a, np.int32)
_P16 = lambda a: _keep(a, np.uint16)
def _PD(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return C.c_void_p(a.ctypes.data)


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,shift", [((192, 144), (3, 2)), ((208, 120), (-9, 7)), ((64, 48), (1, 0))])
def test_cutree_propagate_matches_reference(depth, size, shift):
    if not la_available(depth):
        pytest.skip("no reference lookahead binary")
    W, H = size
    frames = synth_clip(W, H, 4, depth, seed=500 + depth + W, shift=shift)
    hdr, ref_frames, ref_triples = run_reference(depth, frames, PROPS, 1)
    ora = Oracle(depth)
    g = Geometry(W, H)
    for t, rt in zip(PROPS, ref_triples):
        _, p0, b, p1, referenced, seed = t
        pr = rt["prop"]
        assert pr["referenced"] == referenced
        before = [a.astype(np.uint16) for a in pr["before"]]
        pb = before[0]; a0 = pb if p0 == b else before[1]; a1 = pb if p1 == b else before[2]
        got = oracle_propagate(ora, g, b - p0, p1 - b, pr["weightb"], pr["fpsFactor"], referenced, ref_frames[b]["intraCost"], rt["lowresCosts"],
                               ref_frames[b]["invQ"], rt["mvs0"], rt["mvs1"] if p1 > b else None, pb, a0, a1)
        for k, name in enumerate(("b", "p0", "p1")):
            assert np.array_equal(got[k].astype(np.int32), pr["after"][k]), "propagateCost of %s after %s" % (name, t)
        # ... and Lookahead::cuTreeFinish on picture b with the propagated costs: the qp offsets as doubles, identical
        fin = rt["finish"]
        fps = int(256.0 / pr["fpsFactor"])            # (int)(CLIP_DURATION(averageDuration) / CLIP_DURATION(frame duration) * 256): the inverse of the propagate step's factor
        wd = 1.0 - fin["weightedCostDelta"] if fin["ref0Distance"] and fin["weightedCostDelta"] > 0 else 0.0
        out = np.full(g.ncu, -777.0)
        ora.me_lib.xo_cutree_finish(g.ncu, _P32(ref_frames[b]["intraCost"]), _P32(ref_frames[b]["invQ"]), _P16(pr["after"][0].astype(np.uint16)), _PD(fin["qpAq"]), fps,
                                    C.c_double(wd), C.c_double(fin["strength"]), _PD(out))
        touched = out != -777.0
        assert touched.any() and np.array_equal(out[touched], fin["qpCuTree"][touched]), "cuTreeFinish after %s" % (t,)


def fade_clip(W, H, n, depth, seed):
    """a static rough texture (expensive to predict intra) fading towards a flat grey, scale AND offset changing per picture, plus a little
    noise: LookaheadTLD::weightsAnalyse (slicetype.cpp:919-1020) chooses weights for most reference distances"""
    rng = np.random.default_rng(seed)
    pm = (1 << depth) - 1
    tex = rng.random((H, W))
    tex = (tex + np.roll(tex, 1, 0) + np.roll(tex, 1, 1)) / 3
    out = []
    for f in range(n):
        a = 1.0 - 0.2 * f
        fr = (tex * a + (1 - a) * 0.55) * pm + rng.normal(0, 1.0 * (1 << (depth - 8)), (H, W))
        out.append(np.clip(np.rint(fr), 0, pm).astype(np.uint8 if depth == 8 else np.uint16))
    return out


WP_TRIPLES = [(0, 1, 1, 0), (0, 2, 2, 0), (0, 1, 2, 0), (1, 2, 3, 0), (0, 3, 3, 0), (1, 3, 3, 0)]


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size", [(192, 144), (208, 120)])
def test_lookahead_cost_with_weighted_reference_matches_reference(depth, size):
    """--weightp (on by default): P and B estimates whose list-0 search runs in the weighted copy of p0 the reference built"""
    if not la_available(depth):
        pytest.skip("no reference lookahead binary")
    W, H = size
    frames = fade_clip(W, H, 4, depth, seed=40 + depth + W)
    hdr, ref_frames, ref_triples = run_reference(depth, frames, WP_TRIPLES, 6)      # weightp + the quarter-sum input (see ref_lookahead.cpp)
    assert sum(rt["isWeighted"] for rt in ref_triples) >= 3, "the clip must make the reference choose weights"
    ora = Oracle(depth)
    g = Geometry(W, H)
    planes = [lowres_planes_oracle(ora, fr, g) for fr in frames]
    # with weightp the reference allocates the AQ factor array (all zero without an AQ pass): the AQ-scaled outputs use it as it is
    inv_q = [f["invQ"] if f["invQ"][0] >= 0 else None for f in ref_frames]
    intra = [oracle_intra(ora, pl, g, q) for pl, q in zip(planes, inv_q)]
    for t, rt in zip(WP_TRIPLES, ref_triples):
        p0, b, p1, _ = t
        wpl = rt["wplanes"].astype(planes[0].dtype) if rt["isWeighted"] else None
        o = oracle_frame_cost(ora, planes[b], planes[p0], planes[p1] if p1 > b else None, g, intra[b]["intraCost"], inv_q[b], {}, (1, 1), ref0w_planes=wpl)
        for k in ("mvs0", "mvc0", "lowresCosts", "rowSatds") + (("mvs1", "mvc1") if p1 > b else ()):
            assert np.array_equal(o[k], rt[k]), "%s of estimate %s (weighted %d)" % (k, t, rt["isWeighted"])
        norm = o["costEst"] * 100 // 130 if p1 > b else o["costEst"]
        assert (norm, o["costEstAq"]) == (rt["costEstNorm"], rt["costEstAq"])


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,rows,aq", [((192, 144), 3, 0), ((208, 184), 5, 1), ((320, 256), 10, 1), ((192, 144), 9, 0)])
def test_lookahead_cost_with_slices_matches_reference(depth, size, rows, aq):
    """--lookahead-slices: every slice of `rows` block rows is its own reverse sweep (no predictor crosses its lower edge)"""
    if not la_available(depth):
        pytest.skip("no reference lookahead binary")
    W, H = size
    frames = synth_clip(W, H, 3, depth, seed=60 + depth + W + rows, shift=(3, -4))
    triples = [(0, 1, 1, 0, rows), (0, 1, 2, 0, rows), (0, 2, 2, 0, rows)]
    hdr, ref_frames, ref_triples = run_reference(depth, frames, triples, aq)
    ora = Oracle(depth)
    g = Geometry(W, H)
    planes = [lowres_planes_oracle(ora, fr, g) for fr in frames]
    inv_q = [f["invQ"] if aq else None for f in ref_frames]
    intra = [oracle_intra(ora, pl, g, q) for pl, q in zip(planes, inv_q)]
    for t, rt in zip(triples, ref_triples):
        p0, b, p1 = t[:3]
        o = oracle_frame_cost(ora, planes[b], planes[p0], planes[p1] if p1 > b else None, g, intra[b]["intraCost"], inv_q[b], {}, (1, 1), rows_per_slice=rows)
        whole = oracle_frame_cost(ora, planes[b], planes[p0], planes[p1] if p1 > b else None, g, intra[b]["intraCost"], inv_q[b], {}, (1, 1))
        if g.hcu // rows > 1:
            assert not np.array_equal(o["mvc0"], whole["mvc0"])                 # the slicing changes predictors at the slice edges
        for k in ("mvs0", "mvc0", "lowresCosts", "rowSatds") + (("mvs1", "mvc1") if p1 > b else ()):
            assert np.array_equal(o[k], rt[k]), "%s of estimate %s in slices of %d rows" % (k, t, rows)
        norm = o["costEst"] * 100 // 130 if p1 > b else o["costEst"]
        assert (norm, o["costEstAq"]) == (rt["costEstNorm"], rt["costEstAq"])
        if p1 == b:
            assert o["intraMbs"] == rt["intraMbs"]


# --hme (slicetype.cpp:4430-4439, 4483-4575): (method of the quarter-resolution level, method of the half-resolution level, their ranges); 0 = diamond, 1 = hexagon, 2 = uneven multi-hexagon, 3 = star, 5 = exhaustive
HME_TRIPLES = [(0, 1, 1, 0), (0, 2, 2, 0), (0, 2, 3, 1), (0, 1, 2, 0), (1, 2, 3, 0), (0, 3, 3, 0), (0, 1, 3, 0)]


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,aq,shift,hme", [((192, 144), 0, (3, 2), (1, 2, 16, 32)), ((208, 120), 1, (-6, 4), (2, 2, 16, 32)), ((320, 176), 1, (12, -8), (1, 1, 8, 12)),
                                               ((136, 72), 0, (-5, 9), (2, 1, 24, 48)), ((64, 48), 1, (1, 0), (1, 2, 16, 32)),
                                               # diamond (0) and exhaustive (5) levels: the exhaustive search of an --hme reference is cut to +-range around the ZERO vector (motion.cpp:1598-1605)
                                               ((192, 144), 1, (3, -2), (0, 5, 16, 6)), ((208, 120), 0, (-6, 4), (5, 0, 5, 32)), ((136, 72), 1, (2, 3), (0, 0, 16, 32)),
                                               ((320, 176), 0, (9, -4), (5, 2, 4, 24)), ((64, 48), 1, (1, 0), (1, 5, 16, 3)),
                                               # star (3) levels (motion.cpp:1328-1436, 387-629): the large shifts send blocks through the raster (best distance > 5), whose window is the picture
                                               ((192, 144), 1, (3, -2), (3, 1, 16, 32)), ((208, 120), 0, (-6, 4), (1, 3, 16, 32)), ((320, 176), 1, (44, -28), (3, 3, 8, 12)),
                                               ((136, 72), 0, (-25, 19), (3, 2, 24, 48)), ((256, 160), 0, (36, 30), (0, 3, 16, 32))])
def test_hme_lookahead_cost_matches_reference(depth, size, aq, shift, hme):
    if not la_available(depth):
        pytest.skip("no reference lookahead binary")
    W, H = size
    frames = synth_clip(W, H, 4, depth, seed=900 + depth + W, shift=shift)
    hdr, ref_frames, ref_triples = run_reference(depth, frames, HME_TRIPLES, aq, hme=hme)
    ora = Oracle(depth)
    g = Geometry(W, H)
    assert (g.wcu4, g.hcu4) == (hdr["wcu4"], hdr["hcu4"])
    planes, lower, intra = [], [], []
    for f, fr in enumerate(frames):
        pl = lowres_planes_oracle(ora, fr, g)
        lo = lowerres_planes_oracle(ora, pl, g)
        assert np.array_equal(pl.astype(np.int32), ref_frames[f]["planes"]), "lowres planes of frame %d" % f
        assert np.array_equal(lo.astype(np.int32), ref_frames[f]["lowerPlanes"]), "quarter-resolution planes of frame %d" % f
        planes.append(pl); lower.append(lo)
        intra.append(oracle_intra(ora, pl, g, ref_frames[f]["invQ"] if aq else None))
    cache = {}
    rasters = C.c_long.in_dll(ora.me_lib, "xo_la_star_rasters")
    rasters.value = 0
    for t, rt in zip(HME_TRIPLES, ref_triples):
        p0, b, p1, keep = t
        if not keep:
            cache = {k: v for k, v in cache.items() if k[0] != b}
        do = tuple(int(v) for v in rt["doSearch"])
        st = {}
        if not do[0]:
            st["mvs0"], st["mvc0"] = (a.copy() for a in cache[(b, 0, b - p0)])
        if p1 > b and not do[1]:
            st["mvs1"], st["mvc1"] = (a.copy() for a in cache[(b, 1, p1 - b)])
        o = oracle_frame_cost(ora, planes[b], planes[p0], planes[p1] if p1 > b else None, g, intra[b]["intraCost"], ref_frames[b]["invQ"] if aq else None, st, do,
                              hme=dict(fenc=lower[b], ref0=lower[p0], ref1=lower[p1] if p1 > b else None, method=hme[:2], range=hme[2:]))
        for l in range(2 if p1 > b else 1):
            if do[l]:
                assert np.array_equal(o["lmvs%d" % l], rt["lmvs%d" % l]) and np.array_equal(o["lmvc%d" % l], rt["lmvc%d" % l]), "quarter-resolution list %d of estimate %s" % (l, t)
        for k in ("mvs0", "mvc0", "lowresCosts", "rowSatds") + (("mvs1", "mvc1") if p1 > b else ()):
            assert np.array_equal(o[k], rt[k]), "%s of estimate %s" % (k, t)
        norm = o["costEst"] * 100 // 130 if p1 > b else o["costEst"]
        assert (norm, o["costEstAq"]) == (rt["costEstNorm"], rt["costEstAq"]), "totals of estimate %s" % (t,)
        cache[(b, 0, b - p0)] = (o["mvs0"], o["mvc0"])
        if p1 > b:
            cache[(b, 1, p1 - b)] = (o["mvs1"], o["mvc1"])
    if 3 in hme[:2] and max(abs(shift[0]), abs(shift[1])) >= 19:
        assert rasters.value > 0, "no block of this clip reached the raster refinement of the star search"


def test_reference_cannot_run_a_sea_level_of_hme(tmp_path):
    """--hme-search sea: the lookahead's MotionEstimate never gets integral planes (MotionEstimate::integral[] = NULL, motion.cpp:115; set only by Search::predInterSearch,
    search.cpp), and the SEA search reads them (motion.cpp:1509-1541, 1564) -- the reference encoder dies on its first P estimate.  That is why neither the oracle nor the
    producer has a sea level for --hme: there is no reference behaviour to match."""
    import os, subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "x265e2e_8")
    if not os.path.exists(exe):
        pytest.skip("no oracle/_ref/x265e2e_8 (built where the reference is present)")
    env = dict(os.environ, X265LAGPU="0", X265TME="0", X265TMEGPU="0")
    r = subprocess.run([exe, "none", "960", "544", "3", "superfast", str(tmp_path / "sea.hevc"), "hme=1", "hme-search=sea,hex,hex"], capture_output=True, env=env, timeout=600)
    assert r.returncode < 0, "the reference survived an SEA level of --hme (return code %d): the producer should offer it then" % r.returncode
