"""Lookahead frame costs against the committed outputs of the REFERENCE's own classes (tests/golden/lookahead_*.npz, made by
tests/make_golden_lookahead.py): the oracle restatement on the CPU, and the HIP batch on the GPU -- no reference needed at run time."""
import os

import numpy as np
import pytest

from depths import GOLDEN_DEPTHS

from backends import Oracle
from lookahead_util import Geometry, lowres_planes_oracle, oracle_frame_cost, oracle_intra

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def clips(depth):
    z = np.load(os.path.join(GOLD, "lookahead_%d.npz" % depth))
    for ci in range(int(z["nclips"])):
        pre = "c%d_" % ci
        yield z, pre, [f for f in z[pre + "frames"]], bool(z[pre + "aq"]), [tuple(int(v) for v in t) for t in z["triples"]]


def replay(z, pre, triples, estimate):
    """walks the estimates in order with the reference's cache rule; estimate(t, doSearch, cached) -> dict of outputs"""
    cache = {}
    for ti, (p0, b, p1, keep) in enumerate(triples):
        if not keep:
            cache = {k: v for k, v in cache.items() if k[0] != b}
        hdr = z[pre + "t%d_hdr" % ti]
        do = (int(hdr[0]), int(hdr[1]))
        assert do == (int((b, 0, b - p0) not in cache), int(p1 > b and (b, 1, p1 - b) not in cache))
        st = {}
        if not do[0]:
            st["mvs0"], st["mvc0"] = (a.copy() for a in cache[(b, 0, b - p0)])
        if p1 > b and not do[1]:
            st["mvs1"], st["mvc1"] = (a.copy() for a in cache[(b, 1, p1 - b)])
        o = estimate((p0, b, p1), do, st)
        for k in ("mvs0", "mvc0", "lowresCosts", "rowSatds") + (("mvs1", "mvc1") if p1 > b else ()):
            assert np.array_equal(o[k], z[pre + "t%d_%s" % (ti, k)]), "%s of estimate %s" % (k, (p0, b, p1))
        norm = o["costEst"] * 100 // 130 if p1 > b else o["costEst"]
        assert (norm, o["costEstAq"]) == (int(hdr[2]), int(hdr[3])), "totals of estimate %s" % ((p0, b, p1),)
        if p1 == b:
            assert o["intraMbs"] == int(hdr[4])
        cache[(b, 0, b - p0)] = (o["mvs0"], o["mvc0"])
        if p1 > b:
            cache[(b, 1, p1 - b)] = (o["mvs1"], o["mvc1"])


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_oracle_matches_golden(depth):
    ora = Oracle(depth)
    for z, pre, frames, aq, triples in clips(depth):
        H, W = frames[0].shape
        g = Geometry(W, H)
        planes = [lowres_planes_oracle(ora, f, g) for f in frames]
        inv_q = z[pre + "invQ"].astype(np.int32) if aq else None
        intra = [oracle_intra(ora, planes[f], g, inv_q[f] if aq else None) for f in range(len(frames))]
        for k in ("intraCost", "intraMode", "lowresCosts", "rowSatds"):
            assert np.array_equal(np.stack([i[k] for i in intra]), z[pre + "intra_" + k]), "intra " + k
        replay(z, pre, triples, lambda t, do, st: oracle_frame_cost(ora, planes[t[1]], planes[t[0]], planes[t[2]] if t[2] > t[1] else None, g,
                                                                    intra[t[1]]["intraCost"], inv_q[t[1]] if aq else None, st, do))


@pytest.mark.gpu
@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_hip_matches_golden(depth):
    import x265hip  # noqa: F401
    from x265hip_pkg.frame import LA_TASK
    from x265hip_pkg.lookahead import LookaheadBatch
    for z, pre, frames, aq, triples in clips(depth):
        H, W = frames[0].shape
        N = len(frames)
        lb = LookaheadBatch(depth, W, H, N, 4)
        t, g = lb.t, lb.g
        lb.upload(frames); lb.build_lowres()
        if aq:
            lb.d_invq = lb.api.to_device(z[pre + "invQ"].astype(np.int32).reshape(-1))
        lb.intra(); t.cuda.synchronize()
        assert np.array_equal(lb.d_intra_cost.cpu().numpy().reshape(N, g.ncu), z[pre + "intra_intraCost"])
        assert np.array_equal(lb.d_intra_mode.cpu().numpy().reshape(N, g.ncu), z[pre + "intra_intraMode"])
        assert np.array_equal(lb.d_intra_lc.cpu().numpy().view(np.uint16).reshape(N, g.ncu), z[pre + "intra_lowresCosts"])
        assert np.array_equal(lb.d_intra_rows.cpu().numpy().reshape(N, g.hcu), z[pre + "intra_rowSatds"])

        def estimate(tr, do, st):
            tk = np.zeros(1, LA_TASK)
            tk[0]["p0"], tk[0]["b"], tk[0]["p1"] = tr
            tk[0]["doSearch"] = do; tk[0]["mvSlot"] = (0, 1); tk[0]["outSlot"] = 0
            mv = np.zeros((2, g.ncu * 2), np.int16); mc = np.zeros((2, g.ncu), np.int32)
            for l in (0, 1):
                if "mvs%d" % l in st:
                    mv[l] = st["mvs%d" % l].astype(np.int16); mc[l] = st["mvc%d" % l]
            lb.d_mvs[:mv.size] = t.from_numpy(mv.reshape(-1)).cuda(); lb.d_mv_costs[:mc.size] = t.from_numpy(mc.reshape(-1)).cuda()
            lb.n_tasks, lb.tasks_host, lb.d_tasks = 1, tk, lb.api.to_device(tk)
            lb.costs(); t.cuda.synchronize()
            mv = lb.d_mvs[:mv.size].cpu().numpy().reshape(2, -1).astype(np.int32); mc = lb.d_mv_costs[:mc.size].cpu().numpy().reshape(2, -1)
            sm = lb.d_sums.cpu().numpy()[:3]
            return dict(mvs0=mv[0], mvc0=mc[0], mvs1=mv[1], mvc1=mc[1], lowresCosts=lb.d_lc[:g.ncu].cpu().numpy().view(np.uint16).astype(np.int32),
                        rowSatds=lb.d_rows[:g.hcu].cpu().numpy(), costEst=int(sm[0]), costEstAq=int(sm[1]), intraMbs=int(sm[2]))
        replay(z, pre, triples, estimate)
