"""GPU parity of the lookahead frame-cost batch (x265hip_lookahead_intra_batch / _cost_batch, with the lowres pictures built
on the device by x265hip_frame_init_lowres + x265hip_extend_pic_border) against the restated reference path
(oracle/x265_oracle_la.c, itself pinned to the reference's Lowres / Lookahead classes): every MV, cost and total identical."""
import numpy as np
import pytest

from depths import DEPTHS

import x265hip  # noqa: F401
from x265hip_pkg.frame import FrameApi, LA_TASK
from backends import Oracle
from lookahead_util import (MARGIN_X, MARGIN_Y, Geometry, lookahead_cost_row, lowres_planes_oracle, oracle_frame_cost, oracle_intra, pad_full,
                            synth_clip)

pytestmark = pytest.mark.gpu


def device_lowres(api, ora, frames, g):
    """Lowres::init on the device: full-res padded pictures -> 4 half-pel planes per picture, borders extended"""
    t = api.torch
    n = len(frames)
    full = np.stack([pad_full(ora, f, g) for f in frames])
    d_full = api.to_device(full.reshape(-1))
    d_low = t.zeros(n * 4 * g.plane_elems, dtype=api.pixel_t, device="cuda")
    esz = d_low.element_size()
    import ctypes as C
    for f in range(n):
        src = C.c_void_p(d_full.data_ptr() + (f * full.shape[1] + g.full_stride * MARGIN_Y + MARGIN_X) * esz)
        dst = [C.c_void_p(d_low.data_ptr() + ((f * 4 + k) * g.plane_elems + g.origin) * esz) for k in range(4)]
        api.h.check(api.lib.x265hip_frame_init_lowres(api.stream(), src, C.c_ssize_t(g.full_stride), dst[0], dst[1], dst[2], dst[3],
                                                      C.c_ssize_t(g.stride), g.lw, g.lh))
    api.extend_pic_border(d_low, g.origin, g.stride, g.lw, g.lh, MARGIN_X, MARGIN_Y, n_pictures=4 * n, picture_elems=g.plane_elems)
    return d_low


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,aq,shift", [((192, 144), 0, (3, 2)), ((208, 120), 1, (-6, 4)), ((64, 48), 1, (1, 0)), ((32, 16), 0, (1, 1)), ((16, 48), 1, (0, 1)), ((640, 368), 1, (5, -3))])
def test_lookahead_batch_matches_oracle(depth, size, aq, shift):
    api, ora = FrameApi(depth), Oracle(depth)
    t = api.torch
    W, H = size
    N = 4
    frames = synth_clip(W, H, N, depth, seed=300 + depth + W, shift=shift)
    g = Geometry(W, H)
    d_low = device_lowres(api, ora, frames, g)
    planes = [lowres_planes_oracle(ora, f, g) for f in frames]
    got = d_low.cpu().numpy().view(planes[0].dtype).reshape(N, 4, g.plane_elems)
    for f in range(N):
        assert np.array_equal(got[f], planes[f]), "lowres planes of picture %d" % f
    rng = np.random.default_rng(7 + W)
    inv_q = rng.integers(160, 360, (N, g.ncu)).astype(np.int32) if aq else None
    d_invq = api.to_device(inv_q.reshape(-1)) if aq else None
    # ---- intra ----
    d_ic = t.zeros(N * g.ncu, dtype=t.int32, device="cuda"); d_im = t.zeros(N * g.ncu, dtype=t.uint8, device="cuda")
    d_lc = t.zeros(N * g.ncu, dtype=t.int16, device="cuda"); d_rs = t.full((N * g.hcu,), -7, dtype=t.int32, device="cuda")
    d_sm = t.full((N * 2,), -7, dtype=t.int64, device="cuda")
    api.lookahead_intra_batch(d_low, g.plane_elems, g.stride, g.origin, g.wcu, g.hcu, N, d_invq, d_ic, d_im, d_lc, d_rs, d_sm)
    t.cuda.synchronize()
    intra = [oracle_intra(ora, planes[f], g, inv_q[f] if aq else None) for f in range(N)]
    ic = d_ic.cpu().numpy().reshape(N, g.ncu); im = d_im.cpu().numpy().reshape(N, g.ncu)
    lc = d_lc.cpu().numpy().view(np.uint16).reshape(N, g.ncu); rs = d_rs.cpu().numpy().reshape(N, g.hcu); sm = d_sm.cpu().numpy().reshape(N, 2)
    for f in range(N):
        assert np.array_equal(ic[f], intra[f]["intraCost"]) and np.array_equal(im[f], intra[f]["intraMode"]), "intra costs / modes of picture %d" % f
        assert np.array_equal(lc[f], intra[f]["lowresCosts"]) and np.array_equal(rs[f], intra[f]["rowSatds"]), "intra lowresCosts / rowSatds %d" % f
        assert (int(sm[f][0]), int(sm[f][1])) == (intra[f]["costEst"], intra[f]["costEstAq"])
    # ---- frame costs: one batch of independent estimates, then a second batch that reuses cached list-0 searches ----
    row, half = lookahead_cost_row(ora)
    d_row = api.to_device(row.view(np.int16))
    est = [(0, 1, 1), (0, 2, 2), (0, 3, 3), (0, 1, 2), (1, 2, 3), (0, 1, 3), (2, 3, 3)]
    tasks = np.zeros(len(est), LA_TASK)
    nslot = 0
    for i, (p0, b, p1) in enumerate(est):
        tasks[i]["p0"], tasks[i]["b"], tasks[i]["p1"] = p0, b, p1
        tasks[i]["doSearch"] = (1, 1 if p1 > b else 0)
        tasks[i]["mvSlot"] = (nslot, nslot + 1 if p1 > b else 0)
        nslot += 2 if p1 > b else 1
        tasks[i]["outSlot"] = i
    # second batch: B estimates (0,2,3) and (1,3,... no: reuse the list-0 result the P estimate (0,2,2) left in its slot
    est2 = [(0, 2, 3)]
    tasks2 = np.zeros(len(est2), LA_TASK)
    tasks2[0]["p0"], tasks2[0]["b"], tasks2[0]["p1"] = est2[0]
    tasks2[0]["doSearch"] = (0, 1); tasks2[0]["mvSlot"] = (int(tasks[1]["mvSlot"][0]), nslot); tasks2[0]["outSlot"] = len(est)
    nslot += 1
    nout = len(est) + len(est2)
    d_mvs = t.full((nslot * g.ncu * 2,), 0x7fff, dtype=t.int16, device="cuda"); d_mvc = t.full((nslot * g.ncu,), -1, dtype=t.int32, device="cuda")
    d_lc2 = t.zeros(nout * g.ncu, dtype=t.int16, device="cuda"); d_rs2 = t.full((nout * g.hcu,), -7, dtype=t.int32, device="cuda")
    d_sm2 = t.full((nout * 3,), -7, dtype=t.int64, device="cuda")
    for tk in (tasks, tasks2):
        d_tasks = api.to_device(tk)
        api.lookahead_cost_batch(d_low, g.plane_elems, g.stride, g.origin, g.wcu, g.hcu, d_tasks, len(tk), d_ic, d_invq, d_row, half, d_mvs, d_mvc, d_lc2, d_rs2, d_sm2)
        t.cuda.synchronize()
    mvs = d_mvs.cpu().numpy().reshape(nslot, g.ncu * 2).astype(np.int32); mvc = d_mvc.cpu().numpy().reshape(nslot, g.ncu)
    lc2 = d_lc2.cpu().numpy().view(np.uint16).reshape(nout, g.ncu); rs2 = d_rs2.cpu().numpy().reshape(nout, g.hcu); sm2 = d_sm2.cpu().numpy().reshape(nout, 3)
    results = {}
    for tk in list(tasks) + list(tasks2):
        p0, b, p1 = int(tk["p0"]), int(tk["b"]), int(tk["p1"])
        st = {}
        if not tk["doSearch"][0]:
            prev = results[(0, 2, 2)]
            st["mvs0"], st["mvc0"] = prev["mvs0"].copy(), prev["mvc0"].copy()
        o = oracle_frame_cost(ora, planes[b], planes[p0], planes[p1] if p1 > b else None, g, intra[b]["intraCost"], inv_q[b] if aq else None, st,
                              (int(tk["doSearch"][0]), int(tk["doSearch"][1])))
        results[(p0, b, p1)] = o
        s0, s1, out = int(tk["mvSlot"][0]), int(tk["mvSlot"][1]), int(tk["outSlot"])
        what = "estimate (%d,%d,%d)" % (p0, b, p1)
        assert np.array_equal(mvs[s0], o["mvs0"]) and np.array_equal(mvc[s0], o["mvc0"]), "list-0 MVs / costs of " + what
        if p1 > b:
            assert np.array_equal(mvs[s1], o["mvs1"]) and np.array_equal(mvc[s1], o["mvc1"]), "list-1 MVs / costs of " + what
        assert np.array_equal(lc2[out], o["lowresCosts"]), "lowresCosts of " + what
        assert np.array_equal(rs2[out], o["rowSatds"]), "rowSatds of " + what
        assert [int(v) for v in sm2[out]] == [o["costEst"], o["costEstAq"], o["intraMbs"]], "totals of " + what


@pytest.mark.parametrize("depth", DEPTHS)
def test_cutree_propagate_matches_oracle(depth):
    """x265hip_cutree_propagate on the arrays the lookahead batch left in HBM against the oracle's estimateCUPropagate (pinned to the reference)"""
    from x265hip_pkg.lookahead import LookaheadBatch
    from lookahead_util import oracle_propagate
    ora = Oracle(depth)
    W, H, N = 208, 136, 4
    frames = synth_clip(W, H, N, depth, seed=77 + depth, shift=(-7, 5))
    est = [(0, 1, 1), (0, 2, 3), (1, 2, 3), (0, 3, 3), (0, 1, 3)]
    lb = LookaheadBatch(depth, W, H, N, len(est))
    t, g = lb.t, lb.g
    rng = np.random.default_rng(5 + depth)
    inv_q = rng.integers(160, 360, (N, g.ncu)).astype(np.int32)
    lb.d_invq = lb.api.to_device(inv_q.reshape(-1))
    lb.upload(frames); lb.build_lowres(); lb.intra(); lb.set_estimates(est); lb.costs(); t.cuda.synchronize()
    ic = lb.d_intra_cost.cpu().numpy().reshape(N, g.ncu)
    lc = lb.d_lc.cpu().numpy().view(np.uint16).reshape(-1, g.ncu)
    mvs = lb.d_mvs.cpu().numpy().reshape(-1, 2 * g.ncu).astype(np.int32)
    ws = t.zeros(2 * g.ncu, dtype=t.int64, device="cuda")
    for i, (p0, b, p1) in enumerate(est):
        for referenced, fps in ((1, 0.8), (0, 1.0)):
            prop = np.where(rng.random((N, g.ncu)) < 0.06, rng.integers(65000, 65536, (N, g.ncu)), rng.integers(0, 6000, (N, g.ncu))).astype(np.uint16)
            d_prop = lb.api.to_device(prop.view(np.int16).reshape(-1)).view(N, g.ncu)
            lb.api.cutree_propagate(g.wcu, g.hcu, b - p0, p1 - b, 1, fps, referenced, lb.d_intra_cost[b * g.ncu:(b + 1) * g.ncu], lb.d_lc[i * g.ncu:(i + 1) * g.ncu],
                                    lb.d_invq[b * g.ncu:(b + 1) * g.ncu], lb.d_mvs[2 * i * 2 * g.ncu:(2 * i + 1) * 2 * g.ncu],
                                    lb.d_mvs[(2 * i + 1) * 2 * g.ncu:(2 * i + 2) * 2 * g.ncu] if p1 > b else None,
                                    d_prop[b], d_prop[p0], d_prop[p1] if p1 > b else None, ws)
            t.cuda.synchronize()
            got = d_prop.cpu().numpy().view(np.uint16)
            pb = prop[b].copy()
            exp = oracle_propagate(ora, g, b - p0, p1 - b, 1, fps, referenced, ic[b], lc[i].astype(np.int32), inv_q[b], mvs[2 * i], mvs[2 * i + 1] if p1 > b else None,
                                   pb, prop[p0].copy(), prop[p1].copy() if p1 > b else pb)
            assert np.array_equal(got[b], exp[0]) and np.array_equal(got[p0], exp[1]), "cuTree step of %s (referenced %d)" % ((p0, b, p1), referenced)
            if p1 > b:
                assert np.array_equal(got[p1], exp[2]), "cuTree step of %s: list-1 reference" % ((p0, b, p1),)
            untouched = [f for f in range(N) if f not in (p0, b, p1)]
            assert all(np.array_equal(got[f], prop[f]) for f in untouched)


@pytest.mark.parametrize("depth", DEPTHS)
def test_lookahead_on_extreme_pictures(depth):
    """lowres pictures of 0 / PIXEL_MAX blocks (every 8x8 difference 0 or the largest of the depth): the packed SATD of the search and of the finish stage at the limit of
    its 16-bit lanes (a 4x4 coefficient of 16 * PIXEL_MAX)"""
    api, ora = FrameApi(depth), Oracle(depth)
    t = api.torch
    W, H, N = 208, 136, 3
    rng = np.random.default_rng(1234 + depth)
    pm = (1 << depth) - 1
    dt = np.uint8 if depth == 8 else np.uint16
    frames = [(np.kron(rng.integers(0, 2, (H // 16 + 1, W // 16 + 1)), np.ones((16, 16), np.int64))[:H, :W] * pm).astype(dt) for _ in range(N)]
    frames[2] = (pm - frames[1]).astype(dt)                                       # the inverse of its reference
    g = Geometry(W, H)
    planes = [lowres_planes_oracle(ora, f, g) for f in frames]
    d_low = api.to_device(np.stack(planes).reshape(-1))
    intra = [oracle_intra(ora, planes[f], g, None) for f in range(N)]
    d_ic = api.to_device(np.stack([i["intraCost"] for i in intra]).reshape(-1))
    row, half = lookahead_cost_row(ora)
    d_row = api.to_device(row.view(np.int16))
    est = [(0, 1, 1), (0, 1, 2), (1, 2, 2)]
    tasks = np.zeros(len(est), LA_TASK)
    for i, (p0, b, p1) in enumerate(est):
        tasks[i]["p0"], tasks[i]["b"], tasks[i]["p1"] = p0, b, p1
        tasks[i]["doSearch"] = (1, 1 if p1 > b else 0); tasks[i]["mvSlot"] = (2 * i, 2 * i + 1); tasks[i]["outSlot"] = i
    d_tasks = api.to_device(tasks)
    d_mvs = t.zeros(2 * len(est) * g.ncu * 2, dtype=t.int16, device="cuda"); d_mvc = t.zeros(2 * len(est) * g.ncu, dtype=t.int32, device="cuda")
    d_lc = t.zeros(len(est) * g.ncu, dtype=t.int16, device="cuda"); d_rs = t.zeros(len(est) * g.hcu, dtype=t.int32, device="cuda")
    d_sm = t.zeros(len(est) * 3, dtype=t.int64, device="cuda")
    api.lookahead_cost_batch(d_low, g.plane_elems, g.stride, g.origin, g.wcu, g.hcu, d_tasks, len(est), d_ic, None, d_row, half, d_mvs, d_mvc, d_lc, d_rs, d_sm)
    t.cuda.synchronize()
    mvs = d_mvs.cpu().numpy().reshape(-1, g.ncu * 2).astype(np.int32); mvc = d_mvc.cpu().numpy().reshape(-1, g.ncu)
    lc = d_lc.cpu().numpy().view(np.uint16).reshape(-1, g.ncu); sm = d_sm.cpu().numpy().reshape(-1, 3)
    for i, (p0, b, p1) in enumerate(est):
        o = oracle_frame_cost(ora, planes[b], planes[p0], planes[p1] if p1 > b else None, g, intra[b]["intraCost"], None, {}, (1, 1))
        assert np.array_equal(mvs[2 * i], o["mvs0"]) and np.array_equal(mvc[2 * i], o["mvc0"]) and np.array_equal(lc[i], o["lowresCosts"]), "estimate %s" % (est[i],)
        if p1 > b:
            assert np.array_equal(mvs[2 * i + 1], o["mvs1"]) and np.array_equal(mvc[2 * i + 1], o["mvc1"])
        assert [int(v) for v in sm[i]] == [o["costEst"], o["costEstAq"], o["intraMbs"]]


@pytest.mark.parametrize("depth", DEPTHS)
def test_lookahead_weighted_list0_reference(depth):
    """x265hip_la_task.weighted0: list 0 searched in a weighted copy of p0 (an extra picture of the lowres buffer), the bidirectional average in p0 itself"""
    api, ora = FrameApi(depth), Oracle(depth)
    t = api.torch
    W, H, N = 208, 136, 3
    frames = synth_clip(W, H, N, depth, seed=11 + depth, shift=(4, -2))
    g = Geometry(W, H)
    planes = [lowres_planes_oracle(ora, f, g) for f in frames]
    pm = (1 << depth) - 1
    wplanes = np.clip(planes[0].astype(np.int64) * 3 // 4 + 9 * (1 << (depth - 8)), 0, pm).astype(planes[0].dtype)     # stands for weight_pp(scale 96/128, offset 9)
    allp = np.stack(planes + [wplanes])                                                                                  # picture N = the weighted copy of picture 0
    d_low = api.to_device(allp.reshape(-1))
    intra = [oracle_intra(ora, planes[f], g, None) for f in range(N)]
    d_ic = api.to_device(np.stack([i["intraCost"] for i in intra]).reshape(-1))
    row, half = lookahead_cost_row(ora)
    d_row = api.to_device(row.view(np.int16))
    est = [(0, 1, 1), (0, 1, 2), (0, 2, 2)]
    tasks = np.zeros(len(est), LA_TASK)
    for i, (p0, b, p1) in enumerate(est):
        tasks[i]["p0"], tasks[i]["b"], tasks[i]["p1"] = p0, b, p1
        tasks[i]["doSearch"] = (1, 1 if p1 > b else 0); tasks[i]["mvSlot"] = (2 * i, 2 * i + 1); tasks[i]["outSlot"] = i
        tasks[i]["weighted0"] = N + 1
    d_tasks = api.to_device(tasks)
    d_mvs = t.zeros(2 * len(est) * g.ncu * 2, dtype=t.int16, device="cuda"); d_mvc = t.zeros(2 * len(est) * g.ncu, dtype=t.int32, device="cuda")
    d_lc = t.zeros(len(est) * g.ncu, dtype=t.int16, device="cuda"); d_rs = t.zeros(len(est) * g.hcu, dtype=t.int32, device="cuda")
    d_sm = t.zeros(len(est) * 3, dtype=t.int64, device="cuda")
    api.lookahead_cost_batch(d_low, g.plane_elems, g.stride, g.origin, g.wcu, g.hcu, d_tasks, len(est), d_ic, None, d_row, half, d_mvs, d_mvc, d_lc, d_rs, d_sm)
    t.cuda.synchronize()
    mvs = d_mvs.cpu().numpy().reshape(-1, g.ncu * 2).astype(np.int32); mvc = d_mvc.cpu().numpy().reshape(-1, g.ncu)
    lc = d_lc.cpu().numpy().view(np.uint16).reshape(-1, g.ncu); sm = d_sm.cpu().numpy().reshape(-1, 3)
    for i, (p0, b, p1) in enumerate(est):
        o = oracle_frame_cost(ora, planes[b], planes[p0], planes[p1] if p1 > b else None, g, intra[b]["intraCost"], None, {}, (1, 1), ref0w_planes=wplanes)
        plain = oracle_frame_cost(ora, planes[b], planes[p0], planes[p1] if p1 > b else None, g, intra[b]["intraCost"], None, {}, (1, 1))
        assert not np.array_equal(o["mvc0"], plain["mvc0"])                      # the weighted copy really changes the search
        assert np.array_equal(mvs[2 * i], o["mvs0"]) and np.array_equal(mvc[2 * i], o["mvc0"]) and np.array_equal(lc[i], o["lowresCosts"])
        if p1 > b:
            assert np.array_equal(mvs[2 * i + 1], o["mvs1"]) and np.array_equal(mvc[2 * i + 1], o["mvc1"])
        assert [int(v) for v in sm[i]] == [o["costEst"], o["costEstAq"], o["intraMbs"]]


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,rows", [((208, 184), 5), ((320, 256), 10), ((192, 144), 3), ((192, 144), 9)])
def test_lookahead_slices_match_oracle(depth, size, rows):
    """cooperative lookahead slices (rowsPerSlice): every slice swept by its own workgroup"""
    from x265hip_pkg.lookahead import LookaheadBatch
    ora = Oracle(depth)
    W, H = size
    N = 3
    frames = synth_clip(W, H, N, depth, seed=21 + depth + W + rows, shift=(-3, 5))
    est = [(0, 1, 1), (0, 1, 2), (0, 2, 2), (1, 2, 2)]
    lb = LookaheadBatch(depth, W, H, N, len(est))
    t, g = lb.t, lb.g
    lb.upload(frames); lb.build_lowres(); lb.intra(); lb.set_estimates(est); lb.costs(rows_per_slice=rows); t.cuda.synchronize()
    planes = [lowres_planes_oracle(ora, f, g) for f in frames]
    ic = lb.d_intra_cost.cpu().numpy().reshape(N, g.ncu)
    mvs = lb.d_mvs.cpu().numpy().reshape(-1, 2 * g.ncu).astype(np.int32); mvc = lb.d_mv_costs.cpu().numpy().reshape(-1, g.ncu)
    lc = lb.d_lc.cpu().numpy().view(np.uint16).reshape(-1, g.ncu); rs = lb.d_rows.cpu().numpy().reshape(-1, g.hcu); sm = lb.d_sums.cpu().numpy().reshape(-1, 3)
    for i, (p0, b, p1) in enumerate(est):
        o = oracle_frame_cost(ora, planes[b], planes[p0], planes[p1] if p1 > b else None, g, ic[b], None, {}, (1, 1), rows_per_slice=rows)
        assert np.array_equal(mvs[2 * i], o["mvs0"]) and np.array_equal(mvc[2 * i], o["mvc0"]), "list 0 of %s" % ((p0, b, p1),)
        if p1 > b:
            assert np.array_equal(mvs[2 * i + 1], o["mvs1"]) and np.array_equal(mvc[2 * i + 1], o["mvc1"])
        assert np.array_equal(lc[i], o["lowresCosts"]) and np.array_equal(rs[i], o["rowSatds"])
        assert [int(v) for v in sm[i]] == [o["costEst"], o["costEstAq"], o["intraMbs"]]


@pytest.mark.parametrize("depth", DEPTHS)
def test_cutree_finish_matches_oracle(depth):
    """x265hip_cutree_finish (Lookahead::cuTreeFinish) against the oracle, which equals the reference's doubles exactly on the CPU (test_lookahead_oracle_vs_ref.py).
    The integer part is exact; the device's log2 is allowed 1e-12 against the host's (stated tolerance: float accounting, the one place it applies in this path)."""
    import ctypes as C
    from x265hip_pkg.frame import FrameApi
    api, ora = FrameApi(depth), Oracle(depth)
    t = api.torch
    rng = np.random.default_rng(31 + depth)
    ncu = 34 * 60 + 7
    ic = rng.integers(0, 20000, ncu).astype(np.int32); ic[rng.random(ncu) < 0.05] = 0          # blocks without intra cost are left alone
    iq = rng.integers(100, 400, ncu).astype(np.int32)
    prop = np.where(rng.random(ncu) < 0.1, rng.integers(60000, 65536, ncu), rng.integers(0, 8000, ncu)).astype(np.uint16)
    aq = rng.normal(0, 2.0, ncu)
    P = lambda x: C.c_void_p(x.data_ptr())
    for fps, wcd, dist, strength in ((12, 0.26, 1, 2.0), (320, 0.0, 2, 2.0), (256, 0.4, 0, 1.5)):
        d_out = t.full((ncu,), -777.0, dtype=t.float64, device="cuda")
        d_ic, d_iq, d_prop, d_aq = api.to_device(ic), api.to_device(iq), api.to_device(prop.view(np.int16)), api.to_device(aq)      # kept alive until the launch has run
        api.h.check(api.lib.x265hip_cutree_finish(api.stream(), ncu, P(d_ic), P(d_iq), P(d_prop), P(d_aq), fps, C.c_double(wcd), dist, C.c_double(strength), P(d_out)))
        t.cuda.synchronize()
        exp = np.full(ncu, -777.0)
        wd = 1.0 - wcd if dist and wcd > 0 else 0.0
        ora.me_lib.xo_cutree_finish(ncu, C.c_void_p(ic.ctypes.data), C.c_void_p(iq.ctypes.data), C.c_void_p(prop.ctypes.data), C.c_void_p(aq.ctypes.data), fps,
                                    C.c_double(wd), C.c_double(strength), C.c_void_p(exp.ctypes.data))
        got = d_out.cpu().numpy()
        assert np.array_equal(got == -777.0, exp == -777.0) and (exp == -777.0).sum() > 20
        assert np.max(np.abs(got - exp)) <= 1e-12, np.max(np.abs(got - exp))


# --hme: (method of the quarter-resolution level, method of the half-resolution level, their ranges); 0 = diamond, 1 = hexagon, 2 = uneven multi-hexagon, 3 = star, 5 = exhaustive (the reference's default: hex, umh, 16, 32)
@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("size,aq,shift,hme", [((192, 144), 0, (3, 2), (1, 2, 16, 32)), ((208, 120), 1, (-6, 4), (2, 2, 16, 32)), ((320, 176), 1, (12, -8), (1, 1, 8, 12)),
                                               ((136, 72), 0, (-5, 9), (2, 1, 24, 48)), ((64, 48), 1, (1, 0), (1, 2, 16, 32)), ((960, 544), 1, (14, -6), (1, 2, 16, 32)),
                                               # diamond (0) and exhaustive (5) levels
                                               ((192, 144), 1, (3, -2), (0, 5, 16, 6)), ((208, 120), 0, (-6, 4), (5, 0, 5, 32)), ((136, 72), 1, (2, 3), (0, 0, 16, 32)), ((320, 176), 0, (9, -4), (5, 2, 4, 24)),
                                               # star (3) levels; the large shifts send blocks through the raster refinement (whole-picture window, one mv cost in four at the doubled vector)
                                               ((192, 144), 1, (3, -2), (3, 1, 16, 32)), ((208, 120), 0, (-6, 4), (1, 3, 16, 32)), ((320, 176), 1, (44, -28), (3, 3, 8, 12)),
                                               ((136, 72), 0, (-25, 19), (3, 2, 24, 48)), ((256, 160), 0, (36, 30), (0, 3, 16, 32)), ((960, 544), 1, (50, -38), (3, 3, 16, 32))])
def test_hme_lookahead_batch_matches_oracle(depth, size, aq, shift, hme):
    """x265hip_lookahead_cost_batch_hme against the restated --hme sweep (oracle xo_lowres_frame_cost_hme, pinned to the reference by test_lookahead_oracle_vs_ref.py): the
    quarter-resolution MVs / costs (Lowres::lowerResMvs / lowerResMvCosts) and everything the half-resolution sweep leaves, identical"""
    from lookahead_util import lowerres_planes_oracle
    api, ora = FrameApi(depth), Oracle(depth)
    t = api.torch
    W, H = size
    N = 4
    frames = synth_clip(W, H, N, depth, seed=700 + depth + W, shift=shift)
    g = Geometry(W, H)
    planes = [lowres_planes_oracle(ora, f, g) for f in frames]
    lower = [lowerres_planes_oracle(ora, p, g) for p in planes]
    view = np.uint8 if depth == 8 else np.int16
    d_low = api.to_device(np.stack(planes).reshape(-1).view(view)); d_low4 = api.to_device(np.stack(lower).reshape(-1).view(view))
    rng = np.random.default_rng(17 + W)
    inv_q = rng.integers(160, 360, (N, g.ncu)).astype(np.int32) if aq else None
    d_invq = api.to_device(inv_q.reshape(-1)) if aq else None
    intra = [oracle_intra(ora, planes[f], g, inv_q[f] if aq else None) for f in range(N)]
    d_ic = api.to_device(np.stack([it["intraCost"] for it in intra]).reshape(-1))
    row, half = lookahead_cost_row(ora)
    d_row = api.to_device(row.view(np.int16))
    est = [(0, 1, 1), (0, 2, 2), (0, 1, 2), (1, 2, 3), (0, 1, 3), (2, 3, 3)]
    tasks = np.zeros(len(est), LA_TASK)
    nslot = 0
    for i, (p0, b, p1) in enumerate(est):
        tasks[i]["p0"], tasks[i]["b"], tasks[i]["p1"] = p0, b, p1
        tasks[i]["doSearch"] = (1, 1 if p1 > b else 0)
        tasks[i]["mvSlot"] = (nslot, nslot + 1 if p1 > b else 0)
        nslot += 2 if p1 > b else 1
        tasks[i]["outSlot"] = i
    est2 = [(0, 2, 3)]                                  # list 0 (and its quarter-resolution MVs) as the P estimate (0,2,2) left them
    tasks2 = np.zeros(1, LA_TASK)
    tasks2[0]["p0"], tasks2[0]["b"], tasks2[0]["p1"] = est2[0]
    tasks2[0]["doSearch"] = (0, 1); tasks2[0]["mvSlot"] = (int(tasks[1]["mvSlot"][0]), nslot); tasks2[0]["outSlot"] = len(est)
    nslot += 1
    nout = len(est) + 1
    d_mvs = t.full((nslot * g.ncu * 2,), 0x7fff, dtype=t.int16, device="cuda"); d_mvc = t.full((nslot * g.ncu,), -1, dtype=t.int32, device="cuda")
    d_mvs4 = t.full((nslot * g.ncu4 * 2,), 0x7fff, dtype=t.int16, device="cuda"); d_mvc4 = t.full((nslot * g.ncu4,), -1, dtype=t.int32, device="cuda")
    d_lc = t.zeros(nout * g.ncu, dtype=t.int16, device="cuda"); d_rs = t.full((nout * g.hcu,), -7, dtype=t.int32, device="cuda"); d_sm = t.full((nout * 3,), -7, dtype=t.int64, device="cuda")
    for tk in (tasks, tasks2):
        d_tasks = api.to_device(tk)
        api.lookahead_cost_batch_hme(d_low, g.plane_elems, g.stride, g.origin, g.wcu, g.hcu, d_tasks, len(tk), d_ic, d_invq, d_row, half, d_mvs, d_mvc, d_lc, d_rs, d_sm,
                                     d_low4, g.plane_elems4, g.stride4, g.origin4, g.wcu4, g.hcu4, hme[:2], hme[2:], d_mvs4, d_mvc4)
        t.cuda.synchronize()
    mvs = d_mvs.cpu().numpy().reshape(nslot, g.ncu * 2).astype(np.int32); mvc = d_mvc.cpu().numpy().reshape(nslot, g.ncu)
    mvs4 = d_mvs4.cpu().numpy().reshape(nslot, g.ncu4 * 2).astype(np.int32); mvc4 = d_mvc4.cpu().numpy().reshape(nslot, g.ncu4)
    lc = d_lc.cpu().numpy().view(np.uint16).reshape(nout, g.ncu); rs = d_rs.cpu().numpy().reshape(nout, g.hcu); sm = d_sm.cpu().numpy().reshape(nout, 3)
    results = {}
    for tk in list(tasks) + list(tasks2):
        p0, b, p1 = int(tk["p0"]), int(tk["b"]), int(tk["p1"])
        do = (int(tk["doSearch"][0]), int(tk["doSearch"][1]))
        st = {}
        if not do[0]:
            st["mvs0"], st["mvc0"] = results[(0, 2, 2)]["mvs0"].copy(), results[(0, 2, 2)]["mvc0"].copy()
        o = oracle_frame_cost(ora, planes[b], planes[p0], planes[p1] if p1 > b else None, g, intra[b]["intraCost"], inv_q[b] if aq else None, st, do,
                              hme=dict(fenc=lower[b], ref0=lower[p0], ref1=lower[p1] if p1 > b else None, method=hme[:2], range=hme[2:]))
        results[(p0, b, p1)] = o
        what = "estimate (%d,%d,%d)" % (p0, b, p1)
        for l in range(2 if p1 > b else 1):
            s = int(tk["mvSlot"][l])
            if do[l]:
                assert np.array_equal(mvs4[s], o["lmvs%d" % l]) and np.array_equal(mvc4[s], o["lmvc%d" % l]), "quarter-resolution list %d of %s" % (l, what)
            assert np.array_equal(mvs[s], o["mvs%d" % l]) and np.array_equal(mvc[s], o["mvc%d" % l]), "list-%d MVs / costs of %s" % (l, what)
        out = int(tk["outSlot"])
        assert np.array_equal(lc[out], o["lowresCosts"]) and np.array_equal(rs[out], o["rowSatds"]), "lowresCosts / rowSatds of " + what
        assert [int(v) for v in sm[out]] == [o["costEst"], o["costEstAq"], o["intraMbs"]], "totals of " + what


@pytest.mark.parametrize("depth", DEPTHS)
def test_la_host_cutree_step_matches_oracle(depth):
    """x265hip_la_cutree_propagate (the host form integration/lookahead_adapter.cpp binds as Lookahead::estimateCUPropagate): host arrays in, the two references' accumulated
    costs back -- P and B steps, referenced or not, saturating cells, MVs that leave the picture -- against the oracle's estimateCUPropagate (pinned to the reference)."""
    import ctypes as C
    import x265hip
    from lookahead_util import oracle_propagate

    class Desc(C.Structure):
        _fields_ = [("distP0", C.c_int), ("distP1", C.c_int), ("weightedBiPred", C.c_int), ("referenced", C.c_int), ("fpsFactor", C.c_double)] + \
                   [(k, C.c_void_p) for k in ("intraCost", "lowresCosts", "invQscale", "mvs0", "mvs1", "propB", "prop0", "prop1")]

    lib = x265hip.HipLib(depth, fill_table=False).lib
    ora = Oracle(depth)
    g = Geometry(208, 136)
    ctx, la = C.c_void_p(), C.c_void_p()
    assert lib.x265hip_ctx_create(0, C.byref(ctx)) == 0
    lib.x265hip_la_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
    assert lib.x265hip_la_create(ctx, g.wcu, g.hcu, g.stride, g.plane_elems, g.origin, 4, C.byref(la)) == 0
    lib.x265hip_la_cutree_propagate.argtypes = [C.c_void_p, C.c_void_p]
    lib.x265hip_la_destroy.argtypes = [C.c_void_p]; lib.x265hip_ctx_destroy.argtypes = [C.c_void_p]
    rng = np.random.default_rng(31 + depth)
    P = lambda a: a.ctypes.data  # noqa: E731
    try:
        for (d0, d1, referenced, wb, fps) in [(1, 0, 1, 0, 1.0), (2, 1, 0, 1, 0.8), (1, 2, 1, 1, 1.25), (3, 0, 1, 1, 0.5), (1, 1, 0, 0, 1.0)]:
            intra = rng.integers(0, 4000, g.ncu).astype(np.int32); intra[rng.random(g.ncu) < 0.05] = 0
            lists = rng.integers(1, 4, g.ncu) if d1 else np.ones(g.ncu, np.int64)
            lc = (np.minimum(rng.integers(0, 5000, g.ncu), 16383) | (lists << 14)).astype(np.uint16)
            invq = rng.integers(160, 360, g.ncu).astype(np.int32)
            mv = [np.where(rng.random((g.ncu, 2)) < 0.3, 0, rng.integers(-300, 301, (g.ncu, 2))).astype(np.int16) for _ in range(2)]
            prop = [np.where(rng.random(g.ncu) < 0.05, rng.integers(65000, 65536, g.ncu), rng.integers(0, 6000, g.ncu)).astype(np.uint16) for _ in range(3)]
            if not referenced:
                prop[0][:g.wcu] = 0                                  # the caller's memset of the first row (slicetype.cpp:3866-3867)
            exp = oracle_propagate(ora, g, d0, d1, wb, fps, referenced, intra, lc.astype(np.int32), invq, mv[0].reshape(-1).astype(np.int32),
                                   mv[1].reshape(-1).astype(np.int32) if d1 else None, prop[0], prop[1].copy(), prop[2].copy() if d1 else prop[0])
            got = [a.copy() for a in prop]
            d = Desc(d0, d1, wb, referenced, fps, P(intra), P(lc), P(invq), P(mv[0]), P(mv[1]) if d1 else None, P(got[0]), P(got[1]), P(got[2]) if d1 else None)
            rc = lib.x265hip_la_cutree_propagate(la, C.byref(d))
            assert rc == 0, lib.x265hip_last_error()
            assert np.array_equal(got[0], prop[0]), "picture b's own costs are read, not written"
            assert np.array_equal(got[1], exp[1]), "list-0 reference of step %s" % ((d0, d1, referenced),)
            if d1:
                assert np.array_equal(got[2], exp[2]), "list-1 reference of step %s" % ((d0, d1, referenced),)
            else:
                assert np.array_equal(got[2], prop[2])
            assert not np.array_equal(got[1], prop[1])
        bad = Desc(0, 0, 0, 1, 1.0, P(intra), P(lc), P(invq), P(mv[0]), None, P(got[0]), P(got[1]), None)
        assert lib.x265hip_la_cutree_propagate(la, C.byref(bad)) != 0                       # distP0 < 1
    finally:
        lib.x265hip_la_destroy(la); lib.x265hip_ctx_destroy(ctx)
