"""SURVEY 8(f1) at the data level, GPU side: the task list of a reference encode run with --threaded-me -- every PU of every CTU, with the window,
predictor and up to 12 candidates Search::puMotionEstimation built for it (fixtures from oracle/ref_tme.cpp, no reference needed here) --
through x265hip_me_batch: the MV, the cost and the MV cost must be the reference's own (MEData.mv / mvCost, encoder/threadedme.h:122-130)."""
import numpy as np
import pytest

from depths import DEPTHS, GOLDEN_DEPTHS

import x265hip  # noqa: F401
from x265hip_pkg.frame import FrameApi, ME_TASK, ME_RESULT, mvcost_row
from tme_util import TmeFixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("planes", [True, False])
@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_me_batch_replays_the_threaded_me_task_list(depth, planes):
    fx, api = TmeFixture(depth), FrameApi(depth)
    T = api.torch
    c = fx.col
    half = 1 << 14
    d_planes, d_phase, rows = {}, {}, {}
    for pid, pl in fx.planes.items():
        d_planes[pid] = api.to_device(pl["px"])
        if planes:
            d_phase[pid] = T.zeros(16 * pl["px"].size, dtype=d_planes[pid].dtype, device="cuda")
            api.subpel_planes(d_planes[pid], pl["stride"], pl["rows"], d_phase[pid], pl["px"].size)
    checked = 0
    for (pid, w, h, qp, method, subme, merange), idx in fx.groups().items():
        if not planes and len(idx) > 400:
            idx = idx[::3]                           # the in-kernel interpolation form is slow on thousands of 8x8 PUs; a third of them
        pl = fx.planes[pid]
        n = len(idx)
        t = np.zeros(n, ME_TASK)
        cur = np.concatenate([fx.block(i) for i in idx])
        t["curOff"] = np.arange(n) * (w * h)
        t["refOff"] = pl["origin"] + c["blockOffset"][idx]
        t["mvmin"][:, 0] = c["mnx"][idx]; t["mvmin"][:, 1] = c["mny"][idx]; t["mvmax"][:, 0] = c["mxx"][idx]; t["mvmax"][:, 1] = c["mxy"][idx]
        t["qmvp"][:, 0] = c["qmvpx"][idx]; t["qmvp"][:, 1] = c["qmvpy"][idx]
        t["mvc"] = fx.mvc[idx]; t["numCand"] = c["numCand"][idx]; t["mvpFrom"] = -1
        if qp not in rows:
            rows[qp] = api.to_device(mvcost_row(depth, qp, half).view(np.int16))
        d_t, d_cur = api.to_device(t), api.to_device(cur)
        d_res = T.zeros(n * ME_RESULT.itemsize, dtype=T.uint8, device="cuda")
        api.me_batch(w, h, d_cur, w, d_planes[pid], pl["stride"], d_t, n, rows[qp], half, merange, method, subme, d_res,
                     planes=d_phase.get(pid), plane_elems=pl["px"].size if planes else 0)
        T.cuda.synchronize()
        r = d_res.cpu().numpy().view(ME_RESULT)
        bad = np.nonzero((r["mv"][:, 0] != c["outx"][idx]) | (r["mv"][:, 1] != c["outy"][idx]) | (r["cost"] != c["cost"][idx]) | (r["mvcost"] != c["mvcost"][idx]))[0]
        assert len(bad) == 0, "%dx%d plane %d qp %d: %d of %d tasks differ, first: call %d hip %s reference (%d, %d, %d)" % (
            w, h, pid, qp, len(bad), n, idx[bad[0]], r[bad[0]], c["outx"][idx[bad[0]]], c["outy"][idx[bad[0]]], c["cost"][idx[bad[0]]])
        checked += n
    assert checked > 600


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_me_batch_chroma_replays_the_searches_of_pred_inter_search(depth):
    """The predInterSearch call form: chroma SATD terms in every sub-pel cost (subme 3 / 4, 4:2:0).  Fixtures from a regular reference encode
    (P and B pictures, several references, up to 12 candidates): x265hip_me_batch_chroma must return the reference's MV, cost and MV cost."""
    from tme_util import MecFixture
    fx, api = MecFixture(depth), FrameApi(depth)
    T = api.torch
    c = fx.col
    half = 1 << 14
    d_planes, d_phase, rows = {}, {}, {}
    for pid, pl in fx.planes.items():
        d_planes[pid] = api.to_device(pl["px"])
    checked = 0
    keys = {}
    for i in range(len(fx)):
        keys.setdefault((int(c["plane"][i]), int(c["cbPlane"][i]), int(c["crPlane"][i]), int(c["w"][i]), int(c["h"][i]), int(c["qp"][i]), int(c["method"][i]),
                         int(c["subme"][i]), int(c["merange"][i])), []).append(i)
    for (pid, cbid, crid, w, h, qp, method, subme, merange), idx in keys.items():
        pl, cb = fx.planes[pid], fx.planes[cbid]
        if pid not in d_phase:
            d_phase[pid] = T.zeros(16 * pl["px"].size, dtype=d_planes[pid].dtype, device="cuda")
            api.subpel_planes(d_planes[pid], pl["stride"], pl["rows"], d_phase[pid], pl["px"].size)
        n, cw, ch = len(idx), w // 2, h // 2
        t = np.zeros(n, ME_TASK)
        blocks = [fx.blocks(i) for i in idx]
        cur = np.concatenate([b[0] for b in blocks]); cur_cb = np.concatenate([b[1] for b in blocks]); cur_cr = np.concatenate([b[2] for b in blocks])
        t["curOff"] = np.arange(n) * (w * h)
        t["refOff"] = pl["origin"] + c["blockOffset"][idx]
        t["mvmin"][:, 0] = c["mnx"][idx]; t["mvmin"][:, 1] = c["mny"][idx]; t["mvmax"][:, 0] = c["mxx"][idx]; t["mvmax"][:, 1] = c["mxy"][idx]
        t["qmvp"][:, 0] = c["qmvpx"][idx]; t["qmvp"][:, 1] = c["qmvpy"][idx]
        t["mvc"] = fx.mvc[idx]; t["numCand"] = c["numCand"][idx]; t["mvpFrom"] = -1
        if qp not in rows:
            rows[qp] = api.to_device(mvcost_row(depth, qp, half).view(np.int16))
        d_t, d_cur, d_ccb, d_ccr = api.to_device(t), api.to_device(cur), api.to_device(cur_cb), api.to_device(cur_cr)
        d_co = api.to_device((np.arange(n) * (cw * ch)).astype(np.int32))
        d_ro = api.to_device((cb["origin"] + c["chromaOffset"][idx]).astype(np.int32))
        d_res = T.zeros(n * ME_RESULT.itemsize, dtype=T.uint8, device="cuda")
        api.me_batch_chroma(w, h, d_cur, w, d_planes[pid], pl["stride"], d_t, n, rows[qp], half, merange, method, subme, d_res, d_phase[pid], pl["px"].size,
                            d_ccb, d_ccr, cw, d_planes[cbid], d_planes[crid], cb["stride"], d_co, d_ro)
        T.cuda.synchronize()
        r = d_res.cpu().numpy().view(ME_RESULT)
        bad = np.nonzero((r["mv"][:, 0] != c["outx"][idx]) | (r["mv"][:, 1] != c["outy"][idx]) | (r["cost"] != c["cost"][idx]) | (r["mvcost"] != c["mvcost"][idx]))[0]
        assert len(bad) == 0, "%dx%d method %d subme %d: %d of %d tasks differ, first: call %d hip %s reference (%d, %d, %d)" % (
            w, h, method, subme, len(bad), n, idx[bad[0]], r[bad[0]], c["outx"][idx[bad[0]]], c["outy"][idx[bad[0]]], c["cost"][idx[bad[0]]])
        checked += n
    assert checked == len(fx)


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_diamond_batch_replays_the_predictor_searches_of_threaded_me(depth):
    """MotionEstimate::diamondSearch (motion.cpp:631-773): the recorded calls of ThreadedME's first stage (the CTU and its four sub-CUs, range 32)
    through x265hip_diamond_batch -- full-pel MV and cost must be the reference's; plus synthetic windows cut by the picture edge (the per-point
    branches) against the oracle, which the golden test pins to the same records."""
    from tme_util import DiaFixture
    from backends import Oracle
    fx, api, ora = DiaFixture(depth), FrameApi(depth), Oracle(depth)
    T = api.torch
    c = fx.col
    half = 1 << 14
    rows, rows_h = {}, {}
    checked = 0
    keys = {}
    for i in range(len(fx)):
        keys.setdefault((int(c["plane"][i]), int(c["w"][i]), int(c["h"][i]), int(c["qp"][i])), []).append(i)
    for (pid, w, h, qp), idx in keys.items():
        pl = fx.planes[pid]
        n = len(idx)
        t = np.zeros(n, ME_TASK)
        cur = np.concatenate([fx.block(i) for i in idx])
        t["curOff"] = np.arange(n) * (w * h)
        t["refOff"] = pl["origin"] + c["blockOffset"][idx]
        t["mvmin"][:, 0] = c["mnx"][idx]; t["mvmin"][:, 1] = c["mny"][idx]; t["mvmax"][:, 0] = c["mxx"][idx]; t["mvmax"][:, 1] = c["mxy"][idx]
        t["qmvp"][:, 0] = c["mvpx"][idx]; t["qmvp"][:, 1] = c["mvpy"][idx]
        if qp not in rows:
            rows_h[qp] = mvcost_row(depth, qp, half)
            rows[qp] = api.to_device(rows_h[qp].view(np.int16))
        d_t, d_cur, d_ref = api.to_device(t), api.to_device(cur), api.to_device(pl["px"])
        d_res = T.zeros(n * ME_RESULT.itemsize, dtype=T.uint8, device="cuda")
        api.diamond_batch(w, h, d_cur, w, d_ref, pl["stride"], d_t, n, rows[qp], half, d_res)
        T.cuda.synchronize()
        r = d_res.cpu().numpy().view(ME_RESULT)
        bad = np.nonzero((r["mv"][:, 0] != c["outx"][idx]) | (r["mv"][:, 1] != c["outy"][idx]) | (r["cost"] != c["cost"][idx]))[0]
        assert len(bad) == 0, "%dx%d plane %d: %d of %d differ, first: call %d hip %s reference (%d, %d, %d)" % (
            w, h, pid, len(bad), n, idx[bad[0]], r[bad[0]], c["outx"][idx[bad[0]]], c["outy"][idx[bad[0]]], c["cost"][idx[bad[0]]])
        checked += n
        # the same PUs with narrow / lopsided windows and a non-zero MVD origin: oracle as the expectation
        rng = np.random.default_rng(depth * 100 + w)
        t2 = t.copy()
        for k in range(n):
            t2["mvmin"][k] = (-int(rng.integers(0, 20)), -int(rng.integers(0, 20))); t2["mvmax"][k] = (int(rng.integers(0, 20)), int(rng.integers(0, 20)))
            t2["qmvp"][k] = (int(rng.integers(-40, 41)), int(rng.integers(-40, 41)))
        d_t2 = api.to_device(t2)
        api.diamond_batch(w, h, d_cur, w, d_ref, pl["stride"], d_t2, n, rows[qp], half, d_res)
        T.cuda.synchronize()
        r = d_res.cpu().numpy().view(ME_RESULT)
        for k in range(0, n, 3):
            exp = ora.diamond(w, h, cur, w, int(t2["curOff"][k]), pl["px"], pl["stride"], int(t2["refOff"][k]),
                              [int(t2["mvmin"][k][0]), int(t2["mvmin"][k][1]), int(t2["mvmax"][k][0]), int(t2["mvmax"][k][1])], (int(t2["qmvp"][k][0]), int(t2["qmvp"][k][1])), rows_h[qp])
            got = (int(r["mv"][k][0]), int(r["mv"][k][1]), int(r["cost"][k]))
            assert got == exp, "edge window, task %d: hip %s oracle %s (bounds %s %s)" % (k, got, exp, t2["mvmin"][k], t2["mvmax"][k])
    assert checked >= 300


def test_amvp_batch_replays_the_get_pmv_calls_of_reference_encodes():
    """x265hip_amvp_batch against the recorded CUData::getPMV calls (tests/golden/amvp.npz): both AMVP candidates, numMvc and the candidate list."""
    import os
    from tme_util import GOLD
    from x265hip_pkg.frame import AMVP_TASK, AMVP_RESULT
    rows = np.load(os.path.join(GOLD, "amvp.npz"))["calls"]
    api = FrameApi(8)
    T = api.torch
    # one launch per slice context (current POC, temporal flag, POC lists)
    ctx = {}
    for i, r in enumerate(rows):
        ctx.setdefault((int(r[2]), int(r[3])) + tuple(int(v) for v in r[6:38]), []).append(i)
    checked = 0
    for key, idx in ctx.items():
        n = len(idx)
        t = np.zeros(n, AMVP_TASK)
        R = rows[idx]
        nb = R[:, 38:92].reshape(n, 6, 9)
        t["nb"]["mv"][:, :, 0, 0] = nb[:, :, 0]; t["nb"]["mv"][:, :, 0, 1] = nb[:, :, 1]; t["nb"]["mv"][:, :, 1, 0] = nb[:, :, 2]; t["nb"]["mv"][:, :, 1, 1] = nb[:, :, 3]
        t["nb"]["refIdx"][:, :, 0] = nb[:, :, 4]; t["nb"]["refIdx"][:, :, 1] = nb[:, :, 5]; t["nb"]["available"] = nb[:, :, 8]
        t["list"] = R[:, 0]; t["refIdx"] = R[:, 1]; t["colPOC"] = R[:, 92]; t["colRefPOC"] = R[:, 93]
        d_t = api.to_device(t)
        d_out = T.zeros(n * AMVP_RESULT.itemsize, dtype=T.uint8, device="cuda")
        api.amvp_batch(d_t, n, key[0], key[1], [key[2:18], key[18:34]], d_out)
        T.cuda.synchronize()
        o = d_out.cpu().numpy().view(AMVP_RESULT)
        exp_amvp = R[:, 94:98].reshape(n, 2, 2)
        assert np.array_equal(o["amvp"], exp_amvp), "AMVP candidates differ (POC %d)" % key[0]
        assert np.array_equal(o["numMvc"], R[:, 98])
        exp_mvc = R[:, 99:121].reshape(n, 11, 2)
        assert np.array_equal(o["mvc"], exp_mvc), "candidate lists differ (POC %d)" % key[0]
        checked += n
    assert checked == len(rows) >= 20000


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_select_mvp_and_mvp_bits_replay_the_reference_records(depth):
    """x265hip_select_mvp_batch against the recorded Search::selectMVP calls (index; the two SADs against the oracle's), x265hip_mvp_bits_batch against the
    recorded checkBestMVP and updateMVP calls."""
    from tme_util import MvpSelFixture, u32, lam64
    from backends import Oracle
    from x265hip_pkg.frame import SELECT_TASK, SELECT_RESULT, MVP_BITS
    fx, api, ora = MvpSelFixture(depth), FrameApi(depth), Oracle(depth)
    T = api.torch
    S = fx.select
    groups = {}
    for i, r in enumerate(S):
        groups.setdefault((int(r[0]), int(r[1]), int(r[2])), []).append(i)
    d_phase = {}
    checked = 0
    for (pid, w, h), idx in groups.items():
        pl = fx.planes[pid]
        if pid not in d_phase:
            d_ref = api.to_device(pl["px"])
            d_phase[pid] = T.zeros(16 * pl["px"].size, dtype=d_ref.dtype, device="cuda")
            api.subpel_planes(d_ref, pl["stride"], pl["rows"], d_phase[pid], pl["px"].size)
        n = len(idx)
        t = np.zeros(n, SELECT_TASK)
        cur = np.concatenate([fx.block(i) for i in idx])
        t["curOff"] = np.arange(n) * (w * h); t["refOff"] = pl["origin"] + S[idx, 3]
        t["amvp"] = S[idx, 4:8].reshape(n, 2, 2); t["clip"] = S[idx, 8:12]
        d_t, d_cur = api.to_device(t), api.to_device(cur)
        d_out = T.zeros(n * SELECT_RESULT.itemsize, dtype=T.uint8, device="cuda")
        api.select_mvp_batch(w, h, d_cur, w, d_phase[pid], pl["px"].size, pl["stride"], d_t, n, d_out)
        T.cuda.synchronize()
        o = d_out.cpu().numpy().view(SELECT_RESULT)
        assert np.array_equal(o["mvpIdx"], S[idx, 13]), "selectMVP %dx%d plane %d" % (w, h, pid)
        for k in range(0, n, 7):
            _, costs = ora.select_mvp(w, h, fx.block(idx[k]), pl["px"], pl["stride"], pl["origin"] + int(S[idx[k], 3]), S[idx[k], 4:8], S[idx[k], 8:12])
            assert tuple(o["cost"][k]) == tuple(int(v) for v in costs)
        checked += n
    assert checked == len(S)
    bits_row = np.zeros(2 * 32768 + 1, np.float32)
    api.h.check(api.lib.x265hip_mvbits_row(32768, bits_row.ctypes.data_as(__import__("ctypes").c_void_p)))
    d_bits = api.to_device(bits_row.view(np.int32))
    for rows, upd in ((fx.check, False), (fx.update, True)):
        lams = {}
        for i, r in enumerate(rows):
            lams.setdefault(lam64(r[8], r[9]) if upd else lam64(r[9], r[10]), []).append(i)
        for lam, idx in lams.items():
            R = rows[idx]; n = len(idx)
            rec = np.zeros(n, MVP_BITS)
            if upd:
                # updateMVP alone: make both AMVP candidates the new base so that the checkBestMVP step that follows changes nothing
                rec["amvp"][:, 0, 0] = R[:, 0]; rec["amvp"][:, 0, 1] = R[:, 1]; rec["amvp"][:, 1] = rec["amvp"][:, 0]
                rec["mv"][:, 0] = R[:, 2]; rec["mv"][:, 1] = R[:, 3]; rec["alter"][:, 0] = R[:, 4]; rec["alter"][:, 1] = R[:, 5]
                rec["useAlter"] = 1; rec["bits"] = R[:, 6].astype(np.uint32); rec["cost"] = R[:, 7].astype(np.uint32)
            else:
                rec["amvp"] = R[:, 0:4].reshape(n, 2, 2); rec["mv"][:, 0] = R[:, 4]; rec["mv"][:, 1] = R[:, 5]
                rec["mvpIdx"] = R[:, 6]; rec["bits"] = R[:, 7].astype(np.uint32); rec["cost"] = R[:, 8].astype(np.uint32)
            d_rec = api.to_device(rec)
            api.mvp_bits_batch(d_rec, n, d_bits, 32768, lam)
            T.cuda.synchronize()
            o = d_rec.cpu().numpy().view(MVP_BITS)
            if upd:
                assert np.array_equal(o["bits"], R[:, 10].astype(np.uint32)) and np.array_equal(o["cost"], R[:, 11].astype(np.uint32))
            else:
                assert np.array_equal(o["mvpIdx"], R[:, 11]) and np.array_equal(o["bits"], R[:, 12].astype(np.uint32)) and np.array_equal(o["cost"], R[:, 13].astype(np.uint32))


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_whole_pu_motion_estimation_calls_through_the_hip_entry_points(depth):
    """The recorded Search::puMotionEstimation calls (tests/golden/pu_*.npz) with every piece of the PU's chain on the GPU -- x265hip_amvp_batch,
    x265hip_select_mvp_batch, x265hip_me_batch (both searches), x265hip_mvp_bits_batch, x265hip_bidir_satd_batch -- each as one batch over all calls that wait for
    it, composed by the glue of tests/tme_pu.py: the MEData records must be the reference's."""
    import tme_pu
    planes, calls = tme_pu.load_fixture(depth)
    api = FrameApi(depth)
    ex = tme_pu.HipExecutor(api, depth, planes)
    cs = [tme_pu.decode(c) for c in calls]
    outs = tme_pu.replay_batched(cs, planes, np.uint8 if depth == 8 else np.uint16, ex.execute)
    n, kinds = 0, set()
    for ci, (c, res) in enumerate(zip(cs, outs)):
        for pi, o in enumerate(res):
            e = tme_pu.expected(c, pi)
            assert tme_pu.same(o, e), "call %d partition %d (part %d): hip chain %s reference %s" % (ci, pi, c["part"], o, e)
            kinds.add((e["ref"][0] >= 0, e["ref"][1] >= 0)); n += 1
    assert n > 1300 and kinds == {(True, False), (False, True), (True, True)}
    assert ex.launches["me"] < 40 and ex.launches["get_pmv"] < 20, "the requests were not batched: %s" % ex.launches


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_tme_frame_steps_whole_pictures_to_the_references_tables(depth):
    """x265hip_tme_frame: every puMotionEstimation call of every CTU of a P and a B picture (tests/golden/tmectu_*.npz: 4 CTUs x 255 schedule entries x 2 pictures),
    stepped on the device through x265hip_tme_schedule -- the MEData table after the run must hold, at every slot the reference wrote, what the reference wrote.
    Inputs assembled from the records: the table entries the reference found where this picture had not written yet, m_areaBestMV, the lookahead's MVs, the
    reference picture's own table, the temporal neighbour of every PU."""
    import tme_pu
    from x265hip_pkg.frame import INTER_CHOICE, TME_TEMPORAL
    planes, calls = tme_pu.load_fixture(depth, "tmectu")
    api = FrameApi(depth)
    T = api.torch
    steps = api.tme_schedule(64, 8, rect=True, amp=False)
    nS = len(steps)
    cs = [tme_pu.decode(c) for c in calls]
    W = H = 128
    bits_row = np.zeros(2 * 32768 + 1, np.float32)
    api.h.check(api.lib.x265hip_mvbits_row(32768, bits_row.ctypes.data_as(__import__("ctypes").c_void_p)))
    d_bits = api.to_device(bits_row.view(np.int32))
    d_plane, d_phase = {}, {}
    def dev_plane(pid):
        if pid not in d_plane:
            g, px = planes[pid]
            d_plane[pid] = api.to_device(px)
            d_phase[pid] = T.zeros(16 * px.size, dtype=d_plane[pid].dtype, device="cuda")
            api.subpel_planes(d_plane[pid], int(g[1]), int(g[2]), d_phase[pid], px.size)
        return d_plane[pid], d_phase[pid]
    checked = 0
    for poc in sorted({c["curPOC"] for c in cs}):
        fc = [c for c in cs if c["curPOC"] == poc]
        assert len(fc) == 4 * nS
        c0 = fc[0]
        g = planes[int(c0["planeIds"][0][0][0])][0]
        stride, rows, origin = int(g[1]), int(g[2]), int(g[3])
        # the source picture: rebuilt from the PU blocks of the 2Nx2N 64x64 calls
        cur = np.zeros(stride * rows, planes[int(c0["planeIds"][0][0][0])][1].dtype)
        table = np.zeros((4, 593), INTER_CHOICE); table["ref"] = -1
        written = np.zeros((4, 593), bool)
        area = np.zeros((4, 5, 2, 16, 2), np.int16)                   # [ctu][area][list][X265HIP_MAX_REF][x, y]
        temporal = np.zeros((4, nS, 2), TME_TEMPORAL); temporal["nb"]["refIdx"] = -1
        nlist = 1 if c0["isP"] else 2
        ref_tab = [[np.zeros((4, 593), INTER_CHOICE) for _ in range(4)] for _ in range(2)]
        ref_tab_on = [[False] * 4 for _ in range(2)]
        for l in range(2):
            for r in range(4): ref_tab[l][r]["ref"] = -1
        low = [[np.zeros((H // 16) * (W // 16) * 2, np.int16) for _ in range(4)] for _ in range(2)]
        qps, qp_of = [], np.zeros((4, nS), np.uint8)
        per_ctu = {k: [c for c in fc if (c["cuY"] // 64) * 2 + c["cuX"] // 64 == k] for k in range(4)}
        for ctu, lst in per_ctu.items():
            assert len(lst) == nS
            for k, c in enumerate(lst):
                st = steps[k]
                assert (c["part"], c["finalIdx"], c["puOffset"]) == (int(st["part"]), int(st["finalIdx"]), int(st["puOffset"]))
                area[ctu, c["area"], :, :4] = c["areaBest"]
                for d in range(5):
                    slot = c["nbIdx"][d]
                    if slot >= 0 and not written[ctu, slot]:
                        rec = c["nbRec"][d]
                        table[ctu, slot]["mv"] = rec[0:4].reshape(2, 2); table[ctu, slot]["ref"] = rec[4:6]
                subs = list(c["subs"])
                for pi in range(c["numPart"]):
                    x, y, w, h = (int(v) for v in c["geo"][pi])
                    if c["part"] == 0 and w == 64:
                        for yy in range(64):
                            cur[origin + (y + yy) * stride + x: origin + (y + yy) * stride + x + 64] = c["blocks"][pi][yy * 64:(yy + 1) * 64]
                    for l in range(nlist):
                        for r in range(c["numRef"][l]):
                            k5 = next(s for s in subs if s[0] == 5); subs.remove(k5); k5 = k5[1]
                            assert (int(k5[0]), int(k5[1])) == (l, r)
                            nb5 = k5[38 + 45:38 + 54]
                            tp = temporal[ctu, k, pi]
                            tp["nb"]["refIdx"] = (nb5[4], nb5[5])
                            for ll in range(2):
                                if nb5[4 + ll] != -1:
                                    tp["nb"]["mv"][ll] = (nb5[2 * ll], nb5[2 * ll + 1])
                            tp["colPOC"][l] = k5[92]; tp["colRefPOC"][l] = k5[93]
                            rr = c["refRec"][pi][l][r]
                            if rr[4] != -3:
                                ref_tab_on[l][r] = True
                                e = ref_tab[l][r][ctu, c["finalIdx"] + pi * c["puOffset"]]
                                e["mv"] = rr[0:4].reshape(2, 2); e["ref"] = rr[4:6]
                            k10 = [s for s in subs if s[0] == 10]
                            if x + (w >> 1) < W and y + (h >> 1) < H:
                                k10 = k10[0]; subs.remove(k10); k10 = k10[1]
                                idx = ((y + h // 2) >> 4) * (W // 16) + ((x + w // 2) >> 4)
                                assert int(k10[2]) % 2 == 0 and int(k10[3]) % 2 == 0
                                low[l][r][2 * idx] = int(k10[2]) // 2; low[l][r][2 * idx + 1] = int(k10[3]) // 2
                            k2 = next(s for s in subs if s[0] == 2); subs.remove(k2)
                            qp = int(k2[1][14])
                            if (qp, c["lam"]) not in qps: qps.append((qp, c["lam"]))
                            qp_of[ctu, k] = qps.index((qp, c["lam"]))
                            nxt = [s for s in subs if s[0] in (2, 5)]
                            if nxt and nxt[0][0] == 2: subs.remove(nxt[0])          # the second search
                    written[ctu, c["finalIdx"] + pi * c["puOffset"]] = True
        d_rows = api.to_device(np.concatenate([mvcost_row(depth, q, 1 << 15) for (q, _) in qps]).view(np.int16))       # the cost table: one row per qp of the picture
        refs = [[], []]
        for l in range(nlist):
            for r in range(c0["numRef"][l]):
                me_p, me_ph = dev_plane(int(c0["planeIds"][l][r][0])); _, rec_ph = dev_plane(int(c0["planeIds"][l][r][1]))
                refs[l].append(dict(me_plane=me_p, me_phase=me_ph, recon_phase=rec_ph, ref_table=api.to_device(ref_tab[l][r].reshape(-1)) if ref_tab_on[l][r] else None,
                                    lowres_mv=api.to_device(low[l][r])))
        d_table = api.to_device(table.reshape(-1))
        api.tme_frame(is_p=c0["isP"], num_ref=c0["numRef"], cur_poc=poc, temporal_mvp=c0["temporal"], ref_poc=[c0["refPOC"][:16], c0["refPOC"][16:]], merange=c0["merange"],
                      method=c0["method"], subme=c0["subme"], lams=[lm for (_, lm) in qps], qp_index=api.to_device(qp_of.reshape(-1)), width=W, height=H, ctu=64, lowres_blocks_x=W // 16, cur=api.to_device(cur), stride=stride, origin=origin,
                      plane_elems=stride * rows, refs=refs, table=d_table, area_best=api.to_device(area.reshape(-1)), temporal=api.to_device(temporal.reshape(-1)),
                      cost_rows=d_rows, cost_half=1 << 15, bits_row=d_bits, bits_half=32768, steps=steps)
        out = d_table.cpu().numpy().view(INTER_CHOICE).reshape(4, 593)
        # the LAST write of a slot is what the table holds
        last = {}
        for ctu, lst in per_ctu.items():
            for c in lst:
                for pi in range(c["numPart"]):
                    last[(ctu, c["finalIdx"] + pi * c["puOffset"])] = tme_pu.expected(c, pi)
        kinds = set()
        for (ctu, slot), e in last.items():
            o = out[ctu, slot]
            got = dict(mv=[(int(o["mv"][0][0]), int(o["mv"][0][1])), (int(o["mv"][1][0]), int(o["mv"][1][1]))], mvp=[(int(o["mvp"][0][0]), int(o["mvp"][0][1])), (int(o["mvp"][1][0]), int(o["mvp"][1][1]))],
                       mvCost=[int(o["mvCost"][0]), int(o["mvCost"][1])], ref=[int(o["ref"][0]), int(o["ref"][1])], bits=int(o["bits"]), cost=int(o["cost"]))
            assert tme_pu.same(got, e), "POC %d CTU %d slot %d: device %s reference %s" % (poc, ctu, slot, got, e)
            kinds.add((e["ref"][0] >= 0, e["ref"][1] >= 0)); checked += 1
    assert checked >= 2 * 4 * 400
