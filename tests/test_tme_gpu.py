"""SURVEY 8(f1) at the data level, GPU side: the task list of a reference encode run with --threaded-me -- every PU of every CTU, with the window,
predictor and up to 12 candidates Search::puMotionEstimation built for it (fixtures from oracle/ref_tme.cpp, no reference needed here) --
through x265hip_me_batch: the MV, the cost and the MV cost must be the reference's own (MEData.mv / mvCost, encoder/threadedme.h:122-130)."""
import numpy as np
import pytest

import x265hip  # noqa: F401
from x265hip_pkg.frame import FrameApi, ME_TASK, ME_RESULT, mvcost_row
from tme_util import TmeFixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("planes", [True, False])
@pytest.mark.parametrize("depth", [8, 10])
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
