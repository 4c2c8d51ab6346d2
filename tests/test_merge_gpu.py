"""x265hip_inter_merge_batch on the GPU against the oracle's restatement of the tail of Search::puMotionEstimation (search.cpp:258-556): several
references per list searched with x265hip_me_batch, then per PU the best reference of each list, the bidirectional candidate (pixelavg of the two
predictions at SATD, and its zero-MV form) and the final choice."""
import numpy as np
import pytest

from depths import DEPTHS

import x265hip  # noqa: F401
from x265hip_pkg.frame import FrameApi, ME_TASK, ME_RESULT, INTER_CHOICE, mvcost_row, mvbits_row, rd_lambda
from x265hip_pkg.synth import frame_pair
from backends import Oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("nref", [(3, 0), (2, 2), (1, 1), (4, 3)])
def test_merge_matches_oracle(depth, nref):
    api, ora = FrameApi(depth), Oracle(depth)
    T = api.torch
    rng = np.random.default_rng(17 * depth + 5 * nref[0] + nref[1])
    W, H, margin, qp, merange, method, subme = 256, 128, 96, 30, 16, 1, 2
    half, bhalf = 1 << 13, 1 << 13
    cur, _, stride, _ = frame_pair(W, H, depth, 3, margin=margin, max_shift=4)
    cur_f = cur.reshape(-1)
    refs = [[None] * 4, [None] * 4]
    _, base, _, _ = frame_pair(W, H, depth, 3, margin=margin, max_shift=5, noise=0.5)
    pm = (1 << depth) - 1
    for l in range(2):
        for r in range(nref[l]):                                     # each reference: the displaced picture + its own strong noise (averaging two of them pays: bidir wins often)
            sigma = (6 + 5 * r) * (1 << (depth - 8)) * (1.0 if (l, r) != (1, 0) else 1.4)
            refs[l][r] = np.clip(base.astype(np.float64) + rng.normal(0, sigma, base.shape), 0, pm).astype(base.dtype).reshape(-1)
    d_cur = api.to_device(cur_f)
    d_ref = [[api.to_device(x) if x is not None else None for x in refs[l]] for l in range(2)]
    pe = cur_f.size
    d_pl = [[None] * 4, [None] * 4]
    for l in range(2):
        for r in range(nref[l]):
            d_pl[l][r] = T.zeros(16 * pe, dtype=d_cur.dtype, device="cuda")
            api.subpel_planes(d_ref[l][r], stride, cur.shape[0], d_pl[l][r], pe)
    row = mvcost_row(depth, qp, half); d_row = api.to_device(row.view(np.int16))
    bits = mvbits_row(depth, bhalf); d_bits = api.to_device(bits.view(np.int32)).view(T.float32)
    lam = rd_lambda(depth, qp)
    used = set()
    for (w, h) in [(8, 8), (16, 16), (32, 32), (64, 64)]:
        nx, ny = W // w, H // h
        n = nx * ny
        t = np.zeros(n, ME_TASK)
        by, bx = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
        x, y = (bx * w).reshape(-1), (by * h).reshape(-1)
        t["curOff"] = t["refOff"] = (margin + y) * stride + margin + x
        t["mvmin"][:, 0] = -((64 + 8 + x - 1) << 2); t["mvmin"][:, 1] = -((64 + 8 + y - 1) << 2)
        t["mvmax"][:, 0] = (W + 8 - x - 1) << 2; t["mvmax"][:, 1] = (H + 8 - y - 1) << 2
        t["flags"] = 1; t["mvpFrom"] = -1
        t["qmvp"] = rng.integers(-12, 13, (n, 2))
        d_t = api.to_device(t)
        d_res = [[None] * 4, [None] * 4]
        res = [[None] * 4, [None] * 4]
        for l in range(2):
            for r in range(nref[l]):
                d_res[l][r] = T.zeros(n * ME_RESULT.itemsize, dtype=T.uint8, device="cuda")
                api.me_batch(w, h, d_cur, stride, d_ref[l][r], stride, d_t, n, d_row, half, merange, method, subme, d_res[l][r], planes=d_pl[l][r], plane_elems=pe)
        d_out = T.zeros(n * INTER_CHOICE.itemsize, dtype=T.uint8, device="cuda")
        api.inter_merge_batch(w, h, d_cur, stride, stride, d_t, n, [d_res[0][:nref[0]], d_res[1][:nref[1]]], None, [d_pl[0][:nref[0]], d_pl[1][:nref[1]]], pe,
                              d_bits, bhalf, lam, True, max(W, H), d_out)
        T.cuda.synchronize()
        for l in range(2):
            for r in range(nref[l]):
                res[l][r] = d_res[l][r].cpu().numpy().view(ME_RESULT)
        got = d_out.cpu().numpy().view(INTER_CHOICE)
        for i in range(n):
            mv = np.zeros((8, 2), np.int32); mvp = np.zeros((8, 2), np.int32); cost = np.zeros(8, np.int32); mvc = np.zeros(8, np.int32)
            for l in range(2):
                for r in range(nref[l]):
                    k = 4 * l + r
                    mv[k] = res[l][r][i]["mv"]; mvp[k] = t[i]["qmvp"]; cost[k] = res[l][r][i]["cost"]; mvc[k] = res[l][r][i]["mvcost"]
            o, mco = ora.inter_merge(w, h, nref, mv, mvp, cost, mvc, bits, lam, True, max(W, H), list(t[i]["mvmin"]) + list(t[i]["mvmax"]),
                                     cur_f, stride, int(t[i]["curOff"]), refs[0] + refs[1], stride, int(t[i]["refOff"]))
            g = got[i]
            mine = [int(g["mv"][0][0]), int(g["mv"][0][1]), int(g["mv"][1][0]), int(g["mv"][1][1]), int(g["mvp"][0][0]), int(g["mvp"][0][1]), int(g["mvp"][1][0]), int(g["mvp"][1][1]),
                    int(g["ref"][0]), int(g["ref"][1]), int(g["bits"]), int(g["cost"])]
            assert mine == [int(v) for v in o] and [int(v) for v in g["mvCost"]] == [int(v) for v in mco], "PU %dx%d #%d refs %s: hip %s oracle %s" % (w, h, i, nref, mine, list(o))
            used.add((int(g["ref"][0]) >= 0, int(g["ref"][1]) >= 0))
    if nref[1]:
        assert len(used) >= 2, "the clip should exercise more than one outcome (uni / bi): %s" % used
