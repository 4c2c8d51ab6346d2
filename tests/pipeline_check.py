"""Test-side checker of a FramePipeline step: randomly sampled PUs / TUs of the last step() against the oracle (bit-exact).
Lives under tests/ because it drives the oracle; the product package never sees an oracle object."""
import numpy as np

from x265hip_pkg.frame import ME_RESULT  # noqa: F401
from x265hip_pkg.pipeline import LEVELS


def check_sample(pipe, oracle, rng, per_level=40, n_tu=40):
    """Compare randomly sampled PUs / TUs of the last step() with the oracle (bit-exact). Returns #checked."""
    res = {lv: pipe.results(lv) for lv in LEVELS}
    checked = 0
    for lv in LEVELS:
        t = pipe.tasks_host[lv]
        for i in rng.choice(len(t), size=min(per_level, len(t)), replace=False):
            tk = t[i]
            qmvp = (0, 0) if tk["mvpFrom"] < 0 else tuple(int(v) for v in res[2 * lv][tk["mvpFrom"]]["mv"])
            d = pipe.merange << 2
            lx0, ly0, lx1, ly1 = int(tk["mvmin"][0]), int(tk["mvmin"][1]), int(tk["mvmax"][0]), int(tk["mvmax"][1])
            b = [min(lx1, max(lx0, qmvp[0] - d)) >> 2, min(ly1, max(ly0, qmvp[1] - d)) >> 2,
                 min(lx1, max(lx0, qmvp[0] + d)) >> 2, min(ly1, max(ly0, qmvp[1] + d)) >> 2]
            b[3] = max(b[3], b[1])
            exp = oracle.me(lv, lv, pipe.cur_host, pipe.stride, int(tk["curOff"]), pipe.ref_host, pipe.stride, int(tk["refOff"]),
                            b, qmvp, [], pipe.merange, pipe.method, pipe.subme, pipe.cost_row_host)
            got = (int(res[lv][i]["mv"][0]), int(res[lv][i]["mv"][1]), int(res[lv][i]["cost"]))
            assert got == exp, "ME level %d task %d: hip %s oracle %s" % (lv, i, got, exp)
            checked += 1
    if getattr(pipe, "rect", False):
        for (w, h), t in pipe.rect_host.items():
            lv = max(w, h)
            rr = pipe.rect_results(w, h)
            for i in rng.choice(len(t), size=min(max(per_level // 2, 4), len(t)), replace=False):
                tk = t[i]
                qmvp = tuple(int(v) for v in res[lv][tk["mvpFrom"]]["mv"])
                d = pipe.merange << 2
                lx0, ly0, lx1, ly1 = int(tk["mvmin"][0]), int(tk["mvmin"][1]), int(tk["mvmax"][0]), int(tk["mvmax"][1])
                b = [min(lx1, max(lx0, qmvp[0] - d)) >> 2, min(ly1, max(ly0, qmvp[1] - d)) >> 2, min(lx1, max(lx0, qmvp[0] + d)) >> 2, min(ly1, max(ly0, qmvp[1] + d)) >> 2]
                b[3] = max(b[3], b[1])
                exp = oracle.me(w, h, pipe.cur_host, pipe.stride, int(tk["curOff"]), pipe.ref_host, pipe.stride, int(tk["refOff"]), b, qmvp, [], pipe.merange, pipe.method,
                                pipe.subme, pipe.cost_row_host)
                got = (int(rr[i]["mv"][0]), int(rr[i]["mv"][1]), int(rr[i]["cost"]))
                assert got == exp, "rect PU %dx%d task %d: hip %s oracle %s" % (w, h, i, got, exp)
                checked += 1
    n = 1 << pipe.tu_log2
    coeff = pipe.d_coeff.cpu().numpy().reshape(-1, n * n)
    numsig = pipe.d_numsig.cpu().numpy()
    sse = pipe.d_sse.cpu().numpy() if pipe.recon else None
    rec = pipe.d_recon.cpu().numpy().view(pipe.cur_host.dtype) if pipe.recon else None
    for i in rng.choice(len(pipe.tu_host), size=min(n_tu, len(pipe.tu_host)), replace=False):
        tk = pipe.tu_host[i]
        mv = tuple(int(v) for v in res[pipe.mv_level][tk["mvFrom"]]["mv"])
        e_ns, e_coeff, _, e_rec, e_sse = oracle.tq_tu(pipe.tu_log2, pipe.cur_host, pipe.stride, int(tk["curOff"]), pipe.ref_host, pipe.stride,
                                                      int(tk["refOff"]), mv, pipe.qp, 85, want_recon=pipe.recon)
        assert int(numsig[i]) == e_ns and np.array_equal(coeff[i], e_coeff), "TU %d: coefficients differ from the oracle" % i
        if pipe.recon:
            o = int(tk["reconOff"])
            got = np.concatenate([rec[o + y * pipe.stride: o + y * pipe.stride + n] for y in range(n)])
            assert np.array_equal(got, e_rec) and int(sse[i]) == e_sse, "TU %d: reconstruction differs from the oracle" % i
        checked += 1
    return checked


def check_sample_refs(pipe, oracle, rng, per_level=10, n_tu=10):
    """The several-references form of check_sample: per sampled PU the search in every reference (its own parent chain), the choice among them
    (xo_inter_merge) and, per sampled TU, the coefficients compensated from the chosen reference -- all against the oracle."""
    R = pipe.refs
    res = {lv: [pipe.results(lv, r) for r in range(R)] for lv in LEVELS}
    ch = {lv: pipe.choices(lv) for lv in LEVELS}
    checked = 0
    for lv in LEVELS:
        t = pipe.tasks_host[lv]
        for i in rng.choice(len(t), size=min(per_level, len(t)), replace=False):
            tk = t[i]
            mv = np.zeros((8, 2), np.int32); mvp = np.zeros((8, 2), np.int32); cost = np.zeros(8, np.int32); mvc = np.zeros(8, np.int32)
            for r in range(R):
                qmvp = (0, 0) if tk["mvpFrom"] < 0 else tuple(int(v) for v in res[2 * lv][r][tk["mvpFrom"]]["mv"])
                d = pipe.merange << 2
                lx0, ly0, lx1, ly1 = int(tk["mvmin"][0]), int(tk["mvmin"][1]), int(tk["mvmax"][0]), int(tk["mvmax"][1])
                b = [min(lx1, max(lx0, qmvp[0] - d)) >> 2, min(ly1, max(ly0, qmvp[1] - d)) >> 2, min(lx1, max(lx0, qmvp[0] + d)) >> 2, min(ly1, max(ly0, qmvp[1] + d)) >> 2]
                b[3] = max(b[3], b[1])
                exp = oracle.me(lv, lv, pipe.cur_host, pipe.stride, int(tk["curOff"]), pipe.refs_host[r], pipe.stride, int(tk["refOff"]), b, qmvp, [],
                                pipe.merange, pipe.method, pipe.subme, pipe.cost_row_host)
                g = res[lv][r][i]
                assert (int(g["mv"][0]), int(g["mv"][1]), int(g["cost"])) == exp, "level %d task %d reference %d: hip %s oracle %s" % (lv, i, r, g, exp)
                mv[r] = g["mv"]; mvp[r] = qmvp; cost[r] = g["cost"]; mvc[r] = g["mvcost"]
            o, mco = oracle.inter_merge(lv, lv, (R, 0), mv, mvp, cost, mvc, pipe.bits_row_host, pipe.rd_lambda, False, max(pipe.W, pipe.H),
                                        list(tk["mvmin"]) + list(tk["mvmax"]), pipe.cur_host, pipe.stride, int(tk["curOff"]), pipe.refs_host + [None] * (8 - R), pipe.stride, int(tk["refOff"]))
            g = ch[lv][i]
            mine = [int(g["mv"][0][0]), int(g["mv"][0][1]), int(g["mv"][1][0]), int(g["mv"][1][1]), int(g["mvp"][0][0]), int(g["mvp"][0][1]), int(g["mvp"][1][0]), int(g["mvp"][1][1]),
                    int(g["ref"][0]), int(g["ref"][1]), int(g["bits"]), int(g["cost"])]
            assert mine == [int(v) for v in o], "level %d task %d: choice hip %s oracle %s" % (lv, i, mine, list(o))
            checked += 1
    n = 1 << pipe.tu_log2
    coeff = pipe.d_coeff.cpu().numpy().reshape(-1, n * n)
    numsig = pipe.d_numsig.cpu().numpy()
    for i in rng.choice(len(pipe.tu_host), size=min(n_tu, len(pipe.tu_host)), replace=False):
        tk = pipe.tu_host[i]
        c = ch[pipe.mv_level][tk["mvFrom"]]
        r = int(c["ref"][0])
        e_ns, e_coeff, _, _, _ = oracle.tq_tu(pipe.tu_log2, pipe.cur_host, pipe.stride, int(tk["curOff"]), pipe.refs_host[r], pipe.stride, int(tk["refOff"]),
                                              (int(c["mv"][0][0]), int(c["mv"][0][1])), pipe.qp, 85)
        assert int(numsig[i]) == e_ns and np.array_equal(coeff[i], e_coeff), "TU %d (reference %d): coefficients differ from the oracle" % (i, r)
        checked += 1
    return checked


def check_host_batch(hb, oracle, rng, cost_row, bits_row=None, rd_lambda=None, per_shape=8, n_tu=10):
    """The C++ host's batch (x265hip_pkg.host_batch.HostBatch: any number of references in either list, with or without the rectangular and the asymmetric PUs) against the
    oracle: per sampled PU of every searched shape the search in every reference of every list (seeded by that reference's own chain), with several references or a B picture
    the choice among them -- with the bidirectional candidate where the reference has one (xo_inter_merge) --, and per sampled TU the coefficients compensated from the list and
    reference its PU chose.  Returns #checked."""
    R0, R1 = hb.refs, getattr(hb, "refs1", 0)
    lists = [(0, r) for r in range(R0)] + [(1, r) for r in range(R1)]
    planes = {(0, r): hb.refs_host[r] for r in range(R0)}
    planes.update({(1, r): hb.refs1_host[r] for r in range(R1)})
    need_choice = R0 > 1 or R1 > 0
    sq = {lv: {k: hb.shape_results(lv, lv, k[1], k[0]) for k in lists} for lv in LEVELS}
    shapes = [(lv, lv, hb.tasks_host[lv], sq[lv], sq.get(2 * lv)) for lv in LEVELS]
    for table in ([hb.rect_host] if hb.rect else []) + ([hb.amp_host] if getattr(hb, "amp", False) else []):
        for (w, h), t in table.items():
            shapes.append((w, h, t, {k: hb.shape_results(w, h, k[1], k[0]) for k in lists}, sq[max(w, h)]))
    checked = 0
    d = hb.merange << 2
    for (w, h, t, res, parent) in shapes:
        ch = hb.choices(w, h) if need_choice else None
        bidir = R1 > 0 and w != h and max(w, h) > 8           # search.cpp:421-422: not for 2Nx2N, not inside an 8x8 CU
        for i in rng.choice(len(t), size=min(per_shape, len(t)), replace=False):
            tk = t[i]
            mv = np.zeros((8, 2), np.int32); mvp = np.zeros((8, 2), np.int32); cost = np.zeros(8, np.int32); mvc = np.zeros(8, np.int32)
            for (l, r) in lists:
                qmvp = (0, 0) if tk["mvpFrom"] < 0 else tuple(int(v) for v in parent[(l, r)][tk["mvpFrom"]]["mv"])
                lx0, ly0, lx1, ly1 = int(tk["mvmin"][0]), int(tk["mvmin"][1]), int(tk["mvmax"][0]), int(tk["mvmax"][1])
                b = [min(lx1, max(lx0, qmvp[0] - d)) >> 2, min(ly1, max(ly0, qmvp[1] - d)) >> 2, min(lx1, max(lx0, qmvp[0] + d)) >> 2, min(ly1, max(ly0, qmvp[1] + d)) >> 2]
                b[3] = max(b[3], b[1])
                exp = oracle.me(w, h, hb.cur_host, hb.stride, int(tk["curOff"]), planes[(l, r)], hb.stride, int(tk["refOff"]), b, qmvp, [], hb.merange, hb.method, hb.subme, cost_row)
                g = res[(l, r)][i]
                assert (int(g["mv"][0]), int(g["mv"][1]), int(g["cost"])) == exp, "%dx%d task %d list %d reference %d: hip %s oracle %s" % (w, h, i, l, r, g, exp)
                if r < 4:
                    mv[4 * l + r] = g["mv"]; mvp[4 * l + r] = qmvp; cost[4 * l + r] = g["cost"]; mvc[4 * l + r] = g["mvcost"]
            if need_choice and R0 <= 4 and R1 <= 4:          # the oracle's merge takes up to 4 references per list
                rp = [planes.get((k // 4, k % 4)) for k in range(8)]
                o, _ = oracle.inter_merge(w, h, (R0, R1), mv, mvp, cost, mvc, bits_row, rd_lambda, bidir, max(hb.W, hb.H), list(tk["mvmin"]) + list(tk["mvmax"]),
                                          hb.cur_host, hb.stride, int(tk["curOff"]), rp, hb.stride, int(tk["refOff"]))
                g = ch[i]
                mine = [int(g["mv"][0][0]), int(g["mv"][0][1]), int(g["mv"][1][0]), int(g["mv"][1][1]), int(g["mvp"][0][0]), int(g["mvp"][0][1]), int(g["mvp"][1][0]), int(g["mvp"][1][1]),
                        int(g["ref"][0]), int(g["ref"][1]), int(g["bits"]), int(g["cost"])]
                assert mine == [int(v) for v in o], "%dx%d task %d: choice hip %s oracle %s" % (w, h, i, mine, list(o))
            checked += 1
    n = 1 << hb.tu_log2
    coeff, numsig = hb.coeffs()
    coeff = coeff.reshape(-1, n * n)
    chm = hb.choices(hb.mv_level) if need_choice else None
    for i in rng.choice(len(hb.tu_host), size=min(n_tu, len(hb.tu_host)), replace=False):
        tk = hb.tu_host[i]
        if need_choice:
            c = chm[tk["mvFrom"]]
            assert (int(c["ref"][0]) >= 0) != (int(c["ref"][1]) >= 0), "a 2Nx2N PU is uni-directional here"
            l = 0 if int(c["ref"][0]) >= 0 else 1
            key = (l, int(c["ref"][l])); mv = (int(c["mv"][l][0]), int(c["mv"][l][1]))
        else:
            key = (0, 0); mv = tuple(int(v) for v in sq[hb.mv_level][(0, 0)][tk["mvFrom"]]["mv"])
        e_ns, e_coeff, _, _, _ = oracle.tq_tu(hb.tu_log2, hb.cur_host, hb.stride, int(tk["curOff"]), planes[key], hb.stride, int(tk["refOff"]), mv, hb.qp, 85)
        assert int(numsig[i]) == e_ns and np.array_equal(coeff[i], e_coeff), "TU %d (list %d reference %d): coefficients differ from the oracle" % (i, key[0], key[1])
        checked += 1
    return checked
