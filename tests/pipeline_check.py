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
