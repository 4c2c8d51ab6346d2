"""Frame-level smoke: a tiny ME -> TQ (+recon) batch on cuda:0, sampled against the oracle."""
import numpy as np


def smoke_frame():
    import x265hip  # noqa: F401
    from x265hip_pkg.frame import mvcost_row
    from x265hip_pkg.pipeline import FramePipeline
    from x265hip_pkg.synth import frame_pair
    from backends import Oracle
    from pipeline_check import check_sample
    for depth in (8, 10):
        row = mvcost_row(depth, 28, 1 << 15)
        pipe = FramePipeline(depth, 128, 64, 2, qp=28, merange=16, method=1, subme=2, tu_log2=5, recon=True, cost_row=row)
        pipe.upload([frame_pair(128, 64, depth, s, margin=pipe.margin, max_shift=6)[:2] for s in range(2)])
        pipe.step()
        pipe.torch.cuda.synchronize()
        n = check_sample(pipe, Oracle(depth), np.random.default_rng(depth), per_level=6, n_tu=6)
        assert n >= 20
    smoke_lookahead()


def smoke_lookahead():
    """a P and a B frame-cost estimate on 4 small pictures, every MV / cost / total against the oracle"""
    import x265hip  # noqa: F401
    from x265hip_pkg.lookahead import LookaheadBatch
    from backends import Oracle
    from lookahead_util import Geometry, lowres_planes_oracle, oracle_frame_cost, oracle_intra, synth_clip
    for depth in (8, 10):
        ora = Oracle(depth)
        frames = synth_clip(128, 96, 4, depth, seed=3)
        lb = LookaheadBatch(depth, 128, 96, 4, 2)
        lb.upload(frames); lb.build_lowres(); lb.intra()
        est = [(0, 2, 2), (0, 1, 3)]
        lb.set_estimates(est); lb.costs(); lb.t.cuda.synchronize()
        g = Geometry(128, 96)
        planes = [lowres_planes_oracle(ora, f, g) for f in frames]
        ic = lb.d_intra_cost.cpu().numpy().reshape(4, g.ncu)
        mvs = lb.d_mvs.cpu().numpy().reshape(-1, 2 * g.ncu); sums = lb.d_sums.cpu().numpy().reshape(-1, 3)
        for i, (p0, b, p1) in enumerate(est):
            it = oracle_intra(ora, planes[b], g)
            assert np.array_equal(ic[b], it["intraCost"]), "smoke: lookahead intra costs"
            o = oracle_frame_cost(ora, planes[b], planes[p0], planes[p1] if p1 > b else None, g, it["intraCost"], None)
            assert np.array_equal(mvs[2 * i], o["mvs0"]) and [int(v) for v in sums[i]] == [o["costEst"], o["costEstAq"], o["intraMbs"]], "smoke: lookahead estimate"
