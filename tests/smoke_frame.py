"""Frame-level smoke: a tiny ME -> TQ (+recon) batch on cuda:0, sampled against the oracle."""
import numpy as np


def smoke_frame():
    import x265hip  # noqa: F401
    from x265hip_pkg.frame import mvcost_row
    from x265hip_pkg.pipeline import FramePipeline
    from x265hip_pkg.synth import frame_pair
    from backends import Oracle
    for depth in (8, 10):
        row = mvcost_row(depth, 28, 1 << 15)
        pipe = FramePipeline(depth, 128, 64, 2, qp=28, merange=16, method=1, subme=2, tu_log2=5, recon=True, cost_row=row)
        pipe.upload([frame_pair(128, 64, depth, s, margin=pipe.margin, max_shift=6)[:2] for s in range(2)])
        pipe.step()
        pipe.torch.cuda.synchronize()
        n = pipe.check_sample(Oracle(depth), np.random.default_rng(depth), per_level=6, n_tu=6)
        assert n >= 20
