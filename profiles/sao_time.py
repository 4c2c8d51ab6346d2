"""frame-level SAO statistics: time per 1080p / 4K picture and the algorithmic HBM rate (2 planes read once + 1280 B per CTU written)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import x265hip  # noqa
import torch
from x265hip_pkg.frame import FrameApi
for depth in (8, 10):
    api = FrameApi(depth)
    for (W, H) in ((1920, 1080), (3840, 2160)):
        rng = np.random.default_rng(1)
        pm = (1 << depth) - 1
        dt = np.uint8 if depth == 8 else np.uint16
        f = rng.integers(0, pm + 1, W * H).astype(dt); r = np.clip(f.astype(np.int32) + rng.integers(-4, 5, W * H), 0, pm).astype(dt)
        d_f, d_r = api.to_device(f), api.to_device(r)
        n = ((W + 63) // 64) * ((H + 63) // 64)
        d_out = torch.zeros(n * 320, dtype=torch.int32, device="cuda")
        P = lambda x: C.c_void_p(x.data_ptr())
        def run(): api.h.check(api.lib.x265hip_sao_stats_frame(api.stream(), P(d_f), P(d_r), C.c_ssize_t(W), W, H, 64, 0, 0, P(d_out)))
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        alg = W * H * 2 * f.itemsize + n * 1280
        print("%d bit %dx%d: %.4f ms per picture, %.1f GB/s algorithmic (%.1f %% of 8 TB/s), %.0f Mpx/s" % (depth, W, H, ms, alg / ms / 1e6, alg / ms / 1e6 / 80, W * H / ms / 1e3))
