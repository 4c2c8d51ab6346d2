"""SEA path timing: integral planes of a reference stack + x265hip_me_batch_sea over the 16x16 PUs of 8 frames (1080p, 8 bit)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import x265hip  # noqa
import torch
from x265hip_pkg.frame import FrameApi, mvcost_row, ME_RESULT
from x265hip_pkg.pipeline import FramePipeline
from x265hip_pkg.synth import frame_pair
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
merange = int(sys.argv[2]) if len(sys.argv) > 2 else 16
api = FrameApi(depth)
row = mvcost_row(depth, 28, 1 << 15)
pipe = FramePipeline(depth, 1920, 1088, 8, qp=28, merange=merange, method=1, subme=2, tu_log2=5, cost_row=row, api=api)
pipe.upload([frame_pair(1920, 1088, depth, s, margin=pipe.margin, max_shift=24)[:2] for s in range(8)])
pipe.step(); torch.cuda.synchronize()
rows = pipe.F * (pipe.H + 2 * pipe.margin)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
integ = [None]
def planes(): integ[0] = api.sea_integral_planes(pipe.d_ref, pipe.stride, rows)
ms = timed(planes)
px = pipe.stride * rows
bpp = 1 if depth == 8 else 2
print("integral planes: %.3f ms for %d x %d (%.1f Mpx): %.1f GB/s algorithmic (read %d + write 48 B/px)" % (ms, pipe.stride, rows, px / 1e6, px * (bpp + 48) / ms / 1e6, bpp))
d_int, ie = integ[0]
for lv in (16, 32, 64):
    n = len(pipe.tasks_host[lv])
    res = torch.zeros(n * ME_RESULT.itemsize, dtype=torch.uint8, device="cuda")
    def sea(): api.me_batch_sea(lv, lv, pipe.d_cur, pipe.stride, pipe.d_ref, pipe.stride, pipe.d_tasks[lv], n, pipe.d_cost, pipe.half, merange, 2, res, d_int, ie,
                                planes=pipe.d_planes, plane_elems=pipe.plane_elems)
    ms = timed(sea, 3)
    print("SEA %dx%d merange %d: %.3f ms for %d PUs (%.1f Mpx/s)" % (lv, lv, merange, ms, n, n * lv * lv / ms / 1e3))
