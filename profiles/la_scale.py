"""lookahead cost batch: time vs number of estimates in the batch (latency chain vs throughput)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x265hip  # noqa
import torch
from x265hip_pkg.lookahead import LookaheadBatch, minigop_estimates, pan_clip
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
N = int(sys.argv[4]) if len(sys.argv) > 4 else 32
ROWS = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # lookahead slices: block rows per slice (0 = one sweep per picture)
print("depth %d, %dx%d, %d pictures, rows per slice %d" % (depth, W, H, N, ROWS))
est = minigop_estimates(N, 3)
lb = LookaheadBatch(depth, W, H, N, 4 * len(est))
lb.upload(pan_clip(W, H, N, depth, seed=11)); lb.build_lowres(); lb.intra()
for n in (1, 8, 32, 128, len(est), 2 * len(est), 4 * len(est)):
    lb.set_estimates((est * 4)[:n])
    lb.costs(ROWS); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): lb.costs(ROWS)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("estimates %4d: %.3f ms  (%.1f us per estimate)" % (n, ms, ms * 1e3 / n))
