import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import numpy as np
import x265hip
from x265hip_pkg.synth import frame_pair
from x265hip_pkg.frame import FrameApi, ME_TASK, ME_RESULT
from backends import Oracle
depth = int(sys.argv[1])
api, ora = FrameApi(depth), Oracle(depth)
W, H, margin = 320, 192, 96
cur, ref, stride, (dx, dy) = frame_pair(W, H, depth, 10, margin=margin, max_shift=10)
cur_f, ref_f = cur.reshape(-1), ref.reshape(-1)
d_cur, d_ref = api.to_device(cur_f), api.to_device(ref_f)
half = 1 << 13
row = ora.mvcost_row(28, half); d_row = api.to_device(row.view(np.int16))
dt = cur_f.dtype
w, h = 8, 8
for (qx, qy) in [(-8, -7), (-7, -8), (-7, -7), (0, 0)]:
    t = np.zeros(1, ME_TASK)
    px, py = 64, 48
    off = (margin + py) * stride + margin + px
    t[0]["mvpFrom"] = -1; t[0]["curOff"] = off; t[0]["refOff"] = off; t[0]["mvmin"] = (-40, -40); t[0]["mvmax"] = (40, 40); t[0]["qmvp"] = (qx, qy)
    d_t = api.to_device(t); d_res = api.torch.zeros(8192, dtype=api.torch.uint8, device="cuda")
    api.me_batch(w, h, d_cur, stride, d_ref, stride, d_t, 1, d_row, half, 4, 0, 7, d_res)
    api.torch.cuda.synchronize()
    got = d_res.cpu().numpy().view(dt)[:w * h].reshape(h, w)
    so = off + (qx >> 2) + (qy >> 2) * stride
    xf, yf = qx & 3, qy & 3
    buf = np.zeros(w * h, dt)
    if xf == 0 and yf == 0:
        exp = np.array([ref_f[so + y * stride: so + y * stride + w] for y in range(h)])
    else:
        kind = "hpp" if yf == 0 else ("vpp" if xf == 0 else "hvpp")
        exp = ora.interp(kind, 8, w, h, ref_f, stride, so, buf, w, xf if kind != "vpp" else yf, yf).reshape(h, w)
    print("q", qx, qy, "equal", np.array_equal(got, exp))
    if not np.array_equal(got, exp):
        print(got[:4]); print(exp[:4])
