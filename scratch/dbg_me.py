import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'oracle')
import numpy as np
import x265hip
from x265hip_pkg.synth import frame_pair
from x265hip_pkg.frame import FrameApi, ME_TASK, ME_RESULT
from backends import Oracle
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 8
api, ora = FrameApi(depth), Oracle(depth)
W, H, margin = 320, 192, 96
cur, ref, stride, (dx, dy) = frame_pair(W, H, depth, 10, margin=margin, max_shift=10)
cur_f, ref_f = cur.reshape(-1), ref.reshape(-1)
d_cur, d_ref = api.to_device(cur_f), api.to_device(ref_f)
half = 1 << 13
row = ora.mvcost_row(28, half); d_row = api.to_device(row.view(np.int16))
for (w, h) in [(8, 8), (16, 16), (64, 64), (4, 8)]:
    for subme in (0, 1, 2, 7):
        for fx in range(4):
            for fy in range(4):
                t = np.zeros(1, ME_TASK)
                px, py = 64, 48
                off = (margin + py) * stride + margin + px
                qmvp = (4 * dx + fx, 4 * dy + fy)
                t[0]["mvpFrom"] = -1; t[0]["curOff"] = off; t[0]["refOff"] = off
                t[0]["mvmin"] = (-40, -40); t[0]["mvmax"] = (40, 40); t[0]["qmvp"] = qmvp
                d_t = api.to_device(t)
                d_res = api.torch.zeros(16, dtype=api.torch.uint8, device="cuda")
                for method in ([int(m) for m in os.environ.get("METHODS", "0,1").split(",")]):
                    print("run", w, h, subme, fx, fy, method, flush=True)
                    api.me_batch(w, h, d_cur, stride, d_ref, stride, d_t, 1, d_row, half, 4, method, subme, d_res)
                    api.torch.cuda.synchronize()
                    r = d_res.cpu().numpy().view(ME_RESULT)[0]
                    exp = ora.me(w, h, cur_f, stride, off, ref_f, stride, off, [-40, -40, 40, 40], qmvp, [], 4, method, subme, row)
                    got = (int(r["mv"][0]), int(r["mv"][1]), int(r["cost"]))
                    if got != exp:
                        print("MISMATCH %dx%d subme %d frac (%d,%d) method %d: hip %s oracle %s mvp %s" % (w, h, subme, fx, fy, method, got, exp, qmvp))
print("done")
