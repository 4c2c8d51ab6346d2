"""Probe: test_me_batch_matches_oracle's loop at one depth / method with the phase planes, every mismatch printed (not only the first)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import test_me_gpu as T
from test_me_gpu import *
depth, method = int(sys.argv[1]), int(sys.argv[2])
api, ora = T.FrameApi(depth), T.Oracle(depth)
rng = np.random.default_rng(77 * depth + method)
W, H, margin = 320, 192, 96
half = 1 << 13
for seed in range(2):
    cur, ref, stride, (dx, dy) = T.frame_pair(W, H, depth, 10 + seed, margin=margin, max_shift=10 if seed else 28)
    cur_f, ref_f = cur.reshape(-1), ref.reshape(-1)
    d_cur, d_ref = api.to_device(cur_f), api.to_device(ref_f)
    pe = cur_f.size
    d_pl = api.torch.zeros(16 * pe, dtype=d_ref.dtype, device="cuda")
    api.subpel_planes(d_ref, stride, cur.shape[0], d_pl, pe)
    for (w, h) in T.PUS:
        merange = int(rng.choice([4, 9, 16] if method == 5 else [8, 16, 57])); qp = int(rng.choice([22, 28, 37])); subme = int(rng.integers(0, 8))
        n = 24 if w * h <= 1024 else 10
        tasks = T.make_tasks(rng, W, H, margin, stride, w, h, n, dx, dy, merange)
        row = ora.mvcost_row(qp, half)
        d_tasks, d_row = api.to_device(tasks), api.to_device(row.view(np.int16))
        d_res = api.torch.zeros(n * T.ME_RESULT.itemsize, dtype=api.torch.uint8, device="cuda")
        api.me_batch(w, h, d_cur, stride, d_ref, stride, d_tasks, n, d_row, half, merange, method, subme, d_res, planes=d_pl, plane_elems=pe)
        api.torch.cuda.synchronize()
        res = d_res.cpu().numpy().view(T.ME_RESULT)
        bad = 0
        for i in range(n):
            tk = tasks[i]
            bounds = [int(tk["mvmin"][0]), int(tk["mvmin"][1]), int(tk["mvmax"][0]), int(tk["mvmax"][1])]
            mvc = [int(v) for v in tk["mvc"][:2 * int(tk["numCand"])]]
            exp = ora.me(w, h, cur_f, stride, int(tk["curOff"]), ref_f, stride, int(tk["refOff"]), bounds, (int(tk["qmvp"][0]), int(tk["qmvp"][1])), mvc, merange, method, subme, row)
            got = (int(res[i]["mv"][0]), int(res[i]["mv"][1]), int(res[i]["cost"]))
            if got != exp:
                bad += 1
                print("  seed %d PU %dx%d task %d subme %d merange %d qp %d: hip %s oracle %s mvp %s" % (seed, w, h, i, subme, merange, qp, got, exp, tk["qmvp"]))
        print("seed %d PU %dx%d subme %d merange %d: %d / %d differ" % (seed, w, h, subme, merange, bad, n), flush=True)
