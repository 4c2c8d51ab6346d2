"""Probe: the one 64x64 STAR task that differs at 12 bit (test_me_batch_matches_oracle[3-12-True], seed 1, task 4) under every subme, with and without the phase planes."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import test_me_gpu as T
depth, method = int(sys.argv[1]) if len(sys.argv) > 1 else 12, 3
api, ora = T.FrameApi(depth), T.Oracle(depth)
rng = np.random.default_rng(77 * 12 + method)      # the geometry of the 12-bit test at any depth
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
        merange = int(rng.choice([8, 16, 57])); qp = int(rng.choice([22, 28, 37])); subme = int(rng.integers(0, 8))
        n = 24 if w * h <= 1024 else 10
        tasks = T.make_tasks(rng, W, H, margin, stride, w, h, n, dx, dy, merange)
        if not (seed == 1 and (w, h) == (64, 64)):
            continue
        row = ora.mvcost_row(qp, half)
        one = tasks[4:5].copy()
        print("task", {k: one[0][k] for k in one.dtype.names if k != "mvc"}, "mvc", one[0]["mvc"][:2 * int(one[0]["numCand"])], "shift", dx, dy)
        for sm in range(8):
            for planes in (False, True):
                d_tasks, d_row = api.to_device(one), api.to_device(row.view(np.int16))
                d_res = api.torch.zeros(T.ME_RESULT.itemsize, dtype=api.torch.uint8, device="cuda")
                api.me_batch(w, h, d_cur, stride, d_ref, stride, d_tasks, 1, d_row, half, merange, method, sm, d_res, planes=d_pl if planes else None, plane_elems=pe if planes else 0)
                api.torch.cuda.synchronize()
                res = d_res.cpu().numpy().view(T.ME_RESULT)
                tk = one[0]
                bounds = [int(tk["mvmin"][0]), int(tk["mvmin"][1]), int(tk["mvmax"][0]), int(tk["mvmax"][1])]
                mvc = [int(v) for v in tk["mvc"][:2 * int(tk["numCand"])]]
                exp = ora.me(w, h, cur_f, stride, int(tk["curOff"]), ref_f, stride, int(tk["refOff"]), bounds, (int(tk["qmvp"][0]), int(tk["qmvp"][1])), mvc, merange, method, sm, row)
                got = (int(res[0]["mv"][0]), int(res[0]["mv"][1]), int(res[0]["cost"]))
                print("subme %d planes %d: hip %s oracle %s %s" % (sm, planes, got, exp, "" if got == exp else "  <-- differs"))
