"""GPU parity of the batched motion search (x265hip_me_batch) against the restated reference driver
(oracle/x265_oracle_me.c, itself pinned to the real motion.cpp): MV and cost must be identical."""
import numpy as np
import pytest

from depths import DEPTHS

import x265hip  # noqa: F401
from x265hip_pkg.synth import frame_pair
from x265hip_pkg.frame import FrameApi, ME_TASK, ME_RESULT
from backends import Oracle

pytestmark = pytest.mark.gpu

PUS = [(8, 8), (16, 16), (32, 32), (64, 64), (16, 8), (8, 16), (32, 16), (16, 32), (64, 32), (32, 64), (16, 12), (12, 16),
       (16, 4), (4, 16), (32, 24), (24, 32), (32, 8), (8, 32), (64, 48), (48, 64), (64, 16), (16, 64), (8, 4), (4, 8)]


def make_tasks(rng, W, H, margin, stride, w, h, n, dx, dy, merange):
    t = np.zeros(n, ME_TASK)
    for i in range(n):
        px = int(rng.integers(0, (W - w) // 4 + 1)) * 4
        py = int(rng.integers(0, (H - h) // 4 + 1)) * 4
        off = (margin + py) * stride + margin + px
        r = rng.random()
        if r < 0.6:
            qmvp = (4 * dx + int(rng.integers(-9, 10)), 4 * dy + int(rng.integers(-9, 10)))
        elif r < 0.8:
            qmvp = (0, 0)
        else:
            qmvp = (int(rng.integers(-120, 121)), int(rng.integers(-120, 121)))
        lim = margin - 16
        fx, fy = qmvp[0] >> 2, qmvp[1] >> 2
        t[i]["curOff"] = off; t[i]["refOff"] = off
        t[i]["mvmin"] = (max(fx - merange, -px - lim), max(fy - merange, -py - lim))
        t[i]["mvmax"] = (min(fx + merange, W - px - w + lim), min(fy + merange, H - py - h + lim))
        t[i]["qmvp"] = qmvp
        nc = int(rng.integers(0, 5))
        t[i]["numCand"] = nc; t[i]["mvpFrom"] = -1
        t[i]["mvc"][:2 * nc] = rng.integers(-80, 81, 2 * nc)
    return t


@pytest.mark.parametrize("planes", [False, True])
@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("method", [0, 1, 2, 3, 5])      # DIA, HEX, UMH, STAR, FULL
def test_me_batch_matches_oracle(depth, method, planes):
    api, ora = FrameApi(depth), Oracle(depth)
    rng = np.random.default_rng(77 * depth + method)
    W, H, margin = 320, 192, 96
    half = 1 << 13
    for seed in range(2):
        cur, ref, stride, (dx, dy) = frame_pair(W, H, depth, 10 + seed, margin=margin, max_shift=10 if seed else 28)
        cur_f, ref_f = cur.reshape(-1), ref.reshape(-1)
        d_cur, d_ref = api.to_device(cur_f), api.to_device(ref_f)
        d_pl, pe = None, 0
        if planes:
            pe = cur_f.size
            d_pl = api.torch.zeros(16 * pe, dtype=d_ref.dtype, device="cuda")
            api.subpel_planes(d_ref, stride, cur.shape[0], d_pl, pe)
        for (w, h) in PUS:
            merange = int(rng.choice([4, 9, 16] if method == 5 else [8, 16, 57]))
            qp = int(rng.choice([22, 28, 37]))
            subme = int(rng.integers(0, 8))
            n = 24 if w * h <= 1024 else 10
            tasks = make_tasks(rng, W, H, margin, stride, w, h, n, dx, dy, merange)
            row = ora.mvcost_row(qp, half)
            d_tasks, d_row = api.to_device(tasks), api.to_device(row.view(np.int16))
            d_res = api.torch.zeros(n * ME_RESULT.itemsize, dtype=api.torch.uint8, device="cuda")
            api.me_batch(w, h, d_cur, stride, d_ref, stride, d_tasks, n, d_row, half, merange, method, subme, d_res, planes=d_pl, plane_elems=pe)
            api.torch.cuda.synchronize()
            res = d_res.cpu().numpy().view(ME_RESULT)
            for i in range(n):
                tk = tasks[i]
                bounds = [int(tk["mvmin"][0]), int(tk["mvmin"][1]), int(tk["mvmax"][0]), int(tk["mvmax"][1])]
                mvc = [int(v) for v in tk["mvc"][:2 * int(tk["numCand"])]]
                exp = ora.me(w, h, cur_f, stride, int(tk["curOff"]), ref_f, stride, int(tk["refOff"]), bounds,
                             (int(tk["qmvp"][0]), int(tk["qmvp"][1])), mvc, merange, method, subme, row)
                got = (int(res[i]["mv"][0]), int(res[i]["mv"][1]), int(res[i]["cost"]))
                assert got == exp, "PU %dx%d task %d method %d subme %d merange %d: hip %s oracle %s (mvp %s)" % (
                    w, h, i, method, subme, merange, got, exp, tk["qmvp"])


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("method", [1, 3])      # HEX, STAR
def test_me_batch_on_extreme_pictures(depth, method):
    """Pictures of 0 / PIXEL_MAX blocks: every difference is 0 or the largest the depth has -- the 16-bit lanes of the packed Hadamard transforms, the cost keys of the
    64x64 search and the sums of the sub-pel stage at their limits (a 64x64 SAD of 4096 * PIXEL_MAX, 4x4 coefficients of 16 * PIXEL_MAX)"""
    api, ora = FrameApi(depth), Oracle(depth)
    rng = np.random.default_rng(990 + depth + method)
    W, H, margin = 320, 192, 96
    half = 1 << 13
    pm = (1 << depth) - 1
    cur, ref, stride, (dx, dy) = frame_pair(W, H, depth, 31, margin=margin, max_shift=6)
    blocks = lambda b: np.kron(rng.integers(0, 2, (cur.shape[0] // b + 1, cur.shape[1] // b + 1)), np.ones((b, b), np.int64))[:cur.shape[0], :cur.shape[1]] * pm  # noqa: E731
    cur = blocks(8).astype(cur.dtype); ref = (pm - blocks(4)).astype(ref.dtype)
    ref[margin + 40:margin + 150, margin + 30:margin + 290] = pm - cur[margin + 40:margin + 150, margin + 30:margin + 290]      # a region where every pixel differs by PIXEL_MAX
    cur_f, ref_f = np.ascontiguousarray(cur).reshape(-1), np.ascontiguousarray(ref).reshape(-1)
    d_cur, d_ref = api.to_device(cur_f), api.to_device(ref_f)
    pe = cur_f.size
    d_pl = api.torch.zeros(16 * pe, dtype=d_ref.dtype, device="cuda")
    api.subpel_planes(d_ref, stride, cur.shape[0], d_pl, pe)
    for (w, h) in [(64, 64), (32, 32), (16, 16), (8, 8), (64, 32), (16, 32), (8, 4), (12, 16)]:
        for subme in (2, 4):
            merange, qp = 16, 51
            n = 12
            tasks = make_tasks(rng, W, H, margin, stride, w, h, n, dx, dy, merange)
            row = ora.mvcost_row(qp, half)
            d_tasks, d_row = api.to_device(tasks), api.to_device(row.view(np.int16))
            d_res = api.torch.zeros(n * ME_RESULT.itemsize, dtype=api.torch.uint8, device="cuda")
            api.me_batch(w, h, d_cur, stride, d_ref, stride, d_tasks, n, d_row, half, merange, method, subme, d_res, planes=d_pl, plane_elems=pe)
            api.torch.cuda.synchronize()
            res = d_res.cpu().numpy().view(ME_RESULT)
            for i in range(n):
                tk = tasks[i]
                bounds = [int(tk["mvmin"][0]), int(tk["mvmin"][1]), int(tk["mvmax"][0]), int(tk["mvmax"][1])]
                mvc = [int(v) for v in tk["mvc"][:2 * int(tk["numCand"])]]
                exp = ora.me(w, h, cur_f, stride, int(tk["curOff"]), ref_f, stride, int(tk["refOff"]), bounds, (int(tk["qmvp"][0]), int(tk["qmvp"][1])), mvc, merange, method, subme, row)
                got = (int(res[i]["mv"][0]), int(res[i]["mv"][1]), int(res[i]["cost"]))
                assert got == exp, "PU %dx%d task %d method %d subme %d: hip %s oracle %s" % (w, h, i, method, subme, got, exp)


@pytest.mark.parametrize("depth", DEPTHS)
def test_star_search_from_a_start_outside_the_window(depth):
    """The zero MV that wins the start stage is clipped in y only (motion.cpp:1000-1003): with a predictor far from zero and a small range the search starts OUTSIDE its
    window, and the star rounds -- each point tested against the window side it moves towards only (:406-448) -- cost points outside it.  The 64x64 level keeps its window
    in LDS: such a PU must take the general path (star64_body.inc: startInside).  Pictures that hardly move, predictors 10-18 pixels off, range 8."""
    api, ora = FrameApi(depth), Oracle(depth)
    rng = np.random.default_rng(4242 + depth)
    W, H, margin = 320, 192, 96
    half = 1 << 13
    cur, ref, stride, (dx, dy) = frame_pair(W, H, depth, 55, margin=margin, max_shift=1)
    cur_f, ref_f = cur.reshape(-1), ref.reshape(-1)
    d_cur, d_ref = api.to_device(cur_f), api.to_device(ref_f)
    pe = cur_f.size
    d_pl = api.torch.zeros(16 * pe, dtype=d_ref.dtype, device="cuda")
    api.subpel_planes(d_ref, stride, cur.shape[0], d_pl, pe)
    outside = 0
    for (w, h) in [(64, 64), (32, 32), (16, 16), (64, 32)]:
        merange, n = 8, 16
        tasks = make_tasks(rng, W, H, margin, stride, w, h, n, dx, dy, merange)
        for i in range(n):
            qmvp = (int(rng.choice([-1, 1])) * int(rng.integers(40, 72)), int(rng.choice([-1, 1])) * int(rng.integers(40, 72)))
            px = (int(tasks[i]["curOff"]) % stride) - margin; py = int(tasks[i]["curOff"]) // stride - margin
            fx, fy, lim = qmvp[0] >> 2, qmvp[1] >> 2, margin - 16
            tasks[i]["qmvp"] = qmvp; tasks[i]["numCand"] = 0
            tasks[i]["mvmin"] = (max(fx - merange, -px - lim), max(fy - merange, -py - lim))
            tasks[i]["mvmax"] = (min(fx + merange, W - px - w + lim), min(fy + merange, H - py - h + lim))
        row = ora.mvcost_row(28, half)
        d_tasks, d_row = api.to_device(tasks), api.to_device(row.view(np.int16))
        d_res = api.torch.zeros(n * ME_RESULT.itemsize, dtype=api.torch.uint8, device="cuda")
        api.me_batch(w, h, d_cur, stride, d_ref, stride, d_tasks, n, d_row, half, merange, 3, 2, d_res, planes=d_pl, plane_elems=pe)
        api.torch.cuda.synchronize()
        res = d_res.cpu().numpy().view(ME_RESULT)
        for i in range(n):
            tk = tasks[i]
            bounds = [int(tk["mvmin"][0]), int(tk["mvmin"][1]), int(tk["mvmax"][0]), int(tk["mvmax"][1])]
            exp = ora.me(w, h, cur_f, stride, int(tk["curOff"]), ref_f, stride, int(tk["refOff"]), bounds, (int(tk["qmvp"][0]), int(tk["qmvp"][1])), [], merange, 3, 2, row)
            got = (int(res[i]["mv"][0]), int(res[i]["mv"][1]), int(res[i]["cost"]))
            assert got == exp, "PU %dx%d task %d: hip %s oracle %s (mvp %s window %s)" % (w, h, i, got, exp, tk["qmvp"], bounds)
            outside += not (bounds[0] <= 0 <= bounds[2])
    assert outside > 40


SEA_UNDEFINED = {(8, 4), (4, 8), (8, 32), (32, 8)}      # the reference's own result is not a function of the inputs (see test_me_oracle_vs_ref.py)


@pytest.mark.parametrize("planes", [False, True])
@pytest.mark.parametrize("depth", DEPTHS)
def test_me_batch_sea_matches_oracle(depth, planes):
    """SEA: integral planes built on the device (x265hip_sea_integral_planes) + x265hip_me_batch_sea against the oracle"""
    api, ora = FrameApi(depth), Oracle(depth)
    rng = np.random.default_rng(401 * depth)
    W, H, margin = 320, 192, 96
    half = 1 << 13
    cur, ref, stride, (dx, dy) = frame_pair(W, H, depth, 21, margin=margin, max_shift=12)
    rows = cur.shape[0]
    cur_f, ref_f = cur.reshape(-1), ref.reshape(-1)
    d_cur, d_ref = api.to_device(cur_f), api.to_device(ref_f)
    d_int, ie = api.sea_integral_planes(d_ref, stride, rows)
    integral = ora.sea_integral_planes(ref_f, stride, margin * stride + margin, H, margin, margin)
    got = d_int.cpu().numpy().view(np.uint32).reshape(12, ie)
    BW = [32, 32, 32, 24, 16, 16, 16, 12, 8, 8, 4, 4]; BH = [32, 24, 8, 32, 16, 12, 4, 16, 32, 8, 16, 4]
    for k in range(12):                                   # compare where the reference defines the plane: rows 1 .. rows-BH, columns 0 .. stride-BW-1
        a = got[k].reshape(rows, stride)[1:rows - BH[k], :stride - BW[k]]
        b = integral[k].reshape(rows, stride)[1:rows - BH[k], :stride - BW[k]]
        assert np.array_equal(a, b), "integral plane %d" % k
    d_pl, pe = None, 0
    if planes:
        pe = cur_f.size
        d_pl = api.torch.zeros(16 * pe, dtype=d_ref.dtype, device="cuda")
        api.subpel_planes(d_ref, stride, rows, d_pl, pe)
    for (w, h) in PUS:
        if (w, h) in SEA_UNDEFINED:
            with pytest.raises(RuntimeError):
                api.me_batch_sea(w, h, d_cur, stride, d_ref, stride, d_cur, 1, d_cur, half, 8, 2, d_cur, d_int, ie)
            continue
        merange = int(rng.choice([6, 12, 20]))
        qp = int(rng.choice([22, 28, 37])); subme = int(rng.integers(0, 8))
        n = 12 if w * h <= 1024 else 6
        tasks = make_tasks(rng, W, H, margin, stride, w, h, n, dx, dy, merange)
        row = ora.mvcost_row(qp, half)
        d_tasks, d_row = api.to_device(tasks), api.to_device(row.view(np.int16))
        d_res = api.torch.zeros(n * ME_RESULT.itemsize, dtype=api.torch.uint8, device="cuda")
        api.me_batch_sea(w, h, d_cur, stride, d_ref, stride, d_tasks, n, d_row, half, merange, subme, d_res, d_int, ie, planes=d_pl, plane_elems=pe)
        api.torch.cuda.synchronize()
        res = d_res.cpu().numpy().view(ME_RESULT)
        for i in range(n):
            tk = tasks[i]
            bounds = [int(tk["mvmin"][0]), int(tk["mvmin"][1]), int(tk["mvmax"][0]), int(tk["mvmax"][1])]
            mvc = [int(v) for v in tk["mvc"][:2 * int(tk["numCand"])]]
            exp = ora.me(w, h, cur_f, stride, int(tk["curOff"]), ref_f, stride, int(tk["refOff"]), bounds,
                         (int(tk["qmvp"][0]), int(tk["qmvp"][1])), mvc, merange, 4, subme, row, integral=integral)
            got_i = (int(res[i]["mv"][0]), int(res[i]["mv"][1]), int(res[i]["cost"]))
            assert got_i == exp, "SEA PU %dx%d task %d subme %d merange %d: hip %s oracle %s (mvp %s)" % (w, h, i, subme, merange, got_i, exp, tk["qmvp"])


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("method", [1, 3])               # HEX, STAR (with its raster over the whole +-128 window)
def test_me_batch_merange_128(depth, method):
    """BASELINE configs[4] searches with --merange 128: MVDs reach the edge of (and leave) the +-512 quarter-pel cost slice the kernels keep
    in LDS, the STAR raster covers 52 x 52 placements, and the `tmv << 3` quirk indexes the cost row far outside the slice."""
    api, ora = FrameApi(depth), Oracle(depth)
    rng = np.random.default_rng(1280 + depth + method)
    W, H, margin, merange = 448, 320, 160, 128
    half = 1 << 14
    cur, ref, stride, (dx, dy) = frame_pair(W, H, depth, 55, margin=margin, max_shift=40)
    cur_f, ref_f = cur.reshape(-1), ref.reshape(-1)
    d_cur, d_ref = api.to_device(cur_f), api.to_device(ref_f)
    pe = cur_f.size
    row = ora.mvcost_row(28, half)
    d_row = api.to_device(row.view(np.int16))
    for planes in (True, False):
        d_pl = None
        if planes:
            d_pl = api.torch.zeros(16 * pe, dtype=d_ref.dtype, device="cuda")
            api.subpel_planes(d_ref, stride, cur.shape[0], d_pl, pe)
        for (w, h) in [(64, 64), (32, 32), (16, 16), (8, 8), (64, 32), (16, 64)]:
            if not planes and w * h < 1024:
                continue
            n = 10 if planes else 4
            subme = int(rng.integers(2, 5))
            tasks = make_tasks(rng, W, H, margin, stride, w, h, n, dx, dy, merange)
            for i in range(0, n, 2):                      # every other task: a far-off predictor, so the first STAR pass lands far away and the raster runs
                px_, py_ = (int(tasks[i]["curOff"]) % stride) - margin, (int(tasks[i]["curOff"]) // stride) - margin
                q = (int(rng.integers(-300, 301)), int(rng.integers(-300, 301)))
                lim = margin - 16
                tasks[i]["qmvp"] = q
                tasks[i]["mvmin"] = (max((q[0] >> 2) - merange, -px_ - lim), max((q[1] >> 2) - merange, -py_ - lim))
                tasks[i]["mvmax"] = (min((q[0] >> 2) + merange, W - px_ - w + lim), min((q[1] >> 2) + merange, H - py_ - h + lim))
            d_tasks = api.to_device(tasks)
            d_res = api.torch.zeros(n * ME_RESULT.itemsize, dtype=api.torch.uint8, device="cuda")
            api.me_batch(w, h, d_cur, stride, d_ref, stride, d_tasks, n, d_row, half, merange, method, subme, d_res, planes=d_pl, plane_elems=pe if planes else 0)
            api.torch.cuda.synchronize()
            res = d_res.cpu().numpy().view(ME_RESULT)
            for i in range(n):
                tk = tasks[i]
                bounds = [int(tk["mvmin"][0]), int(tk["mvmin"][1]), int(tk["mvmax"][0]), int(tk["mvmax"][1])]
                mvc = [int(v) for v in tk["mvc"][:2 * int(tk["numCand"])]]
                exp = ora.me(w, h, cur_f, stride, int(tk["curOff"]), ref_f, stride, int(tk["refOff"]), bounds,
                             (int(tk["qmvp"][0]), int(tk["qmvp"][1])), mvc, merange, method, subme, row)
                got = (int(res[i]["mv"][0]), int(res[i]["mv"][1]), int(res[i]["cost"]))
                assert got == exp, "merange 128: PU %dx%d task %d method %d subme %d planes %s: hip %s oracle %s (mvp %s)" % (
                    w, h, i, method, subme, planes, got, exp, tk["qmvp"])


def test_me_batch_plane_buffer_beyond_4gb():
    """A reference stack whose 16-slot phase-plane buffer is larger than 4 GB (4 padded 8K 10-bit pictures: 4.6 GB) cannot be addressed
    with the 32-bit byte offsets of the size-specialised kernels; x265hip_me_batch must take the generic kernels and still be exact.
    The searched picture is the LAST of the stack, so every address lies beyond 2^32 bytes in the higher slots."""
    depth = 10
    api, ora = FrameApi(depth), Oracle(depth)
    T = api.torch
    rng = np.random.default_rng(4096)
    W, H, margin, F = 7680, 4352, 96, 4
    c4, r4, _, (dx, dy) = frame_pair(W // 4, H // 4, depth, 77, margin=0, max_shift=20)      # one quarter-size pair, tiled 4 x 4: same motion everywhere
    cur = np.pad(np.tile(c4, (4, 4)), margin, mode="edge"); ref = np.pad(np.tile(r4, (4, 4)), margin, mode="edge")
    stride, rows = W + 2 * margin, H + 2 * margin
    plane = stride * rows
    cur_f, ref_f = np.ascontiguousarray(cur).reshape(-1), np.ascontiguousarray(ref).reshape(-1)
    d_cur = api.to_device(cur_f)
    d_ref = T.zeros(F * plane, dtype=d_cur.dtype, device="cuda")
    d_ref[(F - 1) * plane:].copy_(api.to_device(ref_f))
    pe = F * plane
    assert 16 * pe * 2 >= 1 << 32
    d_pl = T.empty(16 * pe, dtype=d_ref.dtype, device="cuda")
    api.subpel_planes(d_ref, stride, F * rows, d_pl, pe)
    half = 1 << 13
    row = ora.mvcost_row(28, half)
    d_row = api.to_device(row.view(np.int16))
    for (w, h) in [(64, 64), (32, 32), (16, 16), (8, 8)]:
        n = 12
        tasks = make_tasks(rng, W, H, margin, stride, w, h, n, dx, dy, 57)
        host = tasks.copy()
        tasks["refOff"] += (F - 1) * plane
        d_tasks = api.to_device(tasks)
        d_res = T.zeros(n * ME_RESULT.itemsize, dtype=T.uint8, device="cuda")
        api.me_batch(w, h, d_cur, stride, d_ref, stride, d_tasks, n, d_row, half, 57, 3, 3, d_res, planes=d_pl, plane_elems=pe)
        T.cuda.synchronize()
        res = d_res.cpu().numpy().view(ME_RESULT)
        for i in range(n):
            tk = host[i]
            bounds = [int(tk["mvmin"][0]), int(tk["mvmin"][1]), int(tk["mvmax"][0]), int(tk["mvmax"][1])]
            mvc = [int(v) for v in tk["mvc"][:2 * int(tk["numCand"])]]
            exp = ora.me(w, h, cur_f, stride, int(tk["curOff"]), ref_f, stride, int(tk["refOff"]), bounds,
                         (int(tk["qmvp"][0]), int(tk["qmvp"][1])), mvc, 57, 3, 3, row)
            got = (int(res[i]["mv"][0]), int(res[i]["mv"][1]), int(res[i]["cost"]))
            assert got == exp, ">4GB planes: PU %dx%d task %d: hip %s oracle %s" % (w, h, i, got, exp)


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("method", [0, 1, 3, 5])        # DIA, HEX, STAR, FULL
def test_me_batch_chroma_matches_oracle(depth, method):
    """x265hip_me_batch_chroma (the predInterSearch call form: chroma SATD terms in every sub-pel cost, motion.cpp:1805-1865) against the oracle, which
    is pinned on recorded reference searches (test_tme_golden.py): all 24 PU shapes (the terms apply where the 4:2:0 block is a multiple of 4x4),
    subme 3..7 (and 2, where bChromaSATD stays off), random predictors and candidates."""
    api, ora = FrameApi(depth), Oracle(depth)
    T = api.torch
    rng = np.random.default_rng(911 * depth + method)
    W, H, margin = 320, 192, 96
    half = 1 << 13
    cur, ref, stride, (dx, dy) = frame_pair(W, H, depth, 31, margin=margin, max_shift=12)
    # chroma planes: any content does for parity; half-size pictures with half the margin, the same layout rule
    cm = margin // 2
    ccb, rcb, cstr, _ = frame_pair(W // 2, H // 2, depth, 32, margin=cm, max_shift=6)
    ccr, rcr, _, _ = frame_pair(W // 2, H // 2, depth, 33, margin=cm, max_shift=6)
    cur_f, ref_f = cur.reshape(-1), ref.reshape(-1)
    cc = (ccb.reshape(-1), ccr.reshape(-1)); rc = (rcb.reshape(-1), rcr.reshape(-1))
    d_cur, d_ref = api.to_device(cur_f), api.to_device(ref_f)
    d_cc = [api.to_device(x) for x in cc]; d_rc = [api.to_device(x) for x in rc]
    pe = cur_f.size
    d_pl = T.zeros(16 * pe, dtype=d_ref.dtype, device="cuda")
    api.subpel_planes(d_ref, stride, cur.shape[0], d_pl, pe)
    for (w, h) in PUS:
        merange = int(rng.choice([4, 9] if method == 5 else [8, 16, 57]))
        qp = int(rng.choice([22, 28, 37]))
        subme = int(rng.integers(2, 8))
        n = 16 if w * h <= 1024 else 8
        tasks = make_tasks(rng, W, H, margin, stride, w, h, n, dx, dy, merange)
        offc = np.zeros(n, np.int32)
        for i in range(n):
            px, py = (int(tasks[i]["curOff"]) % stride) - margin, (int(tasks[i]["curOff"]) // stride) - margin
            offc[i] = (cm + py // 2) * cstr + cm + px // 2
        row = ora.mvcost_row(qp, half)
        d_tasks, d_row, d_off = api.to_device(tasks), api.to_device(row.view(np.int16)), api.to_device(offc)
        d_res = T.zeros(n * ME_RESULT.itemsize, dtype=T.uint8, device="cuda")
        api.me_batch_chroma(w, h, d_cur, stride, d_ref, stride, d_tasks, n, d_row, half, merange, method, subme, d_res, d_pl, pe,
                            d_cc[0], d_cc[1], cstr, d_rc[0], d_rc[1], cstr, d_off, d_off)
        T.cuda.synchronize()
        res = d_res.cpu().numpy().view(ME_RESULT)
        for i in range(n):
            tk = tasks[i]
            bounds = [int(tk["mvmin"][0]), int(tk["mvmin"][1]), int(tk["mvmax"][0]), int(tk["mvmax"][1])]
            mvc = [int(v) for v in tk["mvc"][:2 * int(tk["numCand"])]]
            exp = ora.me_chroma(w, h, cur_f, stride, int(tk["curOff"]), ref_f, stride, int(tk["refOff"]), bounds, (int(tk["qmvp"][0]), int(tk["qmvp"][1])), mvc,
                                merange, method, subme, row, cc, cstr, int(offc[i]), rc, cstr, int(offc[i]))
            got = (int(res[i]["mv"][0]), int(res[i]["mv"][1]), int(res[i]["cost"]))
            assert got == exp, "chroma: PU %dx%d task %d method %d subme %d merange %d: hip %s oracle %s (mvp %s)" % (w, h, i, method, subme, merange, got, exp, tk["qmvp"])


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("method", [1, 3])
def test_me_batch_cost_row_per_task(depth, method):
    """X265HIP_ME_ROWS: PUs of CUs with different qps in one launch -- every task names its row of a cost table; the results must be those of the oracle searching
    with that qp's row (sizes incl. 64x64, whose STAR search goes through the window kernel)."""
    from x265hip_pkg.frame import mvcost_row
    api, ora = FrameApi(depth), Oracle(depth)
    T = api.torch
    rng = np.random.default_rng(17 * depth + method)
    W, H, margin, half = 320, 192, 96, 1 << 14
    cur, ref, stride, (dx, dy) = frame_pair(W, H, depth, 77, margin=margin, max_shift=12)
    cur_f, ref_f = cur.reshape(-1), ref.reshape(-1)
    d_cur, d_ref = api.to_device(cur_f), api.to_device(ref_f)
    pe = cur_f.size
    d_pl = T.zeros(16 * pe, dtype=d_ref.dtype, device="cuda")
    api.subpel_planes(d_ref, stride, cur.shape[0], d_pl, pe)
    qps = [22, 30, 37, 45]
    rows = [mvcost_row(depth, q, half) for q in qps]
    d_table = api.to_device(np.concatenate(rows).view(np.int16))
    for (w, h) in [(64, 64), (32, 32), (16, 8), (8, 8)]:
        n = 24
        tasks = make_tasks(rng, W, H, margin, stride, w, h, n, dx, dy, 24)
        which = rng.integers(0, len(qps), n)
        tasks["flags"] = 2 | (which.astype(np.int16) << 8)                  # X265HIP_ME_ROWS | row << 8
        d_tasks = api.to_device(tasks)
        d_res = T.zeros(n * ME_RESULT.itemsize, dtype=T.uint8, device="cuda")
        api.me_batch_rows(w, h, d_cur, stride, d_ref, stride, d_tasks, n, d_table, half, 24, method, 3, d_res, planes=d_pl, plane_elems=pe)
        T.cuda.synchronize()
        res = d_res.cpu().numpy().view(ME_RESULT)
        for i in range(n):
            tk = tasks[i]
            bounds = [int(tk["mvmin"][0]), int(tk["mvmin"][1]), int(tk["mvmax"][0]), int(tk["mvmax"][1])]
            mvc = [int(v) for v in tk["mvc"][:2 * int(tk["numCand"])]]
            exp = ora.me(w, h, cur_f, stride, int(tk["curOff"]), ref_f, stride, int(tk["refOff"]), bounds, (int(tk["qmvp"][0]), int(tk["qmvp"][1])), mvc, 24, method, 3, rows[int(which[i])])
            got = (int(res[i]["mv"][0]), int(res[i]["mv"][1]), int(res[i]["cost"]))
            assert got == exp, "row per task: PU %dx%d task %d qp %d: hip %s oracle %s" % (w, h, i, qps[int(which[i])], got, exp)
