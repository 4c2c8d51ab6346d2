"""The C++ host of the frame-batched path (include/x265hip_ctx.h: x265hip_batch_*, csrc/xh_ctx.cpp) with everything preset slow asks of the motion search in ONE run:
several list-0 references, the rectangular PUs of every CU, sub-batches of whole pictures on their own streams -- exhaustively against the oracle at 256x128 AND at the full size of the three BASELINE workloads (one reference, square PUs: every search and every TU of whole 1080p / 4K / 8K pictures), sampled at 4K with the preset's four references + rectangles, equal to the Python
plumbing (FramePipeline) where that can run the same configuration, and independent of how the batch is cut into streams."""
import ctypes as C

import numpy as np
import pytest

from depths import DEPTHS

import x265hip
from x265hip_pkg.frame import mvcost_row, mvbits_row, rd_lambda
from x265hip_pkg.host_batch import HostBatch, LEVELS
from x265hip_pkg.pipeline import FramePipeline
from x265hip_pkg.synth import frame_pair
from backends import Oracle
from pipeline_check import check_host_batch

pytestmark = pytest.mark.gpu


def pairs_for(W, H, depth, F, refs, seed0=300, refs1=0):
    out = []
    for s in range(F):
        cur, ref, _, _ = frame_pair(W, H, depth, seed=seed0 + s, margin=96, max_shift=14)
        p = [cur, ref]
        rng = np.random.default_rng(9000 + s)
        for r in range(1, refs):
            # an older picture of the same scene: the picture area displaced a little further + its own noise, borders replicated as extendPicBorder does
            inner = np.roll(ref[96:96 + H, 96:96 + W], (2 * r, -3 * r), (0, 1)).astype(np.int32) + rng.integers(-3 * r, 3 * r + 1, (H, W)) * (1 << (depth - 8))
            inner = np.clip(inner, 0, (1 << depth) - 1).astype(ref.dtype)
            p.append(np.pad(inner, 96, mode="edge"))
        for r in range(refs1):
            # list 1: later pictures of the scene -- displaced the other way
            inner = np.roll(ref[96:96 + H, 96:96 + W], (-2 * (r + 1), 3 * (r + 1) - 1), (0, 1)).astype(np.int32) + rng.integers(-2 * (r + 1), 2 * (r + 1) + 1, (H, W)) * (1 << (depth - 8))
            inner = np.clip(inner, 0, (1 << depth) - 1).astype(ref.dtype)
            p.append(np.pad(inner, 96, mode="edge"))
        out.append(tuple(p))
    return out


def make(depth, W, H, F, **kw):
    lib = x265hip.HipLib(depth, fill_table=False).lib          # (maps torch's HIP runtime before the library: the Python plumbing of the same step runs in this process too)
    return HostBatch(lib, depth, W, H, F, **kw)


# (depth, method, subme, list-0 references, rect, streams, amp, list-1 references): P pictures as the presets medium .. slower define them, then B pictures
@pytest.mark.parametrize("depth,method,subme,refs,rect,streams,amp,refs1", [(8, 1, 2, 1, False, 1, False, 0), (10, 3, 3, 1, True, 1, False, 0), (8, 3, 3, 3, False, 2, False, 0),
                                                                            (10, 3, 3, 4, True, 2, False, 0), (8, 1, 2, 2, True, 3, False, 0), (8, 3, 4, 5, True, 1, False, 0),
                                                                            (10, 3, 4, 2, True, 2, True, 0), (8, 3, 3, 1, False, 1, True, 0),
                                                                            (8, 1, 2, 1, False, 2, False, 1), (10, 3, 3, 2, True, 1, False, 2), (8, 3, 3, 2, True, 2, True, 1)])
def test_host_batch_matches_oracle(depth, method, subme, refs, rect, streams, amp, refs1):
    W, H, F, qp, merange = 256, 128, 3, 28, 24
    hb = make(depth, W, H, F, qp=qp, merange=merange, method=method, subme=subme, tu_log2=4, refs=refs, rect=rect, streams=streams, amp=amp, refs1=refs1)
    try:
        pairs = pairs_for(W, H, depth, F, refs, refs1=refs1)
        hb.upload(pairs)
        for f in (0, F - 1):        # the device pads the picture the way the host-padded planes are padded
            for which in range(1 + refs + refs1):
                assert np.array_equal(hb.device_plane(which, f), pairs[f][which].reshape(-1)), "plane %d of picture %d" % (which, f)
        hb.step(); hb.sync()
        # EXHAUSTIVE at this size: every PU of every shape in every reference, every choice, every TU against the oracle (seconds of CPU)
        n = check_host_batch(hb, Oracle(depth), np.random.default_rng(depth + refs), mvcost_row(depth, qp, 1 << 15), mvbits_row(depth, 1 << 14), rd_lambda(depth, qp),
                             per_shape=1 << 30, n_tu=1 << 30)
        ctus = F * (W // 64) * (H // 64)
        assert n == ctus * (85 + (340 if rect else 0) + (168 if amp else 0)) + F * (W // 16) * (H // 16)
        if refs1 and (rect or amp):         # the B picture's split PUs really took the bidirectional candidate somewhere (and a 2Nx2N PU never)
            both = sum(int(((c["ref"][:, 0] >= 0) & (c["ref"][:, 1] >= 0)).sum()) for c in (hb.choices(w, h) for (w, h) in list(hb.rect_host) + list(hb.amp_host) if max(w, h) > 8))
            assert both > 0
            assert all(int(((c["ref"][:, 0] >= 0) & (c["ref"][:, 1] >= 0)).sum()) == 0 for c in (hb.choices(lv) for lv in LEVELS))
    finally:
        hb.close()


@pytest.mark.parametrize("depth", DEPTHS)
def test_streams_do_not_change_the_bytes_and_python_plumbing_agrees(depth):
    """pictures are independent: 1, 2, 3 and 5 sub-batches of a 5-picture batch give the same records and coefficients; so does the Python pipeline (torch tensors, one stream)"""
    W, H, F, qp, merange, method, subme = 192, 128, 5, 30, 20, 3, 3
    pairs = pairs_for(W, H, depth, F, 2)
    ref_out = None
    # (mode 16: the phase planes in groups of two pictures -- 2 + 2 + 1 -- with two references, the rectangles, the choice among references and the TQ stage behind them)
    for streams, band, mode in ((1, 0, 0), (2, 0, 0), (3, 0, 0), (5, 0, 0), (1, 0, 16), (2, 0, 16), (3, 0, 16)):
        hb = make(depth, W, H, F, qp=qp, merange=merange, method=method, subme=subme, tu_log2=5, refs=2, rect=True, streams=streams, band_rows=band)
        try:
            hb.set_fused(mode)
            hb.upload(pairs)
            hb.step(); hb.step(); hb.sync()                     # twice: a second pass over the resident planes gives the same bytes
            out = [hb.results(lv, r).tobytes() for lv in LEVELS for r in range(2)] + [hb.rect_results(w, h, r).tobytes() for (w, h) in sorted(hb.rect_host) for r in range(2)]
            out += [hb.choices(lv).tobytes() for lv in LEVELS] + [hb.choices(w, h).tobytes() for (w, h) in sorted(hb.rect_host)]
            co, ns = hb.coeffs()
            out += [co.tobytes(), ns.tobytes()]
        finally:
            hb.close()
        if ref_out is None:
            ref_out = out
        else:
            assert out == ref_out, "%d streams / mode %d change the results" % (streams, mode)
    pipe = FramePipeline(depth, W, H, F, qp=qp, merange=merange, method=method, subme=subme, tu_log2=5, cost_row=mvcost_row(depth, qp, 1 << 15), refs=2)
    pipe.upload(pairs); pipe.step(); pipe.torch.cuda.synchronize()
    k = 0
    for lv in LEVELS:
        for r in range(2):
            assert pipe.results(lv, r).tobytes() == ref_out[k], "level %d reference %d" % (lv, r)
            k += 1
    assert pipe.d_coeff.cpu().numpy().tobytes() == ref_out[-2] and pipe.d_numsig.cpu().numpy().astype(np.uint32).tobytes() == ref_out[-1]


def test_two_streams_are_joined_when_something_reads_and_one_stream_steps_give_the_same():
    """streams = 2 alternates the 64x64 level between its two streams and leaves them unjoined between steps: a read straight behind the steps (no sync call) must see both
    sub-batches' results; x265hip_batch_step_one_stream on the same batch gives the same bytes, and the two kinds of step may be mixed"""
    W, H, F = 192, 128, 6
    pairs = pairs_for(W, H, 10, F, 1)
    outs = []
    for streams, plan in ((1, "ss"), (2, "ssss"), (2, "sos"), (2, "o"), (2, "soso")):
        hb = make(10, W, H, F, qp=30, merange=20, method=3, subme=3, tu_log2=5, streams=streams)
        try:
            hb.upload(pairs)
            for c in plan:
                hb.step() if c == "s" else hb.step_one_stream()
            out = [hb.results(lv).tobytes() for lv in LEVELS]                    # no sync in between: the reads join the streams themselves
            co, ns = hb.coeffs()
            outs.append(out + [co.tobytes(), ns.tobytes()])
            hb.upload(pairs)                                                     # an upload behind unjoined passes waits for them too
            hb.step(); hb.sync()
            assert [hb.results(lv).tobytes() for lv in LEVELS] == out
        finally:
            hb.close()
    assert all(o == outs[0] for o in outs[1:])


@pytest.mark.parametrize("depth", DEPTHS)
def test_launch_forms_give_the_same_bytes_and_the_library_refuses_the_dropped_experiments(depth):
    """x265hip_batch_set_mode: the 64x64 level with or without its start-stage launch gives the same bytes; the measured-loss forms of earlier rounds (fused lower levels,
    tiled phase planes, band-major schedule: profiles/r03_fused_ab.txt, r03_tiled_ab.txt, r03_band_major_ab.txt) are gone and the library says so instead of silently
    running something else"""
    W, H, F = 320, 192, 3
    pairs = pairs_for(W, H, depth, F, 1, seed0=520)
    outs = []
    hb = make(depth, W, H, F, qp=27, merange=57, method=3, subme=3, tu_log2=5)
    try:
        for flags in (1, 2, 8, 12):
            with pytest.raises(RuntimeError, match="measured losses"):
                hb.set_fused(flags)
    finally:
        hb.close()
    with pytest.raises(RuntimeError):
        make(depth, W, H, F, qp=27, merange=57, method=3, subme=3, tu_log2=5, band_rows=2)
    # (mode 4: the 64x64 level with its start-stage launch, the form of rounds 1-2; mode 16: the phase planes in groups of two pictures -- what a batch does by itself when its
    #  plane buffer outgrows the kernels' 32-bit offsets, 8K 10 bit beyond two pictures -- here 2 + 1 pictures)
    for mode, streams in ((0, 1), (0, 2), (4, 1), (4, 2), (16, 1), (16, 2), (20, 2)):
        hb = make(depth, W, H, F, qp=27, merange=57, method=3, subme=3, tu_log2=5, streams=streams)
        try:
            hb.set_fused(mode)
            hb.upload(pairs)
            hb.step(); hb.step(); hb.sync()
            co, ns = hb.coeffs()
            outs.append([hb.results(lv).tobytes() for lv in LEVELS] + [co.tobytes(), ns.tobytes()])
        finally:
            hb.close()
    assert all(o == outs[0] for o in outs[1:])


def test_stage_timing_and_names():
    hb = make(8, 256, 128, 4, method=3, subme=3, merange=24, rect=True, streams=2)
    try:
        hb.upload(pairs_for(256, 128, 8, 4, 1))
        assert hb.kernel_names() == ["planes", "me64", "rect64", "me32", "rect32", "me16", "rect16", "me8", "rect8", "tq"]
        hb.set_timing(True)
        for _ in range(3):
            hb.step()
        k = hb.read_kernel_timing()                              # star64_kernel alone: inside its stage
        t = hb.read_timing()
        assert set(t) == set(hb.kernel_names()) and all(0 < v < 50 for v in t.values()), t
        assert k is not None and 0 < k <= t["me64"] * 1.05, (k, t)
        hb.set_timing(False); hb.step(); hb.sync()
        with pytest.raises(RuntimeError):
            hb.read_timing()                                     # nothing was timed since the last read
        assert hb.read_kernel_timing() is None
    finally:
        hb.close()


def test_preset_slow_search_at_4k_10bit_full_size():
    """BASELINE configs[2] with the preset's own search load in one run of the C++ host: 3840x2176 10-bit, STAR, subme 3, merange 57, 4 list-0 references, the rectangular
    PUs of every CU (425 PUs per CTU), 2 pictures on 2 streams: a second pass gives the same bytes, and PUs of every shape and reference, the choices among the
    references and TUs, sampled at random, equal the oracle's."""
    depth, W, H, F, refs = 10, 3840, 2176, 2, 4
    hb = make(depth, W, H, F, qp=28, merange=57, method=3, subme=3, tu_log2=5, refs=refs, rect=True, streams=2)
    try:
        hb.upload(pairs_for(W, H, depth, F, refs, seed0=700))
        hb.step(); hb.sync()
        first = [hb.results(lv, r).tobytes() for lv in LEVELS for r in range(refs)] + [hb.choices(w, h).tobytes() for (w, h) in sorted(hb.rect_host)]
        co1, ns1 = hb.coeffs()
        hb.step(); hb.sync()
        assert first == [hb.results(lv, r).tobytes() for lv in LEVELS for r in range(refs)] + [hb.choices(w, h).tobytes() for (w, h) in sorted(hb.rect_host)]
        co2, ns2 = hb.coeffs()
        assert np.array_equal(co1, co2) and np.array_equal(ns1, ns2)
        n = check_host_batch(hb, Oracle(depth), np.random.default_rng(17), mvcost_row(depth, 28, 1 << 15), mvbits_row(depth, 1 << 14), rd_lambda(depth, 28), per_shape=5, n_tu=10)
        assert n >= 12 * 5 + 10
    finally:
        hb.close()


def test_8k_batch_keeps_its_planes_in_groups_of_pictures():
    """BASELINE configs[4]'s picture size with FOUR pictures in one batch: the 16-slot plane buffer (4.6 GB per reference) outgrows the 32-bit byte offsets of the size-specialised
    kernels, so the batch keeps its planes in groups of two pictures (csrc/xh_ctx.cpp, plane groups) instead of falling back to the generic kernels.  PUs of every level and TUs
    sampled over all four pictures equal the oracle's, and a second pass gives the same bytes."""
    depth, W, H, F = 10, 7680, 4352, 4
    hb = make(depth, W, H, F, qp=28, merange=128, method=3, subme=4, tu_log2=5, refs=1, rect=False, streams=2)
    try:
        hb.upload(pairs_for(W, H, depth, F, 1, seed0=1300))
        hb.step(); hb.sync()
        first = [hb.results(lv).tobytes() for lv in LEVELS]
        co1, ns1 = hb.coeffs()
        hb.step(); hb.sync()
        assert first == [hb.results(lv).tobytes() for lv in LEVELS]
        co2, ns2 = hb.coeffs()
        assert np.array_equal(co1, co2) and np.array_equal(ns1, ns2)
        n = check_host_batch(hb, Oracle(depth), np.random.default_rng(23), mvcost_row(depth, 28, 1 << 15), per_shape=24, n_tu=24)
        assert n >= 4 * 24 + 24
        # the samples must reach the second group (pictures 2 and 3)
        per64 = (W // 64) * (H // 64)
        assert len(hb.tasks_host[64]) == F * per64
    finally:
        hb.close()


# ---- the headline configuration, EVERY PU and EVERY TU of whole 4K pictures against the oracle (the oracle on all host cores: forked workers over slices of the task lists) ----
_FULL = {}


def _full_pus(arg):
    lv, lo, hi = arg
    hb, res, parent, row = _FULL["hb"], _FULL["res"][lv], _FULL["res"].get(2 * lv), _FULL["row"]
    ora = _FULL["ora"]
    t, d = hb["tasks"][lv], hb["merange"] << 2
    bad = []
    for i in range(lo, hi):
        tk = t[i]
        qmvp = (0, 0) if tk["mvpFrom"] < 0 else (int(parent[tk["mvpFrom"]]["mv"][0]), int(parent[tk["mvpFrom"]]["mv"][1]))
        lx0, ly0, lx1, ly1 = int(tk["mvmin"][0]), int(tk["mvmin"][1]), int(tk["mvmax"][0]), int(tk["mvmax"][1])
        b = [min(lx1, max(lx0, qmvp[0] - d)) >> 2, min(ly1, max(ly0, qmvp[1] - d)) >> 2, min(lx1, max(lx0, qmvp[0] + d)) >> 2, min(ly1, max(ly0, qmvp[1] + d)) >> 2]
        b[3] = max(b[3], b[1])
        exp = ora.me(lv, lv, hb["cur"], hb["stride"], int(tk["curOff"]), hb["ref"], hb["stride"], int(tk["refOff"]), b, qmvp, [], hb["merange"], hb["method"], hb["subme"], row)
        g = res[i]
        if (int(g["mv"][0]), int(g["mv"][1]), int(g["cost"])) != exp:
            bad.append((lv, i, (int(g["mv"][0]), int(g["mv"][1]), int(g["cost"])), exp))
    return hi - lo, bad


def _full_tus(arg):
    lo, hi = arg
    hb, ora = _FULL["hb"], _FULL["ora"]
    mvs, coeff, numsig = _FULL["res"][hb["mv_level"]], _FULL["coeff"], _FULL["numsig"]
    bad = []
    for i in range(lo, hi):
        tk = hb["tu"][i]
        mv = (int(mvs[tk["mvFrom"]]["mv"][0]), int(mvs[tk["mvFrom"]]["mv"][1]))
        e_ns, e_coeff, _, _, _ = ora.tq_tu(hb["tu_log2"], hb["cur"], hb["stride"], int(tk["curOff"]), hb["ref"], hb["stride"], int(tk["refOff"]), mv, hb["qp"], 85)
        if int(numsig[i]) != e_ns or not np.array_equal(coeff[i], e_coeff):
            bad.append(i)
    return hi - lo, bad


# BASELINE configs[1], [2] (= [3] per GPU) and [4] as bench.py's workloads define them: (depth, W, H, pictures, method, subme, merange)
@pytest.mark.parametrize("depth,W,H,F,method,subme,merange", [(8, 1920, 1088, 4, 1, 2, 57), (10, 3840, 2176, 2, 3, 3, 57), (10, 7680, 4352, 1, 3, 4, 128)])
def test_baseline_workloads_every_pu_and_tu_of_whole_pictures(depth, W, H, F, method, subme, merange):
    """The three BASELINE workloads at FULL size as bench.py runs them (one reference, the 85 square PUs of every CTU, 32x32 TUs; the headline: 3840x2176 10-bit, STAR, subme 3,
    merange 57, 2 pictures on 2 streams): ALL motion searches -- MV and cost; 346,800 at 4K, 693,600 at 8K -- and ALL TUs -- every quantised coefficient and numSig -- equal the
    oracle's (north_star: bit-identical quantised-coefficient output on 4K pictures).  The oracle runs on the host's cores (forked workers; a few CPU-minutes in all at 8K)."""
    import multiprocessing as mp
    import os
    qp = 28
    hb = make(depth, W, H, F, qp=qp, merange=merange, method=method, subme=subme, tu_log2=5, refs=1, rect=False, streams=min(F, 2))
    try:
        hb.upload(pairs_for(W, H, depth, F, 1, seed0=820))
        hb.step(); hb.sync()
        n = 32
        coeff, numsig = hb.coeffs()
        _FULL.update(hb=dict(tasks={lv: hb.tasks_host[lv] for lv in LEVELS}, tu=hb.tu_host, cur=hb.cur_host, ref=hb.refs_host[0], stride=hb.stride, merange=hb.merange,
                             method=hb.method, subme=hb.subme, mv_level=hb.mv_level, tu_log2=hb.tu_log2, qp=hb.qp),
                     res={lv: hb.shape_results(lv, lv, 0, 0) for lv in LEVELS}, coeff=coeff.reshape(-1, n * n), numsig=numsig, row=mvcost_row(depth, qp, 1 << 15), ora=Oracle(depth))
    finally:
        hb.close()
    jobs = []
    for lv in LEVELS:
        nt, step = len(_FULL["hb"]["tasks"][lv]), {64: 24, 32: 128, 16: 1024, 8: 4096}[lv]
        jobs += [(lv, lo, min(nt, lo + step)) for lo in range(0, nt, step)]
    ntu = len(_FULL["hb"]["tu"])
    tjobs = [(lo, min(ntu, lo + 256)) for lo in range(0, ntu, 256)]
    workers = max(2, (os.cpu_count() or 4) - 2)
    with mp.get_context("fork").Pool(workers) as pool:
        pus = pool.map(_full_pus, jobs, chunksize=1)
        tus = pool.map(_full_tus, tjobs, chunksize=1)
    _FULL.clear()
    bad = [b for _, bl in pus for b in bl]
    assert not bad, "%d of %d searches differ from the oracle, first (level, task, hip, oracle): %s" % (len(bad), sum(c for c, _ in pus), bad[0])
    badt = [b for _, bl in tus for b in bl]
    assert not badt, "%d of %d TUs differ from the oracle, first: TU %d" % (len(badt), ntu, badt[0])
    assert sum(c for c, _ in pus) == F * (W // 64) * (H // 64) * 85 and sum(c for c, _ in tus) == F * (W // 32) * (H // 32)


def _preset_pus(arg):
    """one slice of one PU shape: the search in every reference (seeded by that reference's own chain) and the choice among the references"""
    (w, h), lo, hi = arg
    F_ = _FULL
    hb, ora, R = F_["hb"], F_["ora"], F_["refs"]
    t, res, parent, ch = F_["tasks"][(w, h)], F_["res"][(w, h)], F_["res"][(max(w, h), max(w, h))] if w != h else F_["res"].get((2 * w, 2 * h)), F_["choice"][(w, h)]
    d = hb["merange"] << 2
    bad = []
    for i in range(lo, hi):
        tk = t[i]
        mv = np.zeros((8, 2), np.int32); mvp = np.zeros((8, 2), np.int32); cost = np.zeros(8, np.int32); mvc = np.zeros(8, np.int32)
        for r in range(R):
            qmvp = (0, 0) if tk["mvpFrom"] < 0 else (int(parent[r][tk["mvpFrom"]]["mv"][0]), int(parent[r][tk["mvpFrom"]]["mv"][1]))
            lx0, ly0, lx1, ly1 = int(tk["mvmin"][0]), int(tk["mvmin"][1]), int(tk["mvmax"][0]), int(tk["mvmax"][1])
            b = [min(lx1, max(lx0, qmvp[0] - d)) >> 2, min(ly1, max(ly0, qmvp[1] - d)) >> 2, min(lx1, max(lx0, qmvp[0] + d)) >> 2, min(ly1, max(ly0, qmvp[1] + d)) >> 2]
            b[3] = max(b[3], b[1])
            exp = ora.me(w, h, hb["cur"], hb["stride"], int(tk["curOff"]), hb["planes"][r], hb["stride"], int(tk["refOff"]), b, qmvp, [], hb["merange"], hb["method"], hb["subme"], F_["row"])
            g = res[r][i]
            if (int(g["mv"][0]), int(g["mv"][1]), int(g["cost"])) != exp:
                bad.append(("search", w, h, i, r, (int(g["mv"][0]), int(g["mv"][1]), int(g["cost"])), exp))
            mv[r] = g["mv"]; mvp[r] = qmvp; cost[r] = g["cost"]; mvc[r] = g["mvcost"]
        rp = [hb["planes"][k] if k < R else None for k in range(8)]
        o, _ = ora.inter_merge(w, h, (R, 0), mv, mvp, cost, mvc, F_["bits"], F_["lam"], False, max(hb["W"], hb["H"]), list(tk["mvmin"]) + list(tk["mvmax"]),
                               hb["cur"], hb["stride"], int(tk["curOff"]), rp, hb["stride"], int(tk["refOff"]))
        g = ch[i]
        mine = [int(g["mv"][0][0]), int(g["mv"][0][1]), int(g["mv"][1][0]), int(g["mv"][1][1]), int(g["mvp"][0][0]), int(g["mvp"][0][1]), int(g["mvp"][1][0]), int(g["mvp"][1][1]),
                int(g["ref"][0]), int(g["ref"][1]), int(g["bits"]), int(g["cost"])]
        if mine != [int(v) for v in o]:
            bad.append(("choice", w, h, i, mine, [int(v) for v in o]))
    return (hi - lo), bad


def test_preset_slow_every_pu_of_a_whole_4k_picture_in_every_reference():
    """BASELINE configs[2] with the preset's own load, EXHAUSTIVELY once: one 3840x2176 10-bit picture, STAR, subme 3, merange 57, 4 list-0 references, the 425 PUs of every CTU
    (squares + 2NxN / Nx2N of every CU): all 2040 x 425 x 4 searches -- MV and cost, each seeded by its own reference's chain -- and all 2040 x 425 choices among the references
    (MV, MVP, reference, bits, cost) equal the oracle's.  The oracle runs on every host core (forked workers over slices of the task lists)."""
    import multiprocessing as mp
    import os
    depth, W, H, F, refs, qp = 10, 3840, 2176, 1, 4, 28
    hb = make(depth, W, H, F, qp=qp, merange=57, method=3, subme=3, tu_log2=5, refs=refs, rect=True, streams=1)
    try:
        hb.upload(pairs_for(W, H, depth, F, refs, seed0=910))
        hb.step(); hb.sync()
        shapes = [(lv, lv) for lv in LEVELS] + sorted(hb.rect_host)
        tasks = {(lv, lv): hb.tasks_host[lv] for lv in LEVELS}
        tasks.update(hb.rect_host)
        _FULL.update(hb=dict(cur=hb.cur_host, planes=[hb.refs_host[r] for r in range(refs)], stride=hb.stride, merange=hb.merange, method=hb.method, subme=hb.subme, W=hb.W, H=hb.H),
                     refs=refs, tasks=tasks, res={sh: [hb.shape_results(sh[0], sh[1], r, 0) for r in range(refs)] for sh in shapes}, choice={sh: hb.choices(sh[0], sh[1]) for sh in shapes},
                     row=mvcost_row(depth, qp, 1 << 15), bits=mvbits_row(depth, 1 << 14), lam=rd_lambda(depth, qp), ora=Oracle(depth))
    finally:
        hb.close()
    jobs = []
    for sh in shapes:
        nt = len(tasks[sh])
        step = max(16, min(2048, (24 * 64 * 64) // (sh[0] * sh[1])))           # slices of about equal pixel count
        jobs += [(sh, lo, min(nt, lo + step)) for lo in range(0, nt, step)]
    jobs.sort(key=lambda j: -(j[0][0] * j[0][1] * (j[2] - j[1])))
    with mp.get_context("fork").Pool(max(2, (os.cpu_count() or 4) - 2)) as pool:
        out = pool.map(_preset_pus, jobs, chunksize=1)
    _FULL.clear()
    bad = [b for _, bl in out for b in bl]
    assert not bad, "%d PUs differ from the oracle, first: %s" % (len(bad), bad[0])
    assert sum(c for c, _ in out) == (W // 64) * (H // 64) * 425


def test_a_context_hands_the_large_blocks_of_a_destroyed_batch_to_its_next_one():
    """x265hip_batch_destroy keeps the batch's large device blocks (>= 64 MiB: the phase planes) with the context, the next x265hip_batch_create of the same context takes them
    again (a long-lived host that creates and destroys batches does not unmap and map multi-GB ranges each time -- the pattern behind round 5's runtime fault); a smaller batch
    does not get a block more than a quarter too large; x265hip_ctx_trim gives the kept blocks back.  The bytes computed on a reused block are the first batch's."""
    import ctypes as C
    depth, W, H, F = 8, 1920, 1088, 2                      # phase planes: 16 x 2 x 2112 x 1280 bytes = 86 MB
    lib = x265hip.HipLib(depth, fill_table=False).lib
    lib.x265hip_ctx_trim.restype = C.c_size_t
    lib.x265hip_batch_device_ptr.restype = C.c_void_p
    hb = HostBatch(lib, depth, W, H, F, qp=30, merange=16, method=1, subme=2, tu_log2=5)
    pairs = pairs_for(W, H, depth, F, 1)
    try:
        hb.upload(pairs); hb.step(); hb.sync()
        first = [hb.results(lv).tobytes() for lv in LEVELS]
        p_first = lib.x265hip_batch_device_ptr(hb.batch, 200)
        assert p_first
        lib.x265hip_batch_destroy.restype = None
        lib.x265hip_batch_destroy(hb.batch); hb.batch = C.c_void_p()
        # the same geometry again on the same context: the phase planes' block comes back
        assert lib.x265hip_batch_create(hb.ctx, C.byref(hb.desc), C.byref(hb.batch)) == 0, lib.x265hip_last_error()
        assert lib.x265hip_batch_device_ptr(hb.batch, 200) == p_first, "the kept block was not reused"
        hb.upload(pairs); hb.step(); hb.sync()
        assert [hb.results(lv).tobytes() for lv in LEVELS] == first
        lib.x265hip_batch_destroy(hb.batch); hb.batch = C.c_void_p()
        kept = lib.x265hip_ctx_trim(hb.ctx)
        assert kept >= 16 * F * (W + 192) * (H + 192) and lib.x265hip_ctx_trim(hb.ctx) == 0
    finally:
        hb.close()
