"""Frame pipeline on the GPU: sampled oracle parity on small frames for every search method, and -- at the
bench's full 1080p size -- size-independent properties plus a random sample against the oracle."""
import numpy as np
import pytest

from depths import DEPTHS

import x265hip  # noqa: F401
from x265hip_pkg.frame import mvcost_row
from x265hip_pkg.pipeline import FramePipeline, LEVELS
from x265hip_pkg.synth import frame_pair
from backends import Oracle
from pipeline_check import check_sample

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("planes", [True, False])
@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("method,subme,tu", [(1, 2, 5), (3, 3, 4), (0, 0, 3), (1, 7, 2)])
def test_small_frames_match_oracle(depth, method, subme, tu, planes):
    row = mvcost_row(depth, 28, 1 << 15)
    pipe = FramePipeline(depth, 256, 128, 2, qp=28, merange=24, method=method, subme=subme, tu_log2=tu, recon=True, cost_row=row,
                         use_planes=planes)
    pipe.upload([frame_pair(256, 128, depth, 40 + s, margin=pipe.margin, max_shift=14)[:2] for s in range(2)])
    pipe.step()
    pipe.torch.cuda.synchronize()
    assert check_sample(pipe, Oracle(depth), np.random.default_rng(depth + method), per_level=12, n_tu=16) >= 40


def test_full_size_properties_and_sample():
    """BASELINE configs[1] size: 1920x1088 8-bit, HEX, subme 2, merange 57."""
    depth, W, H = 8, 1920, 1088
    row = mvcost_row(depth, 28, 1 << 15)
    pipe = FramePipeline(depth, W, H, 2, qp=28, merange=57, method=1, subme=2, tu_log2=5, recon=True, cost_row=row)
    T = pipe.torch
    cur0, ref0, _, (dx, dy) = frame_pair(W, H, depth, 1, margin=pipe.margin, max_shift=24)
    # frame 1: reference identical to the source -> the search must return zero MVs, zero residual, zero SSE
    pipe.upload([(cur0, ref0), (cur0, cur0)])
    pipe.step(); T.cuda.synchronize()
    first = {lv: pipe.results(lv).copy() for lv in LEVELS}
    coeff1 = pipe.d_coeff.clone(); sse1 = pipe.d_sse.clone(); rec1 = pipe.d_recon.clone()
    # determinism / idempotence: a second pass over the same planes gives identical bytes
    pipe.step(); T.cuda.synchronize()
    for lv in LEVELS:
        assert np.array_equal(first[lv], pipe.results(lv))
    assert T.equal(coeff1, pipe.d_coeff) and T.equal(sse1, pipe.d_sse) and T.equal(rec1, pipe.d_recon)
    # identical frames
    for lv in LEVELS:
        r = first[lv]; n = len(r) // 2
        assert not r["mv"][n:].any(), "identical frames must give zero motion at level %d" % lv
    ntu = len(pipe.tu_host) // 2
    assert int(pipe.d_numsig[ntu:].abs().sum()) == 0 and int(pipe.d_coeff.view(-1, 1024)[ntu:].abs().sum()) == 0
    assert int(pipe.d_sse[ntu:].sum()) == 0
    # the global motion is found by the bulk of the large PUs of frame 0
    r64 = first[64][:len(first[64]) // 2]
    hit = np.mean((np.abs(r64["mv"][:, 0] - 4 * dx) <= 4) & (np.abs(r64["mv"][:, 1] - 4 * dy) <= 4))
    assert hit > 0.8, "only %.0f%% of the 64x64 PUs found the synthetic motion (%d,%d)" % (100 * hit, dx, dy)
    # checksum of checksums: per-TU SSE must add up to the plane-level squared error of the reconstruction
    m, s = pipe.margin, pipe.stride
    cur = pipe.d_cur.view(2, H + 2 * m, s)[:, m:m + H, m:m + W].to(T.int64)
    rec = pipe.d_recon.view(2, H + 2 * m, s)[:, m:m + H, m:m + W].to(T.int64)
    assert int(((cur - rec) ** 2).sum()) == int(pipe.d_sse.sum())
    # and a random sample of PUs / TUs bit-exact against the oracle
    assert check_sample(pipe, Oracle(depth), np.random.default_rng(3), per_level=25, n_tu=25) == 125


def _full_size(depth, W, H, method, subme, merange, pair0, min_hit):
    """Size-independent properties of one step at a BASELINE configuration's own size (2 frame pairs: a moving one and an identical one)
    plus >= 100 randomly sampled PUs / TUs bit-exact against the oracle."""
    row = mvcost_row(depth, 28, 1 << 15)
    pipe = FramePipeline(depth, W, H, 2, qp=28, merange=merange, method=method, subme=subme, tu_log2=5, recon=True, cost_row=row)
    T = pipe.torch
    cur0, ref0, (dx, dy) = pair0(pipe.margin)
    pipe.upload([(cur0, ref0), (cur0, cur0)])
    pipe.step(); T.cuda.synchronize()
    first = {lv: pipe.results(lv).copy() for lv in LEVELS}
    coeff1 = pipe.d_coeff.clone(); sse1 = pipe.d_sse.clone(); rec1 = pipe.d_recon.clone()
    pipe.step(); T.cuda.synchronize()                      # idempotence: a second pass over the same planes gives identical bytes
    for lv in LEVELS:
        assert np.array_equal(first[lv], pipe.results(lv))
    assert T.equal(coeff1, pipe.d_coeff) and T.equal(sse1, pipe.d_sse) and T.equal(rec1, pipe.d_recon)
    for lv in LEVELS:                                      # identical frames: zero motion, zero residual, zero error
        r = first[lv]; n = len(r) // 2
        assert not r["mv"][n:].any(), "identical frames must give zero motion at level %d" % lv
    ntu = len(pipe.tu_host) // 2
    assert int(pipe.d_numsig[ntu:].abs().sum()) == 0 and int(pipe.d_coeff.view(-1, 1024)[ntu:].abs().sum()) == 0
    assert int(pipe.d_sse[ntu:].sum()) == 0
    r64 = first[64][:len(first[64]) // 2]                  # the synthetic global motion is found by the bulk of the large PUs
    hit = np.mean((np.abs(r64["mv"][:, 0] - 4 * dx) <= 4) & (np.abs(r64["mv"][:, 1] - 4 * dy) <= 4))
    assert hit > min_hit, "only %.0f%% of the 64x64 PUs found the synthetic motion (%d,%d)" % (100 * hit, dx, dy)
    m, s = pipe.margin, pipe.stride                        # checksum of checksums: per-TU SSE adds up to the plane-level squared error
    cur = pipe.d_cur.view(2, H + 2 * m, s)[:, m:m + H, m:m + W].to(T.int64)
    rec = pipe.d_recon.view(2, H + 2 * m, s)[:, m:m + H, m:m + W].to(T.int64)
    assert int(((cur - rec) ** 2).sum()) == int(pipe.d_sse.sum())
    assert check_sample(pipe, Oracle(depth), np.random.default_rng(W), per_level=25, n_tu=25) == 125


def test_full_size_4k_10bit_slow():
    """BASELINE configs[2]: 3840x2176 (CTU-aligned 2160p) 10-bit, preset slow = STAR, subme 3, merange 57."""
    def pair0(margin):
        cur, ref, _, mv = frame_pair(3840, 2176, 10, 1, margin=margin, max_shift=24)
        return cur, ref, mv
    _full_size(10, 3840, 2176, 3, 3, 57, pair0, 0.8)


def test_full_size_8k_10bit_slower_merange_128():
    """BASELINE configs[4]: 7680x4352 10-bit, preset slower = STAR, subme 4, --merange 128.  The picture is a 4K synthetic pair tiled
    2 x 2 (the same global motion in every tile), which keeps the host-side generation short."""
    def pair0(margin):
        c, r, _, mv = frame_pair(3840, 2176, 10, 2, margin=0, max_shift=40)
        return (np.ascontiguousarray(np.pad(np.tile(c, (2, 2)), margin, mode="edge")),
                np.ascontiguousarray(np.pad(np.tile(r, (2, 2)), margin, mode="edge")), mv)
    _full_size(10, 7680, 4352, 3, 4, 128, pair0, 0.7)


@pytest.mark.parametrize("depth", DEPTHS)
def test_phase_planes_equal_the_interpolation_primitives(depth):
    """x265hip_subpel_planes: every phase plane equals luma_hpp / luma_vpp / luma_hvpp of the oracle, block by block."""
    from x265hip_pkg.frame import FrameApi
    api, ora = FrameApi(depth), Oracle(depth)
    T = api.torch
    rng = np.random.default_rng(depth)
    W, H, m = 192, 96, 32
    _, ref, stride, _ = frame_pair(W, H, depth, 9, margin=m, max_shift=4)
    rows = H + 2 * m
    ref_f = ref.reshape(-1)
    d_ref = api.to_device(ref_f)
    pe = stride * rows
    d_pl = T.zeros(16 * pe, dtype=d_ref.dtype, device="cuda")
    api.subpel_planes(d_ref, stride, rows, d_pl, pe)
    T.cuda.synchronize()
    pl = d_pl.cpu().numpy().view(ref_f.dtype).reshape(16, rows, stride)
    for f in range(1, 16):
        xf, yf = f & 3, f >> 2
        for _ in range(6):
            w, h = [(8, 8), (16, 16), (32, 32), (64, 64), (16, 4), (4, 16)][int(rng.integers(0, 6))]
            x = int(rng.integers(8, stride - w - 8)); y = int(rng.integers(8, rows - h - 8))
            buf = np.zeros(w * h, ref_f.dtype)
            kind = "hpp" if yf == 0 else ("vpp" if xf == 0 else "hvpp")
            exp = ora.interp(kind, 8, w, h, ref_f, stride, y * stride + x, buf, w, xf if kind != "vpp" else yf, yf).reshape(h, w)
            assert np.array_equal(pl[f, y:y + h, x:x + w], exp), "phase (%d,%d) block %dx%d at (%d,%d)" % (xf, yf, w, h, x, y)


@pytest.mark.parametrize("depth", DEPTHS)
def test_reconstruction_becomes_the_next_reference_on_device(depth):
    """TQ recon -> extendPicBorder -> phase planes, all in HBM, equals the oracle's chain on the host copy."""
    from x265hip_pkg.frame import FrameApi
    ora = Oracle(depth)
    row = mvcost_row(depth, 30, 1 << 15)
    W, H, F = 192, 128, 2
    pipe = FramePipeline(depth, W, H, F, qp=30, merange=16, method=1, subme=2, tu_log2=4, recon=True, cost_row=row)
    pairs = [frame_pair(W, H, depth, 70 + s, margin=pipe.margin, max_shift=9)[:2] for s in range(F)]
    pipe.upload(pairs)
    # the recon plane starts as garbage everywhere (margins included): only the F pictures proper get written by the TQ stage
    pipe.d_recon.fill_(0x55)
    pipe.step()
    api, T, m = pipe.api, pipe.torch, pipe.margin
    T.cuda.synchronize()
    rec_before = pipe.d_recon.cpu().numpy().view(pipe.cur_host.dtype).copy()
    api.extend_pic_border(pipe.d_recon, m * pipe.stride + m, pipe.stride, W, H, m, m, n_pictures=F, picture_elems=pipe.plane)
    T.cuda.synchronize()
    got = pipe.d_recon.cpu().numpy().view(pipe.cur_host.dtype)
    exp = rec_before.copy()
    for f in range(F):
        exp[f * pipe.plane:(f + 1) * pipe.plane] = ora.extend_pic_border(rec_before[f * pipe.plane:(f + 1) * pipe.plane], pipe.stride, W, H, m, m)
    assert np.array_equal(got, exp)
    # ... and its phase planes are the interpolation primitives of that reconstructed, extended picture
    rows = F * (H + 2 * m)
    d_pl = T.zeros(16 * pipe.plane * F, dtype=pipe.d_recon.dtype, device="cuda")
    api.subpel_planes(pipe.d_recon, pipe.stride, rows, d_pl, pipe.plane * F)
    T.cuda.synchronize()
    pl = d_pl.cpu().numpy().view(got.dtype).reshape(16, rows, pipe.stride)
    assert np.array_equal(pl[0].reshape(-1), got)
    rng = np.random.default_rng(depth)
    for f in (2, 7, 9, 15):
        xf, yf = f & 3, f >> 2
        x, y = int(rng.integers(8, pipe.stride - 40)), int(rng.integers(8, rows - 40))
        kind = "hpp" if yf == 0 else ("vpp" if xf == 0 else "hvpp")
        buf = np.zeros(32 * 32, got.dtype)
        e = ora.interp(kind, 8, 32, 32, got, pipe.stride, y * pipe.stride + x, buf, 32, xf if kind != "vpp" else yf, yf).reshape(32, 32)
        assert np.array_equal(pl[f, y:y + 32, x:x + 32], e)


@pytest.mark.parametrize("depth,method,subme,refs", [(8, 1, 2, 3), (10, 3, 3, 4), (8, 3, 3, 2)])
def test_several_references_match_oracle(depth, method, subme, refs):
    """preset medium searches 3 references, slow 4 (param.cpp:567-587): every reference its own predictor chain down the pyramid, the choice per PU by the
    reference's bit / cost rule (x265hip_inter_merge_batch), every TU compensated from the reference its PU chose."""
    from pipeline_check import check_sample_refs
    row = mvcost_row(depth, 28, 1 << 15)
    W, H, F = 256, 128, 2
    pipe = FramePipeline(depth, W, H, F, qp=28, merange=24, method=method, subme=subme, tu_log2=4, cost_row=row, refs=refs)
    rng = np.random.default_rng(depth + refs)
    pm = (1 << depth) - 1
    pairs = []
    for s in range(F):
        cur, ref, _, _ = frame_pair(W, H, depth, 60 + s, margin=pipe.margin, max_shift=10, noise=1.0)
        others = []
        for r in range(1, refs):                         # further references: the first one displaced again + noise that differs from block to block, so the best reference varies
            o = np.roll(ref, (2 * r, -3 * r), (0, 1)).astype(np.float64)
            sig = np.kron(rng.uniform(0.5, 6.0, (o.shape[0] // 32 + 1, o.shape[1] // 32 + 1)), np.ones((32, 32)))[:o.shape[0], :o.shape[1]]
            others.append(np.clip(o + rng.normal(0, 1, o.shape) * sig * (1 << (depth - 8)), 0, pm).astype(ref.dtype))
        base_sig = np.kron(rng.uniform(0.5, 6.0, (ref.shape[0] // 32 + 1, ref.shape[1] // 32 + 1)), np.ones((32, 32)))[:ref.shape[0], :ref.shape[1]]
        ref0 = np.clip(ref.astype(np.float64) + rng.normal(0, 1, ref.shape) * base_sig * (1 << (depth - 8)), 0, pm).astype(ref.dtype)
        pairs.append((cur, ref0) + tuple(others))
    pipe.upload(pairs)
    pipe.step(); pipe.torch.cuda.synchronize()
    assert check_sample_refs(pipe, Oracle(depth), np.random.default_rng(1), per_level=10, n_tu=16) >= 50
    chosen = {int(r) for lv in LEVELS for r in pipe.choices(lv)["ref"][:, 0]}
    assert len(chosen) >= 2, "the clip should make more than one reference win: %s" % chosen


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("method,subme", [(1, 2), (3, 3)])
def test_rectangular_partitions_match_oracle(depth, method, subme):
    """param bEnableRectInter (preset slow and up): the 2NxN / Nx2N PUs of every CU of the pyramid (64x32 ... 4x8, 425 PUs per CTU with the squares), each
    seeded by its CU's 2Nx2N result -- sampled PUs of every shape against the oracle."""
    row = mvcost_row(depth, 28, 1 << 15)
    pipe = FramePipeline(depth, 256, 128, 2, qp=28, merange=24, method=method, subme=subme, tu_log2=4, cost_row=row, rect=True)
    assert sum(len(t) for t in pipe.rect_host.values()) + sum(len(t) for t in pipe.tasks_host.values()) == 2 * (256 // 64) * (128 // 64) * 425
    pipe.upload([frame_pair(256, 128, depth, 140 + s, margin=pipe.margin, max_shift=14)[:2] for s in range(2)])
    pipe.step()
    pipe.torch.cuda.synchronize()
    assert check_sample(pipe, Oracle(depth), np.random.default_rng(3 * depth + method), per_level=12, n_tu=8) >= 8 * 6 + 40
