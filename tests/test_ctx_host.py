"""include/x265hip_ctx.h, host side (no GPU): the pyramid task lists the C++ batch builds are, byte for byte, the lists the Python pipeline
builds; argument checking; the context refuses to exist without a device."""
import ctypes as C

import numpy as np
import pytest

from depths import DEPTHS

import x265hip  # noqa: F401
from x265hip_pkg.frame import ME_TASK, TU_TASK
from x265hip_pkg.pipeline import pyramid_tasks, LEVELS


from x265hip_pkg.host_batch import BatchDesc as Desc      # the ctypes mirror of x265hip_batch_desc (fields beyond the ones given are zero)


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("geom", [(128, 64, 2, 96, 5), (256, 192, 3, 96, 4), (1920, 1088, 1, 96, 3), (64, 64, 1, 88, 2)])
def test_task_lists_equal_the_python_pipeline(depth, geom):
    W, H, F, margin, tu = geom
    lib = x265hip.HipLib(depth, fill_table=False).lib
    d = Desc(W, H, F, margin, 28, 57, 1, 2, tu, 0, 1)
    tasks, tus, _ = pyramid_tasks(W, H, F, margin, tu)
    for lv in LEVELS:
        n = lib.x265hip_batch_task_count(C.byref(d), lv)
        assert n == len(tasks[lv])
        out = np.zeros(n, ME_TASK)
        assert lib.x265hip_batch_build_me_tasks(C.byref(d), lv, C.c_void_p(out.ctypes.data)) == 0
        assert out.tobytes() == tasks[lv].tobytes(), "level %d" % lv
    n = lib.x265hip_batch_tu_count(C.byref(d))
    assert n == len(tus)
    out = np.zeros(n, TU_TASK)
    assert lib.x265hip_batch_build_tu_tasks(C.byref(d), C.c_void_p(out.ctypes.data)) == 0
    assert out.tobytes() == tus.tobytes()


def test_amp_task_lists_cover_every_cu_the_way_the_four_modes_split_it():
    """x265hip_batch_build_amp_tasks: per CU of 64 / 32 / 16 the eight PUs of 2NxnU, 2NxnD, nLx2N, nRx2N (g_puLookup, encoder/threadedme.h:67-92): a shape and its
    complement tile the CU twice (once per mode of the pair); every PU points at its own CU's 2Nx2N task; a range of CTU rows is a contiguous range of the list"""
    lib = x265hip.HipLib(8, fill_table=False).lib
    W, H, F, margin = 192, 128, 2, 96
    d = Desc(W, H, F, margin, 28, 57, 3, 3, 5, 0, 1, 1, 1, 1, 0, 1, 0)          # ... refs, rect, streams, bandRows, amp, refs1
    stride, plane = W + 2 * margin, (W + 2 * margin) * (H + 2 * margin)
    total = 0
    for lv in (64, 32, 16):
        sq = np.zeros(lib.x265hip_batch_task_count(C.byref(d), lv), ME_TASK)
        assert lib.x265hip_batch_build_me_tasks(C.byref(d), lv, C.c_void_p(sq.ctypes.data)) == 0
        cover = np.zeros((F, H, W), np.int32)
        for (w, h) in ((lv, lv // 4), (lv, 3 * lv // 4), (lv // 4, lv), (3 * lv // 4, lv)):
            n = lib.x265hip_batch_amp_task_count(C.byref(d), w, h)
            assert n == F * (W // lv) * (H // lv) * 2
            t = np.zeros(n, ME_TASK)
            assert lib.x265hip_batch_build_amp_tasks(C.byref(d), w, h, C.c_void_p(t.ctypes.data)) == 0
            total += n
            rows_of_ctus = []
            for k in t:
                off = int(k["curOff"]); f, rem = divmod(off, plane); y, x = divmod(rem, stride); y -= margin; x -= margin
                assert 0 <= x and x + w <= W and 0 <= y and y + h <= H and k["refOff"] == k["curOff"]
                cover[f, y:y + h, x:x + w] += 1
                cu = sq[int(k["mvpFrom"])]
                coff = int(cu["curOff"]) - f * plane; cy, cx = divmod(coff, stride); cy -= margin; cx -= margin
                assert cx <= x and x + w <= cx + lv and cy <= y and y + h <= cy + lv, "the PU lies inside the CU whose 2Nx2N MV seeds it"
                assert list(k["mvmin"]) == list(cu["mvmin"]) and list(k["mvmax"]) == list(cu["mvmax"])
                rows_of_ctus.append(f * (H // 64) + y // 64)
            assert rows_of_ctus == sorted(rows_of_ctus)
        assert (cover == 4).all(), "four modes, each tiling the CU once"
    assert total == F * (W // 64) * (H // 64) * 168
    assert lib.x265hip_batch_amp_task_count(C.byref(d), 8, 2) < 0 and lib.x265hip_batch_amp_task_count(C.byref(d), 32, 16) < 0 and lib.x265hip_batch_amp_task_count(C.byref(d), 64, 32) < 0


def test_bad_descriptors_are_refused():
    lib = x265hip.HipLib(8, fill_table=False).lib
    lib.x265hip_last_error.restype = C.c_char_p
    for bad in (Desc(100, 64, 1, 96, 28, 57, 1, 2, 5, 0, 1), Desc(128, 64, 0, 96, 28, 57, 1, 2, 5, 0, 1), Desc(128, 64, 1, 32, 28, 57, 1, 2, 5, 0, 1),
                Desc(128, 64, 1, 96, 99, 57, 1, 2, 5, 0, 1), Desc(128, 64, 1, 96, 28, 57, 1, 9, 5, 0, 1), Desc(128, 64, 1, 96, 28, 57, 1, 2, 6, 0, 1)):
        assert lib.x265hip_batch_task_count(C.byref(bad), 64) < 0
        out = np.zeros(64, ME_TASK)
        assert lib.x265hip_batch_build_me_tasks(C.byref(bad), 64, C.c_void_p(out.ctypes.data)) < 0
    ok = Desc(128, 64, 1, 96, 28, 57, 1, 2, 5, 0, 1)
    assert lib.x265hip_batch_task_count(C.byref(ok), 48) < 0                  # not a pyramid level


def test_batch_mode_switches_need_a_batch():
    """x265hip_batch_set_fused / _set_timing / _step / _step_one_stream on a null batch are argument errors (no GPU needed)"""
    for depth in (8, 10):
        lib = x265hip.HipLib(depth, fill_table=False).lib
        for mode in (0, 1, 2, 4, 8):
            assert lib.x265hip_batch_set_fused(None, mode) < 0 and lib.x265hip_batch_set_mode(None, mode) < 0
        assert lib.x265hip_batch_set_timing(None, 1) < 0 and lib.x265hip_batch_step(None) < 0 and lib.x265hip_batch_step_one_stream(None) < 0


def test_context_needs_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = x265hip.HipLib(8, fill_table=False).lib
    ctx = C.c_void_p()
    assert lib.x265hip_ctx_create(0, C.byref(ctx)) < 0 and not ctx.value      # no CPU fallback: fails loudly


def test_reference_count_and_picture_size_limits_are_argument_errors():
    """Host-side validation, no GPU: x265hip_tme_frame refuses more than X265HIP_MAX_REF = 16 references per list (MAX_NUM_REF, common/common.h:329) before it looks at a
    device; pictures whose quarter-pel clip limits would not fit int16 are refused by the batch descriptor."""
    from x265hip_pkg.frame import MAX_REF
    lib = x265hip.HipLib(8, fill_table=False).lib
    lib.x265hip_last_error.restype = C.c_char_p
    assert MAX_REF == 16

    class Ref(C.Structure):
        _fields_ = [(n, C.c_void_p) for n in ("mePlane", "mePhase", "reconPhase", "refTable", "lowresMv")]

    class Args(C.Structure):
        _fields_ = [("isP", C.c_int), ("numRef", C.c_int * 2), ("curPOC", C.c_int), ("temporalMvp", C.c_int), ("refPOC", (C.c_int * 16) * 2),
                    ("searchRange", C.c_int), ("searchMethod", C.c_int), ("subpelRefine", C.c_int),
                    ("picWidth", C.c_int), ("picHeight", C.c_int), ("ctuSize", C.c_int), ("lowresBlocksX", C.c_int),
                    ("curPlane", C.c_void_p), ("stride", C.c_ssize_t), ("origin", C.c_int64), ("planeElems", C.c_int64),
                    ("refs", (Ref * MAX_REF) * 2), ("table", C.c_void_p), ("areaBest", C.c_void_p), ("temporal", C.c_void_p),
                    ("nQp", C.c_int), ("qpIndex", C.c_void_p), ("costRows", C.c_void_p), ("costHalfRange", C.c_int), ("lambdas", C.c_uint64 * 64), ("bitsRow", C.c_void_p), ("bitsHalfRange", C.c_int),
                    ("steps", C.c_void_p), ("nSteps", C.c_int), ("workspace", C.c_void_p), ("workspaceBytes", C.c_size_t),
                    ("refLagPixels", C.c_int), ("flags", C.c_int), ("frameParallel", C.c_int), ("ctuFirst", C.c_int), ("ctuCount", C.c_int),
                    ("pirStartCol", C.c_int), ("pirSafeX", C.c_int)]
    lib.x265hip_tme_workspace.restype = C.c_size_t
    buf = np.zeros(4096, np.uint8)
    a = Args()
    a.isP = 1; a.numRef[0] = 17; a.picWidth = a.picHeight = 64; a.ctuSize = 64; a.nQp = 1; a.nSteps = 1
    for f in ("curPlane", "table", "areaBest", "temporal", "costRows", "bitsRow", "steps", "workspace"):
        setattr(a, f, buf.ctypes.data)
    a.workspaceBytes = lib.x265hip_tme_workspace(1)
    assert lib.x265hip_tme_frame(None, C.byref(a)) == -3 and b"17 references" in lib.x265hip_last_error()
    a.numRef[0] = 0
    assert lib.x265hip_tme_frame(None, C.byref(a)) == -3
    for w, h in ((8192, 64), (64, 8192)):
        assert lib.x265hip_batch_task_count(C.byref(Desc(w, h, 1, 96, 28, 57, 1, 2, 5, 0, 1)), 64) < 0
    assert lib.x265hip_batch_task_count(C.byref(Desc(8128, 64, 1, 96, 28, 57, 1, 2, 5, 0, 1)), 64) == 127


@pytest.mark.parametrize("geom", [(128, 64, 2, 96), (256, 192, 3, 96)])
def test_rect_task_lists_equal_the_python_pipeline(geom):
    """the 2NxN / Nx2N task lists of the C++ host (x265hip_batch_build_rect_tasks) are, byte for byte, pipeline.rect_tasks'"""
    from x265hip_pkg.pipeline import rect_tasks
    W, H, F, margin = geom
    lib = x265hip.HipLib(8, fill_table=False).lib
    d = Desc(W, H, F, margin, 28, 57, 3, 3, 5, 0, 1, 4, 1, 2)
    exp = rect_tasks(W, H, F, margin)
    assert len(exp) == 8
    for (w, h), t in exp.items():
        n = lib.x265hip_batch_rect_task_count(C.byref(d), w, h)
        assert n == len(t)
        out = np.zeros(n, ME_TASK)
        assert lib.x265hip_batch_build_rect_tasks(C.byref(d), w, h, C.c_void_p(out.ctypes.data)) == 0
        assert out.tobytes() == t.tobytes(), "%dx%d" % (w, h)
    assert lib.x265hip_batch_rect_task_count(C.byref(d), 64, 64) < 0 and lib.x265hip_batch_rect_task_count(C.byref(d), 64, 16) < 0
    for bad in (Desc(W, H, F, margin, 28, 57, 3, 3, 5, 0, 1, 17, 0, 1), Desc(W, H, F, margin, 28, 57, 3, 3, 5, 0, 1, 1, 0, 9), Desc(W, H, F, margin, 28, 57, 3, 3, 5, 0, 0, 2, 0, 1)):
        assert lib.x265hip_batch_task_count(C.byref(bad), 64) < 0        # too many references / streams; several references without phase planes


def test_filter_producer_refuses_bad_geometry():
    """x265hip_ff_create / x265hip_ff_picture validate before they look at a device: no context, dimensions that are no multiple of 8, CTU sizes HEVC does not have"""
    lib = x265hip.HipLib(8, fill_table=False).lib
    ff = C.c_void_p()
    fake = C.c_void_p(1)                                         # never dereferenced: the geometry is refused first
    assert lib.x265hip_ff_create(None, 128, 64, 64, C.c_ssize_t(320), C.c_ssize_t(160), C.byref(ff)) == -3 and not ff.value
    for (w, h, ctu, sy, sc) in [(100, 64, 64, 320, 160), (128, 60, 64, 320, 160), (128, 64, 48, 320, 160), (128, 64, 64, 100, 160), (128, 64, 64, 320, 32), (8192, 64, 64, 9000, 4500)]:
        assert lib.x265hip_ff_create(fake, w, h, ctu, C.c_ssize_t(sy), C.c_ssize_t(sc), C.byref(ff)) == -3 and not ff.value, (w, h, ctu, sy, sc)
    assert lib.x265hip_ff_picture(None, None) == -3
