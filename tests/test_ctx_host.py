"""include/x265hip_ctx.h, host side (no GPU): the pyramid task lists the C++ batch builds are, byte for byte, the lists the Python pipeline
builds; argument checking; the context refuses to exist without a device."""
import ctypes as C

import numpy as np
import pytest

import x265hip  # noqa: F401
from x265hip_pkg.frame import ME_TASK, TU_TASK
from x265hip_pkg.pipeline import pyramid_tasks, LEVELS


class Desc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("width", "height", "frames", "margin", "qp", "merange", "method", "subme", "tuLog2", "recon", "usePlanes")]


@pytest.mark.parametrize("depth", [8, 10])
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


def test_context_needs_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = x265hip.HipLib(8, fill_table=False).lib
    ctx = C.c_void_p()
    assert lib.x265hip_ctx_create(0, C.byref(ctx)) < 0 and not ctx.value      # no CPU fallback: fails loudly
