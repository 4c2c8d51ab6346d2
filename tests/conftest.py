import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # X265HIP_EMU=1 (tests/emu/run_gpu_suite.sh, tests/test_emu_kernels.py): the `-m gpu` tests against the library's sources compiled for the host (tests/emu: a workgroup as
    # fibers, the cross-lane operations modelled) -- the tests' "device" tensors are CPU tensors then.  Test infrastructure; nothing of the product reads the variable
    if os.environ.get("X265HIP_EMU"):
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import torch_on_host
        torch_on_host.install()
    # tools/fence_run.sh: the tests' own device tensors through the fence allocator too (every block ends at an unmapped page; csrc/xh_fence.h)
    fence = os.environ.get("X265HIP_FENCE_TORCH")
    if fence:
        import torch
        torch.cuda.memory.change_current_allocator(torch.cuda.memory.CUDAPluggableAllocator(fence, "xh_fence_torch_malloc", "xh_fence_torch_free"))


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle .so files are git-ignored build products; build them on demand (seconds)."""
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    if os.path.isdir("/root/reference/source"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
