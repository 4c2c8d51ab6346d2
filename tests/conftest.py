import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
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
