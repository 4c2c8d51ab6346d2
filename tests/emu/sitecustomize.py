"""TEST INFRASTRUCTURE ONLY: with PYTHONPATH=tests/emu and X265HIP_EMU=1 every Python process of a run (a script and the children it starts) maps the tests' device tensors
onto the CPU (torch_on_host.py), so that a SCRIPT -- not only pytest -- can drive the emulated library.  Used by tests/test_emu_kernels.py to run bench.py's line on a few CTUs."""
import os

if os.environ.get("X265HIP_EMU"):
    try:
        import torch_on_host
        torch_on_host.install()
    except Exception:          # (a process without torch: nothing to map)
        pass
