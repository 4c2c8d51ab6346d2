"""The ASan build of the emulated library must catch what it is for: a kernel writing ONE element past a "device" block (the host heap there), and one reading past an LDS array's
end would be reported the same way (static arrays have redzones).  Run by tests/test_emu_kernels.py::test_the_asan_build_catches_an_overrun in a process of its own (it is meant to die);
argv[1] = elements the output buffer is too small by (0: the call must survive)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = C.CDLL(os.path.join(ROOT, "tests", "emu", "_build_asan", "libx265hip_8.so"))
W, H, M = 64, 32, 16
stride, rows = W + 2 * M, H + 2 * M
short = int(sys.argv[1]) if len(sys.argv) > 1 else 1
libc = C.CDLL(None)
libc.malloc.restype = C.c_void_p
libc.malloc.argtypes = [C.c_size_t]
plane = libc.malloc(stride * rows)                                   # (malloc of the preloaded ASan runtime: redzones either side)
C.memset(plane, 7, stride * rows)
out = libc.malloc(16 * stride * rows - short)
lib.x265hip_subpel_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_int, C.c_void_p, C.c_int64]
rc = lib.x265hip_subpel_planes(None, plane, stride, rows, out, stride * rows)
print("survived (rc %d) with an output %d byte(s) short" % (rc, short))
