"""Time of x265hip_transform_batch for 32x32 / 16x16 transforms: the MFMA kernels against the LDS matrix-product kernel (X265HIP_DCT32_VALU=1 selects the latter).
python profiles/micro/idct32_time.py <depth> [N]"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import x265hip  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lib = C.CDLL(x265hip.lib_path(depth))
n = 32640 * (1024 // (N * N))                       # the TUs of eight 4K pictures
rng = np.random.default_rng(1)
coef = torch.from_numpy(rng.integers(-2000, 2000, n * N * N).astype(np.int16)).cuda()
out = torch.zeros(n * N * N, dtype=torch.int16, device="cuda")
fwd = torch.zeros(n * N * N, dtype=torch.int16, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
IP = C.c_ssize_t
for op, name in ((1, "inverse"), (0, "forward")):
    dst = out if op else fwd
    for _ in range(3):
        lib.x265hip_transform_batch(st, op, N, C.c_void_p(coef.data_ptr()), IP(N), None, C.c_void_p(dst.data_ptr()), IP(N), None, n)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        lib.x265hip_transform_batch(st, op, N, C.c_void_p(coef.data_ptr()), IP(N), None, C.c_void_p(dst.data_ptr()), IP(N), None, n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%s %dx%d, %d TUs, %d bit: %.1f us per launch, %.0f GB/s in + out" % (name, N, N, n, depth, dt * 1e6, n * N * N * 4 / dt / 1e9))
