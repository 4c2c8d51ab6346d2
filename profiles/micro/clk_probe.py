"""Does a stream of tiny kernels run at full shader clock?  Samples `rocm-smi --showclocks` while (a) the GPU idles, (b) x265hip_ssim_frame (three tiny
launches, 17 wavefronts in the longest) runs back to back for ~2 s, (c) the 1080p ME step (fills the chip) runs for ~2 s.  Run from the repo root on the GPU box."""
import ctypes as C
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import x265hip  # noqa: E402,F401
from x265hip_pkg.frame import FrameApi  # noqa: E402


def clocks():
    r = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True).stdout
    return [ln.strip() for ln in r.splitlines() if "sclk" in ln or "mclk" in ln][:4]


def sample_during(fn, seconds, label):
    stop = [False]
    out = []

    def sampler():
        time.sleep(seconds / 2)
        out.extend(clocks())
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        fn(); n += 1
    import torch
    torch.cuda.synchronize(); th.join()
    print(label, "calls", n, "->", out)


api = FrameApi(8)
t = api.torch
W, H = 1920, 1080
rng = np.random.default_rng(1)
a = api.to_device(rng.integers(0, 256, W * H).astype(np.uint8)); b = api.to_device(rng.integers(0, 256, W * H).astype(np.uint8))
api.lib.x265hip_ssim_workspace.restype = C.c_size_t
ws = t.zeros(api.lib.x265hip_ssim_workspace(W, H) // 4, dtype=t.float32, device="cuda")
rs = t.zeros(17, dtype=t.float32, device="cuda"); rc = t.zeros(17, dtype=t.int32, device="cuda"); fr = t.zeros(2, dtype=t.float64, device="cuda")
P = lambda x: C.c_void_p(x.data_ptr())


def ssim():
    for _ in range(50):
        api.lib.x265hip_ssim_frame(api.stream(), P(a), C.c_ssize_t(W), P(b), C.c_ssize_t(W), W, H, 64, P(ws), P(rs), P(rc), P(fr))
    t.cuda.synchronize()


big = t.zeros(1 << 28, dtype=t.float32, device="cuda")


def heavy():
    for _ in range(20):
        big.mul_(1.0001)
    t.cuda.synchronize()


print("idle", clocks())
sample_during(ssim, 2.0, "ssim stream")
sample_during(heavy, 2.0, "streaming 1 GiB elementwise")
e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
for label, pre in (("ssim after idle", lambda: time.sleep(1.0)), ("ssim right after heavy work", heavy)):
    pre()
    e0.record()
    for _ in range(50):
        api.lib.x265hip_ssim_frame(api.stream(), P(a), C.c_ssize_t(W), P(b), C.c_ssize_t(W), W, H, 64, P(ws), P(rs), P(rc), P(fr))
    e1.record(); t.cuda.synchronize()
    print(label, "%.4f ms per call" % (e0.elapsed_time(e1) / 50))
