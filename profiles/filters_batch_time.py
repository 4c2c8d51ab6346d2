#!/usr/bin/env python3
"""profiles/filters_batch_time.py: time per stage of the picture-batched in-loop filter entry points (8 pictures of 1080p, 8 bit) next to the per-picture calls."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import x265hip  # noqa
from x265hip_pkg.frame import FrameApi
from deblock_util import I8, U8, DeblockPic, coded_picture, descriptor

depth, W, H, ctu, F = 8, 1920, 1080 - 1080 % 8, 64, 8
api = FrameApi(depth); L, st = api.lib, api.stream()
pic = coded_picture(depth, W, H, ctu, 21)
P = lambda x: C.c_void_p(x.data_ptr())
shapes = [(H, W), (H // 2, W // 2), (H // 2, W // 2)]
elems = [h * w for h, w in shapes]
rec = [api.to_device(np.concatenate([p.reshape(-1)] * F)) for p in pic["planes"]]
rng = np.random.default_rng(5)
src = [api.to_device(np.concatenate([np.clip(p.astype(np.int64) + rng.integers(-3, 4, p.shape), 0, 255).astype(p.dtype).reshape(-1)] * F)) for p in pic["planes"]]; out = [torch.empty_like(x) for x in rec]
arrs = {k: api.to_device(np.ascontiguousarray(pic[k]).reshape(-1)) for k in U8 + I8 + ("mv0", "mv1")}
desc = descriptor(pic, lambda k: arrs[k].data_ptr())
class Job(C.Structure):
    _fields_ = [("pic", DeblockPic), ("Y", C.c_void_p), ("Cb", C.c_void_p), ("Cr", C.c_void_p), ("bsOut", C.c_void_p)]
jobs = (Job * F)()
for f in range(F):
    jobs[f].pic = desc
    jobs[f].Y = rec[0].data_ptr() + f * elems[0]; jobs[f].Cb = rec[1].data_ptr() + f * elems[1]; jobs[f].Cr = rec[2].data_ptr() + f * elems[2]
d_jobs = api.to_device(np.frombuffer(bytes(jobs), np.uint8).copy())
n_ctu = ((W + 63) // 64) * ((H + 63) // 64)
stats = torch.zeros(F * n_ctu * 320, dtype=torch.int32, device="cuda")
prm = api.to_device(np.zeros(F * n_ctu * 6, np.int32))
L.x265hip_ssim_workspace.restype = C.c_size_t
ws = torch.zeros(F * (L.x265hip_ssim_workspace(W, H) // 4), dtype=torch.float32, device="cuda")
nrows = (H + 63) // 64
rs = torch.zeros(F * nrows, dtype=torch.float32, device="cuda"); rc = torch.zeros(F * nrows, dtype=torch.int32, device="cuda"); fr = torch.zeros(2 * F, dtype=torch.float64, device="cuda")
ssd = torch.zeros(F, dtype=torch.int64, device="cuda")

def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

for n in (1, F):
    print("pictures per call:", n)
    print("  deblock   %.4f ms" % t(lambda: L.x265hip_deblock_pictures(st, P(d_jobs), jobs, n, C.c_ssize_t(W), C.c_ssize_t(W // 2))))
    print("  sao_stats %.4f ms" % t(lambda: L.x265hip_sao_stats_pictures(st, P(src[0]), P(rec[0]), C.c_ssize_t(W), W, H, 64, 0, 0, P(stats), n, C.c_int64(elems[0]))))
    print("  sao_apply %.4f ms" % t(lambda: L.x265hip_sao_apply_pictures(st, P(rec[0]), P(out[0]), C.c_ssize_t(W), W, H, 64, P(prm), n, C.c_int64(elems[0]))))
    print("  ssd       %.4f ms" % t(lambda: L.x265hip_plane_ssd_pictures(st, P(src[0]), P(out[0]), C.c_ssize_t(W), W, H, P(ssd), n, C.c_int64(elems[0]), C.c_int64(elems[0]))))
    print("  ssim      %.4f ms" % t(lambda: L.x265hip_ssim_pictures(st, P(out[0]), C.c_ssize_t(W), P(src[0]), C.c_ssize_t(W), W, H, 64, P(ws), P(rs), P(rc), P(fr), n, C.c_int64(elems[0]), C.c_int64(elems[0]))))

# the whole chain of bench.py --filters through the batched entry points, 3 planes
stats3 = [torch.zeros(F * n_ctu * 320, dtype=torch.int32, device="cuda") for _ in range(3)]
prm3 = [api.to_device(np.zeros(F * n_ctu * 6, np.int32)) for _ in range(3)]
ssd3 = torch.zeros(3 * F, dtype=torch.int64, device="cuda")
def chain():
    L.x265hip_deblock_pictures(st, P(d_jobs), jobs, F, C.c_ssize_t(W), C.c_ssize_t(W // 2))
    for c in range(3):
        h, w = shapes[c]; cs = ctu if c == 0 else ctu // 2
        L.x265hip_sao_stats_pictures(st, P(src[c]), P(rec[c]), C.c_ssize_t(w), w, h, cs, 0, 0 if c == 0 else 2, P(stats3[c]), F, C.c_int64(elems[c]))
        L.x265hip_sao_apply_pictures(st, P(rec[c]), P(out[c]), C.c_ssize_t(w), w, h, cs, P(prm3[c]), F, C.c_int64(elems[c]))
        L.x265hip_plane_ssd_pictures(st, P(src[c]), P(out[c]), C.c_ssize_t(w), w, h, C.c_void_p(ssd3.data_ptr() + 8 * F * c), F, C.c_int64(elems[c]), C.c_int64(elems[c]))
    L.x265hip_ssim_pictures(st, P(out[0]), C.c_ssize_t(W), P(src[0]), C.c_ssize_t(W), W, H, ctu, P(ws), P(rs), P(rc), P(fr), F, C.c_int64(elems[0]), C.c_int64(elems[0]))
print("whole chain, %d pictures, 3 planes: %.4f ms" % (F, t(chain)))
for c in (1, 2):
    h, w = shapes[c]
    print("  chroma plane %d sao_stats %.4f ms" % (c, t(lambda: L.x265hip_sao_stats_pictures(st, P(src[c]), P(rec[c]), C.c_ssize_t(w), w, h, 32, 0, 2, P(stats3[c]), F, C.c_int64(elems[c])))))
    print("  chroma plane %d sao_apply %.4f ms" % (c, t(lambda: L.x265hip_sao_apply_pictures(st, P(rec[c]), P(out[c]), C.c_ssize_t(w), w, h, 32, P(prm3[c]), F, C.c_int64(elems[c])))))
    print("  chroma plane %d ssd %.4f ms" % (c, t(lambda: L.x265hip_plane_ssd_pictures(st, P(src[c]), P(out[c]), C.c_ssize_t(w), w, h, C.c_void_p(ssd3.data_ptr() + 8 * F * c), F, C.c_int64(elems[c]), C.c_int64(elems[c])))))
