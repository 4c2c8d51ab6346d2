"""The picture-batched forms of the in-loop filter chain (x265hip_deblock_pictures, _sao_stats_pictures, _sao_apply_pictures, _ssim_pictures, _plane_ssd_pictures):
one launch per stage for F pictures must give, picture by picture, exactly what the per-picture entry points give (which are checked against the oracle and the
reference's own classes elsewhere); the deblocked planes are compared with the oracle here as well."""
import ctypes as C

import numpy as np
import pytest

import x265hip  # noqa: F401
from x265hip_pkg.frame import FrameApi
from backends import Oracle
from deblock_util import I8, U8, DeblockPic, coded_picture, descriptor, run_oracle

pytestmark = pytest.mark.gpu


class Job(C.Structure):
    _fields_ = [("pic", DeblockPic), ("Y", C.c_void_p), ("Cb", C.c_void_p), ("Cr", C.c_void_p), ("bsOut", C.c_void_p)]


@pytest.mark.parametrize("depth,W,H,ctu", [(8, 320, 192, 64), (10, 256, 136, 32)])
def test_batched_chain_equals_per_picture_calls(depth, W, H, ctu):
    api, ora = FrameApi(depth), Oracle(depth)
    T, L, st = api.torch, api.lib, api.stream()
    F = 3
    P = lambda x: C.c_void_p(x.data_ptr())
    pics = [coded_picture(depth, W, H, ctu, 40 + f, slice_p=bool(f & 1)) for f in range(F)]
    shapes = [(H, W), (H // 2, W // 2), (H // 2, W // 2)]
    rng = np.random.default_rng(depth)
    pm = (1 << depth) - 1
    # stacks: plane c of picture f at f * elems[c]
    elems = [h * w for h, w in shapes]
    rec = [api.to_device(np.concatenate([p["planes"][c].reshape(-1) for p in pics])) for c in range(3)]
    src = [api.to_device(np.concatenate([np.clip(p["planes"][c].astype(np.int64) + rng.integers(-3, 4, p["planes"][c].shape), 0, pm).astype(p["planes"][c].dtype).reshape(-1)
                                         for p in pics])) for c in range(3)]
    rec1 = [x.clone() for x in rec]                                    # the per-picture path works on its own copy
    arrs = [{k: api.to_device(np.ascontiguousarray(p[k]).reshape(-1)) for k in U8 + I8 + ("mv0", "mv1")} for p in pics]
    es = rec[0].element_size()
    # ---- deblocking ----
    jobs = (Job * F)()
    for f in range(F):
        jobs[f].pic = descriptor(pics[f], lambda k: arrs[f][k].data_ptr())
        jobs[f].Y = rec[0].data_ptr() + f * elems[0] * es; jobs[f].Cb = rec[1].data_ptr() + f * elems[1] * es; jobs[f].Cr = rec[2].data_ptr() + f * elems[2] * es
        jobs[f].bsOut = None
    d_jobs = api.to_device(np.frombuffer(bytes(jobs), np.uint8).copy())
    api.h.check(L.x265hip_deblock_pictures(st, P(d_jobs), jobs, F, C.c_ssize_t(W), C.c_ssize_t(W // 2)))
    for f in range(F):
        d = descriptor(pics[f], lambda k: arrs[f][k].data_ptr())
        api.h.check(L.x265hip_deblock_frame(st, C.byref(d), C.c_void_p(rec1[0].data_ptr() + f * elems[0] * es), C.c_ssize_t(W), C.c_void_p(rec1[1].data_ptr() + f * elems[1] * es),
                                            C.c_void_p(rec1[2].data_ptr() + f * elems[2] * es), C.c_ssize_t(W // 2), None))
    T.cuda.synchronize()
    for c in range(3):
        assert T.equal(rec[c], rec1[c]), "deblocked plane %d: batch differs from the per-picture calls" % c
    for f in range(F):
        exp = run_oracle(ora, pics[f])
        for c in range(3):
            got = rec[c].cpu().numpy().view(exp[c].dtype)[f * elems[c]:(f + 1) * elems[c]].reshape(shapes[c])
            assert np.array_equal(got, exp[c]), "picture %d plane %d differs from the oracle" % (f, c)
    # ---- SAO statistics, SAO, SSD per plane; SSIM on luma ----
    for c in range(3):
        h, w = shapes[c]
        cs = ctu if c == 0 else ctu // 2
        nctu = ((w + cs - 1) // cs) * ((h + cs - 1) // cs)
        po = 0 if c == 0 else 2
        stats_b = T.zeros(F * nctu * 320, dtype=T.int32, device="cuda"); stats_1 = T.zeros_like(stats_b)
        api.h.check(L.x265hip_sao_stats_pictures(st, P(src[c]), P(rec[c]), C.c_ssize_t(w), w, h, cs, 0, po, P(stats_b), F, C.c_int64(elems[c])))
        prm = np.zeros((F, nctu, 6), np.int32)
        prm[:, :, 0] = rng.integers(-1, 5, (F, nctu)); prm[:, :, 1] = rng.integers(0, 32, (F, nctu)); prm[:, :, 2:] = rng.integers(-7, 8, (F, nctu, 4))
        eo = prm[:, :, 0] <= 3
        prm[:, :, 2:4][eo] = np.abs(prm[:, :, 2:4][eo]); prm[:, :, 4:6][eo] = -np.abs(prm[:, :, 4:6][eo])
        d_prm = api.to_device(prm.reshape(-1))
        out_b = T.zeros_like(rec[c]); out_1 = T.zeros_like(rec[c])
        api.h.check(L.x265hip_sao_apply_pictures(st, P(rec[c]), P(out_b), C.c_ssize_t(w), w, h, cs, P(d_prm), F, C.c_int64(elems[c])))
        ssd_b = T.zeros(F, dtype=T.int64, device="cuda"); ssd_1 = T.zeros(F, dtype=T.int64, device="cuda")
        api.h.check(L.x265hip_plane_ssd_pictures(st, P(src[c]), P(out_b), C.c_ssize_t(w), w, h, P(ssd_b), F, C.c_int64(elems[c]), C.c_int64(elems[c])))
        for f in range(F):
            o = f * elems[c] * es
            api.h.check(L.x265hip_sao_stats_frame(st, C.c_void_p(src[c].data_ptr() + o), C.c_void_p(rec[c].data_ptr() + o), C.c_ssize_t(w), w, h, cs, 0, po,
                                                  C.c_void_p(stats_1.data_ptr() + f * nctu * 320 * 4)))
            api.h.check(L.x265hip_sao_apply_frame(st, C.c_void_p(rec[c].data_ptr() + o), C.c_void_p(out_1.data_ptr() + o), C.c_ssize_t(w), w, h, cs,
                                                  C.c_void_p(d_prm.data_ptr() + f * nctu * 6 * 4)))
            api.h.check(L.x265hip_plane_ssd(st, C.c_void_p(src[c].data_ptr() + o), C.c_void_p(out_1.data_ptr() + o), C.c_ssize_t(w), w, h, C.c_void_p(ssd_1.data_ptr() + 8 * f)))
        T.cuda.synchronize()
        assert T.equal(stats_b, stats_1) and T.equal(out_b, out_1) and T.equal(ssd_b, ssd_1), "plane %d: batch differs from the per-picture calls" % c
        assert int(stats_b.abs().sum()) > 0 and int(ssd_b.sum()) > 0
        if c == 0:
            L.x265hip_ssim_workspace.restype = C.c_size_t
            wsz = L.x265hip_ssim_workspace(w, h) // 4
            nrows = (h + ctu - 1) // ctu
            ws_b = T.zeros(F * wsz, dtype=T.float32, device="cuda"); ws_1 = T.zeros(wsz, dtype=T.float32, device="cuda")
            rs_b = T.zeros(F * nrows, dtype=T.float32, device="cuda"); rc_b = T.zeros(F * nrows, dtype=T.int32, device="cuda"); fr_b = T.zeros(2 * F, dtype=T.float64, device="cuda")
            rs_1 = T.zeros_like(rs_b); rc_1 = T.zeros_like(rc_b); fr_1 = T.zeros_like(fr_b)
            api.h.check(L.x265hip_ssim_pictures(st, P(out_b), C.c_ssize_t(w), P(src[0]), C.c_ssize_t(w), w, h, ctu, P(ws_b), P(rs_b), P(rc_b), P(fr_b), F, C.c_int64(elems[0]), C.c_int64(elems[0])))
            for f in range(F):
                o = f * elems[0] * es
                api.h.check(L.x265hip_ssim_frame(st, C.c_void_p(out_b.data_ptr() + o), C.c_ssize_t(w), C.c_void_p(src[0].data_ptr() + o), C.c_ssize_t(w), w, h, ctu, P(ws_1),
                                                 C.c_void_p(rs_1.data_ptr() + 4 * f * nrows), C.c_void_p(rc_1.data_ptr() + 4 * f * nrows), C.c_void_p(fr_1.data_ptr() + 16 * f)))
            T.cuda.synchronize()
            assert T.equal(rs_b, rs_1) and T.equal(rc_b, rc_1) and T.equal(fr_b, fr_1), "SSIM: batch differs from the per-picture calls"
            assert float(fr_b[1]) > 0
