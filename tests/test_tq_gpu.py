"""GPU parity of the fused inter-TU pipeline (x265hip_tq_batch) against the oracle (xo_tq_tu):
quantised coefficients, numSig, deltaU, reconstruction and SSE must be bit-identical."""
import numpy as np
import pytest

from depths import DEPTHS

import x265hip  # noqa: F401
from x265hip_pkg.synth import frame_pair
from x265hip_pkg.frame import FrameApi, TU_TASK
from backends import Oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("log2n", [2, 3, 4, 5])
def test_tq_batch_matches_oracle(depth, log2n):
    api, ora = FrameApi(depth), Oracle(depth)
    torch = api.torch
    rng = np.random.default_rng(31 * depth + log2n)
    W, H, margin = 256, 128, 48
    N = 1 << log2n
    cur, ref, stride, (dx, dy) = frame_pair(W, H, depth, 20 + log2n, margin=margin, max_shift=8)
    cur_f, ref_f = cur.reshape(-1), ref.reshape(-1)
    d_cur, d_ref = api.to_device(cur_f), api.to_device(ref_f)
    for (qp, add, recon, flat) in [(28, 85, True, True), (22, 171, True, True), (37, 85, False, True), (45, 85, True, True),
                                   (51, 85, True, True), (30, 85, True, False), (4, 171, True, True)]:
        n = 40
        t = np.zeros(n, TU_TASK)
        for i in range(n):
            px = int(rng.integers(0, (W - N) // 4 + 1)) * 4; py = int(rng.integers(0, (H - N) // 4 + 1)) * 4
            off = (margin + py) * stride + margin + px
            t[i]["mvFrom"] = -1; t[i]["curOff"] = off; t[i]["refOff"] = off; t[i]["reconOff"] = i * N * N      # dense, non-overlapping recon blocks
            if rng.random() < 0.8:
                t[i]["mv"] = (4 * dx + int(rng.integers(-6, 7)), 4 * dy + int(rng.integers(-6, 7)))
            else:
                t[i]["mv"] = (int(rng.integers(-60, 61)), int(rng.integers(-60, 61)))
        qc = None if flat else (rng.integers(8, 64, N * N) * 1024).astype(np.int32)
        d_t = api.to_device(t)
        d_coeff = torch.zeros(n * N * N, dtype=torch.int16, device="cuda")
        d_ns = torch.zeros(n, dtype=torch.int32, device="cuda")
        d_du = torch.zeros(n * N * N, dtype=torch.int32, device="cuda")
        d_rec = torch.zeros(n * N * N, dtype=d_cur.dtype, device="cuda") if recon else None
        d_sse = torch.zeros(n, dtype=torch.int64, device="cuda") if recon else None
        d_qc = api.to_device(qc) if qc is not None else None
        api.tq_batch(log2n, d_cur, stride, d_ref, stride, d_t, n, qp, add, d_coeff, d_ns, quant_coeff=d_qc, delta_u=d_du,
                     recon=d_rec, recon_stride=N, sse=d_sse)
        torch.cuda.synchronize()
        coeff = d_coeff.cpu().numpy().reshape(n, N * N); ns = d_ns.cpu().numpy(); du = d_du.cpu().numpy().reshape(n, N * N)
        rec = d_rec.cpu().numpy().view(cur_f.dtype) if recon else None
        sse = d_sse.cpu().numpy() if recon else None
        kinds = set()
        for i in range(n):
            off = int(t[i]["curOff"]); mv = (int(t[i]["mv"][0]), int(t[i]["mv"][1]))
            e_ns, e_coeff, e_du, e_rec, e_sse = ora.tq_tu(log2n, cur_f, stride, off, ref_f, stride, off, mv, qp, add, quant_coeff=qc, want_recon=recon)
            assert int(ns[i]) == e_ns and np.array_equal(coeff[i], e_coeff), "coeff: N=%d qp=%d task %d mv %s" % (N, qp, i, mv)
            assert np.array_equal(du[i], e_du), "deltaU: N=%d qp=%d task %d" % (N, qp, i)
            if recon:
                got = rec[i * N * N:(i + 1) * N * N]
                assert np.array_equal(got, e_rec), "recon: N=%d qp=%d task %d numSig %d" % (N, qp, i, e_ns)
                assert int(sse[i]) == e_sse, "sse: N=%d qp=%d task %d" % (N, qp, i)
            kinds.add(0 if e_ns == 0 else (1 if e_ns == 1 else 2))


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("log2n", [2, 3, 4, 5])
def test_tq_batch_chroma_matches_oracle(depth, log2n):
    """Chroma TUs of a 4:2:0 picture: motion compensation = predInterChromaPixel (predict.cpp:340-380) with the luma MV
    read in eighth-pels; the rest of the chain is unchanged."""
    api, ora = FrameApi(depth), Oracle(depth)
    torch = api.torch
    rng = np.random.default_rng(77 * depth + log2n)
    W, H, margin = 160, 96, 40                        # the chroma plane of a 320x192 picture
    N = 1 << log2n
    cur, ref, stride, (dx, dy) = frame_pair(W, H, depth, 60 + log2n, margin=margin, max_shift=5)
    cur_f, ref_f = cur.reshape(-1), ref.reshape(-1)
    d_cur, d_ref = api.to_device(cur_f), api.to_device(ref_f)
    for (qp, add, recon) in [(29, 85, True), (20, 171, True), (39, 85, False), (6, 85, True)]:
        n = 48
        t = np.zeros(n, TU_TASK)
        for i in range(n):
            px = int(rng.integers(0, (W - N) // 2 + 1)) * 2; py = int(rng.integers(0, (H - N) // 2 + 1)) * 2
            off = (margin + py) * stride + margin + px
            t[i]["mvFrom"] = -1; t[i]["curOff"] = off; t[i]["refOff"] = off; t[i]["reconOff"] = i * N * N
            kind = i % 6
            if kind == 0:   mv = (8 * dx, 8 * dy)                                                    # full-pel
            elif kind == 1: mv = (8 * dx + int(rng.integers(1, 8)), 8 * dy)                           # horizontal only
            elif kind == 2: mv = (8 * dx, 8 * dy + int(rng.integers(1, 8)))                           # vertical only
            elif kind == 3: mv = (int(rng.integers(-100, 101)), int(rng.integers(-100, 101)))
            else:           mv = (8 * dx + int(rng.integers(-12, 13)), 8 * dy + int(rng.integers(-12, 13)))
            t[i]["mv"] = mv
        d_t = api.to_device(t)
        d_coeff = torch.zeros(n * N * N, dtype=torch.int16, device="cuda")
        d_ns = torch.zeros(n, dtype=torch.int32, device="cuda")
        d_du = torch.zeros(n * N * N, dtype=torch.int32, device="cuda")
        d_rec = torch.zeros(n * N * N, dtype=d_cur.dtype, device="cuda") if recon else None
        d_sse = torch.zeros(n, dtype=torch.int64, device="cuda") if recon else None
        api.tq_batch(log2n, d_cur, stride, d_ref, stride, d_t, n, qp, add, d_coeff, d_ns, delta_u=d_du,
                     recon=d_rec, recon_stride=N, sse=d_sse, chroma=True)
        torch.cuda.synchronize()
        coeff = d_coeff.cpu().numpy().reshape(n, N * N); ns = d_ns.cpu().numpy(); du = d_du.cpu().numpy().reshape(n, N * N)
        rec = d_rec.cpu().numpy().view(cur_f.dtype) if recon else None
        sse = d_sse.cpu().numpy() if recon else None
        coded = 0
        for i in range(n):
            off = int(t[i]["curOff"]); mv = (int(t[i]["mv"][0]), int(t[i]["mv"][1]))
            e_ns, e_coeff, e_du, e_rec, e_sse = ora.tq_tu(log2n, cur_f, stride, off, ref_f, stride, off, mv, qp, add, want_recon=recon, chroma=True)
            assert int(ns[i]) == e_ns and np.array_equal(coeff[i], e_coeff), "chroma coeff: N=%d qp=%d task %d mv %s" % (N, qp, i, mv)
            assert np.array_equal(du[i], e_du), "chroma deltaU: N=%d qp=%d task %d" % (N, qp, i)
            if recon:
                assert np.array_equal(rec[i * N * N:(i + 1) * N * N], e_rec), "chroma recon: N=%d qp=%d task %d mv %s" % (N, qp, i, mv)
                assert int(sse[i]) == e_sse, "chroma sse: N=%d qp=%d task %d" % (N, qp, i)
            coded += e_ns > 0
        assert coded or qp > 35


@pytest.mark.parametrize("depth", DEPTHS)
@pytest.mark.parametrize("log2n", [2, 3, 4, 5])
def test_tq_batch_bidirectional_matches_oracle(depth, log2n):
    """Bi-directionally predicted TUs: two 14-bit predictions (Predict::predInterLumaShort: p2s | hps | vps | hps + vss) -> addAvg (predict.cpp:186-211).
    The launch takes the TUs whose PU chose exactly (list-0 reference 1, list-1 reference 0); TUs of other choices keep their output untouched."""
    from x265hip_pkg.frame import INTER_CHOICE
    api, ora = FrameApi(depth), Oracle(depth)
    torch = api.torch
    rng = np.random.default_rng(500 * depth + log2n)
    W, H, margin = 256, 128, 48
    N = 1 << log2n
    cur, ref0, stride, (dx, dy) = frame_pair(W, H, depth, 70 + log2n, margin=margin, max_shift=6)
    _, ref1, _, (ex, ey) = frame_pair(W, H, depth, 70 + log2n, margin=margin, max_shift=6)      # same source picture, another displacement draw
    ref1 = np.ascontiguousarray(ref1[::-1, ::-1]) if np.array_equal(ref0, ref1) else ref1     # (identical draws would make the average trivial)
    cur_f, r0_f, r1_f = cur.reshape(-1), ref0.reshape(-1), ref1.reshape(-1)
    d_cur, d_r0, d_r1 = api.to_device(cur_f), api.to_device(r0_f), api.to_device(r1_f)
    for (qp, recon) in [(27, True), (35, False), (12, True)]:
        n = 48
        t = np.zeros(n, TU_TASK); ch = np.zeros(n, INTER_CHOICE)
        for i in range(n):
            px = int(rng.integers(0, (W - N) // 4 + 1)) * 4; py = int(rng.integers(0, (H - N) // 4 + 1)) * 4
            off = (margin + py) * stride + margin + px
            t[i]["mvFrom"] = i; t[i]["curOff"] = off; t[i]["refOff"] = off; t[i]["reconOff"] = i * N * N
            kind = i % 8
            mv0 = [(0, 0), (4 * dx, 4 * dy), (4 * dx + 2, 4 * dy), (4 * dx, 4 * dy + 1), (4 * dx + 3, 4 * dy + 2)][kind % 5] if kind < 5 else (int(rng.integers(-30, 31)), int(rng.integers(-30, 31)))
            mv1 = (int(rng.integers(-30, 31)), int(rng.integers(-30, 31))) if kind % 2 else (4 * int(rng.integers(-5, 6)), 4 * int(rng.integers(-5, 6)))
            ch[i]["mv"][0] = mv0; ch[i]["mv"][1] = mv1
            ch[i]["ref"] = (1, 0) if i % 6 else ((1, -1) if i % 12 else (0, 0))                   # every sixth PU chose something else
        d_t, d_ch = api.to_device(t), api.to_device(ch)
        d_coeff = torch.full((n * N * N,), 7, dtype=torch.int16, device="cuda")
        d_ns = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        d_du = torch.zeros(n * N * N, dtype=torch.int32, device="cuda")
        d_rec = torch.zeros(n * N * N, dtype=d_cur.dtype, device="cuda") if recon else None
        d_sse = torch.zeros(n, dtype=torch.int64, device="cuda") if recon else None
        api.tq_batch(log2n, d_cur, stride, d_r0, stride, d_t, n, qp, 85, d_coeff, d_ns, delta_u=d_du, recon=d_rec, recon_stride=N, sse=d_sse,
                     choice=d_ch, choice_list=0, choice_ref=1, ref1=d_r1, choice_ref1=0)
        torch.cuda.synchronize()
        coeff = d_coeff.cpu().numpy().reshape(n, N * N); ns = d_ns.cpu().numpy(); du = d_du.cpu().numpy().reshape(n, N * N)
        rec = d_rec.cpu().numpy().view(cur_f.dtype) if recon else None
        sse = d_sse.cpu().numpy() if recon else None
        done = 0
        for i in range(n):
            if tuple(ch[i]["ref"]) != (1, 0):
                assert int(ns[i]) == -1 and (coeff[i] == 7).all(), "a TU of another choice was touched (task %d)" % i
                continue
            off = int(t[i]["curOff"])
            mv0 = (int(ch[i]["mv"][0][0]), int(ch[i]["mv"][0][1])); mv1 = (int(ch[i]["mv"][1][0]), int(ch[i]["mv"][1][1]))
            e_ns, e_coeff, e_du, e_rec, e_sse = ora.tq_tu_bi(log2n, cur_f, stride, off, r0_f, r1_f, stride, off, mv0, mv1, qp, 85, want_recon=recon)
            assert int(ns[i]) == e_ns and np.array_equal(coeff[i], e_coeff), "bi coeff: N=%d qp=%d task %d mv %s %s" % (N, qp, i, mv0, mv1)
            assert np.array_equal(du[i], e_du)
            if recon:
                assert np.array_equal(rec[i * N * N:(i + 1) * N * N], e_rec), "bi recon: N=%d qp=%d task %d mv %s %s" % (N, qp, i, mv0, mv1)
                assert int(sse[i]) == e_sse
            done += 1
        assert done >= 36


@pytest.mark.parametrize("depth", DEPTHS)
def test_tq_batch_intra_4x4_dst_matches_oracle(depth):
    """The chain of an intra luma 4x4 TU (quant.cpp:429-432, 585-603): prediction from a plane the caller filled (MV 0), rounding 171, the DST-VII
    pair instead of the DCT, no DC-only shortcut in the inverse."""
    api, ora = FrameApi(depth), Oracle(depth)
    torch = api.torch
    rng = np.random.default_rng(9 + depth)
    W, H, margin = 128, 64, 16
    cur, pred, stride, _ = frame_pair(W, H, depth, 91, margin=margin, max_shift=2)       # "pred": any plane close to the source does
    cur_f, pred_f = cur.reshape(-1), pred.reshape(-1)
    d_cur, d_pred = api.to_device(cur_f), api.to_device(pred_f)
    for qp in (10, 24, 33, 44):
        n = 96
        t = np.zeros(n, TU_TASK)
        for i in range(n):
            px = int(rng.integers(0, (W - 4) // 4 + 1)) * 4; py = int(rng.integers(0, (H - 4) // 4 + 1)) * 4
            off = (margin + py) * stride + margin + px
            t[i]["mvFrom"] = -1; t[i]["curOff"] = off; t[i]["refOff"] = off; t[i]["reconOff"] = i * 16
        d_t = api.to_device(t)
        d_coeff = torch.zeros(n * 16, dtype=torch.int16, device="cuda"); d_ns = torch.zeros(n, dtype=torch.int32, device="cuda")
        d_du = torch.zeros(n * 16, dtype=torch.int32, device="cuda")
        d_rec = torch.zeros(n * 16, dtype=d_cur.dtype, device="cuda"); d_sse = torch.zeros(n, dtype=torch.int64, device="cuda")
        api.tq_batch(2, d_cur, stride, d_pred, stride, d_t, n, qp, 171, d_coeff, d_ns, delta_u=d_du, recon=d_rec, recon_stride=4, sse=d_sse, dst4=True)
        torch.cuda.synchronize()
        coeff = d_coeff.cpu().numpy().reshape(n, 16); ns = d_ns.cpu().numpy(); rec = d_rec.cpu().numpy().view(cur_f.dtype); sse = d_sse.cpu().numpy()
        for i in range(n):
            off = int(t[i]["curOff"])
            e_ns, e_coeff, e_du, e_rec, e_sse = ora.tq_tu_dst4(cur_f, stride, off, pred_f, stride, off, qp, 171, want_recon=True)
            assert int(ns[i]) == e_ns and np.array_equal(coeff[i], e_coeff), "dst4 coeff: qp=%d task %d" % (qp, i)
            assert np.array_equal(rec[i * 16:(i + 1) * 16], e_rec) and int(sse[i]) == e_sse, "dst4 recon: qp=%d task %d numSig %d" % (qp, i, e_ns)
