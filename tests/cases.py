"""Seeded case generators shared by all parity tests.

Input distributions follow the reference's own harnesses (source/test/pixelharness.cpp:40-89,
mbdstharness.cpp:55-89, ipfilterharness.cpp:36-60, intrapredharness.cpp:31-44): three buffer
kinds -- random, all-minimum, all-maximum -- sliding offsets, odd strides.
A case is (label, method_name, args); every backend in tests/backends.py implements the methods.
"""
import numpy as np

LUMA_PU = [(4, 4), (8, 8), (16, 16), (32, 32), (64, 64), (8, 4), (4, 8), (16, 8), (8, 16), (32, 16), (16, 32),
           (64, 32), (32, 64), (16, 12), (12, 16), (16, 4), (4, 16), (32, 24), (24, 32), (32, 8), (8, 32),
           (64, 48), (48, 64), (64, 16), (16, 64)]
CU = [4, 8, 16, 32, 64]
TU = [4, 8, 16, 32]
KINDS = ("rand", "min", "max")


def pix_buf(rng, depth, n, kind):
    pm = (1 << depth) - 1
    dt = np.uint8 if depth == 8 else np.uint16
    if kind == "min":
        return np.zeros(n, dt)
    if kind == "max":
        return np.full(n, pm, dt)
    return rng.integers(0, pm + 1, n).astype(dt)


def short_buf(rng, n, kind, lo, hi):
    if kind == "min":
        return np.full(n, lo, np.int16)
    if kind == "max":
        return np.full(n, hi, np.int16)
    return rng.integers(lo, hi + 1, n).astype(np.int16)


def cases_pixelcmp(depth, rng, reps=2):
    pm = (1 << depth) - 1
    for kind_a in KINDS:
        for kind_b in KINDS:
            if kind_a != "rand" and kind_b != "rand" and kind_a == kind_b:
                continue
            A = pix_buf(rng, depth, 64 * 80, kind_a)
            B = pix_buf(rng, depth, 100 * 80, kind_b)
            for (w, h) in LUMA_PU:
                for _ in range(reps):
                    sb = int(rng.integers(64, 100))
                    oa, ob = int(rng.integers(0, 16)) * 0 + int(rng.integers(0, 8)) * 4, int(rng.integers(0, 33))
                    yield ("sad %dx%d %s/%s" % (w, h, kind_a, kind_b), "sad", (w, h, A, 64, oa, B, sb, ob))
                    yield ("satd %dx%d %s/%s" % (w, h, kind_a, kind_b), "satd", (w, h, A, 64, oa, B, sb, ob))
                    offs = [int(rng.integers(0, 33 + 3 * sb)) for _ in range(4)]
                    yield ("sad_x3 %dx%d" % (w, h), "sad_x3", (w, h, A, oa, B, sb, offs[:3]))
                    yield ("sad_x4 %dx%d" % (w, h), "sad_x4", (w, h, A, oa, B, sb, offs))
            for n in CU:
                sa, sb = int(rng.integers(64, 80)), int(rng.integers(64, 100))
                oa, ob = int(rng.integers(0, 17)), int(rng.integers(0, 33))
                yield ("sa8d %d" % n, "sa8d", (n, A, sa, oa, B, sb, ob))
                yield ("sse_pp %d" % n, "sse_pp", (n, A, sa, oa, B, sb, ob))
                yield ("psy_cost_pp %d" % n, "psy_cost_pp", (n, A, sa, oa, B, sb, ob))
    for kind in KINDS:
        S1 = short_buf(rng, 80 * 80, kind, -pm - 1, pm)
        S2 = short_buf(rng, 100 * 80, "rand", -pm - 1, pm)
        for n in CU:
            sa, sb = int(rng.integers(64, 80)), int(rng.integers(64, 100))
            yield ("sse_ss %d %s" % (n, kind), "sse_ss", (n, S1, sa, 3, S2, sb, 5))
            yield ("ssd_s %d %s" % (n, kind), "ssd_s", (n, S1, sa, 7))


def cases_blockops(depth, rng):
    pm = (1 << depth) - 1
    dt = np.uint8 if depth == 8 else np.uint16
    for kind in KINDS:
        P0 = pix_buf(rng, depth, 100 * 70, kind)
        P1 = pix_buf(rng, depth, 100 * 70, "rand")
        R = short_buf(rng, 100 * 70, kind, -pm - 1, pm)
        for n in CU:
            st = int(rng.integers(n, 97))
            dsh = np.full(100 * 70, -21846, np.int16)
            dpx = np.full(100 * 70, 0xCD, dt)
            yield ("calcresidual %d %s" % (n, kind), "calcresidual", (n, P0, P1, dsh, st))
            ds, s0, s1 = int(rng.integers(n, 97)), int(rng.integers(n, 97)), int(rng.integers(n, 97))
            yield ("sub_ps %d %s" % (n, kind), "sub_ps", (n, dsh, ds, P0, P1, s0, s1))
            yield ("add_ps %d %s" % (n, kind), "add_ps", (n, dpx, ds, P0, R, s0, s1))
            yield ("copy_ss %d" % n, "copy_ss", (n, dsh, ds, R, s0))
            yield ("copy_ps %d" % n, "copy_ps", (n, dsh, ds, P0, s0))
            yield ("copy_sp %d" % n, "copy_sp", (n, dpx, ds, short_buf(rng, 100 * 70, kind, 0, pm), s0))
            yield ("blockfill_s %d" % n, "blockfill_s", (n, dsh, ds, int(rng.integers(-32768, 32768))))
            yield ("transpose %d" % n, "transpose", (n, dpx, P0, s0))
        for n in TU:
            full = short_buf(rng, 100 * 70, kind, -32768, 32767)
            dsh = np.full(100 * 70, -21846, np.int16)
            st = int(rng.integers(n, 97))
            sh = int(rng.integers(1, 8))
            yield ("cpy2Dto1D_shl %d" % n, "cpy2Dto1D_shl", (n, dsh, full, st, sh))
            yield ("cpy2Dto1D_shr %d" % n, "cpy2Dto1D_shr", (n, dsh, full, st, sh))
            yield ("cpy1Dto2D_shl %d" % n, "cpy1Dto2D_shl", (n, dsh, full, st, sh))
            yield ("cpy1Dto2D_shr %d" % n, "cpy1Dto2D_shr", (n, dsh, full, st, sh))
        A0 = short_buf(rng, 100 * 70, kind, -16384, 16383)
        A1 = short_buf(rng, 100 * 70, "rand", -16384, 16383)
        for (w, h) in LUMA_PU:
            dpx = np.full(100 * 70, 0xCD, dt)
            ds, s0, s1 = int(rng.integers(w, 97)), int(rng.integers(w, 97)), int(rng.integers(w, 97))
            yield ("copy_pp %dx%d" % (w, h), "copy_pp", (w, h, dpx, ds, P0, s0))
            yield ("addAvg %dx%d %s" % (w, h, kind), "addAvg", (w, h, A0, A1, dpx, s0, s1, ds))
            yield ("pixelavg_pp %dx%d %s" % (w, h, kind), "pixelavg_pp", (w, h, dpx, ds, P0, s0, P1, s1))
        # weighted prediction (pixelharness.cpp check_weightp: w0 random, shift >= correction)
        corr = 14 - depth
        for _ in range(3):
            w, h = 16 * int(rng.integers(1, 4)), int(rng.integers(1, 20))
            w0 = int(rng.integers(1, 128)); shift = corr + int(rng.integers(0, 7)); rnd = (1 << (shift - 1)) if shift else 0
            off = int(rng.integers(-128, 128)) << (depth - 8)
            dpx = np.full(100 * 70, 0xCD, dt)
            yield ("weight_sp", "weight_sp", (A0, dpx, 96, 80, w, h, w0, rnd, shift, off))
            rnd_pp = rnd & ~((1 << corr) - 1)
            yield ("weight_pp", "weight_pp", (P0, dpx, 96, w, h, w0, rnd_pp, shift, off))
        dpx = np.full(64 * 64, 0xCD, dt)
        yield ("scale1D", "scale1D_128to64", (dpx, P0))
        yield ("scale2D", "scale2D_64to32", (dpx, P0, int(rng.integers(64, 100))))


def cases_transform(depth, rng, reps=2):
    pm = (1 << depth) - 1
    for kind in KINDS:
        for n in TU:
            for _ in range(reps):
                st = int(rng.integers(n, 70))
                src = short_buf(rng, 70 * 32, kind, -pm, pm)
                yield ("dct %d %s" % (n, kind), "dct", (n, src, st))
                coef = short_buf(rng, n * n, kind, -32768, 32767)
                dsh = np.full(70 * 32, -21846, np.int16)
                yield ("idct %d %s" % (n, kind), "idct", (n, coef, dsh, st))
                # realistic coefficients too (small values: exercises rounding not clipping)
                small = short_buf(rng, n * n, "rand", -600, 600)
                yield ("idct %d small" % n, "idct", (n, small, dsh, st))
        src = short_buf(rng, 70 * 4, kind, -pm, pm)
        yield ("dst4 %s" % kind, "dst4", (src, 9))
        yield ("idst4 %s" % kind, "idst4", (short_buf(rng, 16, kind, -32768, 32767), np.full(70 * 4, -21846, np.int16), 11))
    # quant / nquant / dequant with qp-derived parameters (mbdstharness.cpp:139-290, quant.cpp:465-469,555-568)
    quant_scales = [26214, 23302, 20560, 18396, 16384, 14564]
    inv_scales = [40, 45, 51, 57, 64, 72]
    for kind in KINDS:
        for log2n in (2, 3, 4, 5):
            num = 1 << (2 * log2n)
            for _ in range(reps):
                qp = int(rng.integers(0, 52))
                per, rem = qp // 6, qp % 6
                tshift = 15 - depth - log2n
                qbits = 14 + per + tshift
                coef = short_buf(rng, num, kind, -32768, 32767)
                flat = np.full(num, quant_scales[rem], np.int32)
                lst = (rng.integers(1, 200, num) * quant_scales[rem] // 16).astype(np.int32)
                for qc in (flat, lst):
                    add = (171 if rng.integers(0, 2) else 85) << (qbits - 9)
                    yield ("quant n%d qp%d %s" % (num, qp, kind), "quant", (coef, qc, qbits, add, num))
                    yield ("nquant n%d qp%d %s" % (num, qp, kind), "nquant", (coef, qc, qbits, 1 << (qbits - 1), num))
                q = short_buf(rng, num, kind, -32768, 32767)
                shift = log2n + depth - 9 + 5  # QUANT_IQUANT_SHIFT - QUANT_SHIFT - transformShift
                shift = 20 - 14 - tshift
                yield ("dequant_normal n%d qp%d" % (num, qp), "dequant_normal", (q, num, inv_scales[rem] << per, shift))
                deq = (rng.integers(1, 256, num) * inv_scales[rem]).astype(np.int32)
                yield ("dequant_scaling n%d qp%d" % (num, qp), "dequant_scaling", (q, deq, num, per, shift))
                qs = short_buf(rng, num, "rand", -3, 3)
                yield ("count_nonzero %d" % num, "count_nonzero", (1 << log2n, qs))
                resi = short_buf(rng, 70 * 32, "rand", -2, 2)
                yield ("copy_cnt %d" % num, "copy_cnt", (1 << log2n, resi, int(rng.integers(1 << log2n, 70))))
                off = rng.integers(0, 300, num).astype(np.uint16)
                rs = rng.integers(0, 1 << 20, num).astype(np.uint32)
                yield ("denoise %d" % num, "denoise_dct", (coef, rs, off, num))


def cases_interp(depth, rng, reps=1):
    """ipfilterharness.cpp:62-554: source stride random (we keep it >= width+taps so the read
    region stays inside the buffer), dst stride random, every coeffIdx, isRowExt 0/1."""
    dt = np.uint8 if depth == 8 else np.uint16
    for kind in KINDS:
        src = pix_buf(rng, depth, 200 * 90, kind)
        ssrc = short_buf(rng, 200 * 90, kind, -16384 + (0 if kind != "min" else 0), 16383)
        for taps in (8, 4):
            sizes = LUMA_PU if taps == 8 else [(w // 2, h // 2) for (w, h) in LUMA_PU if (w, h) != (4, 4)]
            nidx = 4 if taps == 8 else 8
            for (w, h) in sizes:
                for _ in range(reps):
                    ss = int(rng.integers(w + 8, 110)); ds = int(rng.integers(w, 100))
                    so = 4 * ss + 8
                    idx = int(rng.integers(1, nidx)); idy = int(rng.integers(1, nidx))
                    dpx = np.full(100 * 80, 0xCD, dt); dsh = np.full(100 * 80, -12851, np.int16)
                    L = "%dtap %dx%d %s" % (taps, w, h, kind)
                    yield ("hpp " + L, "interp", ("hpp", taps, w, h, src, ss, so, dpx, ds, idx))
                    yield ("vpp " + L, "interp", ("vpp", taps, w, h, src, ss, so, dpx, ds, idx))
                    yield ("hps " + L, "interp", ("hps", taps, w, h, src, ss, so, dsh, ds, idx, int(rng.integers(0, 2))))
                    yield ("vps " + L, "interp", ("vps", taps, w, h, src, ss, so, dsh, ds, idx))
                    yield ("vsp " + L, "interp", ("vsp", taps, w, h, ssrc, ss, so, dpx, ds, idx))
                    yield ("vss " + L, "interp", ("vss", taps, w, h, ssrc, ss, so, dsh, ds, idx))
                    yield ("p2s " + L, "interp", ("p2s", taps, w, h, src, ss, so, dsh, ds, 0))
                    if taps == 8:
                        yield ("hvpp " + L, "interp", ("hvpp", taps, w, h, src, ss, so, dpx, ds, idx, idy))
        # idx 0 (copy taps) once per family
        dpx = np.full(100 * 80, 0xCD, dt)
        yield ("hpp idx0", "interp", ("hpp", 8, 16, 16, src, 40, 200, dpx, 33, 0))
        yield ("vpp idx0", "interp", ("vpp", 4, 8, 8, src, 40, 200, dpx, 33, 0))


def cases_intra(depth, rng, reps=2):
    dt = np.uint8 if depth == 8 else np.uint16
    for kind in KINDS:
        for n in TU:
            for _ in range(reps):
                nb = pix_buf(rng, depth, 4 * n + 1 + 16, kind)
                flt = np.full(4 * n + 1 + 16, 0xCD, dt)
                yield ("intra_filter %d %s" % (n, kind), "intra_filter", (n, nb, flt))
                ds = int(rng.integers(n, 70))
                for mode in range(35):
                    bf = int(rng.integers(0, 2)) if n <= 16 else 0
                    if mode == 0:
                        bf = 0
                    dpx = np.full(70 * 32, 0xCD, dt)
                    yield ("intra_pred %d m%d f%d %s" % (n, mode, bf, kind), "intra_pred", (n, nb, dpx, ds, mode, bf))
                nb2 = pix_buf(rng, depth, 4 * n + 1 + 16, "rand")
                yield ("allangs %d %s" % (n, kind), "intra_allangs", (n, nb, nb2, int(n <= 16)))


def cases_extras(depth, rng, reps=2):
    """cu[].lowpass_dct (lowpassdct.cpp:34-116) and pu[].ads (pixel.cpp:121-165; the reference's harness has no test for ads)."""
    pm = (1 << depth) - 1
    for kind in KINDS:
        for n in (8, 16, 32):
            for _ in range(reps):
                st = int(rng.integers(n, 70))
                src = short_buf(rng, 70 * 32, kind, -pm, pm)
                yield ("lowpass_dct %d %s" % (n, kind), "lowpass_dct", (n, src, st))
    for n in (8, 16, 32):      # full int16 range: the 2x2 sums and the 8x8 block sum wrap in int16 like the reference's
        yield ("lowpass_dct %d wide" % n, "lowpass_dct", (n, short_buf(rng, 70 * 32, "rand", -32768, 32767), 40))
    for kind in KINDS:         # reference-plane preparation: extendPicBorder (pixel.cpp:1044-1058) and its row slot (ipfilter.cpp:59-77)
        for _ in range(reps):
            w, h = int(rng.integers(8, 90)), int(rng.integers(4, 40))
            mx, my = int(rng.choice([4, 16, 40, 96])), int(rng.integers(1, 20))
            stride = w + 2 * mx + 4 * int(rng.integers(0, 3))
            plane = pix_buf(rng, depth, stride * (h + 2 * my), kind if kind != "rand" else "rand")
            yield ("extend_pic_border %dx%d %s" % (w, h, kind), "extend_pic_border", (plane, stride, w, h, mx, my))
            rows = pix_buf(rng, depth, stride * h, "rand")
            yield ("extend_row_border %dx%d" % (w, h), "extend_row_border", (rows, stride, w, h, mx))
    for kind in KINDS:         # intra mode scan: sa8d of 35 predictions per CU (search.cpp:1655-1745)
        for n in (4, 8, 16, 32, 64):
            for _ in range(reps):
                stride = n + int(rng.integers(0, 40))
                src = pix_buf(rng, depth, stride * n + 64, kind)
                off = int(rng.integers(0, 32))
                nb_ref = pix_buf(rng, depth, 4 * n + 1, "rand" if kind == "rand" else kind)
                nb_filt = pix_buf(rng, depth, 4 * n + 1, "rand")
                yield ("intra_costs %d %s" % (n, kind), "intra_costs", (n, src, stride, off, nb_ref, nb_filt))
    dt = np.uint8 if depth == 8 else np.uint16
    for kind in KINDS:         # lookahead plane preparation: frame_init_lowres_core (pixel.cpp:596-622)
        for _ in range(reps):
            w, h = int(rng.integers(4, 70)), int(rng.integers(2, 30))
            ss, ds = 2 * w + 2 + int(rng.integers(0, 9)), w + int(rng.integers(0, 9))
            src = pix_buf(rng, depth, ss * (2 * h + 2), kind)
            outs = [np.full(ds * h, 0xCD, dt) for _ in range(4)]
            yield ("frame_init_lowres %dx%d %s" % (w, h, kind), "frame_init_lowres", (src, ss, outs[0], outs[1], outs[2], outs[3], ds, w, h))
    for (w, h) in LUMA_PU:
        for _ in range(reps):
            width = int(rng.integers(1, 140))
            delta = int(rng.integers(8, 300))
            base = int(rng.integers(0, w * h * pm + 1))
            span = width + delta + w
            sums = (base + rng.integers(-4000, 4000, span)).clip(0, None).astype(np.uint32)
            enc = (base + rng.integers(-3000, 3000, 4)).clip(0, None).astype(np.int32)
            cost = rng.integers(0, 600, width).astype(np.uint16)
            thresh = int(rng.integers(500, 9000))
            yield ("ads %dx%d" % (w, h), "ads", (w, h, enc, sums, delta, cost, width, thresh))


FAMILIES = {
    "extras": cases_extras,
    "pixelcmp": cases_pixelcmp,
    "blockops": cases_blockops,
    "transform": cases_transform,
    "interp": cases_interp,
    "intra": cases_intra,
}


def run_case(backend, method, args):
    out = getattr(backend, method)(*args)
    return out if isinstance(out, tuple) else (out,)


def same(a, b):
    if len(a) != len(b):
        return False
    for x, y in zip(a, b):
        if isinstance(x, np.ndarray) or isinstance(y, np.ndarray):
            if not np.array_equal(np.asarray(x), np.asarray(y)):
                return False
        elif int(x) != int(y):
            return False
    return True
