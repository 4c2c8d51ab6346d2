"""Shared plumbing of the deblocking tests: a random but well-formed coded picture (CU quadtree, prediction partitions, transform quadtree, modes, cbf,
QPs, references, motion vectors -- the per-partition arrays of the reference's CUData, CTU after CTU in z-scan order), the driver of
oracle/_ref/x265deblock_* and the oracle call."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
U8 = ("log2CUSize", "cuDepth", "partSize", "tuDepth", "predMode", "cbfLuma", "tqBypass")
I8 = ("qp", "refIdx0", "refIdx1")


def dbk_bin(depth):
    return os.path.join(ROOT, "oracle", "_ref", "x265deblock_%d" % depth)


def morton(lx, ly):
    z = 0
    for b in range(4):
        z |= ((lx >> b) & 1) << (2 * b) | ((ly >> b) & 1) << (2 * b + 1)
    return z


def chroma_shifts(csp):
    """(horizontal, vertical) subsampling shifts of X265_CSP_I420 = 1 / I422 = 2 / I444 = 3"""
    return (0 if csp == 3 else 1), (1 if csp in (0, 1) else 0)


def coded_picture(depth, W, H, ctu, seed, slice_p=False, bypass=False, qp_range=(18, 46), csp=1):
    """W, H multiples of 8.  Returns dict(planes=[Y, Cb, Cr], arrays..., refPic, params)"""
    rng = np.random.default_rng(seed)
    nx, ny, upc = (W + ctu - 1) // ctu, (H + ctu - 1) // ctu, ctu // 4
    n = nx * ny * upc * upc
    a = {k: np.zeros(n, np.uint8) for k in U8}
    a.update({k: np.full(n, -1 if k != "qp" else 30, np.int8) for k in I8})
    a["mv0"], a["mv1"] = np.zeros((n, 2), np.int32), np.zeros((n, 2), np.int32)
    pm = (1 << depth) - 1
    Y = np.zeros((H, W), np.int64)
    ref_pic = np.array([[3, 5, 3, 9] + [20 + i for i in range(12)], [5, 3, 11, 3] + [40 + i for i in range(12)]], np.int32)   # duplicates on purpose

    def part(x, y):          # luma position -> array index
        return ((y // ctu) * nx + x // ctu) * upc * upc + morton((x % ctu) // 4, (y % ctu) // 4)

    def tu_tree(x, y, size, d, cu_idx_set):
        split = size > 32 or (size > 4 and d < 3 and rng.random() < 0.4)
        if split:
            h = size // 2
            for (dx, dy) in ((0, 0), (h, 0), (0, h), (h, h)):
                tu_tree(x + dx, y + dy, h, d + 1, cu_idx_set)
            return
        cbf = int(rng.random() < 0.5)
        for yy in range(y, y + size, 4):
            for xx in range(x, x + size, 4):
                i = part(xx, yy)
                a["tuDepth"][i] = d; a["cbfLuma"][i] = cbf << d

    def cu_tree(x, y, size, d):
        if x >= W or y >= H:
            return
        inside = x + size <= W and y + size <= H
        if not inside or (size > 8 and rng.random() < (0.75 if size > 16 else 0.5)):
            h = size // 2
            for (dx, dy) in ((0, 0), (h, 0), (0, h), (h, h)):
                cu_tree(x + dx, y + dy, h, d + 1)
            return
        r = rng.random()
        mode = 2 if r < 0.3 else (5 if r < 0.45 else 1)                       # MODE_INTRA / MODE_SKIP / MODE_INTER (cudata.h:58-64)
        if mode == 2:
            ps = 3 if (size == 8 and rng.random() < 0.4) else 0
        elif mode == 5:
            ps = 0
        else:
            ps = int(rng.choice([0, 1, 2, 4, 5, 6, 7] if size >= 16 else [0, 1, 2]))
        qp = int(rng.integers(qp_range[0], qp_range[1] + 1)) if rng.random() < 0.9 else int(rng.choice([0, 51]))
        byp = int(bypass and rng.random() < 0.25)
        # prediction units: (x0, y0, w, h) relative, partTable of cudata.cpp:146-157
        q, hh, t = size // 4, size // 2, size - size // 4
        pus = {0: [(0, 0, size, size)], 1: [(0, 0, size, hh), (0, hh, size, hh)], 2: [(0, 0, hh, size), (hh, 0, hh, size)],
               3: [(0, 0, hh, hh), (hh, 0, hh, hh), (0, hh, hh, hh), (hh, hh, hh, hh)], 4: [(0, 0, size, q), (0, q, size, t)], 5: [(0, 0, size, t), (0, t, size, q)],
               6: [(0, 0, q, size), (q, 0, t, size)], 7: [(0, 0, t, size), (t, 0, q, size)]}[ps]
        base = rng.integers(-40, 41, 2)
        for (px, py, pw, ph) in pus:
            if mode == 2:
                r0 = r1 = -1; m0 = m1 = (0, 0)
            else:
                kind = 0 if slice_p else int(rng.integers(0, 3))            # list 0 only / list 1 only / both
                r0 = int(rng.integers(0, 4)) if kind != 1 else -1
                r1 = int(rng.integers(0, 4)) if kind != 0 else -1
                m0 = base + rng.integers(-5, 6, 2) if rng.random() < 0.8 else base
                m1 = base + rng.integers(-5, 6, 2) if rng.random() < 0.8 else m0
            for yy in range(y + py, y + py + ph, 4):
                for xx in range(x + px, x + px + pw, 4):
                    i = part(xx, yy)
                    a["refIdx0"][i], a["refIdx1"][i] = r0, r1
                    a["mv0"][i], a["mv1"][i] = (m0 if r0 >= 0 else (7, -9)), (m1 if r1 >= 0 else (-3, 8))     # garbage where unused, like stale CUData
        for yy in range(y, y + size, 4):
            for xx in range(x, x + size, 4):
                i = part(xx, yy)
                a["log2CUSize"][i] = size.bit_length() - 1; a["cuDepth"][i] = d; a["partSize"][i] = ps; a["predMode"][i] = mode; a["qp"][i] = qp; a["tqBypass"][i] = byp
        tu_tree(x, y, size, 0, None)
        if mode == 2 and ps == 3:                                            # intra NxN: the transform tree splits at least once
            for yy in range(y, y + size, 4):
                for xx in range(x, x + size, 4):
                    i = part(xx, yy)
                    if a["tuDepth"][i] == 0:
                        a["tuDepth"][i] = 1; a["cbfLuma"][i] = (int(a["cbfLuma"][i]) & 1) << 1
        # content: a level per CU, a gentle ramp, now and then a step at a transform edge, little noise
        level = rng.integers(pm // 8, pm - pm // 8)
        gy, gx = np.mgrid[0:size, 0:size]
        blk = level + (gx * rng.integers(-2, 3) + gy * rng.integers(-2, 3)) * (1 << (depth - 8)) // 2
        Y[y:y + size, x:x + size] = blk

    for cy in range(ny):
        for cx in range(nx):
            cu_tree(cx * ctu, cy * ctu, ctu, 0)
    step = (rng.integers(-6, 7, (H // 4, W // 4)) * (rng.random((H // 4, W // 4)) < 0.6)).repeat(4, 0).repeat(4, 1) * (1 << (depth - 8))
    noise = rng.integers(-1, 2, (H, W)) * (1 << (depth - 8))
    big = rng.random((H // 8, W // 8)) < 0.05                               # a few strong edges that must stay unfiltered
    Yf = np.clip(Y // 3 + pm // 3 + step + noise + big.repeat(8, 0).repeat(8, 1) * (pm // 3), 0, pm)
    k = np.random.default_rng(seed + 1)
    hs, vs = chroma_shifts(csp)
    sub = Yf[::1 << vs, ::1 << hs]
    Cb = np.clip(sub // 2 + pm // 4 + k.integers(-3, 4, sub.shape) * (1 << (depth - 8)), 0, pm)
    Cr = np.clip(pm - sub // 2 - pm // 4 + k.integers(-3, 4, sub.shape) * (1 << (depth - 8)), 0, pm)
    if depth == 8 and seed % 3 == 0:                                        # extremes: saturated neighbours
        Yf[: H // 4] = np.where(k.random((H // 4, W)) < 0.5, pm, pm - 2)
    dt = np.uint8 if depth == 8 else np.uint16
    return dict(a, planes=[Yf.astype(dt), Cb.astype(dt), Cr.astype(dt)], refPic=ref_pic, W=W, H=H, ctu=ctu, depth=depth, csp=csp, slice_p=int(slice_p), bypass=int(bypass),
                beta_div2=int(rng.integers(-3, 4)), tc_div2=int(rng.integers(-3, 4)), cb_off=int(rng.integers(-6, 7)), cr_off=int(rng.integers(-6, 7)))


def run_reference(pic):
    with tempfile.TemporaryDirectory() as td:
        inp, out = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        upc2 = (pic["ctu"] // 4) ** 2
        nctu = len(pic["qp"]) // upc2
        with open(inp, "wb") as f:
            for p in pic["planes"]:
                f.write(p.tobytes())
            for c in range(nctu):
                s = slice(c * upc2, (c + 1) * upc2)
                for k in U8 + I8:
                    f.write(pic[k][s].tobytes())
                f.write(pic["mv0"][s].tobytes()); f.write(pic["mv1"][s].tobytes())
            f.write(pic["refPic"].tobytes())
        r = subprocess.run([dbk_bin(pic["depth"]), str(pic["W"]), str(pic["H"]), str(pic["ctu"]), inp, out, str(pic["slice_p"]), str(pic["beta_div2"]), str(pic["tc_div2"]),
                            str(pic["cb_off"]), str(pic["cr_off"]), str(pic["bypass"])], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, X265REF_SLICE_ROWS=",".join(str(r) for r in pic.get("slice_rows", ())), X265REF_CSP=str(pic.get("csp", 1))))
        assert r.returncode == 0, r.stderr[-2000:]
        d = np.fromfile(out, np.uint16)
    W, H = pic["W"], pic["H"]
    hs, vs = chroma_shifts(pic.get("csp", 1))
    cw, ch = W >> hs, H >> vs
    return [d[:W * H].reshape(H, W), d[W * H:W * H + cw * ch].reshape(ch, cw), d[W * H + cw * ch:].reshape(ch, cw)]


class DeblockPic(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("width", "height", "ctuSize", "sliceIsP", "betaOffsetDiv2", "tcOffsetDiv2", "cbQpOffset", "crQpOffset", "tqBypassEnabled")] + \
               [(k, C.c_void_p) for k in ("log2CUSize", "partSize", "tuDepth", "predMode", "cbfLuma", "tqBypass", "qp", "refIdx0", "refIdx1", "mv0", "mv1")] + \
               [("refPic", C.c_int32 * 32), ("sliceFirstRow", C.c_void_p), ("chromaFormat", C.c_int)]


def slice_first_row(pic):
    """--slices: one byte per CTU row (+ a 0), non-zero where pic["slice_rows"] says a slice begins; None for one slice"""
    rows = pic.get("slice_rows", ())
    if not rows:
        return None
    n = (pic["H"] + pic["ctu"] - 1) // pic["ctu"]
    a = np.zeros(n + 1, np.uint8)
    for r in rows:
        if 0 < r < n:
            a[r] = 1
    return a


def descriptor(pic, ptr):
    """ptr(name) -> address of that array (host or device)"""
    d = DeblockPic(pic["W"], pic["H"], pic["ctu"], pic["slice_p"], pic["beta_div2"], pic["tc_div2"], pic["cb_off"], pic["cr_off"], pic["bypass"])
    for k in ("log2CUSize", "partSize", "tuDepth", "predMode", "cbfLuma", "tqBypass", "qp", "refIdx0", "refIdx1", "mv0", "mv1"):
        setattr(d, k, ptr(k))
    d.refPic[:] = [int(v) for v in pic["refPic"].reshape(-1)]
    d.chromaFormat = int(pic.get("csp", 1))
    return d


def run_oracle(ora, pic, want_bs=False):
    planes = [np.ascontiguousarray(p.copy()) for p in pic["planes"]]
    keep = {k: np.ascontiguousarray(pic[k]) for k in U8 + I8 + ("mv0", "mv1")}
    d = descriptor(pic, lambda k: keep[k].ctypes.data)
    sfr = slice_first_row(pic)
    if sfr is not None:
        d.sliceFirstRow = sfr.ctypes.data
    W, H = pic["W"], pic["H"]
    bs = np.zeros((2, H // 4, W // 4), np.uint8)
    P = lambda x: C.c_void_p(x.ctypes.data)
    ora.lib.xo_deblock_frame(C.byref(d), P(planes[0]), C.c_ssize_t(W), P(planes[1]), P(planes[2]), C.c_ssize_t(planes[1].shape[1]), P(bs))
    return (planes, bs) if want_bs else planes
