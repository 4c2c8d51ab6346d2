"""Shared plumbing of the lookahead frame-cost tests: synthetic clips, the lowres picture geometry of the reference
(picyuv.cpp:88-93, lowres.cpp:84-103), the oracle-side drivers and the parser of oracle/_ref/x265la_* output."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CTU = 64
MARGIN_X, MARGIN_Y = CTU + 32, CTU + 16          # PicYuv::create (picyuv.cpp:91-92); the lowres planes reuse them (lowres.cpp:87,102-103)


def la_bin(depth):
    return os.path.join(ROOT, "oracle", "_ref", "x265la_%d" % depth)


def la_available(depth):
    return os.path.exists(la_bin(depth))


def synth_clip(W, H, n, depth, seed, shift=(3, 2), noise=2):
    """n frames of a smooth texture panning by `shift` full-res pixels per frame, plus noise and one local change"""
    rng = np.random.default_rng(seed)
    pad = 16 + max(abs(shift[0]), abs(shift[1])) * n
    base = rng.integers(0, 256, (H + 2 * pad, W + 2 * pad)).astype(np.float32)
    for _ in range(3):
        base = (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(np.roll(base, 1, 0), 1, 1)) / 4
    base = (base - base.min()) / (base.max() - base.min())
    pm = (1 << depth) - 1
    frames = []
    for f in range(n):
        y0, x0 = pad + shift[1] * f, pad + shift[0] * f
        fr = base[y0:y0 + H, x0:x0 + W] * pm + rng.normal(0, noise * (1 << (depth - 8)), (H, W))
        if f == n // 2:                                        # a patch of new content: intra wins there
            ph, pw = min(24, H - H // 4), min(40, W - W // 3)
            fr[H // 4:H // 4 + ph, W // 3:W // 3 + pw] = rng.integers(0, pm + 1, (ph, pw))
        frames.append(np.clip(np.rint(fr), 0, pm).astype(np.uint8 if depth == 8 else np.uint16))
    return frames


class Geometry:
    def __init__(self, W, H):
        self.W, self.H = W, H
        self.full_stride = (W + CTU - 1) // CTU * CTU + 2 * MARGIN_X
        self.full_rows = (H + CTU - 1) // CTU * CTU + 2 * MARGIN_Y
        self.wcu, self.hcu = (W // 2 + 7) >> 3, (H // 2 + 7) >> 3
        self.lw, self.lh = self.wcu * 8, self.hcu * 8
        s = W // 2 + 2 * MARGIN_X
        self.stride = s + (32 - (s & 31)) % 32 if (s & 31) else s
        self.rows = self.lh + 2 * MARGIN_Y
        self.plane_elems = self.stride * self.rows
        self.origin = self.stride * MARGIN_Y + MARGIN_X
        self.ncu = self.wcu * self.hcu
        # --hme: the quarter-resolution planes (lowres.cpp:171-188: rows lumaStride / 2 apart, planesize / 2 pixels, pixel (0,0) at padoffset / 2) on the
        # Lookahead::m_4x4Width x m_4x4Height grid (slicetype.cpp:1112-1113)
        self.stride4, self.plane_elems4, self.origin4 = self.stride // 2, self.plane_elems // 2, self.origin // 2
        self.wcu4, self.hcu4 = (W // 4 + 7) >> 3, (H // 4 + 7) >> 3
        self.ncu4 = self.wcu4 * self.hcu4


def pad_full(ora, frame, g):
    """the source picture inside its padded allocation, borders replicated (what ref_lookahead.cpp builds with the reference's extendPicBorder)"""
    buf = np.zeros((g.full_rows, g.full_stride), frame.dtype)
    buf[MARGIN_Y:MARGIN_Y + g.H, MARGIN_X:MARGIN_X + g.W] = frame
    return ora.extend_pic_border(buf.reshape(-1), g.full_stride, g.W, g.H, MARGIN_X, MARGIN_Y)


def lowres_planes_oracle(ora, frame, g):
    """Lowres::init (lowres.cpp:381-391): frameInitLowres + extendPicBorder on the four planes -> array [4, plane_elems]"""
    full = pad_full(ora, frame, g)
    z = [np.zeros(g.plane_elems, frame.dtype) for _ in range(4)]
    src = full[g.full_stride * MARGIN_Y + MARGIN_X:]
    dst = [p[g.origin:] for p in z]
    L = ora.lib
    ip = C.c_ssize_t
    P = lambda a: C.c_void_p(a.ctypes.data)
    L.xo_frame_init_lowres(P(src), P(dst[0]), P(dst[1]), P(dst[2]), P(dst[3]), ip(g.full_stride), ip(g.stride), g.lw, g.lh)
    return np.stack([ora.extend_pic_border(p, g.stride, g.lw, g.lh, MARGIN_X, MARGIN_Y) for p in z])


def lowerres_planes_oracle(ora, planes, g):
    """Lowres::init with --hme (lowres.cpp:393-403): frameInitLowerRes of the full-pel lowres plane + extendPicBorder by half the margins -> array [4, plane_elems4]"""
    z = [np.zeros(g.plane_elems4, planes.dtype) for _ in range(4)]
    dst = [p[g.origin4:] for p in z]
    ip = C.c_ssize_t
    P = lambda a: C.c_void_p(a.ctypes.data)
    ora.lib.xo_frame_init_lowres(P(planes[0][g.origin:]), P(dst[0]), P(dst[1]), P(dst[2]), P(dst[3]), ip(g.stride), ip(g.stride4), g.lw // 2, g.lh // 2)
    # pixel (0,0) sits MARGIN_Y rows (not MARGIN_Y / 2) below the start of the plane: the rows above the extended border stay zero
    skip = (MARGIN_Y - MARGIN_Y // 2) * g.stride4
    assert g.origin4 == skip + (MARGIN_Y // 2) * g.stride4 + MARGIN_X // 2
    out = []
    for p in z:
        q = p.copy()
        q[skip:] = ora.extend_pic_border(p[skip:].copy(), g.stride4, g.lw // 2, g.lh // 2, MARGIN_X // 2, MARGIN_Y // 2)
        out.append(q)
    return np.stack(out)


class XoLaHme(C.Structure):
    _fields_ = [("fenc", C.c_void_p), ("ref0", C.c_void_p), ("ref1", C.c_void_p), ("stride", C.c_ssize_t), ("wcu", C.c_int), ("hcu", C.c_int),
                ("method", C.c_int * 2), ("range", C.c_int * 2), ("mvs", C.c_void_p * 2), ("mvCosts", C.c_void_p * 2)]


def _P(a, off=0):
    return C.c_void_p(a.ctypes.data + off * a.itemsize)


def oracle_intra(ora, planes, g, inv_q=None):
    L = ora.me_lib
    ic, im, lc = (np.zeros(g.ncu, np.int32) for _ in range(3))
    rs = np.zeros(g.hcu, np.int32); sums = np.zeros(2, np.int64)
    L.xo_lowres_intra_estimate(_P(planes[0], g.origin), C.c_ssize_t(g.stride), g.wcu, g.hcu, _P(inv_q) if inv_q is not None else None,
                               _P(ic), _P(im), _P(lc), _P(rs), _P(sums))
    return dict(intraCost=ic, intraMode=im, lowresCosts=lc, rowSatds=rs, costEst=int(sums[0]), costEstAq=int(sums[1]))


def lookahead_cost_row(ora, half=1 << 13):
    return ora.mvcost_row(int(ora.me_lib.xo_lookahead_qp()), half), half


def oracle_frame_cost(ora, fenc_planes, ref0_planes, ref1_planes, g, intra_cost, inv_q, state=None, do_search=(1, 1), ref0w_planes=None, rows_per_slice=0, hme=None):
    """state = dict(mvs0, mvc0, mvs1, mvc1) carried between estimates that share a reference distance (in/out).
    hme = dict(fenc, ref0, ref1 (quarter-resolution plane arrays [4, plane_elems4]; ref1 None in a P estimate), method (m0, m1), range (r0, r1)): the --hme sweep;
    the result then carries lmvs0 / lmvc0 / lmvs1 / lmvc1 (Lowres::lowerResMvs / lowerResMvCosts of the searched lists)"""
    L = ora.me_lib
    row, half = lookahead_cost_row(ora)
    st = state if state is not None else {}
    for k, n in (("mvs0", 2 * g.ncu), ("mvc0", g.ncu), ("mvs1", 2 * g.ncu), ("mvc1", g.ncu)):
        st.setdefault(k, np.zeros(n, np.int32))
    lc = np.zeros(g.ncu, np.int32); rs = np.zeros(g.hcu, np.int32); sums = np.zeros(3, np.int64)
    VP = C.c_void_p * 4
    r0 = VP(*[ref0_planes[k].ctypes.data + g.origin * ref0_planes.itemsize for k in range(4)])
    r1 = VP(*[ref1_planes[k].ctypes.data + g.origin * ref1_planes.itemsize for k in range(4)]) if ref1_planes is not None else None
    rw = VP(*[ref0w_planes[k].ctypes.data + g.origin * ref0w_planes.itemsize for k in range(4)]) if ref0w_planes is not None else None
    hs, low = None, {}
    if hme is not None:
        for k, n in (("lmvs0", 2 * g.ncu4), ("lmvc0", g.ncu4), ("lmvs1", 2 * g.ncu4), ("lmvc1", g.ncu4)):
            low[k] = np.zeros(n, np.int32)
        q0 = VP(*[hme["ref0"][k].ctypes.data + g.origin4 * hme["ref0"].itemsize for k in range(4)])
        q1 = VP(*[hme["ref1"][k].ctypes.data + g.origin4 * hme["ref1"].itemsize for k in range(4)]) if hme.get("ref1") is not None else None
        hs = XoLaHme(hme["fenc"][0].ctypes.data + g.origin4 * hme["fenc"].itemsize, C.cast(q0, C.c_void_p), C.cast(q1, C.c_void_p) if q1 is not None else None, g.stride4, g.wcu4, g.hcu4,
                     (C.c_int * 2)(*hme["method"]), (C.c_int * 2)(*hme["range"]), (C.c_void_p * 2)(low["lmvs0"].ctypes.data, low["lmvs1"].ctypes.data),
                     (C.c_void_p * 2)(low["lmvc0"].ctypes.data, low["lmvc1"].ctypes.data))
    L.xo_lowres_frame_cost_hme(_P(fenc_planes[0], g.origin), r0, r1, rw, C.c_ssize_t(g.stride), g.wcu, g.hcu, _P(intra_cost),
                               _P(inv_q) if inv_q is not None else None, _P(row, half), int(do_search[0]), int(do_search[1]), int(rows_per_slice),
                               _P(st["mvs0"]), _P(st["mvc0"]), _P(st["mvs1"]), _P(st["mvc1"]), _P(lc), _P(rs), _P(sums), C.byref(hs) if hs is not None else None)
    return dict(low, mvs0=st["mvs0"].copy(), mvc0=st["mvc0"].copy(), mvs1=st["mvs1"].copy(), mvc1=st["mvc1"].copy(), lowresCosts=lc, rowSatds=rs,
                costEst=int(sums[0]), costEstAq=int(sums[1]), intraMbs=int(sums[2]))


def run_reference(depth, frames, triples, aq, hme=None):
    """oracle/_ref/x265la_<depth> on the clip -> (header dict, per-frame dicts, per-triple dicts); hme = (method0, method1, range0, range1) turns --hme on"""
    H, W = frames[0].shape
    with tempfile.TemporaryDirectory() as td:
        inp, out = os.path.join(td, "in.raw"), os.path.join(td, "out.bin")
        np.stack(frames).tofile(inp)
        args = [la_bin(depth), str(W), str(H), str(len(frames)), inp, out, str(int(aq))] + \
               [("prop:" + ",".join(str(v) for v in t[1:])) if t[0] == "prop" else ",".join(str(v) for v in t) for t in triples]
        env = dict(os.environ)
        env.pop("X265LA_HME", None)
        if hme is not None:
            env["X265LA_HME"] = ",".join(str(int(v)) for v in hme)
        r = subprocess.run(args, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        d = open(out, "rb").read()
    recs, off = [], 0
    while off < len(d):
        n = int(np.frombuffer(d, np.int64, 1, off)[0]); off += 8
        recs.append(np.frombuffer(d, np.int32, n, off).copy()); off += 4 * n
    h = recs[0]
    hdr = dict(W=h[0], H=h[1], N=h[2], stride=h[3], lw=h[4], lh=h[5], wcu=h[6], hcu=h[7], mx=h[8], my=h[9], depth=h[10], qg=h[11], bframes=h[12])
    if hme is not None:
        hdr["wcu4"], hdr["hcu4"] = h[13], h[14]
    i, per_frame, per_triple = 1, [], []
    for _ in frames:
        fr = dict(planes=np.stack(recs[i:i + 4]))
        i += 4
        if hme is not None:
            fr["lowerPlanes"] = np.stack(recs[i:i + 4]); i += 4
        fr.update(intraCost=recs[i], intraMode=recs[i + 1], lowresCosts=recs[i + 2], rowSatds=recs[i + 3], invQ=recs[i + 4])
        per_frame.append(fr)
        i += 5
    for tr in triples:
        t = recs[i]
        d = dict(p0=t[0], b=t[1], p1=t[2], keep=t[3], doSearch=(t[4], t[5]), score=t[6], costEstNorm=t[7], costEstAq=t[8], intraMbs=t[9],
                 mvs0=recs[i + 1], mvc0=recs[i + 2], mvs1=recs[i + 3], mvc1=recs[i + 4], lowresCosts=recs[i + 5], rowSatds=recs[i + 6])
        i += 7
        if hme is not None:
            d.update(lmvs0=recs[i], lmvc0=recs[i + 1], lmvs1=recs[i + 2], lmvc1=recs[i + 3]); i += 4
        if int(aq) & 2:                          # weightp: flag, then the four weighted planes when weightsAnalyse chose weights
            d["isWeighted"] = int(recs[i][0]); i += 1
            if d["isWeighted"]:
                d["wplanes"] = np.stack(recs[i:i + 4]); i += 4
        if tr[0] == "prop":                      # + header (referenced, seed, fpsFactor bits, weightb) and before / after of propB, prop0, prop1
            ph = recs[i]
            d["prop"] = dict(referenced=int(ph[0]), seed=int(ph[1]), fpsFactor=float(np.array([ph[2], ph[3]], np.int32).view(np.float64)[0]), weightb=int(ph[4]),
                             before=[recs[i + 1], recs[i + 3], recs[i + 5]], after=[recs[i + 2], recs[i + 4], recs[i + 6]])
            i += 7
            dbl = lambda r: np.ascontiguousarray(r, np.int32).view(np.float64)
            fh = dbl(recs[i])                    # Lookahead::cuTreeFinish on picture b: strength, weightedCostDelta, ref0Distance; qpAqOffset in, qpCuTreeOffset out
            d["finish"] = dict(strength=float(fh[0]), weightedCostDelta=float(fh[1]), ref0Distance=int(fh[2]), qpAq=dbl(recs[i + 1]), qpCuTree=dbl(recs[i + 2]))
            i += 3
        per_triple.append(d)
    return hdr, per_frame, per_triple


def oracle_propagate(ora, g, dist_p0, dist_p1, weightb, fps_factor, referenced, intra_cost, lowres_costs, inv_q, mvs0, mvs1, prop_b, prop0, prop1):
    """xo_estimate_cu_propagate on copies of the three propagateCost arrays (uint16); when p1 == b pass prop1 = prop_b (same array, like
    the reference's refCosts[1] = frames[p1]->propagateCost).  Returns the arrays after the step."""
    L = ora.me_lib
    L.xo_estimate_cu_propagate.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int] + [C.c_void_p] * 8
    pb = prop_b.astype(np.uint16).copy(); p0 = pb if prop0 is prop_b else prop0.astype(np.uint16).copy()
    p1 = pb if prop1 is prop_b else (p0 if prop1 is prop0 else prop1.astype(np.uint16).copy())
    lc = lowres_costs.astype(np.uint16)
    m1 = mvs1 if mvs1 is not None else np.zeros(2 * g.ncu, np.int32)
    L.xo_estimate_cu_propagate(g.wcu, g.hcu, dist_p0, dist_p1, weightb, fps_factor, referenced, _P(intra_cost), _P(lc), _P(inv_q),
                               _P(mvs0), _P(m1), _P(pb), _P(p0), _P(p1))
    return pb, p0, p1
