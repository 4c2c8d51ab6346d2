"""Host side of the lookahead frame-cost path (include/x265hip_frame.h, section "lookahead frame costs").

Mirrors the reference objects of that path -- Lowres (common/lowres.cpp:79-250, 349-407), LookaheadTLD::lowresIntraEstimate and
CostEstimateGroup (encoder/slicetype.cpp:755-864, 4230-4640) -- as ONE device-resident batch: source pictures go up once, their
half-resolution planes, intra costs, motion vectors and frame costs are produced and kept in HBM.  torch supplies device memory
and streams only; every computation is a call into libx265hip_<depth>.so.
"""
import ctypes as C

import numpy as np

from .frame import FrameApi, LA_TASK, mvcost_row

CTU = 64
MARGIN_X, MARGIN_Y = CTU + 32, CTU + 16          # PicYuv::create (picyuv.cpp:91-92); Lowres reuses them (lowres.cpp:87,102-103)


class LowresGeometry:
    """picture geometry of PicYuv::create (picyuv.cpp:88-95) and Lowres::create (lowres.cpp:84-103)"""

    def __init__(self, width, height):
        self.W, self.H = width, height
        self.full_stride = (width + CTU - 1) // CTU * CTU + 2 * MARGIN_X
        self.full_rows = (height + CTU - 1) // CTU * CTU + 2 * MARGIN_Y
        self.full_elems = self.full_stride * self.full_rows
        self.full_origin = self.full_stride * MARGIN_Y + MARGIN_X
        self.wcu, self.hcu = (width // 2 + 7) >> 3, (height // 2 + 7) >> 3
        self.ncu = self.wcu * self.hcu
        self.lw, self.lh = self.wcu * 8, self.hcu * 8
        s = width // 2 + 2 * MARGIN_X
        self.stride = s + ((32 - (s & 31)) & 31)
        self.plane_elems = self.stride * (self.lh + 2 * MARGIN_Y)
        self.origin = self.stride * MARGIN_Y + MARGIN_X


class LookaheadBatch:
    """n_frames source pictures on one GPU; estimates are (p0, b, p1) index triples into them."""

    def __init__(self, depth, width, height, n_frames, max_estimates):
        self.api = FrameApi(depth)
        t = self.t = self.api.torch
        g = self.g = LowresGeometry(width, height)
        self.n = n_frames
        self.d_full = t.zeros(n_frames * g.full_elems, dtype=self.api.pixel_t, device="cuda")
        self.d_low = t.zeros(n_frames * 4 * g.plane_elems, dtype=self.api.pixel_t, device="cuda")
        self.d_intra_cost = t.zeros(n_frames * g.ncu, dtype=t.int32, device="cuda")
        self.d_intra_mode = t.zeros(n_frames * g.ncu, dtype=t.uint8, device="cuda")
        self.d_intra_lc = t.zeros(n_frames * g.ncu, dtype=t.int16, device="cuda")
        self.d_intra_rows = t.zeros(n_frames * g.hcu, dtype=t.int32, device="cuda")
        self.d_intra_sums = t.zeros(n_frames * 2, dtype=t.int64, device="cuda")
        self.half = 1 << 14
        row = mvcost_row(depth, self.api.lookahead_qp(), self.half)
        self.d_row = self.api.to_device(row.view(np.int16))
        self.max_est = max_estimates
        self.d_mvs = t.zeros(2 * max_estimates * g.ncu * 2, dtype=t.int16, device="cuda")
        self.d_mv_costs = t.zeros(2 * max_estimates * g.ncu, dtype=t.int32, device="cuda")
        self.d_lc = t.zeros(max_estimates * g.ncu, dtype=t.int16, device="cuda")
        self.d_rows = t.zeros(max_estimates * g.hcu, dtype=t.int32, device="cuda")
        self.d_sums = t.zeros(max_estimates * 3, dtype=t.int64, device="cuda")
        self.d_invq = None

    def upload(self, frames):
        """frames: n_frames arrays height x width (u8 / u16).  Copies them into the padded allocation (one strided copy each)."""
        g, t = self.g, self.t
        view = self.d_full.view(self.n, g.full_rows, g.full_stride)
        for f, fr in enumerate(frames):
            a = np.ascontiguousarray(fr)
            if a.dtype == np.uint16:
                a = a.view(np.int16)
            view[f, MARGIN_Y:MARGIN_Y + g.H, MARGIN_X:MARGIN_X + g.W] = t.from_numpy(a).cuda()

    def build_lowres(self):
        """Lowres::init (lowres.cpp:381-391) for every picture: border replication of the source, frameInitLowres, 4x extendPicBorder"""
        g, api = self.g, self.api
        api.extend_pic_border(self.d_full, g.full_origin, g.full_stride, g.W, g.H, MARGIN_X, MARGIN_Y, n_pictures=self.n, picture_elems=g.full_elems)
        esz = self.d_low.element_size()
        for f in range(self.n):
            src = C.c_void_p(self.d_full.data_ptr() + (f * g.full_elems + g.full_origin) * esz)
            dst = [C.c_void_p(self.d_low.data_ptr() + ((f * 4 + k) * g.plane_elems + g.origin) * esz) for k in range(4)]
            api.h.check(api.lib.x265hip_frame_init_lowres(api.stream(), src, C.c_ssize_t(g.full_stride), dst[0], dst[1], dst[2], dst[3],
                                                          C.c_ssize_t(g.stride), g.lw, g.lh))
        api.extend_pic_border(self.d_low, g.origin, g.stride, g.lw, g.lh, MARGIN_X, MARGIN_Y, n_pictures=4 * self.n, picture_elems=g.plane_elems)

    def intra(self):
        g = self.g
        self.api.lookahead_intra_batch(self.d_low, g.plane_elems, g.stride, g.origin, g.wcu, g.hcu, self.n, self.d_invq, self.d_intra_cost,
                                       self.d_intra_mode, self.d_intra_lc, self.d_intra_rows, self.d_intra_sums)

    def set_estimates(self, triples):
        """independent estimates: every one searches into its own MV slots (2*i, 2*i+1) and writes output slot i"""
        assert len(triples) <= self.max_est
        tk = np.zeros(len(triples), LA_TASK)
        for i, (p0, b, p1) in enumerate(triples):
            tk[i]["p0"], tk[i]["b"], tk[i]["p1"] = p0, b, p1
            tk[i]["doSearch"] = (1, 1 if p1 > b else 0)
            tk[i]["mvSlot"] = (2 * i, 2 * i + 1)
            tk[i]["outSlot"] = i
        self.n_tasks = len(triples)
        self.tasks_host = tk
        self.d_tasks = self.api.to_device(tk)

    def costs(self, rows_per_slice=0):
        """rows_per_slice > 0: the cooperative --lookahead-slices form (every slice its own sweep)"""
        g = self.g
        self.api.lookahead_cost_batch(self.d_low, g.plane_elems, g.stride, g.origin, g.wcu, g.hcu, self.d_tasks, self.n_tasks, self.d_intra_cost,
                                      self.d_invq, self.d_row, self.half, self.d_mvs, self.d_mv_costs, self.d_lc, self.d_rows, self.d_sums,
                                      rows_per_slice=rows_per_slice)

    def frame_scores(self, frame_bias=0):
        """int64 scores as estimateFrameCost returns them: costEst, normalised for B estimates (slicetype.cpp:4454-4459)"""
        s = self.d_sums.cpu().numpy().reshape(-1, 3)[:self.n_tasks, 0]
        isb = self.tasks_host["p1"] > self.tasks_host["b"]
        return np.where(isb, s * 100 // (130 + frame_bias), s)


def pan_clip(width, height, n, depth, seed=1, shift=(5, 3), noise=2.0):
    """n frames of one smooth texture panning by `shift` pixels per frame, plus noise"""
    rng = np.random.default_rng(seed)
    pad = 16 + max(abs(shift[0]), abs(shift[1])) * n
    base = rng.random((height + 2 * pad, width + 2 * pad)).astype(np.float32)
    for cell in (32, 8, 2):
        k = np.ones(cell, np.float32) / cell
        base = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, base) * 0.5 + base * 0.5
        base = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 0, base) * 0.5 + base * 0.5
    base = (base - base.min()) / (base.max() - base.min())
    pm = (1 << depth) - 1
    out = []
    for f in range(n):
        y0, x0 = pad + shift[1] * f, pad + shift[0] * f
        fr = base[y0:y0 + height, x0:x0 + width] * pm + rng.normal(0, noise * (1 << (depth - 8)), (height, width))
        out.append(np.clip(np.rint(fr), 0, pm).astype(np.uint8 if depth == 8 else np.uint16))
    return out


def minigop_estimates(n_frames, bframes=3):
    """the (p0, b, p1) choices a lookahead with `bframes` asks about inside a window of n_frames pictures:
    every P estimate up to distance bframes + 1 and every B estimate whose two references are that close"""
    out = []
    for b in range(1, n_frames):
        for d0 in range(1, bframes + 2):
            if b - d0 >= 0:
                out.append((b - d0, b, b))
        for d0 in range(1, bframes + 1):
            for d1 in range(1, bframes + 2 - d0):
                if b - d0 >= 0 and b + d1 < n_frames:
                    out.append((b - d0, b, b + d1))
    return out
