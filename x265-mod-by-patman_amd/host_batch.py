"""ctypes view of the C++ host of the frame-batched path (include/x265hip_ctx.h: x265hip_ctx_* / x265hip_batch_*): context, resident planes, one step = phase planes ->
ME 64 / 32 / 16 / 8 (every reference; per-PU choice; optionally the rectangular PUs) -> MC / DCT / quant, on sub-batches of whole pictures that run on their own
streams.  No torch in here: this is what a C++ encoder links, driven from Python for bench.py and the tests; every computation is inside libx265hip_<depth>.so."""
import ctypes as C

import numpy as np

from .frame import ME_TASK, ME_RESULT, TU_TASK, INTER_CHOICE

LEVELS = (64, 32, 16, 8)


class BatchDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("width", "height", "frames", "margin", "qp", "merange", "method", "subme", "tuLog2", "recon", "usePlanes", "refs", "rect", "streams", "bandRows", "amp", "refs1")]


class HostBatch:
    def __init__(self, lib, depth, width, height, frames, qp=28, merange=57, method=1, subme=2, tu_log2=5, margin=96, recon=False, use_planes=True, refs=1, rect=False,
                 streams=1, device=0, band_rows=0, amp=False, refs1=0):
        self.lib, self.depth = lib, depth
        self.W, self.H, self.F, self.margin = width, height, frames, margin
        self.qp, self.merange, self.method, self.subme, self.tu_log2, self.recon, self.use_planes = qp, merange, method, subme, tu_log2, recon, use_planes
        self.refs, self.rect, self.streams, self.band_rows, self.amp, self.refs1 = refs, rect, streams, band_rows, amp, refs1
        self.stride = width + 2 * margin
        self.plane = self.stride * (height + 2 * margin)
        self.pixel = np.uint8 if depth == 8 else np.uint16
        lib.x265hip_last_error.restype = C.c_char_p
        lib.x265hip_batch_stage_name.restype = C.c_char_p
        self.desc = BatchDesc(width, height, frames, margin, qp, merange, method, subme, tu_log2, int(recon), int(use_planes), refs, int(rect), streams, band_rows, int(amp), refs1)
        self.ctx, self.batch = C.c_void_p(), C.c_void_p()
        self._ck(lib.x265hip_ctx_create(device, C.byref(self.ctx)), "ctx_create")
        self._ck(lib.x265hip_batch_create(self.ctx, C.byref(self.desc), C.byref(self.batch)), "batch_create")
        self.mv_level = max(1 << tu_log2, 8)
        self.tasks_host = {}
        for lv in LEVELS:
            t = np.zeros(lib.x265hip_batch_task_count(C.byref(self.desc), lv), ME_TASK)
            self._ck(lib.x265hip_batch_build_me_tasks(C.byref(self.desc), lv, C.c_void_p(t.ctypes.data)), "build_me_tasks")
            self.tasks_host[lv] = t
        self.rect_host = {}
        if rect:
            for lv in LEVELS:
                for (w, h) in ((lv, lv // 2), (lv // 2, lv)):
                    t = np.zeros(lib.x265hip_batch_rect_task_count(C.byref(self.desc), w, h), ME_TASK)
                    self._ck(lib.x265hip_batch_build_rect_tasks(C.byref(self.desc), w, h, C.c_void_p(t.ctypes.data)), "build_rect_tasks")
                    self.rect_host[(w, h)] = t
        self.amp_host = {}
        if amp:
            for lv in (64, 32, 16):
                for (w, h) in ((lv, lv // 4), (lv, 3 * lv // 4), (lv // 4, lv), (3 * lv // 4, lv)):
                    t = np.zeros(lib.x265hip_batch_amp_task_count(C.byref(self.desc), w, h), ME_TASK)
                    self._ck(lib.x265hip_batch_build_amp_tasks(C.byref(self.desc), w, h, C.c_void_p(t.ctypes.data)), "build_amp_tasks")
                    self.amp_host[(w, h)] = t
        self.tu_host = np.zeros(lib.x265hip_batch_tu_count(C.byref(self.desc)), TU_TASK)
        self._ck(lib.x265hip_batch_build_tu_tasks(C.byref(self.desc), C.c_void_p(self.tu_host.ctypes.data)), "build_tu_tasks")
        self.stage_names = [lib.x265hip_batch_stage_name(self.batch, i).decode() for i in range(lib.x265hip_batch_stage_count(self.batch))]

    def _ck(self, rc, what):
        if rc < 0:
            raise RuntimeError("x265hip %s: %d %s" % (what, rc, self.lib.x265hip_last_error().decode()))
        return rc

    def close(self):
        if self.batch:
            self.lib.x265hip_batch_destroy.restype = None
            self.lib.x265hip_batch_destroy(self.batch); self.batch = C.c_void_p()
        if self.ctx:
            self.lib.x265hip_ctx_destroy.restype = None
            self.lib.x265hip_ctx_destroy(self.ctx); self.ctx = C.c_void_p()

    @property
    def pixels_per_step(self):
        return self.F * self.W * self.H

    def upload(self, pairs):
        """pairs: F tuples (cur_padded, list-0 references ..., list-1 references ...) of shape (H + 2 margin, W + 2 margin) with replicated borders; the pictures inside the padding
        are handed to x265hip_batch_upload_plane, which pads on the device (extendPicBorder).  The padded host stacks are kept for the CPU baseline."""
        assert len(pairs) == self.F and all(len(p) == 1 + self.refs + self.refs1 for p in pairs)
        m = self.margin
        for f, p in enumerate(pairs):
            for which, a in enumerate(p):
                pic = np.ascontiguousarray(a.reshape(self.H + 2 * m, self.stride)[m:m + self.H, m:m + self.W])
                self._ck(self.lib.x265hip_batch_upload_plane(self.batch, which, f, C.c_void_p(pic.ctypes.data), C.c_ssize_t(self.W)), "upload_plane")
                self.sync()                                  # `pic` must outlive the asynchronous copy
        self.cur_host = np.concatenate([p[0].reshape(-1) for p in pairs])
        self.refs_host = [np.concatenate([p[1 + r].reshape(-1) for p in pairs]) for r in range(self.refs)]
        self.refs1_host = [np.concatenate([p[1 + self.refs + r].reshape(-1) for p in pairs]) for r in range(self.refs1)]
        self.ref_host = self.refs_host[0]

    def device_plane(self, which, frame):
        out = np.zeros(self.plane, self.pixel)
        self._ck(self.lib.x265hip_batch_read_plane(self.batch, which, frame, C.c_void_p(out.ctypes.data)), "read_plane")
        return out

    def step(self):
        self._ck(self.lib.x265hip_batch_step(self.batch), "batch_step")

    def step_one_stream(self):
        """the same pass on the context's stream alone (x265hip_batch_step_one_stream): stages one after the other, for per-stage times"""
        self._ck(self.lib.x265hip_batch_step_one_stream(self.batch), "batch_step_one_stream")

    def sync(self):
        self._ck(self.lib.x265hip_ctx_sync(self.ctx), "ctx_sync")

    def set_fused(self, flags):
        """x265hip_batch_set_mode: X265HIP_BATCH_* flags -- 4 = the 64x64 level with its start-stage launch, 16 = phase planes in groups of two pictures; 0 = the default schedule.
        (1 / 2, the fused lower levels, and 8, the tiled phase planes, were measured losses and left the library in round 5: it refuses them.)"""
        self._ck(self.lib.x265hip_batch_set_mode(self.batch, int(flags)), "batch_set_mode")

    set_mode = set_fused

    def set_timing(self, on):
        self.lib.x265hip_batch_set_timing(self.batch, int(on))

    def read_timing(self):
        """{stage: mean ms over the timed steps since the last call}"""
        ms = (C.c_float * len(self.stage_names))()
        self._ck(self.lib.x265hip_batch_read_timing(self.batch, ms), "read_timing")
        return {n: float(ms[i]) for i, n in enumerate(self.stage_names)}

    def read_kernel_timing(self):
        """mean ms of star64_kernel alone (the longest single kernel of a STAR pass) over the timed steps since the last call, or None when no such launch was timed"""
        ms = C.c_float(0)
        n = self.lib.x265hip_batch_read_kernel_timing(self.batch, C.byref(ms))
        return float(ms.value) if n > 0 else None

    def kernel_names(self):
        return list(self.stage_names)

    def results(self, lv, ref=0):
        return self.shape_results(lv, lv, ref)

    def shape_tasks(self, w, h):
        return self.tasks_host[w] if w == h else self.rect_host[(w, h)] if (w, h) in self.rect_host else self.amp_host[(w, h)]

    def shape_results(self, w, h, ref=0, lst=0):
        out = np.zeros(len(self.shape_tasks(w, h)), ME_RESULT)
        self._ck(self.lib.x265hip_batch_read_results_list(self.batch, w, h, lst, ref, C.c_void_p(out.ctypes.data)), "read_results_list")
        return out

    def rect_results(self, w, h, ref=0):
        return self.shape_results(w, h, ref)

    def choices(self, w, h=None):
        h = w if h is None else h
        out = np.zeros(len(self.shape_tasks(w, h)), INTER_CHOICE)
        self._ck(self.lib.x265hip_batch_read_choices(self.batch, w, h, C.c_void_p(out.ctypes.data)), "read_choices")
        return out

    def coeffs(self):
        n = 1 << self.tu_log2
        co = np.zeros(len(self.tu_host) * n * n, np.int16); ns = np.zeros(len(self.tu_host), np.uint32)
        self._ck(self.lib.x265hip_batch_read_coeffs(self.batch, C.c_void_p(co.ctypes.data), C.c_void_p(ns.ctypes.data)), "read_coeffs")
        return co, ns

    def sub_batch_pictures(self):
        """pictures of sub-batch 0 -- the one the stage events are recorded on"""
        if self.band_rows > 0:
            return self.band_rows * 64 / self.H                     # (a fraction of a picture: the first band)
        S = max(1, min(self.streams, self.F))
        return self.F * 1 // S if S > 1 else self.F

    def algorithmic_bytes(self, whole_batch=False):
        """SURVEY 8(d) compulsory bytes per launch group of ONE sub-batch (what the stage events bracket; whole_batch: of a one-stream pass, step_one_stream): each plane
        byte once + the records the stage writes (16 B per PU and reference, 2 B per coefficient + 4 B per TU); the phase-plane stage reads one padded plane stack and writes
        16, per reference."""
        bpp = 1 if self.depth == 8 else 2
        nf = self.F if whole_batch else self.sub_batch_pictures()
        px = int(nf * self.W * self.H)
        share = nf / self.F
        R = self.refs + self.refs1
        alg = {"me%d" % lv: px * (1 + R) * bpp + int(len(self.tasks_host[lv]) * share) * 16 * R for lv in LEVELS}
        if self.rect:
            for lv in LEVELS:
                alg["rect%d" % lv] = 2 * px * (1 + R) * bpp + int(sum(len(self.rect_host[k]) for k in ((lv, lv // 2), (lv // 2, lv))) * share) * 16 * R
        if self.amp:
            for lv in (64, 32, 16):
                alg["amp%d" % lv] = 4 * px * (1 + R) * bpp + int(sum(len(t) for (w, h), t in self.amp_host.items() if max(w, h) == lv) * share) * 16 * R
        alg["tq"] = px * (2 * bpp + 2) + int(len(self.tu_host) * share) * 4 + (px * bpp if self.recon else 0)
        alg = {k: int(v) for k, v in alg.items()}
        if self.use_planes:
            alg["planes"] = int((self.F if self.band_rows > 0 else nf) * self.plane * bpp * 17 * (self.refs + self.refs1))
        return alg
