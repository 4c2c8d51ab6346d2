"""Frame-batched ME -> MC/DCT/quant pipeline on device-resident planes (the bench / smoke workload).

One step over a batch of F (source, reference) frame pairs:
  S1+S2  x265hip_me_batch for the 2Nx2N PU pyramid of every CTU (64, 32, 16, 8 -> 85 PUs / CTU64), each level
         seeded with its parent's MV the way Analysis::deriveMVsForCTU / computeMVForPUs seed PUs from
         m_areaBestMV (analysis.cpp:161-306); search window per Search::setSearchRange (search.cpp:4969-5021)
         with CUData::clipMv limits (cudata.cpp:2094-2107).
  S0     (use_planes) x265hip_subpel_planes: the 15 quarter-pel phase planes of the reference stack, so that
         every sub-pel candidate of S2 is a plain SATD at an integer offset.
  S3     x265hip_tq_batch on the 2^tu_log2 grid with the MVs of the matching pyramid level:
         MC -> residual -> DCT -> quant (inter rounding 85), coefficients + numSig out.
  S4     optional (recon=True): dequant -> IDCT -> reconstruction + SSE.
All launches go to torch's current stream; nothing is synchronised inside step().
"""
import numpy as np

from .frame import FrameApi, ME_TASK, ME_RESULT, TU_TASK, ME_WINDOW, INTER_CHOICE, mvbits_row, rd_lambda

LEVELS = (64, 32, 16, 8)
CTU = 64


def pyramid_tasks(W, H, F, margin, tu_log2):
    """Host-side task lists of a batch of F padded W x H frame pairs (the numpy twin of x265hip_batch_build_me_tasks / _tu_tasks in
    csrc/xh_ctx.cpp): {level: ME_TASK array}, TU_TASK array, the pyramid level whose MVs drive the TUs."""
    stride = W + 2 * margin
    plane = stride * (H + 2 * margin)
    tasks = {}
    for lv in LEVELS:
        nx, ny = W // lv, H // lv
        t = np.zeros(F * nx * ny, ME_TASK)
        f, by, bx = np.meshgrid(np.arange(F), np.arange(ny), np.arange(nx), indexing="ij")
        x, y = (bx * lv).reshape(-1), (by * lv).reshape(-1)
        f = f.reshape(-1)
        off = f * plane + (margin + y) * stride + margin + x
        t["curOff"] = off
        t["refOff"] = off
        # CUData::clipMv limits in quarter-pels (offset 8, maxCUSize 64)
        t["mvmin"][:, 0] = -((CTU + 8 + x - 1) << 2)
        t["mvmin"][:, 1] = -((CTU + 8 + y - 1) << 2)
        t["mvmax"][:, 0] = (W + 8 - x - 1) << 2
        t["mvmax"][:, 1] = (H + 8 - y - 1) << 2
        t["flags"] = ME_WINDOW
        if lv == CTU:
            t["mvpFrom"] = -1
        else:
            pnx, pny = W // (2 * lv), H // (2 * lv)
            t["mvpFrom"] = f * (pnx * pny) + (by.reshape(-1) // 2) * pnx + (bx.reshape(-1) // 2)
        tasks[lv] = t
    n = 1 << tu_log2
    nx, ny = W // n, H // n
    tu = np.zeros(F * nx * ny, TU_TASK)
    f, by, bx = np.meshgrid(np.arange(F), np.arange(ny), np.arange(nx), indexing="ij")
    x, y, f = (bx * n).reshape(-1), (by * n).reshape(-1), f.reshape(-1)
    off = f * plane + (margin + y) * stride + margin + x
    tu["curOff"] = off; tu["refOff"] = off; tu["reconOff"] = off
    mv_level = max(n, 8)                                     # pyramid level whose MVs drive the TUs
    lnx, lny = W // mv_level, H // mv_level
    tu["mvFrom"] = f * (lnx * lny) + (y // mv_level) * lnx + (x // mv_level)
    return tasks, tu, mv_level


RECT_SHAPES = tuple((w, h) for lv in LEVELS for (w, h) in ((lv, lv // 2), (lv // 2, lv)))     # 2NxN and Nx2N of every CU size: 64x32 ... 4x8


def rect_tasks(W, H, F, margin):
    """Task lists of the rectangular partitions (param bEnableRectInter: preset slow and up, param.cpp:572-587): for every CU of the pyramid its two 2NxN and two
    Nx2N PUs (g_puLookup, encoder/threadedme.h:67-92), each seeded with the MV of its own CU's 2Nx2N search (mvpFrom indexes that level's results).
    {(w, h): ME_TASK array}"""
    stride = W + 2 * margin
    plane = stride * (H + 2 * margin)
    out = {}
    for lv in LEVELS:
        for (w, h) in ((lv, lv // 2), (lv // 2, lv)):
            nx, ny = W // w, H // h
            t = np.zeros(F * nx * ny, ME_TASK)
            f, by, bx = np.meshgrid(np.arange(F), np.arange(ny), np.arange(nx), indexing="ij")
            x, y, f = (bx * w).reshape(-1), (by * h).reshape(-1), f.reshape(-1)
            off = f * plane + (margin + y) * stride + margin + x
            t["curOff"] = off; t["refOff"] = off
            cx, cy = (x // lv) * lv, (y // lv) * lv                 # CUData::clipMv works on the CU's position (cudata.cpp:2094-2107)
            t["mvmin"][:, 0] = -((CTU + 8 + cx - 1) << 2); t["mvmin"][:, 1] = -((CTU + 8 + cy - 1) << 2)
            t["mvmax"][:, 0] = (W + 8 - cx - 1) << 2; t["mvmax"][:, 1] = (H + 8 - cy - 1) << 2
            t["flags"] = ME_WINDOW
            t["mvpFrom"] = f * ((W // lv) * (H // lv)) + (y // lv) * (W // lv) + (x // lv)
            out[(w, h)] = t
    return out


class FramePipeline:
    def __init__(self, depth, width, height, frames, qp=28, merange=57, method=1, subme=2, tu_log2=5, margin=96,
                 recon=False, cost_row=None, api=None, use_planes=True, refs=1, rect=False):
        """refs > 1: every source picture is searched in `refs` list-0 reference pictures (param->maxNumReferences); each reference has its own predictor
        chain down the pyramid (m_areaBestMV[area][list][ref], analysis.cpp:248-306), x265hip_inter_merge_batch picks the reference per PU with the
        reference's bit / cost rule, and the TQ stage compensates every TU from the reference its PU chose."""
        assert width % CTU == 0 and height % CTU == 0, "pad the picture to whole CTUs"
        self.api = api or FrameApi(depth)
        self.torch = self.api.torch
        self.depth, self.W, self.H, self.F = depth, width, height, frames
        self.qp, self.merange, self.method, self.subme, self.tu_log2, self.margin = qp, merange, method, subme, tu_log2, margin
        self.recon = recon
        self.use_planes = use_planes
        self.refs = refs
        self.rect = rect                                         # also search the 2NxN / Nx2N PUs of every CU (425 PUs per CTU instead of 85)
        assert not rect or refs == 1, "rect: one reference"
        assert refs == 1 or use_planes, "several references: the phase planes are required"
        self.d_planes = None
        self.stride = width + 2 * margin
        self.plane = self.stride * (height + 2 * margin)        # elements per padded plane
        self.half = 1 << 15
        if cost_row is None:
            raise ValueError("cost_row (uint16, centred, 2*half+1 entries) must be supplied by the caller")
        assert len(cost_row) == 2 * self.half + 1
        self.cost_row_host = cost_row
        self.d_cost = self.api.to_device(cost_row.view(np.int16))
        self.d_cur = self.d_ref = None
        self._build_tasks()

    # ---- host-side task construction (once) ----
    def _origin(self, f, x, y):
        return f * self.plane + (self.margin + y) * self.stride + self.margin + x

    def _build_tasks(self):
        self.tasks_host, tu, self.mv_level = pyramid_tasks(self.W, self.H, self.F, self.margin, self.tu_log2)
        self.tu_host = tu
        self.overlap_tq = False                                  # step(): TQ on a side stream beside the smaller-PU searches (measured: 2 %, 0.644 vs 0.657 ms)
        n = 1 << self.tu_log2
        T = self.torch
        self.d_tasks = {lv: self.api.to_device(t) for lv, t in self.tasks_host.items()}
        self.d_results = {lv: T.zeros(len(t) * ME_RESULT.itemsize, dtype=T.uint8, device="cuda") for lv, t in self.tasks_host.items()}
        if self.refs > 1:
            self.d_results_ref = [self.d_results] + [{lv: T.zeros(len(t) * ME_RESULT.itemsize, dtype=T.uint8, device="cuda") for lv, t in self.tasks_host.items()}
                                                     for _ in range(self.refs - 1)]
            self.d_choice = {lv: T.zeros(len(t) * INTER_CHOICE.itemsize, dtype=T.uint8, device="cuda") for lv, t in self.tasks_host.items()}
            self.bits_half = 1 << 14
            self.bits_row_host = mvbits_row(self.depth, self.bits_half)
            self.d_bits = self.api.to_device(self.bits_row_host.view(np.int32)).view(T.float32)
            self.rd_lambda = rd_lambda(self.depth, self.qp)
        if self.rect:
            self.rect_host = rect_tasks(self.W, self.H, self.F, self.margin)
            self.d_rect_tasks = {k: self.api.to_device(t) for k, t in self.rect_host.items()}
            self.d_rect_results = {k: T.zeros(len(t) * ME_RESULT.itemsize, dtype=T.uint8, device="cuda") for k, t in self.rect_host.items()}
        self.d_tu = self.api.to_device(tu)
        self.d_coeff = T.zeros(len(tu) * n * n, dtype=T.int16, device="cuda")
        self.d_numsig = T.zeros(len(tu), dtype=T.int32, device="cuda")
        self.d_sse = T.zeros(len(tu), dtype=T.int64, device="cuda") if self.recon else None
        self.d_recon = None

    @property
    def pixels_per_step(self):
        return self.F * self.W * self.H

    def upload(self, pairs):
        """pairs: F tuples (cur_padded, ref_padded [, ref1_padded ...]) of shape (H + 2*margin, W + 2*margin); `refs` reference pictures per source picture."""
        assert len(pairs) == self.F and all(len(p) == 1 + self.refs for p in pairs)
        cur = np.concatenate([p[0].reshape(-1) for p in pairs])
        ref = np.concatenate([p[1].reshape(-1) for p in pairs])
        assert cur.size == self.F * self.plane
        self.cur_host, self.ref_host = cur, ref
        self.d_cur, self.d_ref = self.api.to_device(cur), self.api.to_device(ref)
        if self.refs > 1:
            self.refs_host = [ref] + [np.concatenate([p[1 + r].reshape(-1) for p in pairs]) for r in range(1, self.refs)]
            self.d_refs = [self.d_ref] + [self.api.to_device(x) for x in self.refs_host[1:]]
        if self.recon:
            self.d_recon = self.torch.zeros_like(self.d_cur)
        if self.use_planes and self.d_planes is None:
            # 16 phase-plane slots (slot 0 unused) with the reference stack's own addressing
            self.plane_elems = self.F * self.plane
            self.d_planes = self.torch.empty(16 * self.plane_elems, dtype=self.d_ref.dtype, device="cuda")
            if self.refs > 1:
                self.d_planes_ref = [self.d_planes] + [self.torch.empty(16 * self.plane_elems, dtype=self.d_ref.dtype, device="cuda") for _ in range(self.refs - 1)]

    # ---- device work ----
    def launch_planes(self):
        """Quarter-pel phase planes of the whole reference stack (once per reference picture in an encoder)."""
        self.api.subpel_planes(self.d_ref, self.stride, self.F * (self.H + 2 * self.margin), self.d_planes, self.plane_elems)
        for r in range(1, self.refs):
            self.api.subpel_planes(self.d_refs[r], self.stride, self.F * (self.H + 2 * self.margin), self.d_planes_ref[r], self.plane_elems)

    def launch_me(self, lv):
        if self.refs > 1:
            return self._launch_me_refs(lv)
        parent = None if lv == CTU else self.d_results[2 * lv]
        self.api.me_batch(lv, lv, self.d_cur, self.stride, self.d_ref, self.stride, self.d_tasks[lv], len(self.tasks_host[lv]),
                          self.d_cost, self.half, self.merange, self.method, self.subme, self.d_results[lv], mvp_source=parent,
                          planes=self.d_planes if self.use_planes else None, plane_elems=self.plane_elems if self.use_planes else 0)

    def launch_rect(self, lv):
        """the 2NxN and Nx2N PUs of the CUs of size lv, seeded by the 2Nx2N results of the same CUs"""
        for k in ((lv, lv // 2), (lv // 2, lv)):
            self.api.me_batch(k[0], k[1], self.d_cur, self.stride, self.d_ref, self.stride, self.d_rect_tasks[k], len(self.rect_host[k]), self.d_cost, self.half, self.merange,
                              self.method, self.subme, self.d_rect_results[k], mvp_source=self.d_results[lv],
                              planes=self.d_planes if self.use_planes else None, plane_elems=self.plane_elems if self.use_planes else 0)

    def _launch_me_refs(self, lv):
        """the level searched in every reference (each with its own parent chain), then the per-PU choice"""
        n = len(self.tasks_host[lv])
        for r in range(self.refs):
            parent = None if lv == CTU else self.d_results_ref[r][2 * lv]
            self.api.me_batch(lv, lv, self.d_cur, self.stride, self.d_refs[r], self.stride, self.d_tasks[lv], n, self.d_cost, self.half, self.merange, self.method,
                              self.subme, self.d_results_ref[r][lv], mvp_source=parent, planes=self.d_planes_ref[r], plane_elems=self.plane_elems)
        parents = [[None if lv == CTU else self.d_results_ref[r][2 * lv] for r in range(self.refs)], []]
        self.api.inter_merge_batch(lv, lv, self.d_cur, self.stride, self.stride, self.d_tasks[lv], n, [[self.d_results_ref[r][lv] for r in range(self.refs)], []], parents,
                                   [self.d_planes_ref, []], self.plane_elems, self.d_bits, self.bits_half, self.rd_lambda, False, max(self.W, self.H), self.d_choice[lv])

    def launch_tq(self):
        if self.refs > 1:
            for r in range(self.refs):            # one launch per reference plane: every TU is compensated from the reference its PU chose
                self.api.tq_batch(self.tu_log2, self.d_cur, self.stride, self.d_refs[r], self.stride, self.d_tu, len(self.tu_host), self.qp, 85,
                                  self.d_coeff, self.d_numsig, recon=self.d_recon, recon_stride=self.stride, sse=self.d_sse,
                                  planes=self.d_planes_ref[r], plane_elems=self.plane_elems, choice=self.d_choice[self.mv_level], choice_list=0, choice_ref=r)
            return
        self.api.tq_batch(self.tu_log2, self.d_cur, self.stride, self.d_ref, self.stride, self.d_tu, len(self.tu_host), self.qp, 85,
                          self.d_coeff, self.d_numsig, recon=self.d_recon, recon_stride=self.stride, sse=self.d_sse,
                          mv_source=self.d_results[self.mv_level],
                          planes=self.d_planes if self.use_planes else None, plane_elems=self.plane_elems if self.use_planes else 0)

    # ---- sub-batches on their own streams ----
    def _chunks(self, S):
        """frame ranges of S sub-batches (independent pictures: nothing of one chunk reads anything of another)"""
        b = [self.F * i // S for i in range(S + 1)]
        return [(b[i], b[i + 1]) for i in range(S) if b[i + 1] > b[i]]

    def launch_planes_chunk(self, f0, f1):
        esz = self.d_ref.element_size()
        rows = self.H + 2 * self.margin
        for r in range(self.refs):
            src = self.d_ref if r == 0 else self.d_refs[r]
            dst = self.d_planes if r == 0 else self.d_planes_ref[r]
            self.api.subpel_planes(src[f0 * self.plane:], self.stride, (f1 - f0) * rows, dst[f0 * self.plane:], self.plane_elems)

    def launch_me_chunk(self, lv, f0, f1):
        assert self.refs == 1
        per = (self.W // lv) * (self.H // lv)
        a, n = f0 * per, (f1 - f0) * per
        parent = None if lv == CTU else self.d_results[2 * lv]
        self.api.me_batch(lv, lv, self.d_cur, self.stride, self.d_ref, self.stride, self.d_tasks[lv][a * ME_TASK.itemsize:], n,
                          self.d_cost, self.half, self.merange, self.method, self.subme, self.d_results[lv][a * ME_RESULT.itemsize:], mvp_source=parent,
                          planes=self.d_planes if self.use_planes else None, plane_elems=self.plane_elems if self.use_planes else 0)

    def launch_tq_chunk(self, f0, f1):
        n_ = 1 << self.tu_log2
        per = (self.W // n_) * (self.H // n_)
        a, n = f0 * per, (f1 - f0) * per
        self.api.tq_batch(self.tu_log2, self.d_cur, self.stride, self.d_ref, self.stride, self.d_tu[a * TU_TASK.itemsize:], n, self.qp, 85,
                          self.d_coeff[a * n_ * n_:], self.d_numsig[a:], recon=self.d_recon, recon_stride=self.stride, sse=self.d_sse[a:] if self.d_sse is not None else None,
                          mv_source=self.d_results[self.mv_level],
                          planes=self.d_planes if self.use_planes else None, plane_elems=self.plane_elems if self.use_planes else 0)

    def step_split(self, S, ev=None, skew=1):
        """One pass of the hot path with the batch cut into S sub-batches of whole pictures, each on its own stream: the levels of one picture depend on
        each other (a level's predictor is its parent CU's MV), pictures do not, so the LDS-bound 64x64 search of one sub-batch runs beside the latency-bound
        16x16 / 8x8 searches of another.  Issue order is a software pipeline (stream s is `skew` stages behind stream s - 1)."""
        t = self.torch
        main = t.cuda.current_stream()
        chunks = self._chunks(S)
        if getattr(self, "sub_streams", None) is None or len(self.sub_streams) < len(chunks):
            self.sub_streams = [t.cuda.Stream() for _ in chunks]
            self.ev_sub = [t.cuda.Event() for _ in chunks]
            self.ev_start = t.cuda.Event()
        stages = (["planes"] if self.use_planes else []) + ["me%d" % lv for lv in LEVELS] + ["tq"]
        self.ev_start.record(main)
        for st in self.sub_streams[:len(chunks)]:
            st.wait_event(self.ev_start)
        for tick in range(len(stages) + skew * (len(chunks) - 1)):
            for s, (f0, f1) in enumerate(chunks):
                k = tick - skew * s
                if k < 0 or k >= len(stages):
                    continue
                name, st = stages[k], self.sub_streams[s]
                with t.cuda.stream(st):
                    if ev is not None and s == 0:
                        ev[name][0].record(st)
                    if name == "planes":
                        self.launch_planes_chunk(f0, f1)
                    elif name == "tq":
                        self.launch_tq_chunk(f0, f1)
                    else:
                        self.launch_me_chunk(int(name[2:]), f0, f1)
                    if ev is not None and s == 0:
                        ev[name][1].record(st)
        for s in range(len(chunks)):
            self.ev_sub[s].record(self.sub_streams[s])
            main.wait_event(self.ev_sub[s])

    def step(self, ev=None):
        """One pass of the hot path.  The TQ launch depends on the MVs of one ME level only (mv_level), so it runs on a side
        stream next to the remaining (smaller-PU) ME launches -- a memory/MFMA-side kernel beside VALU-side ones -- and is joined
        before the step ends.  ev: optional {name: (start_event, end_event)} recorded around each launch on the stream it runs on."""
        t = self.torch
        if getattr(self, "splits", 1) > 1:
            return self.step_split(self.splits, ev, getattr(self, "skew", 1))
        main = t.cuda.current_stream()
        if getattr(self, "side", None) is None:
            self.side = t.cuda.Stream()
            self.ev_fork, self.ev_join = t.cuda.Event(), t.cuda.Event()

        def run(name, fn, stream):
            if ev is not None:
                ev[name][0].record(stream)
            fn()
            if ev is not None:
                ev[name][1].record(stream)
        if self.use_planes:
            run("planes", self.launch_planes, main)
        forked = False
        for lv in LEVELS:
            run("me%d" % lv, lambda: self.launch_me(lv), main)
            if self.rect:
                run("rect%d" % lv, lambda: self.launch_rect(lv), main)
            if lv == self.mv_level and self.overlap_tq:
                self.ev_fork.record(main)
                self.side.wait_event(self.ev_fork)
                with t.cuda.stream(self.side):
                    run("tq", self.launch_tq, self.side)
                    self.ev_join.record(self.side)
                forked = True
        if forked:
            main.wait_event(self.ev_join)
        else:
            run("tq", self.launch_tq, main)

    # ---- what bench.py reports about a step ----
    def kernel_names(self):
        """names of the launches of one step, in order (the keys of step()'s event dictionary)"""
        me = [n for lv in LEVELS for n in (["me%d" % lv, "rect%d" % lv] if self.rect else ["me%d" % lv])]
        return (["planes"] if self.use_planes else []) + me + ["tq"]

    def algorithmic_bytes(self):
        """SURVEY 8(d) compulsory bytes per launch: each plane byte once + the records the launch writes (16 B per PU, 2 B per
        coefficient + 4 B per TU); the phase-plane launch reads one padded plane stack and writes 16."""
        bpp = 1 if self.depth == 8 else 2
        px = self.pixels_per_step
        alg = {"me%d" % lv: px * (1 + self.refs) * bpp + len(self.tasks_host[lv]) * 16 * self.refs for lv in LEVELS}
        if self.rect:
            for lv in LEVELS:
                alg["rect%d" % lv] = 2 * (px * 2 * bpp) + sum(len(self.rect_host[k]) for k in ((lv, lv // 2), (lv // 2, lv))) * 16
        alg["tq"] = px * (2 * bpp + 2) + len(self.tu_host) * 4 + (px * bpp if self.recon else 0)
        if self.use_planes:
            alg["planes"] = self.F * self.plane * bpp * 17 * self.refs
        return alg

    # ---- read-back ----
    def results(self, lv, ref=0):
        return (self.d_results_ref[ref][lv] if self.refs > 1 else self.d_results[lv]).cpu().numpy().view(ME_RESULT)

    def rect_results(self, w, h):
        return self.d_rect_results[(w, h)].cpu().numpy().view(ME_RESULT)

    def choices(self, lv):
        return self.d_choice[lv].cpu().numpy().view(INTER_CHOICE)
