"""Python-side plumbing for the frame-level C ABI (include/x265hip_frame.h).

torch supplies device memory and streams only; every computation is a call into libx265hip_<depth>.so.
"""
import ctypes as C

import numpy as np

from .binding import HipLib

ME_TASK = np.dtype([("curOff", "<i4"), ("refOff", "<i4"), ("mvmin", "<i2", 2), ("mvmax", "<i2", 2), ("qmvp", "<i2", 2),
                    ("mvc", "<i2", 24), ("numCand", "<i2"), ("flags", "<i2"), ("mvpFrom", "<i4")])
ME_WINDOW = 1
ME_RESULT = np.dtype([("mv", "<i2", 2), ("cost", "<i4"), ("mvcost", "<i4"), ("reserved", "<i4")])
TU_TASK = np.dtype([("curOff", "<i4"), ("refOff", "<i4"), ("mv", "<i2", 2), ("reconOff", "<i4"), ("mvFrom", "<i4")])
AMVP_NB = np.dtype([("mv", "<i2", (2, 2)), ("refIdx", "i1", 2), ("available", "i1"), ("reserved", "i1")])
AMVP_TASK = np.dtype([("nb", AMVP_NB, 6), ("list", "i1"), ("refIdx", "i1"), ("reserved", "<i2"), ("colPOC", "<i4"), ("colRefPOC", "<i4")])
AMVP_RESULT = np.dtype([("amvp", "<i2", (2, 2)), ("numMvc", "<i2"), ("mvc", "<i2", (11, 2)), ("reserved", "<i2")])
assert AMVP_NB.itemsize == 12 and AMVP_TASK.itemsize == 84 and AMVP_RESULT.itemsize == 56
TME_STEP = np.dtype([("part", "<i2"), ("cuSize", "<i2"), ("cuX", "<i2"), ("cuY", "<i2"), ("puOffset", "<i2"), ("finalIdx", "<i2"), ("neighbor", "<i2", 5), ("numPart", "<i2"),
                     ("pu", "<i2", (2, 4))])
TME_TEMPORAL = np.dtype([("nb", AMVP_NB), ("colPOC", "<i4", 2), ("colRefPOC", "<i4", 2)])
assert TME_STEP.itemsize == 40 and TME_TEMPORAL.itemsize == 28
BIDIR_TASK = np.dtype([("curOff", "<i4"), ("refOff", "<i4"), ("mv0", "<i2", 2), ("mv1", "<i2", 2)])
SELECT_TASK = np.dtype([("curOff", "<i4"), ("refOff", "<i4"), ("amvp", "<i2", (2, 2)), ("clip", "<i4", 4)])
SELECT_RESULT = np.dtype([("mvpIdx", "<i4"), ("cost", "<i4", 2)])
MVP_BITS = np.dtype([("amvp", "<i2", (2, 2)), ("mv", "<i2", 2), ("alter", "<i2", 2), ("mvpIdx", "<i2"), ("useAlter", "<i2"), ("bits", "<u4"), ("cost", "<u4")])
assert SELECT_TASK.itemsize == 32 and SELECT_RESULT.itemsize == 12 and MVP_BITS.itemsize == 28
INTER_CHOICE = np.dtype([("mv", "<i2", (2, 2)), ("mvp", "<i2", (2, 2)), ("mvCost", "<u4", 2), ("ref", "i1", 2), ("reserved", "<i2"), ("bits", "<i4"), ("cost", "<u4")])
assert INTER_CHOICE.itemsize == 36
LA_TASK = np.dtype([("b", "<i4"), ("p0", "<i4"), ("p1", "<i4"), ("doSearch", "<i4", 2), ("mvSlot", "<i4", 2), ("outSlot", "<i4"), ("weighted0", "<i4")])
assert ME_TASK.itemsize == 76 and ME_RESULT.itemsize == 16 and TU_TASK.itemsize == 20 and LA_TASK.itemsize == 36


MAX_REF = 16                   # X265HIP_MAX_REF (include/x265hip_frame.h)


class TmeRef(C.Structure):                  # x265hip_tme_ref (include/x265hip_frame.h)
    _fields_ = [("mePlane", C.c_void_p), ("mePhase", C.c_void_p), ("reconPhase", C.c_void_p), ("refTable", C.c_void_p), ("lowresMv", C.c_void_p)]

class TmeArgs(C.Structure):                 # x265hip_tme_args
    _fields_ = [("isP", C.c_int), ("numRef", C.c_int * 2), ("curPOC", C.c_int), ("temporalMvp", C.c_int), ("refPOC", (C.c_int * 16) * 2),
                ("searchRange", C.c_int), ("searchMethod", C.c_int), ("subpelRefine", C.c_int),
                ("picWidth", C.c_int), ("picHeight", C.c_int), ("ctuSize", C.c_int), ("lowresBlocksX", C.c_int),
                ("curPlane", C.c_void_p), ("stride", C.c_ssize_t), ("origin", C.c_int64), ("planeElems", C.c_int64),
                ("refs", (TmeRef * MAX_REF) * 2), ("table", C.c_void_p), ("areaBest", C.c_void_p), ("temporal", C.c_void_p),
                ("nQp", C.c_int), ("qpIndex", C.c_void_p), ("costRows", C.c_void_p), ("costHalfRange", C.c_int), ("lambdas", C.c_uint64 * 64), ("bitsRow", C.c_void_p), ("bitsHalfRange", C.c_int),
                ("steps", C.c_void_p), ("nSteps", C.c_int), ("workspace", C.c_void_p), ("workspaceBytes", C.c_size_t),
                ("refLagPixels", C.c_int), ("flags", C.c_int), ("frameParallel", C.c_int), ("ctuFirst", C.c_int), ("ctuCount", C.c_int),
                ("pirStartCol", C.c_int), ("pirSafeX", C.c_int)]


class LaHme(C.Structure):                   # x265hip_la_hme (include/x265hip_frame.h)
    _fields_ = [("lowerRes", C.c_void_p), ("planeElems", C.c_int64), ("stride", C.c_ssize_t), ("origin", C.c_int64), ("widthInCU", C.c_int), ("heightInCU", C.c_int),
                ("method", C.c_int * 2), ("range", C.c_int * 2), ("mvs", C.c_void_p), ("mvCosts", C.c_void_p)]


class MeChroma(C.Structure):
    _fields_ = [("curCb", C.c_void_p), ("curCr", C.c_void_p), ("curStrideC", C.c_ssize_t), ("refCb", C.c_void_p), ("refCr", C.c_void_p), ("refStrideC", C.c_ssize_t),
                ("curOffC", C.c_void_p), ("refOffC", C.c_void_p)]


class MergeParams(C.Structure):
    _fields_ = [("numRef", C.c_int * 2), ("results", (C.c_void_p * MAX_REF) * 2), ("mvpSource", (C.c_void_p * MAX_REF) * 2), ("subpelPlanes", (C.c_void_p * MAX_REF) * 2), ("planeElems", C.c_int64),
                ("bitsRow", C.c_void_p), ("bitsHalfRange", C.c_int), ("lambda_", C.c_uint64), ("bidir", C.c_int), ("sourceMaxDim", C.c_int)]


def mvbits_row(depth, half):
    """Host-side MVD bit-size row (x265hip_mvbits_row = BitCost::CalculateLogs); needs no GPU."""
    from .binding import HipLib
    lib = HipLib(depth, fill_table=False)
    out = np.zeros(2 * half + 1, np.float32)
    lib.check(lib.lib.x265hip_mvbits_row(half, C.c_void_p(out.ctypes.data)))
    return out


def rd_lambda(depth, qp):
    from .binding import HipLib
    lib = HipLib(depth, fill_table=False).lib
    lib.x265hip_rd_lambda.restype = C.c_uint64
    return int(lib.x265hip_rd_lambda(qp))


class TqParams(C.Structure):
    _fields_ = [("qp", C.c_int), ("add", C.c_int), ("quantCoeff", C.c_void_p), ("deltaU", C.c_void_p),
                ("subpelPlanes", C.c_void_p), ("planeElems", C.c_int64), ("choice", C.c_void_p), ("choiceList", C.c_int), ("choiceRef", C.c_int), ("chroma", C.c_int), ("refPlane1", C.c_void_p), ("choiceRef1", C.c_int), ("dst4", C.c_int)]


def mvcost_row(depth, qp, half):
    """Host-side cost row from the library (x265hip_mvcost_row); needs no GPU."""
    lib = HipLib(depth, fill_table=False)
    out = np.zeros(2 * half + 1, np.uint16)
    lib.check(lib.lib.x265hip_mvcost_row(qp, half, C.c_void_p(out.ctypes.data)))
    return out


def _dp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class FrameApi:
    def __init__(self, depth):
        import torch
        self.torch = torch
        if not torch.cuda.is_available():
            raise RuntimeError("x265hip: no GPU visible -- the HIP path has no CPU fallback")
        self.depth = depth
        self.h = HipLib(depth, fill_table=False)
        self.lib = self.h.lib
        self.pixel_t = torch.uint8 if depth == 8 else torch.int16   # 16-bit pixels travel as int16 bit patterns

    def stream(self):
        return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

    def to_device(self, arr):
        a = np.ascontiguousarray(arr)
        if a.dtype == np.uint16:
            a = a.view(np.int16)
        elif a.dtype.fields is not None or a.dtype not in (np.uint8, np.int16, np.int32, np.int64, np.uint8):
            a = a.view(np.uint8)
        return self.torch.from_numpy(a.reshape(-1)).cuda()

    def me_batch(self, w, h, cur, cstride, ref, rstride, tasks, n, cost_row, half, merange, method, subme, results, mvp_source=None,
                 planes=None, plane_elems=0):
        self.h.check(self.lib.x265hip_me_batch(self.stream(), w, h, _dp(cur), C.c_ssize_t(cstride), _dp(ref), C.c_ssize_t(rstride),
                                               _dp(tasks), n, _dp(cost_row), half, merange, method, subme, _dp(results), _dp(mvp_source),
                                               _dp(planes), C.c_int64(plane_elems)))

    def me_batch_rows(self, w, h, cur, cstride, ref, rstride, tasks, n, cost_rows, half, merange, method, subme, results, mvp_source=None,
                      planes=None, plane_elems=0):
        """x265hip_me_batch_rows: cost_rows is a table of rows, tasks with X265HIP_ME_ROWS choose theirs (flags bits 8..15)"""
        self.h.check(self.lib.x265hip_me_batch_rows(self.stream(), w, h, _dp(cur), C.c_ssize_t(cstride), _dp(ref), C.c_ssize_t(rstride),
                                                    _dp(tasks), n, _dp(cost_rows), half, merange, method, subme, _dp(results), _dp(mvp_source),
                                                    _dp(planes), C.c_int64(plane_elems)))

    def sea_integral_planes(self, pic_padded, stride, rows):
        """12 SEA integral planes (uint32, laid out like the padded picture) of a picture resident in HBM -> int32 tensor [12 * rows * stride]"""
        t = self.torch
        elems = stride * rows
        planes = t.zeros(12 * elems, dtype=t.int32, device="cuda")
        self.lib.x265hip_sea_integral_workspace.restype = C.c_size_t
        need = int(self.lib.x265hip_sea_integral_workspace(C.c_ssize_t(stride), rows))
        ws = t.empty(need, dtype=t.uint8, device="cuda")
        self.h.check(self.lib.x265hip_sea_integral_planes(self.stream(), _dp(pic_padded), C.c_ssize_t(stride), rows, _dp(planes), C.c_int64(elems), _dp(ws), C.c_size_t(need)))
        return planes, elems

    def me_batch_sea(self, w, h, cur, cstride, ref, rstride, tasks, n, cost_row, half, merange, subme, results, integral, integral_elems, mvp_source=None,
                     planes=None, plane_elems=0):
        self.h.check(self.lib.x265hip_me_batch_sea(self.stream(), w, h, _dp(cur), C.c_ssize_t(cstride), _dp(ref), C.c_ssize_t(rstride),
                                                   _dp(tasks), n, _dp(cost_row), half, merange, subme, _dp(results), _dp(mvp_source),
                                                   _dp(planes), C.c_int64(plane_elems), _dp(integral), C.c_int64(integral_elems)))

    def subpel_planes(self, ref, stride, rows, out_planes, plane_elems):
        self.h.check(self.lib.x265hip_subpel_planes(self.stream(), _dp(ref), C.c_ssize_t(stride), rows, _dp(out_planes), C.c_int64(plane_elems)))

    def extend_pic_border(self, plane, origin_elems, stride, width, height, margin_x, margin_y, n_pictures=1, picture_elems=0):
        """extendPicBorder on pictures resident in HBM; `origin_elems` = element index of pixel (0,0) of picture 0 inside `plane`."""
        org = C.c_void_p(plane.data_ptr() + origin_elems * plane.element_size())
        self.h.check(self.lib.x265hip_extend_pic_border(self.stream(), org, C.c_ssize_t(stride), width, height, margin_x, margin_y,
                                                        n_pictures, C.c_int64(picture_elems)))

    def intra_cost_batch(self, log2_size, src, src_stride, src_off, nb_ref, nb_filt, nb_pitch, n, costs, workspace=None):
        """sa8d of the 35 intra predictions of n CUs (the mode scan of Search::estIntraPredQT); costs: int32 tensor n x 35."""
        self.lib.x265hip_intra_cost_workspace.restype = C.c_size_t
        need = int(self.lib.x265hip_intra_cost_workspace(log2_size, n))
        if need and (workspace is None or workspace.numel() * workspace.element_size() < need):
            workspace = self.torch.empty(need, dtype=self.torch.uint8, device="cuda")
        self.h.check(self.lib.x265hip_intra_cost_batch(self.stream(), log2_size, _dp(src), C.c_ssize_t(src_stride), _dp(src_off), _dp(nb_ref), _dp(nb_filt),
                                                       nb_pitch, n, _dp(costs), _dp(workspace), C.c_size_t(need)))
        return workspace

    def me_batch_chroma(self, w, h, cur, cstride, ref, rstride, tasks, n, cost_row, half, merange, method, subme, results, planes, plane_elems,
                        cur_cb, cur_cr, cstride_c, ref_cb, ref_cr, rstride_c, cur_off_c, ref_off_c, mvp_source=None):
        """x265hip_me_batch_chroma: the search with the chroma SATD terms of subpelCompare (the predInterSearch call form), 4:2:0."""
        ch = MeChroma(_dp(cur_cb), _dp(cur_cr), cstride_c, _dp(ref_cb), _dp(ref_cr), rstride_c, _dp(cur_off_c), _dp(ref_off_c))
        self.h.check(self.lib.x265hip_me_batch_chroma(self.stream(), w, h, _dp(cur), C.c_ssize_t(cstride), _dp(ref), C.c_ssize_t(rstride), _dp(tasks), n,
                                                      _dp(cost_row), half, merange, method, subme, _dp(results), _dp(mvp_source), _dp(planes), C.c_int64(plane_elems), C.byref(ch)))

    def amvp_batch(self, tasks, n, cur_poc, temporal, ref_poc, out):
        """x265hip_amvp_batch: CUData::getPMV for n (PU, list, reference) records; ref_poc = [2][16] ints"""
        class P(C.Structure):
            _fields_ = [("curPOC", C.c_int), ("temporalMvp", C.c_int), ("refPOC", (C.c_int * 16) * 2)]
        p = P(); p.curPOC = int(cur_poc); p.temporalMvp = int(temporal)
        for l in range(2):
            for r in range(16): p.refPOC[l][r] = int(ref_poc[l][r])
        self.h.check(self.lib.x265hip_amvp_batch(self.stream(), _dp(tasks), n, C.byref(p), _dp(out)))

    def tme_schedule(self, ctu=64, min_cu=8, rect=True, amp=False):
        n = self.lib.x265hip_tme_schedule(ctu, min_cu, int(rect), int(amp), None, 0)
        steps = np.zeros(n, TME_STEP)
        assert self.lib.x265hip_tme_schedule(ctu, min_cu, int(rect), int(amp), steps.ctypes.data_as(C.c_void_p), n) == n
        return steps

    def tme_frame(self, *, is_p, num_ref, cur_poc, temporal_mvp, ref_poc, merange, method, subme, lams, qp_index, width, height, ctu, lowres_blocks_x, cur, stride, origin, plane_elems,
                  refs, table, area_best, temporal, cost_rows, cost_half, bits_row, bits_half, steps, flags=0, ref_lag=0, frame_parallel=False):
        """x265hip_tme_frame; refs[l][r] = dict(me_plane, me_phase, recon_phase, ref_table or None, lowres_mv or None) of device tensors; steps: host TME_STEP array"""
        a = TmeArgs()
        a.isP = int(is_p); a.numRef[0], a.numRef[1] = int(num_ref[0]), int(num_ref[1]); a.curPOC = int(cur_poc); a.temporalMvp = int(temporal_mvp)
        for l in range(2):
            for r in range(16): a.refPOC[l][r] = int(ref_poc[l][r])
        a.searchRange, a.searchMethod, a.subpelRefine = int(merange), int(method), int(subme)
        a.nQp = len(lams); a.qpIndex = _dp(qp_index)
        for q in range(len(lams)):
            a.lambdas[q] = int(lams[q])
        a.costRows = _dp(cost_rows)
        a.picWidth, a.picHeight, a.ctuSize, a.lowresBlocksX = int(width), int(height), int(ctu), int(lowres_blocks_x)
        a.curPlane = _dp(cur); a.stride = int(stride); a.origin = int(origin); a.planeElems = int(plane_elems)
        for l in range(2):
            for r in range(MAX_REF):
                d = refs[l][r] if l < len(refs) and r < len(refs[l]) else None
                if d:
                    a.refs[l][r].mePlane = _dp(d["me_plane"]); a.refs[l][r].mePhase = _dp(d["me_phase"]); a.refs[l][r].reconPhase = _dp(d["recon_phase"])
                    a.refs[l][r].refTable = _dp(d.get("ref_table")); a.refs[l][r].lowresMv = _dp(d.get("lowres_mv"))
        a.table = _dp(table); a.areaBest = _dp(area_best); a.temporal = _dp(temporal)
        a.costHalfRange = int(cost_half); a.bitsRow = _dp(bits_row); a.bitsHalfRange = int(bits_half)
        steps = np.ascontiguousarray(steps)
        a.steps = steps.ctypes.data; a.nSteps = len(steps)
        n_ctu = (width // ctu) * (height // ctu)
        self.lib.x265hip_tme_workspace.restype = C.c_size_t
        ws_bytes = self.lib.x265hip_tme_workspace(n_ctu)
        ws = self.torch.zeros(ws_bytes, dtype=self.torch.uint8, device="cuda")
        a.workspace = _dp(ws); a.workspaceBytes = ws_bytes
        a.flags, a.refLagPixels, a.frameParallel = int(flags), int(ref_lag), int(frame_parallel)
        self.h.check(self.lib.x265hip_tme_frame(self.stream(), C.byref(a)))
        self.torch.cuda.synchronize()           # `steps` and the workspace must outlive the launches

    def bidir_satd_batch(self, w, h, cur, cstride, planes0, planes1, plane_elems, rstride, tasks, n, out):
        self.h.check(self.lib.x265hip_bidir_satd_batch(self.stream(), w, h, _dp(cur), C.c_ssize_t(cstride), _dp(planes0), _dp(planes1), C.c_int64(plane_elems), C.c_ssize_t(rstride), _dp(tasks), n, _dp(out)))

    def select_mvp_batch(self, w, h, cur, cstride, planes, plane_elems, rstride, tasks, n, out):
        self.h.check(self.lib.x265hip_select_mvp_batch(self.stream(), w, h, _dp(cur), C.c_ssize_t(cstride), _dp(planes), C.c_int64(plane_elems), C.c_ssize_t(rstride), _dp(tasks), n, _dp(out)))

    def mvp_bits_batch(self, records, n, bits_row, half_range, lam):
        self.h.check(self.lib.x265hip_mvp_bits_batch(self.stream(), _dp(records), n, _dp(bits_row), half_range, C.c_uint64(lam)))

    def diamond_batch(self, w, h, cur, cstride, ref, rstride, tasks, n, cost_row, half_range, results):
        """x265hip_diamond_batch: MotionEstimate::diamondSearch for n PUs (full-pel MV, cost)"""
        self.h.check(self.lib.x265hip_diamond_batch(self.stream(), w, h, _dp(cur), C.c_ssize_t(cstride), _dp(ref), C.c_ssize_t(rstride), _dp(tasks), n,
                                                    _dp(cost_row), half_range, _dp(results)))

    def inter_merge_batch(self, w, h, cur, cstride, rstride, tasks, n, results, mvp_sources, planes, plane_elems, bits_row, bits_half, lam, bidir, source_max_dim, out):
        """x265hip_inter_merge_batch; results / mvp_sources / planes: [list][ref] nested lists of tensors (or None)."""
        p = MergeParams()
        for l in range(2):
            p.numRef[l] = len(results[l])
            for r in range(len(results[l])):
                p.results[l][r] = results[l][r].data_ptr()
                p.mvpSource[l][r] = mvp_sources[l][r].data_ptr() if mvp_sources and mvp_sources[l][r] is not None else None
                p.subpelPlanes[l][r] = planes[l][r].data_ptr() if planes and planes[l][r] is not None else None
        p.planeElems = plane_elems; p.bitsRow = bits_row.data_ptr(); p.bitsHalfRange = bits_half; p.lambda_ = lam; p.bidir = int(bidir); p.sourceMaxDim = source_max_dim
        self.h.check(self.lib.x265hip_inter_merge_batch(self.stream(), w, h, _dp(cur), C.c_ssize_t(cstride), C.c_ssize_t(rstride), _dp(tasks), n, C.byref(p), _dp(out)))

    def lookahead_qp(self):
        return int(self.lib.x265hip_lookahead_qp())

    def lookahead_intra_batch(self, lowres, plane_elems, stride, origin, wcu, hcu, n_frames, inv_qscale, intra_cost, intra_mode, lowres_costs, row_satds, sums):
        """LookaheadTLD::lowresIntraEstimate for n_frames lowres pictures (4 planes each) resident in `lowres`."""
        self.h.check(self.lib.x265hip_lookahead_intra_batch(self.stream(), _dp(lowres), C.c_int64(plane_elems), C.c_ssize_t(stride), C.c_int64(origin), wcu, hcu,
                                                            n_frames, _dp(inv_qscale), _dp(intra_cost), _dp(intra_mode), _dp(lowres_costs), _dp(row_satds), _dp(sums)))

    def lookahead_cost_batch(self, lowres, plane_elems, stride, origin, wcu, hcu, tasks, n_tasks, intra_cost, inv_qscale, cost_row, half,
                             mvs, mv_costs, lowres_costs, row_satds, sums, rows_per_slice=0):
        """CostEstimateGroup::estimateFrameCost for n_tasks (p0, b, p1) choices (LA_TASK records on the device)."""
        self.h.check(self.lib.x265hip_lookahead_cost_batch(self.stream(), _dp(lowres), C.c_int64(plane_elems), C.c_ssize_t(stride), C.c_int64(origin), wcu, hcu,
                                                           _dp(tasks), n_tasks, int(lowres.numel() // (4 * plane_elems)), _dp(intra_cost), _dp(inv_qscale), _dp(cost_row), half, rows_per_slice,
                                                           _dp(mvs), _dp(mv_costs), _dp(lowres_costs), _dp(row_satds), _dp(sums)))

    def lookahead_cost_batch_hme(self, lowres, plane_elems, stride, origin, wcu, hcu, tasks, n_tasks, intra_cost, inv_qscale, cost_row, half,
                                 mvs, mv_costs, lowres_costs, row_satds, sums, lower, plane_elems4, stride4, origin4, wcu4, hcu4, methods, ranges, lower_mvs, lower_mv_costs):
        """the same with --hme (x265hip_la_hme): the quarter-resolution pictures `lower` (same places as `lowres`), the level methods / ranges, the level-0 MV / cost slots"""
        hme = LaHme(lower.data_ptr(), plane_elems4, stride4, origin4, wcu4, hcu4, (C.c_int * 2)(*methods), (C.c_int * 2)(*ranges), lower_mvs.data_ptr(), lower_mv_costs.data_ptr())
        self.h.check(self.lib.x265hip_lookahead_cost_batch_hme(self.stream(), _dp(lowres), C.c_int64(plane_elems), C.c_ssize_t(stride), C.c_int64(origin), wcu, hcu,
                                                               _dp(tasks), n_tasks, int(lowres.numel() // (4 * plane_elems)), _dp(intra_cost), _dp(inv_qscale), _dp(cost_row), half, 0,
                                                               _dp(mvs), _dp(mv_costs), _dp(lowres_costs), _dp(row_satds), _dp(sums), C.byref(hme)))

    def cutree_propagate(self, wcu, hcu, dist_p0, dist_p1, weightb, fps_factor, referenced, intra_cost, lowres_costs, inv_q, mvs0, mvs1, prop_b, prop0, prop1, workspace):
        """one cuTree propagation step (Lookahead::estimateCUPropagate); tensors may be views into the lookahead batch's arrays"""
        self.h.check(self.lib.x265hip_cutree_propagate(self.stream(), wcu, hcu, dist_p0, dist_p1, weightb, C.c_double(fps_factor), referenced,
                                                       _dp(intra_cost), _dp(lowres_costs), _dp(inv_q), _dp(mvs0), _dp(mvs1), _dp(prop_b), _dp(prop0), _dp(prop1),
                                                       _dp(workspace), C.c_size_t(workspace.numel() * workspace.element_size())))

    def frame_init_lowres(self, src, src_stride, d0, dh, dv, dc, dst_stride, width, height):
        self.h.check(self.lib.x265hip_frame_init_lowres(self.stream(), _dp(src), C.c_ssize_t(src_stride), _dp(d0), _dp(dh), _dp(dv), _dp(dc),
                                                        C.c_ssize_t(dst_stride), width, height))

    def tq_batch(self, log2n, cur, cstride, ref, rstride, tasks, n, qp, add, coeff, numsig, quant_coeff=None, delta_u=None,
                 recon=None, recon_stride=0, sse=None, mv_source=None, planes=None, plane_elems=0, choice=None, choice_list=0, choice_ref=0, chroma=False, ref1=None, choice_ref1=0, dst4=False):
        p = TqParams(qp, add, _dp(quant_coeff), _dp(delta_u), _dp(planes), plane_elems if planes is not None else 0, _dp(choice), choice_list, choice_ref, int(chroma),
                     _dp(ref1), choice_ref1, int(dst4))
        self.h.check(self.lib.x265hip_tq_batch(self.stream(), log2n, _dp(cur), C.c_ssize_t(cstride), _dp(ref), C.c_ssize_t(rstride),
                                               _dp(tasks), n, C.byref(p), _dp(coeff), _dp(numsig),
                                               _dp(recon), C.c_ssize_t(recon_stride), _dp(sse), _dp(mv_source)))
