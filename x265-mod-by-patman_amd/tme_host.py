"""ctypes view of the ThreadedME producer for a C++ encoder (include/x265hip_ctx.h: x265hip_tme_create / _picture): host planes and tables in, one picture's
MEData table out.  What oracle/ref_tme_gpu.cpp binds inside the reference encoder; here for Python callers (bench.py's producer leg, tests)."""
import ctypes as C

import numpy as np

from .frame import INTER_CHOICE, TME_TEMPORAL


class HostRef(C.Structure):
    _fields_ = [("mePlane", C.c_void_p), ("reconPlane", C.c_void_p), ("refTable", C.c_void_p), ("lowresMv", C.c_void_p), ("reconKey", C.c_uint64), ("meKey", C.c_uint64),
                ("reconRowsValid", C.c_int), ("meRowsValid", C.c_int)]


class PictureDesc(C.Structure):
    _fields_ = [("isP", C.c_int), ("numRef", C.c_int * 2), ("curPOC", C.c_int), ("temporalMvp", C.c_int), ("refPOC", (C.c_int * 16) * 2),
                ("searchRange", C.c_int), ("searchMethod", C.c_int), ("subpelRefine", C.c_int),
                ("width", C.c_int), ("height", C.c_int), ("lowresBlocksX", C.c_int),
                ("curPlane", C.c_void_p), ("stride", C.c_ssize_t), ("origin", C.c_int64), ("planeElems", C.c_int64),
                ("refs", (HostRef * 16) * 2),
                ("table", C.c_void_p), ("median", C.c_void_p), ("temporal", C.c_void_p),
                ("nQp", C.c_int), ("qps", C.c_int * 64), ("qpIndex", C.c_void_p), ("areaQpIndex", C.c_void_p),
                ("sourceHeight", C.c_int), ("frameThreads", C.c_int), ("flags", C.c_int), ("areaBestOut", C.c_void_p), ("ctuRowFirst", C.c_int), ("ctuRowCount", C.c_int),
                ("pirStartCol", C.c_int), ("pirSafeX", C.c_int)]


class TmeProducer:
    """One producer per picture geometry and partition set (rect / amp as in the preset)."""

    def __init__(self, lib, width, height, ctu=64, min_cu=8, rect=False, amp=False):
        self.lib = lib
        self.ctx = C.c_void_p()
        rc = lib.x265hip_ctx_create(0, C.byref(self.ctx))
        if rc:
            raise RuntimeError("x265hip_ctx_create: %d" % rc)
        self.tme = C.c_void_p()
        rc = lib.x265hip_tme_create(self.ctx, width, height, ctu, min_cu, int(rect), int(amp), C.byref(self.tme))
        if rc:
            raise RuntimeError("x265hip_tme_create: %d" % rc)
        lib.x265hip_tme_entries.restype = C.c_int
        self.entries = int(lib.x265hip_tme_entries(self.tme, None))
        self.width, self.height, self.ctu = width, height, ctu
        self.n_ctu = ((width + ctu - 1) // ctu) * ((height + ctu - 1) // ctu)
        self._keep = None

    def close(self):
        if self.tme:
            self.lib.x265hip_tme_destroy.restype = None
            self.lib.x265hip_tme_destroy(self.tme)
            self.tme = C.c_void_p()
        if self.ctx:
            self.lib.x265hip_ctx_destroy.restype = None
            self.lib.x265hip_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def pin(self, arr):
        """page-lock a long-lived numpy buffer (x265hip_host_register); undone by unpin"""
        self.lib.x265hip_host_register.argtypes = [C.c_void_p, C.c_size_t]
        rc = self.lib.x265hip_host_register(arr.ctypes.data, arr.nbytes)
        if rc:
            raise RuntimeError("x265hip_host_register: %d" % rc)

    def unpin(self, arr):
        self.lib.x265hip_host_unregister.argtypes = [C.c_void_p]
        self.lib.x265hip_host_unregister(arr.ctypes.data)

    def empty_table(self):
        """the table of a picture before its first record: every slot unavailable (FrameData::reinit)"""
        t = np.zeros(self.n_ctu * 593, dtype=INTER_CHOICE)
        t["ref"] = -1
        return t

    def picture(self, cur, refs, stride, origin, table, qp=28, is_p=True, merange=57, method=1, subme=2, cur_poc=1, ref_pocs=((0,), ()), ref_keys=None, flags=0, frame_threads=1,
                rows=None, rows_valid=None, pir=None):
        """cur: padded plane (numpy, pixel dtype); refs: [[plane, ...] of list 0, [...] of list 1]; table: INTER_CHOICE[n_ctu * 593] in / out.
        No temporal neighbours, no lookahead MVs, one qp: what a first P picture after an intra picture looks like.
        rows = (first CTU row, count): a band of the picture (desc.ctuRowFirst / ctuRowCount); rows_valid = plane rows of every reference that are final now (frame threads)."""
        d = PictureDesc()
        d.isP = int(is_p); d.numRef[0] = len(refs[0]); d.numRef[1] = len(refs[1]) if not is_p else 0
        d.curPOC = cur_poc; d.temporalMvp = 0
        for l in range(2):
            for r, p in enumerate(ref_pocs[l]):
                d.refPOC[l][r] = int(p)
        d.searchRange, d.searchMethod, d.subpelRefine = int(merange), int(method), int(subme)
        d.width, d.height, d.lowresBlocksX = self.width, self.height, (self.width // 2 + 7) // 8
        d.curPlane = cur.ctypes.data; d.stride = int(stride); d.origin = int(origin); d.planeElems = int(cur.size)
        for l in range(2):
            for r, p in enumerate(refs[l]):
                d.refs[l][r].mePlane = p.ctypes.data; d.refs[l][r].reconPlane = p.ctypes.data
                d.refs[l][r].reconKey = int(ref_keys[l][r]) if ref_keys else 0          # 0: uploaded and phase-interpolated with every picture
                d.refs[l][r].reconRowsValid = int(rows_valid) if rows_valid else 0
        if rows:
            d.ctuRowFirst, d.ctuRowCount = int(rows[0]), int(rows[1])
        if pir:                                                 # --intra-refresh: (the picture's pirStartCol, the reference's pirEndCol * ctu - 3)
            d.pirStartCol, d.pirSafeX = int(pir[0]), int(pir[1])
        if self._keep is None:                                  # no temporal neighbour anywhere, one qp: the same arrays for every picture
            temporal = np.zeros(self.n_ctu * self.entries * 2, dtype=TME_TEMPORAL)
            temporal["nb"]["refIdx"] = -1
            self._keep = (temporal, np.zeros(self.n_ctu * self.entries, dtype=np.uint8), np.zeros(self.n_ctu * 5, dtype=np.uint8))
        temporal, qp_index, area_qp = self._keep
        d.table = table.ctypes.data; d.temporal = temporal.ctypes.data; d.nQp = 1; d.qps[0] = int(qp)
        d.qpIndex = qp_index.ctypes.data; d.areaQpIndex = area_qp.ctypes.data
        d.flags = int(flags); d.frameThreads = int(frame_threads); d.sourceHeight = self.height
        rc = self.lib.x265hip_tme_picture(self.tme, C.byref(d))
        if rc:
            self.lib.x265hip_last_error.restype = C.c_char_p
            raise RuntimeError("x265hip_tme_picture: %d %s" % (rc, self.lib.x265hip_last_error().decode()))
        return table
