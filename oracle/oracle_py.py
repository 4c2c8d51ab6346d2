"""TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/libx265oracle_{8,10}.so.

All buffers are flat numpy arrays (pixel dtype = uint8 for depth 8, uint16 above; int16 for
shorts); offsets/strides are in elements.  Methods that write return a modified COPY of the
destination buffer that was passed in, so callers can check untouched bytes as well.
The same method vocabulary is implemented by tests/backends.py for the reference process and the
HIP library.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_P = C.c_void_p
_IP = C.c_ssize_t


def _ptr(a, off=0):
    return C.c_void_p(a.ctypes.data + off * a.itemsize)


class Oracle:
    def __init__(self, depth):
        self.depth = depth
        self.pixel = np.uint8 if depth == 8 else np.uint16
        path = os.path.join(HERE, "libx265oracle_%d.so" % depth)
        self.lib = C.CDLL(path)
        L = self.lib
        L.xo_sse_pp.restype = C.c_uint64
        L.xo_sse_ss.restype = C.c_uint64
        L.xo_ssd_s.restype = C.c_uint64
        L.xo_quant.restype = C.c_uint32
        L.xo_nquant.restype = C.c_uint32
        L.xo_copy_count.restype = C.c_uint32
        L.xo_dct_matrix.restype = C.c_void_p
        assert L.xo_bit_depth() == depth
        self.me_lib = C.CDLL(os.path.join(HERE, "libx265oracle_me_%d.so" % depth))
        self.me_lib.xo_lambda.restype = C.c_double

    # ---- pixel compare ----
    def sad(self, w, h, A, sa, oa, B, sb, ob):
        return self.lib.xo_sad(w, h, _ptr(A, oa), _IP(sa), _ptr(B, ob), _IP(sb))

    def satd(self, w, h, A, sa, oa, B, sb, ob):
        return self.lib.xo_satd(w, h, _ptr(A, oa), _IP(sa), _ptr(B, ob), _IP(sb))

    def sa8d(self, n, A, sa, oa, B, sb, ob):
        return self.lib.xo_sa8d(n, _ptr(A, oa), _IP(sa), _ptr(B, ob), _IP(sb))

    def psy_cost_pp(self, n, A, sa, oa, B, sb, ob):
        return self.lib.xo_psy_cost_pp(n, _ptr(A, oa), _IP(sa), _ptr(B, ob), _IP(sb))

    def sse_pp(self, n, A, sa, oa, B, sb, ob):
        return self.lib.xo_sse_pp(n, n, _ptr(A, oa), _IP(sa), _ptr(B, ob), _IP(sb))

    def sse_ss(self, n, A, sa, oa, B, sb, ob):
        return self.lib.xo_sse_ss(n, n, _ptr(A, oa), _IP(sa), _ptr(B, ob), _IP(sb))

    def ssd_s(self, n, A, sa, oa):
        return self.lib.xo_ssd_s(n, _ptr(A, oa), _IP(sa))

    def sad_x3(self, w, h, F, of, R, rs, offs):
        res = np.zeros(4, np.int32)
        self.lib.xo_sad_x3(w, h, _ptr(F, of), _ptr(R, offs[0]), _ptr(R, offs[1]), _ptr(R, offs[2]), _IP(rs), _ptr(res))
        return res[:3].copy()

    def sad_x4(self, w, h, F, of, R, rs, offs):
        res = np.zeros(4, np.int32)
        self.lib.xo_sad_x4(w, h, _ptr(F, of), _ptr(R, offs[0]), _ptr(R, offs[1]), _ptr(R, offs[2]), _ptr(R, offs[3]), _IP(rs), _ptr(res))
        return res

    # ---- block ops ----
    def calcresidual(self, n, fenc, pred, resi, stride):
        r = resi.copy(); self.lib.xo_calcresidual(n, _ptr(fenc), _ptr(pred), _ptr(r), _IP(stride)); return r

    def sub_ps(self, n, dst, ds, s0, s1, ss0, ss1):
        d = dst.copy(); self.lib.xo_sub_ps(n, n, _ptr(d), _IP(ds), _ptr(s0), _ptr(s1), _IP(ss0), _IP(ss1)); return d

    def add_ps(self, n, dst, ds, s0, s1, ss0, ss1):
        d = dst.copy(); self.lib.xo_add_ps(n, n, _ptr(d), _IP(ds), _ptr(s0), _ptr(s1), _IP(ss0), _IP(ss1)); return d

    def copy_pp(self, w, h, dst, ds, src, ss):
        d = dst.copy(); self.lib.xo_copy_pp(w, h, _ptr(d), _IP(ds), _ptr(src), _IP(ss)); return d

    def copy_ss(self, n, dst, ds, src, ss):
        d = dst.copy(); self.lib.xo_copy_ss(n, n, _ptr(d), _IP(ds), _ptr(src), _IP(ss)); return d

    def copy_sp(self, n, dst, ds, src, ss):
        d = dst.copy(); self.lib.xo_copy_sp(n, n, _ptr(d), _IP(ds), _ptr(src), _IP(ss)); return d

    def copy_ps(self, n, dst, ds, src, ss):
        d = dst.copy(); self.lib.xo_copy_ps(n, n, _ptr(d), _IP(ds), _ptr(src), _IP(ss)); return d

    def blockfill_s(self, n, dst, ds, val):
        d = dst.copy(); self.lib.xo_blockfill_s(n, _ptr(d), _IP(ds), C.c_int16(val)); return d

    def cpy2Dto1D_shl(self, n, dst, src, ss, shift):
        d = dst.copy(); self.lib.xo_cpy2Dto1D_shl(n, _ptr(d), _ptr(src), _IP(ss), shift); return d

    def cpy2Dto1D_shr(self, n, dst, src, ss, shift):
        d = dst.copy(); self.lib.xo_cpy2Dto1D_shr(n, _ptr(d), _ptr(src), _IP(ss), shift); return d

    def cpy1Dto2D_shl(self, n, dst, src, ds, shift):
        d = dst.copy(); self.lib.xo_cpy1Dto2D_shl(n, _ptr(d), _ptr(src), _IP(ds), shift); return d

    def cpy1Dto2D_shr(self, n, dst, src, ds, shift):
        d = dst.copy(); self.lib.xo_cpy1Dto2D_shr(n, _ptr(d), _ptr(src), _IP(ds), shift); return d

    def transpose(self, n, dst, src, ss):
        d = dst.copy(); self.lib.xo_transpose(n, _ptr(d), _ptr(src), _IP(ss)); return d

    def addAvg(self, w, h, s0, s1, dst, ss0, ss1, ds):
        d = dst.copy(); self.lib.xo_addAvg(w, h, _ptr(s0), _ptr(s1), _ptr(d), _IP(ss0), _IP(ss1), _IP(ds)); return d

    def pixelavg_pp(self, w, h, dst, ds, s0, ss0, s1, ss1):
        d = dst.copy(); self.lib.xo_pixelavg_pp(w, h, _ptr(d), _IP(ds), _ptr(s0), _IP(ss0), _ptr(s1), _IP(ss1)); return d

    def weight_sp(self, src, dst, ss, ds, w, h, w0, rnd, shift, offset):
        d = dst.copy(); self.lib.xo_weight_sp(_ptr(src), _ptr(d), _IP(ss), _IP(ds), w, h, w0, rnd, shift, offset); return d

    def weight_pp(self, src, dst, stride, w, h, w0, rnd, shift, offset):
        d = dst.copy(); self.lib.xo_weight_pp(_ptr(src), _ptr(d), _IP(stride), w, h, w0, rnd, shift, offset); return d

    def scale1D_128to64(self, dst, src):
        d = dst.copy(); self.lib.xo_scale1D_128to64(_ptr(d), _ptr(src)); return d

    def scale2D_64to32(self, dst, src, stride):
        d = dst.copy(); self.lib.xo_scale2D_64to32(_ptr(d), _ptr(src), _IP(stride)); return d

    # ---- transforms ----
    def dct(self, n, src, stride):
        d = np.zeros(n * n, np.int16); self.lib.xo_dct(n, _ptr(src), _ptr(d), _IP(stride)); return d

    def intra_costs(self, size, src, stride, off, nb_ref, nb_filt):
        c = np.zeros(35, np.int32)
        self.lib.xo_intra_costs(size, _ptr(src, off), _IP(stride), _ptr(nb_ref), _ptr(nb_filt), _ptr(c))
        return c

    def frame_init_lowres(self, src, ss, d0, dh, dv, dc, ds, width, height):
        o = [d.copy() for d in (d0, dh, dv, dc)]
        self.lib.xo_frame_init_lowres(_ptr(src), _ptr(o[0]), _ptr(o[1]), _ptr(o[2]), _ptr(o[3]), _IP(ss), _IP(ds), width, height)
        return tuple(o)

    def extend_pic_border(self, plane, stride, width, height, mx, my):
        d = plane.copy(); self.lib.xo_extend_pic_border(_ptr(d, my * stride + mx), _IP(stride), width, height, mx, my); return d

    def extend_row_border(self, rows, stride, width, height, mx):
        d = rows.copy(); self.lib.xo_extend_row_border(_ptr(d, mx), _IP(stride), width, height, mx); return d

    def lowpass_dct(self, n, src, stride):
        d = np.zeros(n * n, np.int16); self.lib.xo_lowpass_dct(n, _ptr(src), _ptr(d), _IP(stride)); return d

    ADS_X1 = {(4, 4), (8, 8), (16, 12), (12, 16), (16, 4), (4, 16)}
    ADS_X2 = {(8, 4), (4, 8), (16, 8), (8, 16), (32, 16), (16, 32), (64, 32), (32, 64)}

    @classmethod
    def ads_parts(cls, w, h):
        """which ads variant the reference assigns to a PU (pixel.cpp:1122-1146)"""
        return 1 if (w, h) in cls.ADS_X1 else 2 if (w, h) in cls.ADS_X2 else 4

    def ads(self, w, h, enc, sums, delta, cost, width, thresh):
        mvs = np.zeros(max(width, 1), np.int16)
        n = self.lib.xo_ads(self.ads_parts(w, h), w, _ptr(enc), _ptr(sums), delta, _ptr(cost), _ptr(mvs), width, thresh)
        return int(n), mvs[:n].copy()

    def dst4(self, src, stride):
        d = np.zeros(16, np.int16); self.lib.xo_dst4(_ptr(src), _ptr(d), _IP(stride)); return d

    def idct(self, n, src, dst, stride):
        d = dst.copy(); self.lib.xo_idct(n, _ptr(src), _ptr(d), _IP(stride)); return d

    def idst4(self, src, dst, stride):
        d = dst.copy(); self.lib.xo_idst4(_ptr(src), _ptr(d), _IP(stride)); return d

    def quant(self, coef, qc, qbits, add, num):
        du = np.zeros(num, np.int32); q = np.zeros(num, np.int16)
        ns = self.lib.xo_quant(_ptr(coef), _ptr(qc), _ptr(du), _ptr(q), qbits, add, num)
        return ns, q, du

    def nquant(self, coef, qc, qbits, add, num):
        q = np.zeros(num, np.int16)
        ns = self.lib.xo_nquant(_ptr(coef), _ptr(qc), _ptr(q), qbits, add, num)
        return ns, q

    def dequant_normal(self, q, num, scale, shift):
        c = np.zeros(num, np.int16); self.lib.xo_dequant_normal(_ptr(q), _ptr(c), num, scale, shift); return c

    def dequant_scaling(self, q, deq, num, per, shift):
        c = np.zeros(num, np.int16); self.lib.xo_dequant_scaling(_ptr(q), _ptr(deq), _ptr(c), num, per, shift); return c

    def count_nonzero(self, n, q):
        return self.lib.xo_count_nonzero(n, _ptr(q))

    def copy_cnt(self, n, resi, rs):
        c = np.zeros(n * n, np.int16); ns = self.lib.xo_copy_count(n, _ptr(c), _ptr(resi), _IP(rs)); return ns, c

    def denoise_dct(self, coef, ressum, offset, num):
        c = coef.copy(); r = ressum.copy(); self.lib.xo_denoise_dct(_ptr(c), _ptr(r), _ptr(offset), num); return c, r

    def dct_matrix(self, n):
        p = self.lib.xo_dct_matrix(n)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int16)), (n * n,)).copy()

    # ---- interpolation (taps 8 luma / 4 chroma) ----
    def interp(self, kind, taps, w, h, src, ss, so, dst, ds, idx, idx2=0):
        d = dst.copy()
        L = self.lib
        if kind == "hpp": L.xo_interp_hpp(taps, w, h, _ptr(src, so), _IP(ss), _ptr(d), _IP(ds), idx)
        elif kind == "hps": L.xo_interp_hps(taps, w, h, _ptr(src, so), _IP(ss), _ptr(d), _IP(ds), idx, idx2)
        elif kind == "vpp": L.xo_interp_vpp(taps, w, h, _ptr(src, so), _IP(ss), _ptr(d), _IP(ds), idx)
        elif kind == "vps": L.xo_interp_vps(taps, w, h, _ptr(src, so), _IP(ss), _ptr(d), _IP(ds), idx)
        elif kind == "vsp": L.xo_interp_vsp(taps, w, h, _ptr(src, so), _IP(ss), _ptr(d), _IP(ds), idx)
        elif kind == "vss": L.xo_interp_vss(taps, w, h, _ptr(src, so), _IP(ss), _ptr(d), _IP(ds), idx)
        elif kind == "hvpp": L.xo_interp_hvpp(taps, w, h, _ptr(src, so), _IP(ss), _ptr(d), _IP(ds), idx, idx2)
        elif kind == "p2s": L.xo_p2s(w, h, _ptr(src, so), _IP(ss), _ptr(d), _IP(ds))
        else: raise ValueError(kind)
        return d

    # ---- intra ----
    def intra_filter(self, n, samples, filt):
        f = filt.copy(); self.lib.xo_intra_filter(n, _ptr(samples), _ptr(f)); return f

    def intra_pred(self, n, src, dst, ds, mode, bfilter):
        d = dst.copy(); self.lib.xo_intra_pred(n, _ptr(d), _IP(ds), _ptr(src), mode, bfilter); return d

    def intra_allangs(self, n, ref, filt, bluma):
        d = np.zeros(33 * n * n, self.pixel); self.lib.xo_intra_allangs(n, _ptr(d), _ptr(ref), _ptr(filt), bluma); return d

    # ---- motion search driver (x265_oracle_me.c) ----
    def mvcost_row(self, qp, half):
        out = np.zeros(2 * half + 1, np.uint16)
        self.me_lib.xo_mvcost_row(qp, half, _ptr(out))
        return out

    def lambda_tab(self):
        return np.array([self.me_lib.xo_lambda(q) for q in range(70)])

    def sea_integral_planes(self, ref, stride, origin, max_height, pad_x, pad_y):
        """the 12 SEA integral planes of a padded picture (framefilter.cpp:740-833): uint32 array [12, len(ref)], same layout as ref"""
        planes = np.full((12, ref.size), 0xdeadbeef, np.uint32)
        ptrs = (C.c_void_p * 12)(*[planes[k].ctypes.data + origin * 4 for k in range(12)])
        self.me_lib.xo_sea_integral_planes(_ptr(ref, origin), _IP(stride), max_height, pad_x, pad_y, ptrs)
        return planes

    def me(self, w, h, cur, cstride, coff, ref, rstride, roff, bounds, qmvp, mvc, merange, method, subme, costrow, integral=None):
        """costrow: centred uint16 row (as returned by mvcost_row); integral: [12, len(ref)] planes of sea_integral_planes (method 4);
        returns (mvx, mvy, cost)."""
        b = np.asarray(bounds, np.int32); c = np.asarray(mvc, np.int32).reshape(-1)
        out = np.zeros(2, np.int32)
        half = (len(costrow) - 1) // 2
        ip = (C.c_void_p * 12)(*[integral[k].ctypes.data + roff * 4 for k in range(12)]) if integral is not None else None
        cost = self.me_lib.xo_motion_estimate_sea(_ptr(cur, coff), _IP(cstride), w, h, _ptr(ref, roff), _IP(rstride), _ptr(b),
                                                  int(qmvp[0]), int(qmvp[1]), len(c) // 2, _ptr(c) if len(c) else None,
                                                  merange, method, subme, _ptr(costrow, half), _ptr(out), ip)
        return int(out[0]), int(out[1]), int(cost)

    def get_pmv(self, nb, lst, ref_idx, cur_poc, temporal, ref_poc, col_poc, col_ref_poc):
        """CUData::getPMV (xo_get_pmv): nb = 54 ints (6 neighbours x 9, the recorder's layout), ref_poc = 32 ints; returns (amvp[4], mvc pairs)"""
        nb = np.ascontiguousarray(nb, np.int32); rp = np.ascontiguousarray(ref_poc, np.int32)
        amvp = np.zeros(4, np.int32); mvc = np.zeros(32, np.int32)
        n = self.me_lib.xo_get_pmv(_ptr(nb), int(lst), int(ref_idx), int(cur_poc), int(temporal), _ptr(rp), int(col_poc), int(col_ref_poc), _ptr(amvp), _ptr(mvc))
        return amvp, mvc[:2 * n]

    def select_mvp(self, w, h, fenc, ref, rstride, roff, amvp, clip):
        a = np.ascontiguousarray(amvp, np.int32); c = np.ascontiguousarray(clip, np.int32); costs = np.zeros(2, np.int32)
        idx = self.me_lib.xo_select_mvp(w, h, _ptr(fenc), _IP(w), _ptr(ref, roff), _IP(rstride), _ptr(a), _ptr(c), _ptr(costs))
        return int(idx), costs

    def _bits_centre(self):
        if not hasattr(self, "_bits"):
            self._bits = np.zeros(2 * 32768 + 1, np.float32)
            self.me_lib.xo_mvbits_row(32768, _ptr(self._bits))
        return _ptr(self._bits, 32768)

    def mv_bitcost(self, mv, mvp):
        fn = self.me_lib.xo_mv_bitcost; fn.restype = C.c_uint32
        return int(fn(self._bits_centre(), int(mv[0]), int(mv[1]), int(mvp[0]), int(mvp[1])))

    def bidir_satd(self, w, h, fenc, ref0, ref1, rstride, roff, mv0, mv1):
        return int(self.me_lib.xo_bidir_satd(w, h, _ptr(fenc), _IP(w), _ptr(ref0, roff), _IP(rstride), int(mv0[0]), int(mv0[1]), _ptr(ref1, roff), _IP(rstride), int(mv1[0]), int(mv1[1])))

    def check_best_mvp(self, lam, amvp, mv, mvp_idx, bits, cost):
        a = np.ascontiguousarray(amvp, np.int32); io = np.array([mvp_idx, bits, cost], np.uint32)
        self.me_lib.xo_check_best_mvp(self._bits_centre(), C.c_uint64(lam), _ptr(a), int(mv[0]), int(mv[1]), _ptr(io))
        return int(io[0]), int(io[1]), int(io[2])

    def update_mvp(self, lam, amvp, mv, alter, bits, cost):
        io = np.array([bits, cost], np.uint32)
        self.me_lib.xo_update_mvp(self._bits_centre(), C.c_uint64(lam), int(amvp[0]), int(amvp[1]), int(mv[0]), int(mv[1]), int(alter[0]), int(alter[1]), _ptr(io))
        return int(io[0]), int(io[1])

    def diamond(self, w, h, cur, cstride, coff, ref, rstride, roff, bounds, qmvp, costrow):
        """MotionEstimate::diamondSearch (xo_diamond_search): returns (full-pel mvx, mvy, cost)"""
        b = np.asarray(bounds, np.int32)
        out = np.zeros(2, np.int32)
        half = (len(costrow) - 1) // 2
        cost = self.me_lib.xo_diamond_search(_ptr(cur, coff), _IP(cstride), w, h, _ptr(ref, roff), _IP(rstride), _ptr(b), int(qmvp[0]), int(qmvp[1]), _ptr(costrow, half), _ptr(out))
        return int(out[0]), int(out[1]), int(cost)

    def me_chroma(self, w, h, cur, cstride, coff, ref, rstride, roff, bounds, qmvp, mvc, merange, method, subme, costrow, cur_c, cstride_c, coff_c, ref_c, rstride_c, roff_c):
        """me() with the chroma SATD terms (4:2:0): cur_c / ref_c = (Cb, Cr) arrays, coff_c / roff_c = element offset of the PU's chroma block in both"""
        b = np.asarray(bounds, np.int32); c = np.asarray(mvc, np.int32).reshape(-1)
        out = np.zeros(2, np.int32)
        half = (len(costrow) - 1) // 2
        cost = self.me_lib.xo_motion_estimate_chroma(_ptr(cur, coff), _IP(cstride), w, h, _ptr(ref, roff), _IP(rstride), _ptr(b), int(qmvp[0]), int(qmvp[1]), len(c) // 2,
                                                     _ptr(c) if len(c) else None, merange, method, subme, _ptr(costrow, half), _ptr(out),
                                                     _ptr(cur_c[0], coff_c), _ptr(cur_c[1], coff_c), _IP(cstride_c), _ptr(ref_c[0], roff_c), _ptr(ref_c[1], roff_c), _IP(rstride_c))
        return int(out[0]), int(out[1]), int(cost)

    # ---- choice among references + bidirectional candidate (x265_oracle_me.c xo_inter_merge) ----
    def mvbits_row(self, half):
        out = np.zeros(2 * half + 1, np.float32)
        self.me_lib.xo_mvbits_row(half, _ptr(out))
        return out

    def rd_lambda(self, qp):
        self.me_lib.xo_rd_lambda.restype = C.c_uint64
        return int(self.me_lib.xo_rd_lambda(qp))

    def inter_merge(self, w, h, num_ref, mv, mvp, cost, mvcost, bits_row, lam, bidir, source_max_dim, clip, cur, cstride, coff, refs, rstride, roff):
        """mv / mvp: int arrays [8][2] (list * 4 + ref), cost / mvcost [8]; refs: 8 reference planes (or None); returns (out[12], mvCost[2])"""
        nr = np.asarray(num_ref, np.int32); m = np.asarray(mv, np.int32).reshape(-1); p = np.asarray(mvp, np.int32).reshape(-1)
        c = np.asarray(cost, np.int32); mc = np.asarray(mvcost, np.int32); cl = np.asarray(clip, np.int32)
        half = (len(bits_row) - 1) // 2
        ptrs = (C.c_void_p * 8)(*[(r.ctypes.data + roff * r.itemsize) if r is not None else None for r in refs])
        out = np.zeros(12, np.int32); mco = np.zeros(2, np.uint32)
        self.me_lib.xo_inter_merge(w, h, _ptr(nr), _ptr(m), _ptr(p), _ptr(c), _ptr(mc), _ptr(bits_row, half), C.c_uint64(lam), int(bidir), source_max_dim, _ptr(cl),
                                   _ptr(cur, coff), _IP(cstride), ptrs, _IP(rstride), _ptr(out), _ptr(mco))
        return out, mco

    # ---- inter TU pipeline (x265_oracle_me.c xo_tq_tu) ----
    def tq_tu_dst4(self, cur, cstride, coff, pred, pstride, poff, qp, add, want_recon=False):
        coeff = np.zeros(16, np.int16); du = np.zeros(16, np.int32)
        recon = np.zeros(16, self.pixel) if want_recon else None
        sse = C.c_uint64(0)
        fn = self.me_lib.xo_tq_tu_dst4
        fn.restype = C.c_uint32
        ns = fn(_ptr(cur, coff), _IP(cstride), _ptr(pred, poff), _IP(pstride), qp, add, _ptr(coeff), _ptr(du), _ptr(recon) if want_recon else None, _IP(4), C.byref(sse))
        return int(ns), coeff, du, recon, int(sse.value)

    def tq_tu_bi(self, log2n, cur, cstride, coff, ref0, ref1, rstride, roff, mv0, mv1, qp, add, want_recon=False):
        n = 1 << log2n
        coeff = np.zeros(n * n, np.int16); du = np.zeros(n * n, np.int32)
        recon = np.zeros(n * n, self.pixel) if want_recon else None
        sse = C.c_uint64(0)
        fn = self.me_lib.xo_tq_tu_bi
        fn.restype = C.c_uint32
        ns = fn(log2n, _ptr(cur, coff), _IP(cstride), _ptr(ref0, roff), _ptr(ref1, roff), _IP(rstride), int(mv0[0]), int(mv0[1]), int(mv1[0]), int(mv1[1]), qp, add,
                None, _ptr(coeff), _ptr(du), _ptr(recon) if want_recon else None, _IP(n), C.byref(sse))
        return int(ns), coeff, du, recon, int(sse.value)

    def tq_tu(self, log2n, cur, cstride, coff, ref, rstride, roff, mv, qp, add, quant_coeff=None, want_recon=False, chroma=False):
        n = 1 << log2n
        coeff = np.zeros(n * n, np.int16); du = np.zeros(n * n, np.int32)
        recon = np.zeros(n * n, self.pixel) if want_recon else None
        sse = C.c_uint64(0)
        fn = self.me_lib.xo_tq_tu_chroma if chroma else self.me_lib.xo_tq_tu
        fn.restype = C.c_uint32
        ns = fn(log2n, _ptr(cur, coff), _IP(cstride), _ptr(ref, roff), _IP(rstride), int(mv[0]), int(mv[1]), qp, add,
                                  _ptr(quant_coeff) if quant_coeff is not None else None, _ptr(coeff), _ptr(du),
                                  _ptr(recon) if want_recon else None, _IP(n), C.byref(sse))
        return int(ns), coeff, du, recon, int(sse.value)
