"""Whole Search::puMotionEstimation calls of the PU stage of ThreadedME (search.cpp:226-556), recorded by oracle/ref_tme.cpp (kind 9) with everything under them,
and the glue that makes a PU's MEData out of the pieces: getBlkBits / getTUBits, the neighbour records of the CTU's own table, getPMV, selectMVP, the fallback
predictor out of the reference frame's table, the lookahead's MV as extra candidate and second search, setSearchRange, the bit and cost bookkeeping (including the
reference's habit of counting against the last predictor its MotionEstimate object saw), updateMVP, checkBestMVP, the bidirectional candidate, the final record.
The heavy pieces come from a backend: the oracle (tests/test_tme_golden.py) or the HIP batch entry points (tests/test_tme_gpu.py)."""
import numpy as np

HDR = 60            # fixed header ints (ref_tme.cpp kind 9)
SIZE_2Nx2N, SIZE_2NxN, SIZE_Nx2N = 0, 1, 2
HORIZONTAL = (1, 4, 5)      # SIZE_2NxN, SIZE_2NxnU, SIZE_2NxnD (common.h PartSize)
VERTICAL = (2, 6, 7)        # SIZE_Nx2N, SIZE_nLx2N, SIZE_nRx2N


def parse_stream(path):
    """-> planes {id: (geom ints, px uint16)}, calls [dict]"""
    d = np.fromfile(path, np.uint8).tobytes()
    off, planes, calls, cur = 0, {}, [], None
    while off < len(d):
        kind, n = np.frombuffer(d, np.int32, 2, off); off += 8
        ints = np.frombuffer(d, np.int32, n, off).copy(); off += 4 * n
        px = None
        if kind in (1, 3):
            npx = int(ints[1]) * int(ints[2])
            planes[int(ints[0])] = (ints, np.frombuffer(d, np.uint16, npx, off).copy()); off += 2 * npx
            continue
        if kind == 2:
            npx = int(ints[1]) * int(ints[2]) + 2 * int(ints[27]) * int(ints[28])
        elif kind in (4, 6):
            npx = int(ints[1]) * int(ints[2])
        elif kind == 9:
            numPart = int(ints[58])
            g0 = 106 + 15 * numPart + 48 * numPart
            npx = sum(int(ints[g0 + 4 * p + 2]) * int(ints[g0 + 4 * p + 3]) for p in range(numPart))
        else:
            npx = 0
        if npx:
            px = np.frombuffer(d, np.uint16, npx, off).copy(); off += 2 * npx
        if kind == 9:
            cur = {"ints": ints, "px": px, "subs": [], "left": int(ints[-1])}
            calls.append(cur)
        elif cur is not None and cur["left"] > 0:
            cur["subs"].append((int(kind), ints, px)); cur["left"] -= 1
    return planes, calls


def load_fixture(depth):
    """tests/golden/pu_{8,10}.npz (tests/make_golden_tme.py) -> planes, calls as parse_stream returns them (records under a call: ints only)"""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pu_%d.npz" % depth))
    planes = {}
    for k in d.files:
        if k.startswith("plane") and k.endswith("_geom"):
            pid = int(k[5:-5])
            planes[pid] = (d[k], d["plane%d" % pid])
    ci, cis, cp, cps = d["call_ints"], d["call_ints_start"], d["call_px"], d["call_px_start"]
    sk, si, sis, css = d["sub_kind"], d["sub_ints"], d["sub_ints_start"], d["call_sub_start"]
    calls = []
    for n in range(len(cis) - 1):
        subs = [(int(sk[j]), si[sis[j]:sis[j + 1]], None) for j in range(css[n], css[n + 1])]
        calls.append({"ints": ci[cis[n]:cis[n + 1]], "px": cp[cps[n]:cps[n + 1]], "subs": subs, "left": 0})
    return planes, calls


def blk_bits(part, is_p, part_idx, last_mode):
    """Search::getBlkBits (search.cpp:4893-4944)"""
    if part == SIZE_2Nx2N or part == 3:
        return (1 if is_p else 3, 3, 5)
    if is_p:
        return (3, 0, 0)
    if part in HORIZONTAL:
        t = (((0, 0, 3), (0, 0, 0), (0, 0, 0)), ((5, 7, 7), (7, 5, 7), (6, 6, 6)))
    else:
        t = (((0, 2, 3), (0, 0, 0), (0, 0, 0)), ((5, 7, 7), (5, 5, 7), (6, 6, 6)))
    return t[part_idx][last_mode]


def clip_limits(c):
    """CUData::clipMv's limits of the call's CU (cudata.cpp:2094-2107): xmin, ymin, xmax, ymax in quarter-pels"""
    cx, cy, W, H, mcu = c["cuX"], c["cuY"], c["picW"], c["picH"], c["maxCU"]
    return [-((mcu + 8 + cx - 1) << 2), -((mcu + 8 + cy - 1) << 2), (W + 8 - cx - 1) << 2, (H + 8 - cy - 1) << 2]


def search_range(clip, mvp, merange):
    """Search::setSearchRange (search.cpp:4969-5021; no vertical restriction, no intra refresh, one slice) -> full-pel mvmin.x, .y, mvmax.x, .y"""
    d = merange << 2
    mn = [min(clip[2], max(clip[0], mvp[0] - d)), min(clip[3], max(clip[1], mvp[1] - d))]
    mx = [min(clip[2], max(clip[0], mvp[0] + d)), min(clip[3], max(clip[1], mvp[1] + d))]
    b = [mn[0] >> 2, mn[1] >> 2, mx[0] >> 2, mx[1] >> 2]
    b[3] = max(b[3], b[1])
    return b


def decode(call):
    i = call["ints"]
    c = dict(isP=int(i[1]), numRef=(int(i[2]), int(i[3])), curPOC=int(i[4]), temporal=int(i[5]), refPOC=i[6:38].copy(), part=int(i[38]), log2CU=int(i[39]), cuX=int(i[40]), cuY=int(i[41]),
             puOffset=int(i[42]), area=int(i[43]), finalIdx=int(i[44]), nbIdx=[int(v) for v in i[45:50]], merange=int(i[50]), method=int(i[51]), subme=int(i[52]),
             lam=(int(i[53]) & 0xFFFFFFFF) | ((int(i[54]) & 0xFFFFFFFF) << 32), picW=int(i[55]), picH=int(i[56]), maxCU=int(i[57]), numPart=int(i[58]))
    c["areaBest"] = i[60:76].reshape(2, 4, 2)
    c["nbRec"] = i[76:106].reshape(5, 6)
    P = c["numPart"]
    c["out"] = i[106:106 + 15 * P].reshape(P, 15)
    o = 106 + 15 * P
    c["refRec"] = i[o:o + 48 * P].reshape(P, 2, 4, 6); o += 48 * P
    c["geo"] = i[o:o + 4 * P].reshape(P, 4); o += 4 * P
    c["planeIds"] = i[o:o + 16].reshape(2, 4, 2)
    blocks, s = [], 0
    for p in range(P):
        n = int(c["geo"][p][2]) * int(c["geo"][p][3])
        blocks.append(call["px"][s:s + n]); s += n
    c["blocks"] = blocks
    c["subs"] = call["subs"]
    return c


def replay(c, be, planes, dt):
    """-> per partition the MEData the glue produces: dict(mv, mvp, mvCost, ref, bits, cost).  be: backend with get_pmv / select_mvp / me / bidir_satd / bits / getcost /
    mvcost / check_best_mvp / update_mvp (see OracleBackend in test_tme_golden.py)."""
    subs = list(c["subs"])
    nlist = 1 if c["isP"] else 2
    clip = clip_limits(c)
    last_mode = 0
    outs = []

    def take(kind):
        # the next recorded call of this kind, in call order: used for the inputs no recording of pixels can replace (lookahead MV, temporal neighbour, qp of the cost table)
        for k, (kk, ints, px) in enumerate(subs):
            if kk == kind:
                return subs.pop(k)
        return None
    # bestME lives outside the partition loop in the reference (search.cpp:241-243): the second partition of a CU only replaces what the first one left when it is cheaper
    best = [dict(cost=0xFFFFFFFF, ref=-1), dict(cost=0xFFFFFFFF, ref=-1)]
    for pi in range(c["numPart"]):
        x, y, w, h = (int(v) for v in c["geo"][pi])
        fenc = c["blocks"][pi].astype(dt)
        sel_bits = (0, 0, 0)
        for l in range(nlist):
            for r in range(c["numRef"][l]):
                sel_bits = blk_bits(c["part"], c["isP"], pi, last_mode)
                bits = sel_bits[l] + 1 + (r + (1 if r < c["numRef"][l] - 1 else 0))
                mvp = (int(c["areaBest"][l][r][0]), int(c["areaBest"][l][r][1]))
                nb = np.zeros((6, 9), np.int32)
                for d in range(5):
                    rec = c["nbRec"][d]
                    if c["nbIdx"][d] >= 0:
                        nb[d, 0:4] = rec[0:4]; nb[d, 4] = rec[4]; nb[d, 5] = rec[5]; nb[d, 8] = 1 if (rec[4] >= 0 or rec[5] >= 0) else 0
                    else:
                        nb[d, 4] = nb[d, 5] = -1
                k5 = take(5)[1]                       # temporal neighbour + its POCs: inputs
                nb[5] = k5[38 + 45:38 + 54]
                amvp, mvc = be.get_pmv(nb.reshape(-1), l, r, c["curPOC"], c["temporal"], c["refPOC"], int(k5[92]), int(k5[93]))
                given = k5[38:38 + 45].reshape(5, 9)                 # what the reference handed to getPMV: only the reference indices, and the MVs of used lists, are initialised
                for d in range(5):
                    assert nb[d, 4] == given[d, 4] and nb[d, 5] == given[d, 5], "neighbour %d: reference indices differ from what getPMV was given" % d
                    for ll in range(2):
                        assert nb[d, 4 + ll] < 0 or (nb[d, 2 * ll] == given[d, 2 * ll] and nb[d, 2 * ll + 1] == given[d, 2 * ll + 1]), "neighbour %d: MV differs" % d
                amvp = [(int(amvp[0]), int(amvp[1])), (int(amvp[2]), int(amvp[3]))]
                mvc = [int(v) for v in mvc]
                mvp_idx = 0
                me_plane, rec_plane = planes[int(c["planeIds"][l][r][0])], planes[int(c["planeIds"][l][r][1])]
                boff = int(me_plane[0][3]) + y * int(me_plane[0][1]) + x
                if len(mvc):
                    mvp_idx = 0 if amvp[0] == amvp[1] else be.select_mvp(w, h, fenc, rec_plane, boff, amvp, clip)
                    mvp = amvp[mvp_idx]
                else:
                    amvp = [(0, 0), (0, 0)]
                    rr = c["refRec"][pi][l][r]
                    if rr[4] != -3:
                        if rr[4] >= 0 and rr[5] == -1: mvp = (int(rr[0]), int(rr[1]))
                        elif rr[5] >= 0 and rr[4] == -1: mvp = (int(rr[2]), int(rr[3]))
                        elif rr[4] >= 0 and rr[5] >= 0: mvp = (int(rr[2 * l]), int(rr[2 * l + 1]))
                low = None
                if x + (w >> 1) < c["picW"] and y + (h >> 1) < c["picH"]:
                    k10 = take(10)[1]
                    assert (int(k10[0]), int(k10[1])) == (l, r)
                    low = (int(k10[2]), int(k10[3]))
                b_low = False
                if low is not None and low != (0, 0):
                    mvc = mvc + [low[0], low[1]]; b_low = True
                k2 = take(2)[1]
                qp = int(k2[14])
                bounds = search_range(clip, mvp, c["merange"])
                out, satd = be.me(w, h, fenc, me_plane, boff, bounds, mvp, mvc, c["merange"], c["method"], c["subme"], qp)
                last_mvp = mvp
                if b_low and low != mvp:
                    take(2)
                    b_low = False
                    out2, satd2 = be.me(w, h, fenc, me_plane, boff, search_range(clip, low, c["merange"]), low, mvc, c["merange"], c["method"], c["subme"], qp)
                    last_mvp = low
                    if satd2 < satd:
                        out, satd, b_low = out2, satd2, True
                bits += be.bits(out, last_mvp)
                mv_cost = be.mvcost(qp, out, last_mvp)
                cost = ((satd - mv_cost) + be.getcost(c["lam"], bits)) & 0xFFFFFFFF
                if b_low:
                    bits, cost = be.update_mvp(c["lam"], mvp, out, low, bits, cost)
                mvp_idx, bits, cost = be.check_best_mvp(c["lam"], amvp, out, mvp_idx, bits, cost)
                mvp = amvp[mvp_idx]
                if cost < best[l]["cost"]:
                    best[l] = dict(mv=out, mvp=mvp, cost=cost, bits=bits, mvCost=mv_cost, ref=r, rec_plane=rec_plane)
        o = dict(mv=[(0, 0), (0, 0)], mvp=[(0, 0), (0, 0)], mvCost=[0, 0], ref=[-1, -1], bits=0, cost=0)
        bidir_cost, bidir_bits, bmv = 0xFFFFFFFF, 0, None
        restricted = c["log2CU"] == 3 and c["part"] != SIZE_2Nx2N
        if not c["isP"] and not restricted and c["part"] != SIZE_2Nx2N and best[0]["cost"] != 0xFFFFFFFF and best[1]["cost"] != 0xFFFFFFFF:
            boff = int(best[0]["rec_plane"][0][3]) + y * int(best[0]["rec_plane"][0][1]) + x
            satd = be.bidir_satd(w, h, fenc, best[0]["rec_plane"], best[1]["rec_plane"], boff, best[0]["mv"], best[1]["mv"])
            bidir_bits = best[0]["bits"] + best[1]["bits"] + sel_bits[2] - (sel_bits[0] + sel_bits[1])
            bidir_cost = satd + be.getcost(c["lam"], bidir_bits)
            bmv = [best[0]["mv"], best[1]["mv"]]
            try_zero = best[0]["mv"] != (0, 0) or best[1]["mv"] != (0, 0)
            if try_zero:
                zb = search_range(clip, (0, 0), max(c["picW"], c["picH"]))
                zb[3] += 2
                zb = [v << 2 for v in zb]
                for l in range(2):
                    p = best[l]["mvp"]
                    try_zero = try_zero and zb[0] <= p[0] <= zb[2] and zb[1] <= p[1] <= zb[3]
            if try_zero:
                satd = be.bidir_satd(w, h, fenc, best[0]["rec_plane"], best[1]["rec_plane"], boff, (0, 0), (0, 0))
                b0 = best[0]["bits"] - be.bits(best[0]["mv"], best[0]["mvp"]) + be.bits((0, 0), best[0]["mvp"])
                b1 = best[1]["bits"] - be.bits(best[1]["mv"], best[1]["mvp"]) + be.bits((0, 0), best[1]["mvp"])
                cz = satd + be.getcost(c["lam"], b0) + be.getcost(c["lam"], b1)
                if cz < bidir_cost:
                    bmv = [(0, 0), (0, 0)]; bidir_cost = cz; bidir_bits = b0 + b1 + sel_bits[2] - (sel_bits[0] + sel_bits[1])
        if bidir_cost < best[0]["cost"] and bidir_cost < best[1]["cost"]:
            last_mode = 2
            o.update(mv=bmv, mvp=[best[0]["mvp"], best[1]["mvp"]], mvCost=[best[0]["mvCost"], best[1]["mvCost"]], ref=[best[0]["ref"], best[1]["ref"]], bits=bidir_bits, cost=bidir_cost)
        elif best[0]["cost"] <= best[1]["cost"]:
            last_mode = 0
            o["mv"][0] = best[0]["mv"]; o["mvp"][0] = best[0]["mvp"]; o["mvCost"][0] = best[0]["mvCost"]; o["ref"][0] = best[0]["ref"]; o["bits"] = best[0]["bits"]; o["cost"] = best[0]["cost"]
        else:
            last_mode = 1
            o["mv"][1] = best[1]["mv"]; o["mvp"][1] = best[1]["mvp"]; o["mvCost"][1] = best[1]["mvCost"]; o["ref"][1] = best[1]["ref"]; o["bits"] = best[1]["bits"]; o["cost"] = best[1]["cost"]
        outs.append(o)
    return outs


def expected(c, pi):
    """the MEData the reference left at the partition's slot; fields of an unused list are whatever the slot held before (not compared)"""
    e = c["out"][pi]
    return dict(mv=[(int(e[1]), int(e[2])), (int(e[3]), int(e[4]))], mvp=[(int(e[5]), int(e[6])), (int(e[7]), int(e[8]))], mvCost=[int(e[9]) & 0xFFFFFFFF, int(e[10]) & 0xFFFFFFFF],
                ref=[int(e[11]), int(e[12])], bits=int(e[13]), cost=int(e[14]) & 0xFFFFFFFF)


def same(o, e):
    if o["ref"] != e["ref"] or o["bits"] != e["bits"] or o["cost"] != e["cost"]:
        return False
    for l in range(2):
        if e["ref"][l] >= 0 and (o["mv"][l] != e["mv"][l] or o["mvp"][l] != e["mvp"][l] or o["mvCost"][l] != e["mvCost"][l]):
            return False
    return True


class OracleBackend:
    """the pieces of a PU's motion estimation from the oracle (each pinned to the reference on its own: test_tme_golden.py)"""
    def __init__(self, oracle, depth):
        self.o, self.rows, self.dt = oracle, {}, (np.uint8 if depth == 8 else np.uint16)

    def row(self, qp):
        if qp not in self.rows:
            self.rows[qp] = self.o.mvcost_row(qp, 1 << 15)
        return self.rows[qp]

    def get_pmv(self, nb, l, r, cur, temp, refpoc, cp, crp):
        return self.o.get_pmv(nb, l, r, cur, temp, refpoc, cp, crp)

    def select_mvp(self, w, h, fenc, plane, boff, amvp, clip):
        return self.o.select_mvp(w, h, fenc, plane[1], int(plane[0][1]), boff, [amvp[0][0], amvp[0][1], amvp[1][0], amvp[1][1]], clip)[0]

    def me(self, w, h, fenc, plane, boff, bounds, mvp, mvc, merange, method, subme, qp):
        r = self.o.me(w, h, fenc, w, 0, plane[1], int(plane[0][1]), boff, bounds, mvp, mvc, merange, method, subme, self.row(qp))
        return (r[0], r[1]), r[2]

    def bits(self, mv, mvp):
        return self.o.mv_bitcost(mv, mvp)

    def mvcost(self, qp, mv, mvp):
        row = self.row(qp); half = (len(row) - 1) // 2
        return (int(row[half + mv[0] - mvp[0]]) + int(row[half + mv[1] - mvp[1]])) & 0xFFFF

    def getcost(self, lam, bits):
        return ((bits * lam + 128) >> 8) & 0xFFFFFFFF

    def update_mvp(self, lam, amvp, mv, alter, bits, cost):
        return self.o.update_mvp(lam, amvp, mv, alter, bits, cost)

    def check_best_mvp(self, lam, amvp, mv, idx, bits, cost):
        return self.o.check_best_mvp(lam, [amvp[0][0], amvp[0][1], amvp[1][0], amvp[1][1]], mv, idx, bits, cost)

    def bidir_satd(self, w, h, fenc, p0, p1, boff, mv0, mv1):
        return self.o.bidir_satd(w, h, fenc, p0[1], p1[1], int(p0[0][1]), boff, mv0, mv1)
