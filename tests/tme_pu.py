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


def load_fixture(depth, name="pu"):
    """tests/golden/pu_{8,10}.npz / tmectu_{8,10}.npz (tests/make_golden_tme.py) -> planes, calls as parse_stream returns them (records under a call: ints only)"""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "%s_%d.npz" % (name, depth)))
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


def replay_gen(c, planes, dt):
    """Generator form of the glue: every heavy piece is a request `yield (op, args)` answered by send(result); returns (StopIteration.value) per partition the MEData
    the glue produces: dict(mv, mvp, mvCost, ref, bits, cost).  ops: get_pmv / select_mvp / me / bidir_satd / bits / getcost / mvcost / check_best_mvp / update_mvp."""
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
                amvp, mvc = (yield ("get_pmv", (nb.reshape(-1), l, r, c["curPOC"], c["temporal"], c["refPOC"], int(k5[92]), int(k5[93]))))
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
                    mvp_idx = 0 if amvp[0] == amvp[1] else (yield ("select_mvp", (w, h, fenc, rec_plane, boff, amvp, clip)))
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
                out, satd = (yield ("me", (w, h, fenc, me_plane, boff, bounds, mvp, mvc, c["merange"], c["method"], c["subme"], qp)))
                last_mvp = mvp
                if b_low and low != mvp:
                    take(2)
                    b_low = False
                    out2, satd2 = (yield ("me", (w, h, fenc, me_plane, boff, search_range(clip, low, c["merange"]), low, mvc, c["merange"], c["method"], c["subme"], qp)))
                    last_mvp = low
                    if satd2 < satd:
                        out, satd, b_low = out2, satd2, True
                bits += (yield ("bits", (out, last_mvp)))
                mv_cost = (yield ("mvcost", (qp, out, last_mvp)))
                cost = ((satd - mv_cost) + (yield ("getcost", (c["lam"], bits)))) & 0xFFFFFFFF
                if b_low:
                    bits, cost = (yield ("update_mvp", (c["lam"], mvp, out, low, bits, cost)))
                mvp_idx, bits, cost = (yield ("check_best_mvp", (c["lam"], amvp, out, mvp_idx, bits, cost)))
                mvp = amvp[mvp_idx]
                if cost < best[l]["cost"]:
                    best[l] = dict(mv=out, mvp=mvp, cost=cost, bits=bits, mvCost=mv_cost, ref=r, rec_plane=rec_plane)
        o = dict(mv=[(0, 0), (0, 0)], mvp=[(0, 0), (0, 0)], mvCost=[0, 0], ref=[-1, -1], bits=0, cost=0)
        bidir_cost, bidir_bits, bmv = 0xFFFFFFFF, 0, None
        restricted = c["log2CU"] == 3 and c["part"] != SIZE_2Nx2N
        if not c["isP"] and not restricted and c["part"] != SIZE_2Nx2N and best[0]["cost"] != 0xFFFFFFFF and best[1]["cost"] != 0xFFFFFFFF:
            boff = int(best[0]["rec_plane"][0][3]) + y * int(best[0]["rec_plane"][0][1]) + x
            satd = (yield ("bidir_satd", (w, h, fenc, best[0]["rec_plane"], best[1]["rec_plane"], boff, best[0]["mv"], best[1]["mv"])))
            bidir_bits = best[0]["bits"] + best[1]["bits"] + sel_bits[2] - (sel_bits[0] + sel_bits[1])
            bidir_cost = satd + (yield ("getcost", (c["lam"], bidir_bits)))
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
                satd = (yield ("bidir_satd", (w, h, fenc, best[0]["rec_plane"], best[1]["rec_plane"], boff, (0, 0), (0, 0))))
                b0 = best[0]["bits"] - (yield ("bits", (best[0]["mv"], best[0]["mvp"]))) + (yield ("bits", ((0, 0), best[0]["mvp"])))
                b1 = best[1]["bits"] - (yield ("bits", (best[1]["mv"], best[1]["mvp"]))) + (yield ("bits", ((0, 0), best[1]["mvp"])))
                cz = satd + (yield ("getcost", (c["lam"], b0))) + (yield ("getcost", (c["lam"], b1)))
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


def replay(c, be, planes, dt):
    """the glue with a backend that answers every request at once (the oracle)"""
    g = replay_gen(c, planes, dt)
    try:
        op, args = next(g)
        while True:
            op, args = g.send(getattr(be, op)(*args))
    except StopIteration as st:
        return st.value


def replay_batched(cs, planes, dt, execute):
    """the glue of many calls side by side: the pending requests of all calls are handed to execute(op, [args, ...]) -> [result, ...] one op at a time, so that a
    backend can run each of them as ONE batch (the HIP entry points).  -> list of per-call results"""
    gens = [replay_gen(c, planes, dt) for c in cs]
    pending, results = {}, [None] * len(cs)
    for i, g in enumerate(gens):
        try:
            pending[i] = next(g)
        except StopIteration as st:
            results[i] = st.value
    while pending:
        # the op with the most waiting calls first
        ops = {}
        for i, (op, args) in pending.items():
            ops.setdefault(op, []).append(i)
        op = max(ops, key=lambda k: len(ops[k]))
        idx = ops[op]
        answers = execute(op, [pending[i][1] for i in idx])
        for i, a in zip(idx, answers):
            try:
                pending[i] = gens[i].send(a)
            except StopIteration as st:
                results[i] = st.value; del pending[i]
    return results


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


class HipExecutor:
    """execute(op, [args...]) for replay_batched: every op of the glue through the library -- x265hip_amvp_batch, x265hip_select_mvp_batch, x265hip_me_batch,
    x265hip_bidir_satd_batch, x265hip_mvp_bits_batch on the GPU, one launch per group of like requests; the MVD bit / cost tables and RDCost::getCost through the
    library's own host helpers (x265hip_mvbits_row, x265hip_mvcost_row).  Nothing here touches the oracle."""
    def __init__(self, api, depth, planes):
        import ctypes as C
        from x265hip_pkg import frame as F
        self.api, self.F, self.T, self.depth = api, F, api.torch, depth
        self.planes, self.d_plane, self.d_phase = planes, {}, {}
        self.half = 1 << 15
        self.rows, self.d_rows = {}, {}
        self.bits_row = np.zeros(2 * 32768 + 1, np.float32)
        api.h.check(api.lib.x265hip_mvbits_row(32768, self.bits_row.ctypes.data_as(C.c_void_p)))
        self.d_bits = api.to_device(self.bits_row.view(np.int32))
        self.launches = {}

    def _plane(self, p):
        pid = int(p[0][0])
        if pid not in self.d_plane:
            self.d_plane[pid] = self.api.to_device(p[1])
            self.d_phase[pid] = self.T.zeros(16 * p[1].size, dtype=self.d_plane[pid].dtype, device="cuda")
            self.api.subpel_planes(self.d_plane[pid], int(p[0][1]), int(p[0][2]), self.d_phase[pid], p[1].size)
        return pid

    def _row(self, qp):
        if qp not in self.rows:
            self.rows[qp] = self.F.mvcost_row(self.depth, qp, self.half)
            self.d_rows[qp] = self.api.to_device(self.rows[qp].view(np.int16))
        return self.rows[qp]

    def execute(self, op, reqs):
        self.launches[op] = self.launches.get(op, 0) + 1
        return getattr(self, "x_" + op)(reqs)

    # ---- host table lookups (library helpers) ----
    def x_bits(self, reqs):
        b = self.bits_row
        return [int(np.float32(b[32768 + mv[0] - p[0]]) + np.float32(b[32768 + mv[1] - p[1]]) + np.float32(0.5)) for (mv, p) in reqs]

    def x_mvcost(self, reqs):
        out = []
        for (qp, mv, p) in reqs:
            row = self._row(qp)
            out.append((int(row[self.half + mv[0] - p[0]]) + int(row[self.half + mv[1] - p[1]])) & 0xFFFF)
        return out

    def x_getcost(self, reqs):
        return [((bits * lam + 128) >> 8) & 0xFFFFFFFF for (lam, bits) in reqs]

    # ---- device batches ----
    def x_get_pmv(self, reqs):
        F, T = self.F, self.T
        out = [None] * len(reqs)
        groups = {}
        for i, (nb, l, r, cur, temp, refpoc, cp, crp) in enumerate(reqs):
            groups.setdefault((int(cur), int(temp)) + tuple(int(v) for v in refpoc), []).append(i)
        for key, idx in groups.items():
            n = len(idx)
            t = np.zeros(n, F.AMVP_TASK)
            for k, i in enumerate(idx):
                nb = np.asarray(reqs[i][0]).reshape(6, 9)
                t["nb"]["mv"][k, :, 0, 0] = nb[:, 0]; t["nb"]["mv"][k, :, 0, 1] = nb[:, 1]; t["nb"]["mv"][k, :, 1, 0] = nb[:, 2]; t["nb"]["mv"][k, :, 1, 1] = nb[:, 3]
                t["nb"]["refIdx"][k, :, 0] = nb[:, 4]; t["nb"]["refIdx"][k, :, 1] = nb[:, 5]
                t["list"][k] = reqs[i][1]; t["refIdx"][k] = reqs[i][2]; t["colPOC"][k] = reqs[i][6]; t["colRefPOC"][k] = reqs[i][7]
            d_t = self.api.to_device(t)
            d_o = T.zeros(n * F.AMVP_RESULT.itemsize, dtype=T.uint8, device="cuda")
            self.api.amvp_batch(d_t, n, key[0], key[1], [key[2:18], key[18:34]], d_o)
            o = d_o.cpu().numpy().view(F.AMVP_RESULT)
            for k, i in enumerate(idx):
                nm = int(o["numMvc"][k])
                out[i] = (o["amvp"][k].reshape(-1).astype(np.int32), o["mvc"][k][:nm].reshape(-1).astype(np.int32))
        return out

    def x_select_mvp(self, reqs):
        F, T = self.F, self.T
        out = [None] * len(reqs)
        groups = {}
        for i, (w, h, fenc, plane, boff, amvp, clip) in enumerate(reqs):
            groups.setdefault((self._plane(plane), w, h), []).append(i)
        for (pid, w, h), idx in groups.items():
            n = len(idx)
            t = np.zeros(n, F.SELECT_TASK)
            cur = np.concatenate([reqs[i][2] for i in idx])
            t["curOff"] = np.arange(n) * (w * h)
            for k, i in enumerate(idx):
                t["refOff"][k] = reqs[i][4]; t["amvp"][k] = reqs[i][5]; t["clip"][k] = reqs[i][6]
            d_t, d_cur = self.api.to_device(t), self.api.to_device(cur)
            d_o = T.zeros(n * F.SELECT_RESULT.itemsize, dtype=T.uint8, device="cuda")
            pl = self.planes[pid]
            self.api.select_mvp_batch(w, h, d_cur, w, self.d_phase[pid], pl[1].size, int(pl[0][1]), d_t, n, d_o)
            o = d_o.cpu().numpy().view(F.SELECT_RESULT)
            for k, i in enumerate(idx):
                out[i] = int(o["mvpIdx"][k])
        return out

    def x_me(self, reqs):
        F, T = self.F, self.T
        out = [None] * len(reqs)
        groups = {}
        for i, (w, h, fenc, plane, boff, bounds, mvp, mvc, merange, method, subme, qp) in enumerate(reqs):
            groups.setdefault((self._plane(plane), w, h, merange, method, subme, qp), []).append(i)
        for (pid, w, h, merange, method, subme, qp), idx in groups.items():
            n = len(idx)
            self._row(qp)
            t = np.zeros(n, F.ME_TASK)
            cur = np.concatenate([reqs[i][2] for i in idx])
            t["curOff"] = np.arange(n) * (w * h); t["mvpFrom"] = -1
            for k, i in enumerate(idx):
                _, _, _, _, boff, bounds, mvp, mvc, *_ = reqs[i]
                t["refOff"][k] = boff; t["mvmin"][k] = bounds[0:2]; t["mvmax"][k] = bounds[2:4]; t["qmvp"][k] = mvp
                t["numCand"][k] = len(mvc) // 2; t["mvc"][k][:len(mvc)] = mvc
            d_t, d_cur = self.api.to_device(t), self.api.to_device(cur)
            d_r = T.zeros(n * F.ME_RESULT.itemsize, dtype=T.uint8, device="cuda")
            pl = self.planes[pid]
            self.api.me_batch(w, h, d_cur, w, self.d_plane[pid], int(pl[0][1]), d_t, n, self.d_rows[qp], self.half, merange, method, subme, d_r,
                              planes=self.d_phase[pid], plane_elems=pl[1].size)
            r = d_r.cpu().numpy().view(F.ME_RESULT)
            for k, i in enumerate(idx):
                out[i] = ((int(r["mv"][k][0]), int(r["mv"][k][1])), int(r["cost"][k]))
        return out

    def x_bidir_satd(self, reqs):
        F, T = self.F, self.T
        out = [None] * len(reqs)
        groups = {}
        for i, (w, h, fenc, p0, p1, boff, mv0, mv1) in enumerate(reqs):
            groups.setdefault((self._plane(p0), self._plane(p1), w, h), []).append(i)
        for (a, b, w, h), idx in groups.items():
            n = len(idx)
            t = np.zeros(n, F.BIDIR_TASK)
            cur = np.concatenate([reqs[i][2] for i in idx])
            t["curOff"] = np.arange(n) * (w * h)
            for k, i in enumerate(idx):
                t["refOff"][k] = reqs[i][5]; t["mv0"][k] = reqs[i][6]; t["mv1"][k] = reqs[i][7]
            d_t, d_cur = self.api.to_device(t), self.api.to_device(cur)
            d_o = T.zeros(n, dtype=T.int32, device="cuda")
            pl = self.planes[a]
            self.api.bidir_satd_batch(w, h, d_cur, w, self.d_phase[a], self.d_phase[b], pl[1].size, int(pl[0][1]), d_t, n, d_o)
            o = d_o.cpu().numpy()
            for k, i in enumerate(idx):
                out[i] = int(o[k])
        return out

    def _mvp_bits(self, recs, lams):
        F = self.F
        out = [None] * len(recs)
        groups = {}
        for i, lam in enumerate(lams):
            groups.setdefault(lam, []).append(i)
        for lam, idx in groups.items():
            rec = np.zeros(len(idx), F.MVP_BITS)
            for k, i in enumerate(idx):
                for f, v in recs[i].items():
                    rec[f][k] = v
            d = self.api.to_device(rec)
            self.api.mvp_bits_batch(d, len(idx), self.d_bits, 32768, lam)
            o = d.cpu().numpy().view(F.MVP_BITS)
            for k, i in enumerate(idx):
                out[i] = (int(o["mvpIdx"][k]), int(o["bits"][k]), int(o["cost"][k]))
        return out

    def x_check_best_mvp(self, reqs):
        recs = [dict(amvp=np.array(a, np.int16), mv=mv, mvpIdx=idx, bits=bits, cost=cost) for (lam, a, mv, idx, bits, cost) in reqs]
        return self._mvp_bits(recs, [r[0] for r in reqs])

    def x_update_mvp(self, reqs):
        # updateMVP alone: both AMVP slots = the new base, so the checkBestMVP step of the record changes nothing
        recs = [dict(amvp=np.array([a, a], np.int16), mv=mv, alter=alter, useAlter=1, bits=bits, cost=cost) for (lam, a, mv, alter, bits, cost) in reqs]
        return [(b, c) for (_, b, c) in self._mvp_bits(recs, [r[0] for r in reqs])]
