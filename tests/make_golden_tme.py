#!/usr/bin/env python3
"""tests/make_golden_tme.py -- fixtures tests/golden/tme_{8,10}.npz: every MotionEstimate::motionEstimate call of a reference encode run with
--threaded-me (oracle/_ref/x265tme_*, oracle/ref_tme.cpp): source PU, reference plane, window, predictor, candidates -> MV and cost.
Run here (needs /root/reference for `make -C oracle tme`); the .npz files are committed and replayed on the CPU (oracle) and the GPU box."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NI = 29          # ints of a call record in front of the candidates (ref_tme.cpp)
CALL_FIELDS = ("plane w h blockOffset mnx mny mxx mxy qmvpx qmvpy numCand merange method subme qp chromaSatd maxSlices vertRestriction "
               "srcPlane outx outy cost mvcost cbPlane crPlane chromaOffset chromaStride cw ch").split()


DIA_FIELDS = "plane w h blockOffset mnx mny mxx mxy mvpx mvpy qp outx outy cost".split()


def parse(path, dias=None, pmvs=None, sels=None, chks=None, upds=None):
    d = np.fromfile(path, np.uint8).tobytes()
    off, planes, calls, mvcs, blocks = 0, {}, [], [], []
    while off < len(d):
        kind, n = np.frombuffer(d, np.int32, 2, off); off += 8
        ints = np.frombuffer(d, np.int32, n, off).copy(); off += 4 * n
        if kind in (1, 3):
            pid, stride, rows = int(ints[0]), int(ints[1]), int(ints[2])
            px = np.frombuffer(d, np.uint16, stride * rows, off).copy(); off += 2 * stride * rows
            planes[pid] = (ints, px)
        elif kind == 5:
            if pmvs is not None: pmvs.append(ints)
        elif kind == 6:
            npx = int(ints[1]) * int(ints[2])
            px = np.frombuffer(d, np.uint16, npx, off).copy(); off += 2 * npx
            if sels is not None: sels.append((ints, px))
        elif kind == 7:
            if chks is not None: chks.append(ints)
        elif kind == 8:
            if upds is not None: upds.append(ints)
        elif kind == 4:
            npx = int(ints[1]) * int(ints[2])
            px = np.frombuffer(d, np.uint16, npx, off).copy(); off += 2 * npx
            if dias is not None: dias.append((ints, px))
        else:
            w, h = int(ints[1]), int(ints[2])
            npx = w * h + 2 * int(ints[27]) * int(ints[28])          # luma block [+ Cb + Cr blocks of the source PU]
            px = np.frombuffer(d, np.uint16, npx, off).copy(); off += 2 * npx
            calls.append(ints[:NI]); mvcs.append(ints[NI:]); blocks.append(px)
    return planes, calls, mvcs, blocks


def build(depth, args, out):
    exe = os.path.join(ROOT, "oracle", "_ref", "x265tme_%d" % depth)
    with tempfile.TemporaryDirectory() as td:
        raw = os.path.join(td, "tme.bin")
        subprocess.check_call([exe] + args[:4] + [raw] + args[4:], stdout=subprocess.DEVNULL)
        planes, calls, mvcs, blocks = parse(raw)
    dt = np.uint8 if depth == 8 else np.uint16
    calls = np.stack(calls).astype(np.int32)
    mvc = np.zeros((len(calls), 24), np.int16)
    for i, m in enumerate(mvcs):
        assert len(m) <= 24, "more than 12 candidates"
        mvc[i, :len(m)] = m
    starts = np.concatenate([[0], np.cumsum([len(b) for b in blocks])]).astype(np.int64)
    data = {"fields": np.array(CALL_FIELDS), "calls": calls, "mvc": mvc, "fenc": np.concatenate(blocks).astype(dt), "fenc_start": starts,
            "cmdline": np.array(" ".join(args))}
    for pid, (ints, px) in planes.items():
        data["plane%d_geom" % pid] = ints
        data["plane%d" % pid] = px.astype(dt)
    np.savez_compressed(out, **data)
    return calls


def build_dia(depth, args, out):
    """dia_{8,10}.npz: the MotionEstimate::diamondSearch calls (ThreadedME's predictor stage, search.cpp:355-363) of a --threaded-me encode"""
    exe = os.path.join(ROOT, "oracle", "_ref", "x265tme_%d" % depth)
    dias = []
    with tempfile.TemporaryDirectory() as td:
        raw = os.path.join(td, "tme.bin")
        subprocess.check_call([exe] + args[:4] + [raw] + args[4:], stdout=subprocess.DEVNULL)
        planes, _, _, _ = parse(raw, dias)
    dt = np.uint8 if depth == 8 else np.uint16
    calls = np.stack([i for i, _ in dias]).astype(np.int32)
    used = set(int(v) for v in calls[:, 0])
    starts = np.concatenate([[0], np.cumsum([len(b) for _, b in dias])]).astype(np.int64)
    data = {"fields": np.array(DIA_FIELDS), "calls": calls, "fenc": np.concatenate([b for _, b in dias]).astype(dt), "fenc_start": starts, "cmdline": np.array(" ".join(args))}
    for pid, (ints, px) in planes.items():
        if pid in used:
            data["plane%d_geom" % pid] = ints
            data["plane%d" % pid] = px.astype(dt)
    np.savez_compressed(out, **data)
    return calls


def build_mvpsel(depth, args, out):
    """mvpsel_{8,10}.npz: Search::selectMVP calls (source PU, reference plane, both AMVP candidates, clipMv limits -> index) and the checkBestMVP / updateMVP
    records of a --threaded-me encode (ref_tme.cpp kinds 6-8)"""
    exe = os.path.join(ROOT, "oracle", "_ref", "x265tme_%d" % depth)
    sels, chks, upds = [], [], []
    with tempfile.TemporaryDirectory() as td:
        raw = os.path.join(td, "tme.bin")
        subprocess.check_call([exe] + args[:4] + [raw] + args[4:], stdout=subprocess.DEVNULL, env=dict(os.environ, X265TME_SEL="1000000"))
        planes, _, _, _ = parse(raw, None, None, sels, chks, upds)
    dt = np.uint8 if depth == 8 else np.uint16
    rng = np.random.default_rng(11)
    keep = sorted(rng.permutation(len(sels))[:2400].tolist())
    sels = [sels[i] for i in keep]
    calls = np.stack([i for i, _ in sels]).astype(np.int32)
    starts = np.concatenate([[0], np.cumsum([len(b) for _, b in sels])]).astype(np.int64)
    chk = np.unique(np.stack(chks), axis=0); upd = np.unique(np.stack(upds), axis=0)
    data = {"select": calls, "fenc": np.concatenate([b for _, b in sels]).astype(dt), "fenc_start": starts, "check": chk[rng.permutation(len(chk))[:6000]], "update": upd[rng.permutation(len(upd))[:3000]],
            "cmdline": np.array(" ".join(args))}
    for pid in set(int(v) for v in calls[:, 0]):
        ints, px = planes[pid]
        data["plane%d_geom" % pid] = ints
        data["plane%d" % pid] = px.astype(dt)
    np.savez_compressed(out, **data)
    return calls, data["check"], data["update"]


def build_pu(depth, args, out, n_keep=700, keep_all=False):
    """pu_{8,10}.npz: whole Search::puMotionEstimation calls of a --threaded-me encode (ref_tme.cpp kind 9) with the records of everything under them (ints only: the
    source blocks travel with the call record), P and B pictures; a sample of n_keep calls that keeps every bidirectional / list-1 outcome"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tme_pu
    exe = os.path.join(ROOT, "oracle", "_ref", "x265tme_%d" % depth)
    with tempfile.TemporaryDirectory() as td:
        raw = os.path.join(td, "tme.bin")
        subprocess.check_call([exe] + args[:4] + [raw] + args[4:], stdout=subprocess.DEVNULL, env=dict(os.environ, X265TME_PU="1000000"))
        planes, calls = tme_pu.parse_stream(raw)
    rng = np.random.default_rng(3)
    special = [i for i, c in enumerate(calls) if any(tme_pu.decode(c)["out"][p][12] >= 0 for p in range(int(c["ints"][58])))]
    keep = list(range(len(calls))) if keep_all else sorted(set(special[:250]) | set(rng.permutation(len(calls))[:n_keep].tolist()))
    calls = [calls[i] for i in keep]
    dt = np.uint8 if depth == 8 else np.uint16
    data = {"call_ints": np.concatenate([c["ints"] for c in calls]), "call_ints_start": np.concatenate([[0], np.cumsum([len(c["ints"]) for c in calls])]),
            "call_px": np.concatenate([c["px"] for c in calls]).astype(dt), "call_px_start": np.concatenate([[0], np.cumsum([len(c["px"]) for c in calls])]),
            "sub_kind": np.array([k for c in calls for (k, _, _) in c["subs"]], np.int32), "sub_ints": np.concatenate([i for c in calls for (_, i, _) in c["subs"]]),
            "sub_ints_start": np.concatenate([[0], np.cumsum([len(i) for c in calls for (_, i, _) in c["subs"]])]),
            "call_sub_start": np.concatenate([[0], np.cumsum([len(c["subs"]) for c in calls])]), "cmdline": np.array(" ".join(args))}
    used = set()
    for c in calls:
        d = tme_pu.decode(c)
        used |= set(int(v) for v in d["planeIds"].reshape(-1) if v >= 0)
    for pid in used:
        ints, px = planes[pid]
        data["plane%d_geom" % pid] = ints
        data["plane%d" % pid] = px.astype(dt)
    np.savez_compressed(out, **data)
    return calls


def build_sched(out):
    """tme_sched.npz: the order of Search::puMotionEstimation calls inside a CTU (Analysis::computeMVForPUs) with slots, neighbour slots, area index and partition
    rectangles, for three presets (rect / amp off, rect on, rect + amp on): per config an int array [call][cols], CTU by CTU of the first P picture"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tme_pu
    exe = os.path.join(ROOT, "oracle", "_ref", "x265tme_8")
    data = {}
    for name, args in (("medium", ["128", "128", "2", "medium"]), ("slow", ["128", "128", "2", "slow"]), ("slower", ["128", "128", "2", "slower", "bframes=0"])):
        with tempfile.TemporaryDirectory() as td:
            raw = os.path.join(td, "tme.bin")
            subprocess.check_call([exe] + args[:4] + [raw] + args[4:], stdout=subprocess.DEVNULL, env=dict(os.environ, X265TME_PU="10000000"))
            _, calls = tme_pu.parse_stream(raw)
        rows = []
        for c in calls:
            d = tme_pu.decode(c)
            ctu = (d["cuY"] // 64) * 2 + d["cuX"] // 64
            geo = np.zeros(8, np.int32); geo[:4 * d["numPart"]] = d["geo"].reshape(-1)
            rows.append([d["curPOC"], ctu, d["part"], 1 << d["log2CU"], d["cuX"], d["cuY"], d["puOffset"], d["area"], d["finalIdx"]] + d["nbIdx"] + [d["numPart"]] + list(geo))
        rows = np.array(rows, np.int32)
        poc = rows[0, 0]
        data[name] = rows[rows[:, 0] == poc]
    np.savez_compressed(out, **data)
    return {k: len(v) for k, v in data.items()}


def build_amvp(out):
    """amvp.npz: CUData::getPMV calls (ref_tme.cpp kind 5) of a --threaded-me encode and of a regular encode with B pictures and several references; the records
    are bit-depth independent (8-bit harness); identical records are kept once.  Row layout = the recorder's (fixed 99 ints + 22 mvc ints, zero padded)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "x265tme_8")
    rows = []
    for args in (["192", "128", "6", "slow", "bframes=2"], ["192", "128", "8", "slow", "bframes=3", "threaded-me=0", "ref=3"], ["192", "128", "5", "medium", "bframes=0", "threaded-me=0", "ref=2"]):
        pm = []
        with tempfile.TemporaryDirectory() as td:
            raw = os.path.join(td, "tme.bin")
            subprocess.check_call([exe] + args[:4] + [raw] + args[4:], stdout=subprocess.DEVNULL, env=dict(os.environ, X265TME_PMV="200000"))
            parse(raw, None, pm)
        for r in pm:
            row = np.zeros(99 + 22, np.int32); row[:len(r)] = r
            rows.append(row)
    rows = np.unique(np.stack(rows), axis=0)
    if len(rows) > 24000:
        rows = rows[np.random.default_rng(5).permutation(len(rows))[:24000]]
    np.savez_compressed(out, calls=rows)
    return rows


if __name__ == "__main__":
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "oracle"), "tme"])
    # tme_*: --threaded-me encodes (luma-only searches of puMotionEstimation); mec_*: regular encodes whose predInterSearch searches carry the chroma
    # SATD terms (subme >= 3), incl. weighted / several references and B pictures
    for depth, args in (() if "--sched-only" in sys.argv else ((8, ["256", "192", "4", "medium"]), (10, ["256", "192", "4", "slow", "amp=0", "rect=0"]))):
        out = os.path.join(ROOT, "tests", "golden", "dia_%d.npz" % depth)
        c = build_dia(depth, args, out)
        f = {n: c[:, i] for i, n in enumerate(DIA_FIELDS)}
        print("dia", depth, "calls", len(c), "size", os.path.getsize(out), "shapes", sorted({(int(a), int(b)) for a, b in zip(f["w"], f["h"])}), "out range", f["outx"].min(), f["outx"].max(), f["outy"].min(), f["outy"].max(),
              "mvp", np.unique(f["mvpx"]), np.unique(f["mvpy"]))
    for depth in (8, 10):
        # every puMotionEstimation call of every CTU of a P and a B picture, in order: what x265hip_tme_frame steps through (one reference per list)
        out = os.path.join(ROOT, "tests", "golden", "tmectu_%d.npz" % depth)
        c = build_pu(depth, ["128", "128", "3", "slow", "bframes=1", "ref=1"], out, keep_all=True)
        print("tmectu", depth, "calls", len(c), "size", os.path.getsize(out))
    print("sched", build_sched(os.path.join(ROOT, "tests", "golden", "tme_sched.npz")))
    if "--sched-only" in sys.argv: sys.exit(0)
    for depth in (8, 10):
        out = os.path.join(ROOT, "tests", "golden", "pu_%d.npz" % depth)
        c = build_pu(depth, ["128", "128", "6", "slow", "bframes=2", "ref=2"], out)
        print("pu", depth, "calls", len(c), "size", os.path.getsize(out))
    for depth in (8, 10):
        out = os.path.join(ROOT, "tests", "golden", "mvpsel_%d.npz" % depth)
        c, k, u = build_mvpsel(depth, ["128", "128", "5", "slow", "bframes=2"], out)
        print("mvpsel", depth, "select", len(c), "picked", np.bincount(c[:, 13]), "shapes", len({(int(a), int(b)) for a, b in zip(c[:, 1], c[:, 2])}), "check", len(k), "switched", int((k[:, 11] != k[:, 6]).sum()),
              "update", len(u), "size", os.path.getsize(out))
    r = build_amvp(os.path.join(ROOT, "tests", "golden", "amvp.npz"))
    print("amvp calls", len(r), "size", os.path.getsize(os.path.join(ROOT, "tests", "golden", "amvp.npz")), "numMvc", np.bincount(r[:, 98]), "temporal used", int((r[:, 93] != 0).sum() + (r[:, 92] != 0).sum()),
          "lists", np.bincount(r[:, 0]), "refs", np.bincount(r[:, 1]))
    if "--dia-only" in sys.argv: sys.exit(0)
    for name, depth, args in (("tme", 8, ["128", "128", "3", "medium"]), ("tme", 10, ["128", "128", "3", "slow", "amp=0"]),
                              ("mec", 8, ["128", "64", "4", "slow", "threaded-me=0", "rect=0", "amp=0", "bframes=1"]),
                              ("mec", 10, ["128", "64", "4", "slower", "threaded-me=0", "rect=0", "amp=0", "bframes=1", "me=hex"])):
        out = os.path.join(ROOT, "tests", "golden", "%s_%d.npz" % (name, depth))
        c = build(depth, args, out)
        f = {n: c[:, i] for i, n in enumerate(CALL_FIELDS)}
        print(name, depth, "calls", len(c), "size", os.path.getsize(out), "methods", np.unique(f["method"]), "subme", np.unique(f["subme"]), "chromaSatd", np.unique(f["chromaSatd"]),
              "numCand max", f["numCand"].max(), "shapes", sorted({(int(a), int(b)) for a, b in zip(f["w"], f["h"])}), "qp", np.unique(f["qp"]),
              "vert/slices/src", np.unique(f["vertRestriction"]), np.unique(f["maxSlices"]), np.unique(f["srcPlane"]))
