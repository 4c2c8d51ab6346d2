"""Fixtures tests/golden/tme_{8,10}.npz (tests/make_golden_tme.py): the motionEstimate calls of a reference encode run with --threaded-me."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class TmeFixture:
    def __init__(self, depth):
        d = np.load(os.path.join(GOLD, "tme_%d.npz" % depth))
        self.depth = depth
        self.fields = [str(f) for f in d["fields"]]
        self.calls = d["calls"]
        self.col = {n: self.calls[:, i] for i, n in enumerate(self.fields)}
        self.mvc, self.fenc, self.start = d["mvc"], d["fenc"], d["fenc_start"]
        self.planes = {}
        for k in d.files:
            if k.startswith("plane") and k.endswith("_geom"):
                pid = int(k[5:-5])
                g = d[k]
                self.planes[pid] = dict(stride=int(g[1]), rows=int(g[2]), origin=int(g[3]), width=int(g[4]), height=int(g[5]), px=d["plane%d" % pid])

    def __len__(self):
        return len(self.calls)

    def block(self, i):
        return self.fenc[self.start[i]:self.start[i + 1]]

    def groups(self):
        """indices of calls that can share one x265hip_me_batch launch: same reference plane, PU shape, qp (method / subme / merange are the clip's)"""
        keys = {}
        for i in range(len(self.calls)):
            c = self.col
            keys.setdefault((int(c["plane"][i]), int(c["w"][i]), int(c["h"][i]), int(c["qp"][i]), int(c["method"][i]), int(c["subme"][i]), int(c["merange"][i])), []).append(i)
        return keys


class MecFixture(TmeFixture):
    """tests/golden/mec_{8,10}.npz: the searches of Search::predInterSearch in a regular encode (several references, B pictures, candidates),
    with the chroma SATD terms of subpelCompare (subme >= 3); per call also the Cb / Cr source blocks and reference chroma planes."""
    def __init__(self, depth):
        d = np.load(os.path.join(GOLD, "mec_%d.npz" % depth))
        self.depth = depth
        self.fields = [str(f) for f in d["fields"]]
        self.calls = d["calls"]
        self.col = {n: self.calls[:, i] for i, n in enumerate(self.fields)}
        self.mvc, self.fenc, self.start = d["mvc"], d["fenc"], d["fenc_start"]
        self.planes = {}
        for k in d.files:
            if k.startswith("plane") and k.endswith("_geom"):
                pid = int(k[5:-5])
                g = d[k]
                self.planes[pid] = dict(stride=int(g[1]), rows=int(g[2]), origin=int(g[3]), width=int(g[4]), height=int(g[5]), px=d["plane%d" % pid])

    def blocks(self, i):
        """(luma, Cb, Cr) source blocks of call i"""
        c = self.col
        w, h, cw, ch = int(c["w"][i]), int(c["h"][i]), int(c["cw"][i]), int(c["ch"][i])
        b = self.fenc[self.start[i]:self.start[i + 1]]
        return b[:w * h], b[w * h:w * h + cw * ch], b[w * h + cw * ch:]


class DiaFixture:
    """tests/golden/dia_{8,10}.npz: the MotionEstimate::diamondSearch calls of a --threaded-me encode (ThreadedME's predictor stage)"""
    def __init__(self, depth):
        d = np.load(os.path.join(GOLD, "dia_%d.npz" % depth))
        self.depth = depth
        self.fields = [str(f) for f in d["fields"]]
        self.calls = d["calls"]
        self.col = {n: self.calls[:, i] for i, n in enumerate(self.fields)}
        self.fenc, self.start = d["fenc"], d["fenc_start"]
        self.planes = {}
        for k in d.files:
            if k.startswith("plane") and k.endswith("_geom"):
                pid = int(k[5:-5])
                g = d[k]
                self.planes[pid] = dict(stride=int(g[1]), rows=int(g[2]), origin=int(g[3]), width=int(g[4]), height=int(g[5]), px=d["plane%d" % pid])

    def __len__(self):
        return len(self.calls)

    def block(self, i):
        return self.fenc[self.start[i]:self.start[i + 1]]


class MvpSelFixture:
    """tests/golden/mvpsel_{8,10}.npz: Search::selectMVP calls and checkBestMVP / updateMVP records of a --threaded-me encode"""
    def __init__(self, depth):
        d = np.load(os.path.join(GOLD, "mvpsel_%d.npz" % depth))
        self.depth = depth
        self.select, self.check, self.update = d["select"], d["check"], d["update"]
        self.fenc, self.start = d["fenc"], d["fenc_start"]
        self.planes = {}
        for k in d.files:
            if k.startswith("plane") and k.endswith("_geom"):
                pid = int(k[5:-5])
                g = d[k]
                self.planes[pid] = dict(stride=int(g[1]), rows=int(g[2]), origin=int(g[3]), px=d["plane%d" % pid])

    def block(self, i):
        return self.fenc[self.start[i]:self.start[i + 1]]


def u32(v):
    return int(v) & 0xFFFFFFFF


def lam64(lo, hi):
    return u32(lo) | (u32(hi) << 32)
