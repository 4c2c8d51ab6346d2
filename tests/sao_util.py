"""SAO statistics (saoCuStatsBO / E0..E3, encoder/sao.cpp:1774-1937): case generator and the three backends' call forms.
A case = (type, diff[64*64] int16, rec plane, stride, recOff, endX, endY, upBuff1, upBufft, upOff, stats, count)."""
import ctypes as C

import numpy as np

TYPES = {0: "saoCuStatsE0", 1: "saoCuStatsE1", 2: "saoCuStatsE2", 3: "saoCuStatsE3", 4: "saoCuStatsBO"}
NCLASS = {0: 5, 1: 5, 2: 5, 3: 5, 4: 32}


def cases(depth, seed, n=40):
    rng = np.random.default_rng(seed)
    pm = (1 << depth) - 1
    dt = np.uint8 if depth == 8 else np.uint16
    out = []
    for i in range(n):
        t = int(rng.integers(0, 5))
        endX = int(rng.choice([1, 3, 8, 15, 16, 31, 32, 47, 63])) if t in (2, 3) else int(rng.choice([1, 4, 8, 16, 17, 32, 48, 64]))
        endY = int(rng.choice([1, 2, 7, 8, 16, 33, 63])) if t in (2, 3) else int(rng.choice([1, 2, 8, 16, 31, 64]))
        stride = int(rng.choice([endX + 2, 80, 96, 200]))
        stride = max(stride, endX + 2)
        rows = endY + 3
        kind = i % 4                                     # smooth (many ties), noisy, extremes, random
        if kind == 0:
            rec = (np.add.outer(np.arange(rows), np.arange(stride)) // 3 * 5 % (pm + 1)).astype(dt)
        elif kind == 1:
            rec = np.clip(128 * (pm + 1) // 256 + rng.integers(-3, 4, (rows, stride)), 0, pm).astype(dt)
        elif kind == 2:
            rec = rng.choice([0, pm], (rows, stride)).astype(dt)
        else:
            rec = rng.integers(0, pm + 1, (rows, stride)).astype(dt)
        diff = rng.integers(-pm, pm + 1, 64 * 64).astype(np.int16)
        up1 = rng.integers(-1, 2, 70).astype(np.int8); upt = rng.integers(-1, 2, 70).astype(np.int8)
        stats = rng.integers(-1000, 1000, NCLASS[t]).astype(np.int32); count = rng.integers(0, 500, NCLASS[t]).astype(np.int32)
        out.append((t, diff, rec.reshape(-1), stride, stride + 1, endX, endY, up1, upt, 2, stats, count))
    return out


def run_oracle(ora, c):
    t, diff, rec, stride, off, endX, endY, up1, upt, uo, stats, count = c
    s, n, a, b = stats.copy(), count.copy(), up1.copy(), upt.copy()
    P = lambda x, o=0: C.c_void_p(x.ctypes.data + o * x.itemsize)  # noqa: E731
    ora.lib.xo_sao_stats(t, P(diff), P(rec, off), C.c_ssize_t(stride), P(a, uo), P(b, uo), endX, endY, P(s), P(n))
    return s, n, a, b


def run_ref(ref, c):
    t, diff, rec, stride, off, endX, endY, up1, upt, uo, stats, count = c
    o = ref.r.call("sao_stats", [t, stride, off, endX, endY, uo], [diff, rec, up1, upt, stats, count])
    return (np.frombuffer(o[0], np.int32).copy(), np.frombuffer(o[1], np.int32).copy(), np.frombuffer(o[2], np.int8).copy(), np.frombuffer(o[3], np.int8).copy())


def run_hip(lib, c):
    """through the table slot, host pointers, the reference's own signature"""
    t, diff, rec, stride, off, endX, endY, up1, upt, uo, stats, count = c
    s, n, a, b = stats.copy(), count.copy(), up1.copy(), upt.copy()
    P = C.c_void_p
    p = lambda x, o=0: C.c_void_p(x.ctypes.data + o * x.itemsize)  # noqa: E731
    if t in (0, 4):
        lib.scalar(TYPES[t], None, (P, P, C.c_ssize_t, C.c_int, C.c_int, P, P))(p(diff), p(rec, off), stride, endX, endY, p(s), p(n))
    elif t == 2:
        lib.scalar(TYPES[t], None, (P, P, C.c_ssize_t, P, P, C.c_int, C.c_int, P, P))(p(diff), p(rec, off), stride, p(a, uo), p(b, uo), endX, endY, p(s), p(n))
    else:
        lib.scalar(TYPES[t], None, (P, P, C.c_ssize_t, P, C.c_int, C.c_int, P, P))(p(diff), p(rec, off), stride, p(a, uo), endX, endY, p(s), p(n))
    return s, n, a, b
