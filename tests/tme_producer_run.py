"""Helper of test_tme_producer_gpu.py: one picture through x265hip_tme_picture on synthetic planes, prints the SHA-1 of the table.  Run as a script so that the switches
the caller sets (TME_RUN_FLAGS = x265hip_tme_picture_desc.flags: 1 one launch per stage, 2 packed lane groups) differ between runs, each in a fresh process.   python tests/tme_producer_run.py depth preset P|B [method merange]"""
import ctypes as C
import hashlib
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import x265hip  # noqa: E402


def main():
    depth, preset, kind = int(sys.argv[1]), sys.argv[2], sys.argv[3]
    TmeProducer = importlib.import_module("x265-mod-by-patman_amd.tme_host").TmeProducer
    lib = C.CDLL(x265hip.lib_path(depth))
    W, H, margin = 416, int(os.environ.get("TME_RUN_HEIGHT", "240")), 96                     # the bottom CTU row is cut by the picture edge
    stride, rows = W + 2 * margin, ((H + 63) // 64) * 64 + 2 * margin
    rng = np.random.default_rng(11)
    dt = np.uint8 if depth == 8 else np.uint16
    base = rng.integers(0, 1 << depth, (rows // 8 + 2, stride // 8 + 2)).astype(np.int32)
    ref0 = np.kron(base, np.ones((8, 8), dtype=np.int32))[:rows, :stride]
    ref0 = np.clip(ref0 + rng.integers(-6, 7, ref0.shape), 0, (1 << depth) - 1)
    ref1 = np.clip(np.roll(ref0, (-2, 7), axis=(0, 1)) + rng.integers(-5, 6, ref0.shape), 0, (1 << depth) - 1)
    cur = np.clip(np.roll(ref0, (3, -5), axis=(0, 1)) + rng.integers(-4, 5, ref0.shape), 0, (1 << depth) - 1)
    ref0, ref1, cur = (np.ascontiguousarray(a.astype(dt)).reshape(-1) for a in (ref0, ref1, cur))
    rect, amp, method, subme = {"medium": (False, False, 1, 2), "slow": (True, False, 3, 3), "slower": (True, True, 3, 4)}[preset]      # rect / amp / method / subme of the presets (param.cpp:567-608)
    merange = 57
    if len(sys.argv) > 5:
        method, merange = int(sys.argv[4]), int(sys.argv[5])
    prod = TmeProducer(lib, W, H, 64, 8, rect, amp)
    flags = int(os.environ.get("TME_RUN_FLAGS", "0"))
    table = prod.empty_table()
    bands = os.environ.get("TME_RUN_BANDS")
    if bands:
        # frame threads: "whole" = one call on complete references; "rows" = the picture in bands of CTU rows, every reference handed over with exactly the rows the
        # encoder's own rule releases for the band (FrameEncoder::m_refLagRows, frameencoder.cpp:166-171, 1029-1036) -- the rows below them hold garbage in the host copy,
        # so a search (or a phase plane's vertical taps) that looked beyond the rule would change the table
        hpel = {0: 1, 1: 1, 2: 1, 3: 2, 4: 3, 5: 1, 6: 2, 7: 3}[subme]                  # MotionEstimate::hpelIterationCount (motion.cpp:48-58, 155-159)
        lag = 1 + (merange + (1 if method < 2 else 0) + 4 + 2 + (hpel + 1) // 2 + 63) // 64
        n_rows = (H + 63) // 64
        rl = [[ref0, ref1], []] if kind == "P" else [[ref0], [ref1]]
        kw = dict(is_p=kind == "P", method=method, subme=subme, merange=merange, cur_poc=2 if kind == "P" else 1, ref_pocs=((1, 0), ()) if kind == "P" else ((0,), (2,)), flags=flags, frame_threads=3)
        if bands == "whole":
            prod.picture(cur, rl, stride, margin * stride + margin, table, **kw)
        else:
            step = int(bands)
            keys = [[101 + i for i in range(len(rl[0]))], [201 + i for i in range(len(rl[1]))]]
            for r0 in range(0, n_rows, step):
                r1 = min(n_rows, r0 + step)
                last = min(n_rows - 1, r1 - 1 + lag)
                valid = rows if last == n_rows - 1 else margin + (last + 1) * 64
                part = [[p.copy() for p in lst] for lst in rl]
                for lst in part:
                    for p in lst:
                        p[valid * stride:] = rng.integers(0, 1 << depth, p.size - valid * stride).astype(dt)
                prod.picture(cur, part, stride, margin * stride + margin, table, ref_keys=keys, rows=(r0, r1 - r0), rows_valid=valid, **kw)
    elif kind == "P":
        prod.picture(cur, [[ref0, ref1], []], stride, margin * stride + margin, table, method=method, subme=subme, merange=merange, cur_poc=2, ref_pocs=((1, 0), ()), flags=flags)
    else:
        prod.picture(cur, [[ref0], [ref1]], stride, margin * stride + margin, table, is_p=False, method=method, subme=subme, merange=merange, cur_poc=1, ref_pocs=((0,), (2,)), flags=flags)
    used = int((table["ref"] >= 0).any(axis=1).sum())
    bi = int(((table["ref"][:, 0] >= 0) & (table["ref"][:, 1] >= 0)).sum())
    prod.close()
    print("table", hashlib.sha1(table.tobytes()).hexdigest(), used, bi)


if __name__ == "__main__":
    main()
