"""Generates tests/golden/lookahead_{8,10}.npz by running the REFERENCE's Lowres / Lookahead classes (oracle/_ref/x265la_*,
built by `make -C oracle la` from /root/reference) on small synthetic clips.  The fixtures hold inputs (source pictures, AQ factors,
the estimates asked for) and the reference's outputs only.

    python tests/make_golden_lookahead.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
from lookahead_util import run_reference, synth_clip  # noqa: E402

CLIPS = [dict(W=96, H=80, n=4, shift=(2, 1), aq=1), dict(W=160, H=64, n=4, shift=(-4, 2), aq=0)]
TRIPLES = [(0, 1, 1, 0), (0, 2, 2, 0), (0, 2, 3, 1), (1, 2, 3, 0), (0, 3, 3, 0), (0, 1, 3, 0)]


def main():
    for depth in (8, 10):
        out = {"triples": np.array(TRIPLES, np.int32), "nclips": np.array(len(CLIPS))}
        for ci, c in enumerate(CLIPS):
            frames = synth_clip(c["W"], c["H"], c["n"], depth, seed=900 + ci + depth, shift=c["shift"])
            hdr, pf, pt = run_reference(depth, frames, TRIPLES, c["aq"])
            pre = "c%d_" % ci
            out[pre + "frames"] = np.stack(frames); out[pre + "aq"] = np.array(c["aq"])
            out[pre + "invQ"] = np.stack([f["invQ"] for f in pf])
            for k in ("intraCost", "intraMode", "lowresCosts", "rowSatds"):
                out[pre + "intra_" + k] = np.stack([f[k] for f in pf])
            for ti, t in enumerate(pt):
                for k in ("mvs0", "mvc0", "mvs1", "mvc1", "lowresCosts", "rowSatds"):
                    out[pre + "t%d_%s" % (ti, k)] = t[k]
                out[pre + "t%d_hdr" % ti] = np.array([t["doSearch"][0], t["doSearch"][1], t["costEstNorm"], t["costEstAq"], t["intraMbs"]], np.int32)
        path = os.path.join(HERE, "golden", "lookahead_%d.npz" % depth)
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
