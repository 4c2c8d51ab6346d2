#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference C primitives.

Run in the build container (needs oracle/_ref/x265ref_{8,10}, i.e. /root/reference):
    python tests/golden/make_golden.py
Each fixture holds seeded inputs and the outputs the reference produced for them; the fixture is
DATA (inputs + expected outputs) -- no reference source travels.  tests/test_golden.py replays the
cases through oracle/x265_oracle.c (CPU) and through the HIP library (GPU).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from backends import Ref  # noqa: E402
from cases import FAMILIES, run_case  # noqa: E402

SEED = 0x5EED0000
KEEP_EVERY = {"pixelcmp": 9, "blockops": 4, "transform": 3, "interp": 11, "intra": 13, "extras": 1}


def pack(obj, arrays, ids):
    if isinstance(obj, np.ndarray):
        key = ids.get(id(obj))
        if key is None:
            key = "a%d" % len(arrays)
            arrays[key] = obj
            ids[id(obj)] = key
        return {"a": key}
    if isinstance(obj, (list, tuple)):
        return [pack(o, arrays, ids) for o in obj]
    if isinstance(obj, (int, np.integer)):
        return int(obj)
    return obj


def main():
    only = set(sys.argv[1:])                      # optional: regenerate just the named families
    for depth in (8, 10):
        ref = Ref(depth)
        for fam, gen in sorted(FAMILIES.items()):
            if only and fam not in only:
                continue
            rng = np.random.default_rng(SEED + depth)
            arrays, ids, manifest = {}, {}, []
            keep = []  # keep argument arrays alive so id() stays unique
            for k, (label, method, args) in enumerate(gen(depth, rng)):
                if k % KEEP_EVERY[fam]:
                    continue
                outs = run_case(ref, method, args)
                keep.append(args)
                manifest.append({"label": label, "method": method, "args": pack(args, arrays, ids),
                                 "outs": pack(outs, arrays, {})})
            arrays["manifest"] = np.frombuffer(json.dumps(manifest).encode(), np.uint8)
            path = os.path.join(HERE, "%s_%d.npz" % (fam, depth))
            np.savez_compressed(path, **arrays)
            print(path, len(manifest), "cases", os.path.getsize(path) // 1024, "KiB")
        ref.close()


if __name__ == "__main__":
    main()
