"""Loader for tests/golden/*.npz (written by tests/golden/make_golden.py)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _unpack(obj, z):
    if isinstance(obj, dict):
        return z[obj["a"]]
    if isinstance(obj, list):
        return [_unpack(o, z) for o in obj]
    return obj


def load(family, depth):
    z = np.load(os.path.join(GOLDEN, "%s_%d.npz" % (family, depth)))
    manifest = json.loads(bytes(z["manifest"]).decode())
    for c in manifest:
        yield c["label"], c["method"], tuple(_unpack(c["args"], z)), tuple(_unpack(c["outs"], z))
