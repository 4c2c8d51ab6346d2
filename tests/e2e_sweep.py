"""Wider end-to-end sweep than tests/test_e2e_gpu.py (run by hand on a GPU box: python tests/e2e_sweep.py): the reference encoder
with its C table vs with the HIP table, one line per configuration."""
import filecmp
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_e2e_gpu import _env  # noqa: E402

CONFIGS = [
    (8, 64, 64, 2, "ultrafast", []), (8, 128, 64, 3, "medium", []), (10, 128, 64, 2, "medium", []), (8, 64, 64, 2, "slow", []),
    (10, 64, 64, 2, "slower", []), (8, 136, 72, 2, "medium", []), (8, 64, 64, 3, "medium", ["me=dia"]), (8, 64, 64, 3, "medium", ["me=umh"]),
    (8, 64, 64, 3, "medium", ["me=sea", "merange=8"]), (8, 64, 64, 2, "medium", ["me=full", "merange=6"]), (8, 64, 64, 3, "medium", ["weightb=1", "bframes=2"]),
    (8, 64, 64, 2, "medium", ["lowpass-dct=1"]), (8, 64, 64, 2, "medium", ["tskip=1", "rdoq-level=2"]), (8, 64, 64, 2, "medium", ["nr-intra=100", "nr-inter=100"]),
    (8, 64, 64, 2, "medium", ["subme=7"]), (8, 64, 64, 2, "medium", ["scaling-list=default"]), (8, 64, 64, 2, "medium", ["lossless=1"]),
    (8, 64, 64, 2, "medium", ["rd=1"]), (8, 64, 64, 2, "medium", ["rd=5", "rect=1", "amp=1"]), (8, 64, 64, 2, "medium", ["aq-mode=3", "ssim-rd=1"]),
    (8, 64, 64, 2, "medium", ["ctu=32"]), (8, 64, 64, 2, "medium", ["ctu=16", "min-cu-size=8"]), (8, 64, 64, 2, "medium", ["tu-intra-depth=3", "tu-inter-depth=3"]),
    (8, 64, 64, 3, "medium", ["hme=1"]), (8, 64, 64, 2, "medium", ["limit-sao=1"]), (8, 64, 64, 2, "medium", ["sao-non-deblock=1"]),
    (10, 64, 64, 3, "medium", ["me=sea", "merange=8"]), (8, 64, 64, 3, "medium", ["cutree=0"]), (8, 64, 64, 4, "medium", ["b-adapt=2", "bframes=3", "rc-lookahead=4"]),
]


def main():
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        for (depth, w, h, frames, preset, extra) in CONFIGS:
            enc = os.path.join(ROOT, "oracle", "_ref", "x265enc_%d" % depth)
            lib = os.path.join(os.environ.get("X265HIP_LIBDIR", os.path.join(ROOT, "x265-mod-by-patman_amd")), "libx265hip_%d.so" % depth)
            outs = {}
            for mode in ("c", "hip"):
                out = os.path.join(td, mode + ".hevc")
                r = subprocess.run([enc, mode, lib, str(w), str(h), str(frames), preset, out] + extra, env=_env(), capture_output=True, text=True, timeout=1800)
                outs[mode] = (out, r.returncode, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
            same = outs["c"][1] == 0 and outs["hip"][1] == 0 and filecmp.cmp(outs["c"][0], outs["hip"][0], shallow=False)
            bad += 0 if same else 1
            info = json.loads(outs["hip"][2]) if outs["hip"][1] == 0 else outs["hip"][2]
            print("%-9s %d-bit %dx%d x%d %-9s %-40s %s" % ("IDENTICAL" if same else "DIFFERENT", depth, w, h, frames, preset, " ".join(extra), info), flush=True)
    print("%d of %d configurations identical" % (len(CONFIGS) - bad, len(CONFIGS)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
