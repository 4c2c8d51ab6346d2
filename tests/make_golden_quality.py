"""Regenerates tests/golden/quality_{8,10}.npz: pictures the reference ENCODER reconstructed (oracle/_ref/x265enc_*, X265ENC_DUMP) with the
SSIM / PSNR it reported for them (x265_picture.frameData).  Needs /root/reference at build time of oracle/_ref; the fixtures hold data only."""
import os
import pathlib
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from test_quality_oracle_vs_ref import encoder_dump  # noqa: E402

CASES = {8: (136, 72, 3, "medium", ["bframes=1"], 64), 10: (136, 104, 2, "fast", ["ctu=16", "qp=40"], 16)}

if __name__ == "__main__":
    for depth, (w, h, frames, preset, extra, ctu) in CASES.items():
        with tempfile.TemporaryDirectory() as td:
            pics = encoder_dump(pathlib.Path(td), depth, w, h, frames, preset, extra)
        np.savez_compressed(os.path.join(HERE, "golden", "quality_%d.npz" % depth), ctu=ctu,
                            ssim=np.array([p["ssim"] for p in pics]), psnr=np.stack([p["psnr"] for p in pics]),
                            **{"%s%d_%d" % (k, i, c): p[k][c] for i, p in enumerate(pics) for k in ("src", "rec") for c in range(3)})
        print(depth, [p["ssim"] for p in pics])
