"""Regenerates tests/golden/filters_{8,10}.npz: inputs and the outputs of the REFERENCE's own Deblock and SAO classes (oracle/_ref/x265deblock_*,
oracle/_ref/x265sao_*; needs /root/reference at build time of oracle/_ref) for small pictures -- the vectors the GPU box checks x265hip_deblock_frame,
x265hip_sao_stats_frame and x265hip_sao_apply_frame against where the reference itself is absent.  Data only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
from deblock_util import I8, U8, coded_picture, run_reference  # noqa: E402
from test_sao_oracle_vs_ref import sao_apply_reference_420, sao_frame_pair, sao_frame_reference, sao_params  # noqa: E402

DEBLOCK = {8: (136, 72, 64, 2, False, False), 10: (64, 64, 32, 6, True, True)}          # W, H, ctu, seed, P slice, lossless CUs
SAO = {8: (136, 72, 64), 10: (72, 40, 32)}

if __name__ == "__main__":
    for depth in (8, 10):
        W, H, ctu, seed, sp, bp = DEBLOCK[depth]
        pic = coded_picture(depth, W, H, ctu, seed, sp, bp)
        out = run_reference(pic)
        d = {"dbk_" + k: pic[k] for k in U8 + I8 + ("mv0", "mv1", "refPic")}
        d["dbk_params"] = np.array([W, H, ctu, pic["slice_p"], pic["beta_div2"], pic["tc_div2"], pic["cb_off"], pic["cr_off"], pic["bypass"]], np.int32)
        for c in range(3):
            d["dbk_in%d" % c], d["dbk_out%d" % c] = pic["planes"][c], out[c].astype(pic["planes"][c].dtype)
        W, H, ctu = SAO[depth]
        planes = [sao_frame_pair(depth, W, H, 61 + depth), sao_frame_pair(depth, W // 2, H // 2, 62 + depth), sao_frame_pair(depth, W // 2, H // 2, 63 + depth)]
        n = ((W + ctu - 1) // ctu) * ((H + ctu - 1) // ctu)
        rng = np.random.default_rng(depth)
        prm = np.stack([sao_params(rng, n, depth) for _ in range(3)])
        prm[2, :, 0] = prm[1, :, 0]
        for a in range(n):
            if prm[2, a, 0] == 4:
                prm[2, a, 1] = rng.integers(0, 32); prm[2, a, 2:] = rng.integers(-7, 8, 4)
            elif prm[2, a, 0] >= 0:
                prm[2, a, 2:] = (rng.integers(0, 8), rng.integers(0, 8), -rng.integers(0, 8), -rng.integers(0, 8))
        stats = sao_frame_reference(depth, planes[0][0], planes[0][1], ctu, 0, chroma=planes[1:])
        applied = sao_apply_reference_420(depth, planes, ctu, prm)
        d["sao_params"] = np.array([W, H, ctu], np.int32); d["sao_prm"] = prm; d["sao_stats"] = stats
        for c in range(3):
            d["sao_fenc%d" % c], d["sao_rec%d" % c], d["sao_out%d" % c] = planes[c][0], planes[c][1], applied[c].astype(planes[c][1].dtype)
        np.savez_compressed(os.path.join(HERE, "golden", "filters_%d.npz" % depth), **d)
        print(depth, "written")
