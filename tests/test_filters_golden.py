"""The in-loop filters against committed outputs of the REFERENCE's own classes (tests/golden/filters_*.npz, tests/make_golden_filters.py): the oracle on
the CPU (runs where /root/reference is absent), the HIP entry points on the GPU box."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from depths import GOLDEN_DEPTHS

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle")); sys.path.insert(0, HERE)
from oracle_py import Oracle  # noqa: E402
from deblock_util import I8, U8, run_oracle  # noqa: E402
from test_sao_oracle_vs_ref import sao_apply_oracle, sao_frame_oracle  # noqa: E402


def golden(depth):
    return np.load(os.path.join(HERE, "golden", "filters_%d.npz" % depth))


def deblock_picture(g):
    W, H, ctu, sp, beta, tc, cb, cr, byp = (int(v) for v in g["dbk_params"])
    pic = {k: g["dbk_" + k] for k in U8 + I8 + ("mv0", "mv1", "refPic")}
    pic.update(W=W, H=H, ctu=ctu, slice_p=sp, beta_div2=beta, tc_div2=tc, cb_off=cb, cr_off=cr, bypass=byp, planes=[g["dbk_in%d" % c] for c in range(3)])
    return pic


@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_oracle_matches_golden_filter_outputs(depth):
    g, ora = golden(depth), Oracle(depth)
    pic = deblock_picture(g); pic["depth"] = depth
    out = run_oracle(ora, pic)
    for c in range(3):
        assert np.array_equal(out[c], g["dbk_out%d" % c]), "deblock plane %d" % c
    W, H, ctu = (int(v) for v in g["sao_params"])
    for c in range(3):
        cs = ctu if c == 0 else ctu // 2
        assert np.array_equal(sao_frame_oracle(ora, g["sao_fenc%d" % c], g["sao_rec%d" % c], cs, 0, 0 if c == 0 else 2), g["sao_stats"][c]), "SAO statistics plane %d" % c
        assert np.array_equal(sao_apply_oracle(ora, g["sao_rec%d" % c], cs, g["sao_prm"][c]), g["sao_out%d" % c]), "SAO plane %d" % c


@pytest.mark.gpu
@pytest.mark.parametrize("depth", GOLDEN_DEPTHS)
def test_hip_matches_golden_filter_outputs(depth):
    import x265hip  # noqa: F401
    from x265hip_pkg.frame import FrameApi
    from test_deblock_gpu import hip_deblock
    g, api = golden(depth), FrameApi(depth)
    t = api.torch
    pic = deblock_picture(g); pic["depth"] = depth
    out, _ = hip_deblock(api, pic)
    for c in range(3):
        assert np.array_equal(out[c], g["dbk_out%d" % c]), "deblock plane %d" % c
    W, H, ctu = (int(v) for v in g["sao_params"])
    P = lambda x: C.c_void_p(x.data_ptr())
    for c in range(3):
        cs = ctu if c == 0 else ctu // 2
        fenc, rec = np.ascontiguousarray(g["sao_fenc%d" % c]), np.ascontiguousarray(g["sao_rec%d" % c])
        h, w = rec.shape
        d_f, d_r, d_prm = api.to_device(fenc.reshape(-1)), api.to_device(rec.reshape(-1)), api.to_device(np.ascontiguousarray(g["sao_prm"][c]).astype(np.int32).reshape(-1))
        d_st = t.zeros(g["sao_stats"][c].size, dtype=t.int32, device="cuda"); d_o = t.zeros_like(d_r)
        api.h.check(api.lib.x265hip_sao_stats_frame(api.stream(), P(d_f), P(d_r), C.c_ssize_t(w), w, h, cs, 0, 0 if c == 0 else 2, P(d_st)))
        api.h.check(api.lib.x265hip_sao_apply_frame(api.stream(), P(d_r), P(d_o), C.c_ssize_t(w), w, h, cs, P(d_prm)))
        assert np.array_equal(d_st.cpu().numpy().reshape(g["sao_stats"][c].shape), g["sao_stats"][c]), "SAO statistics plane %d" % c
        assert np.array_equal(d_o.cpu().numpy().view(rec.dtype).reshape(h, w), g["sao_out%d" % c]), "SAO plane %d" % c
