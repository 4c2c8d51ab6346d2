"""include/x265hip_ctx.h on the GPU: a C++ host (x265-mod-by-patman_amd/examples/batch_host.cpp: dlopen + the C API, no Python / torch in that
process) uploads unpadded pictures, runs the pyramid batch and reads MVs and coefficients back -- identical to the Python pipeline fed with
the same pictures padded on the host, and (through it) to the oracle."""
import os
import subprocess

import numpy as np
import pytest

import x265hip  # noqa: F401
from x265hip_pkg.binding import HERE, lib_path
from x265hip_pkg.frame import ME_RESULT, mvcost_row
from x265hip_pkg.pipeline import FramePipeline, LEVELS
from x265hip_pkg.synth import frame_pair
from backends import Oracle
from pipeline_check import check_sample

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth,method,subme", [(8, 1, 2), (10, 3, 3)])
def test_cpp_host_drives_the_batch(depth, method, subme, tmp_path):
    W, H, F, qp, merange = 256, 128, 2, 28, 24
    exe = os.path.join(HERE, "build", "batch_host")
    assert os.path.exists(exe), "x265-mod-by-patman_amd/build/batch_host is built by the package Makefile"
    pairs = [frame_pair(W, H, depth, 90 + s, margin=96, max_shift=14)[:2] for s in range(F)]
    raw = np.concatenate([np.concatenate([c[96:96 + H, 96:96 + W].reshape(-1), r[96:96 + H, 96:96 + W].reshape(-1)]) for c, r in pairs])
    inp, outp = str(tmp_path / "in.raw"), str(tmp_path / "out.bin")
    raw.tofile(inp)
    env = dict(os.environ); env.pop("LD_PRELOAD", None)
    r = subprocess.run([exe, lib_path(depth), str(W), str(H), str(F), str(method), str(subme), str(merange), str(qp), inp, outp], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    blob = open(outp, "rb").read()
    # the same pictures through the Python pipeline (planes padded on the host with numpy's edge mode = extendPicBorder)
    pipe = FramePipeline(depth, W, H, F, qp=qp, merange=merange, method=method, subme=subme, tu_log2=5, cost_row=mvcost_row(depth, qp, 1 << 15))
    pipe.upload(pairs); pipe.step(); pipe.torch.cuda.synchronize()
    off = 0
    for lv in LEVELS:
        exp = pipe.results(lv)
        got = np.frombuffer(blob, ME_RESULT, len(exp), off); off += exp.nbytes
        assert np.array_equal(got["mv"], exp["mv"]) and np.array_equal(got["cost"], exp["cost"]) and np.array_equal(got["mvcost"], exp["mvcost"]), "level %d" % lv
    ns = pipe.d_numsig.cpu().numpy().astype(np.uint32)
    co = pipe.d_coeff.cpu().numpy()
    assert np.array_equal(np.frombuffer(blob, np.uint32, len(ns), off), ns); off += ns.nbytes
    assert np.array_equal(np.frombuffer(blob, np.int16, len(co), off), co); off += co.nbytes
    assert off == len(blob)
    assert check_sample(pipe, Oracle(depth), np.random.default_rng(5), per_level=8, n_tu=8) == 40       # ... and that pipeline equals the oracle
