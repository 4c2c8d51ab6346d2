"""examples/multi_gpu_host.cpp on the GPU: two workers (threads, then forked processes) with a context each -- here both on device 0, the only one of the box -- take the
same pictures and must leave the same bytes as examples/batch_host.cpp (whose output tests/test_ctx_gpu.py pins to the Python pipeline and the oracle); a device that does
not exist fails the whole job with the device named."""
import json
import os
import subprocess

import numpy as np
import pytest

import x265hip  # noqa: F401
from x265hip_pkg.binding import HERE, lib_path
from x265hip_pkg.synth import frame_pair

pytestmark = pytest.mark.gpu


def fnv1a(blob):
    h = 1469598103934665603
    for b in blob:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def run(args):
    env = dict(os.environ); env.pop("LD_PRELOAD", None)
    return subprocess.run(args, capture_output=True, text=True, env=env, timeout=600)


@pytest.mark.parametrize("depth,method,subme", [(8, 1, 2), (10, 3, 3)])
def test_workers_agree_with_each_other_and_with_the_single_device_host(depth, method, subme, tmp_path):
    W, H, F, qp, merange = 256, 128, 2, 28, 24
    pairs = [frame_pair(W, H, depth, 90 + s, margin=96, max_shift=14)[:2] for s in range(F)]
    raw = np.concatenate([np.concatenate([c[96:96 + H, 96:96 + W].reshape(-1), r[96:96 + H, 96:96 + W].reshape(-1)]) for c, r in pairs])
    inp, outp = str(tmp_path / "in.raw"), str(tmp_path / "out.bin")
    raw.tofile(inp)
    r = run([os.path.join(HERE, "build", "batch_host"), lib_path(depth), str(W), str(H), str(F), str(method), str(subme), str(merange), str(qp), inp, outp])
    assert r.returncode == 0, r.stdout + r.stderr
    want = "%016x" % fnv1a(open(outp, "rb").read())
    common = [os.path.join(HERE, "build", "multi_gpu_host"), lib_path(depth), "--devices", "0,0", "--width", str(W), "--height", str(H), "--frames", str(F), "--steps", "2", "--warmup", "1",
              "--method", str(method), "--subme", str(subme), "--merange", str(merange), "--qp", str(qp), "--input", inp]
    for extra in ([], ["--procs"], ["--streams", "1", "--inner", "2"]):
        r = run(common + extra)
        assert r.returncode == 0, r.stdout + r.stderr
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["n_gpus"] == 2 and d["devices"] == [0, 0] and d["same_frames"] is True
        assert d["digests"] == [want, want], (extra, d["digests"], want)
        assert d["value"] > 0 and len(d["per_device_ms_per_step"]) == 2


def test_own_pictures_per_worker_and_a_missing_device():
    exe = os.path.join(HERE, "build", "multi_gpu_host")
    r = run([exe, lib_path(10), "--devices", "0,0", "--width", "256", "--height", "128", "--frames", "2", "--steps", "2", "--warmup", "1", "--merange", "24"])
    assert r.returncode == 0, r.stdout + r.stderr
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["digests"][0] != d["digests"][1]                      # worker k searched ITS pictures (global indices k * frames ...)
    r2 = run([exe, lib_path(10), "--devices", "0,0", "--width", "256", "--height", "128", "--frames", "2", "--steps", "2", "--warmup", "1", "--merange", "24", "--same-frames"])
    d2 = json.loads(r2.stdout.strip().splitlines()[-1])
    assert d2["digests"][0] == d2["digests"][1] == d["digests"][0]  # ... and worker 0's are pictures 0 .. frames - 1 either way
    r = run([exe, lib_path(10), "--devices", "0,63", "--width", "128", "--height", "128", "--frames", "1", "--steps", "1"])
    assert r.returncode == 3 and "x265hip_ctx_create on device 63" in r.stderr and not r.stdout.strip()
