"""The three bindings together inside the compiled reference encoder (oracle/_ref/x265e2e_8), no GPU: one library made of the three mocks (tests/mock_{tme,la,ff}_producer.cpp).
The lookahead and filter mocks answer through the oracle (exact), the ThreadedME mock with a pure function of its inputs: switching the lookahead and filter bindings ON beside
the ThreadedME binding must therefore not move the bitstream -- under four frame threads + WPP (ThreadedME and the filters both in bands of CTU rows) and with one frame thread
(whole pictures).  What this adds to the per-binding tests: three contexts in one process, the bindings' locks and thread pools side by side."""
import hashlib
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "x265e2e_8")
ORACLE = os.path.join(ROOT, "oracle", "libx265oracle_me_8.so")
REAL = os.path.join(ROOT, "x265-mod-by-patman_amd", "libx265hip_8.so")

pytestmark = pytest.mark.skipif(not (os.path.exists(EXE) and os.path.exists(ORACLE) and os.path.exists(REAL)), reason="oracle/_ref/x265e2e_8, the oracle or libx265hip_8.so not built")


@pytest.fixture(scope="module")
def mock(tmp_path_factory):
    d = tmp_path_factory.mktemp("mock_all")
    objs = []
    for name, flags in (("tme", []), ("la", ["-DMOCK_NO_COMMON"]), ("ff", ["-DMOCK_NO_COMMON"])):      # one copy of x265hip_ctx_create / _last_error: the ThreadedME mock's
        o = str(d / (name + ".o"))
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-c", "-o", o, os.path.join(ROOT, "tests", "mock_%s_producer.cpp" % name)] + flags, check=True)
        objs.append(o)
    out = str(d / "libmock_all.so")
    subprocess.run(["g++", "-shared", "-o", out] + objs + ["-ldl"], check=True)
    return out


def encode(mock, tmp_path, name, la_ff, env, options):
    outp = str(tmp_path / (name + ".hevc"))
    e = dict(os.environ, X265MOCK_REAL_LIB=REAL, X265MOCK_ORACLE_LIB=ORACLE, X265TMEGPU="1", X265LAGPU=str(la_ff), X265FFGPU=str(la_ff), **env)
    r = subprocess.run([EXE, mock, "640", "368", "12", "medium", outp] + list(options), capture_output=True, text=True, env=e, timeout=300)
    assert r.returncode == 0 and "PROTOCOL VIOLATION" not in r.stderr, r.stderr[-600:]
    info = json.loads(r.stdout.strip().splitlines()[-1])
    info["md5"] = hashlib.md5(open(outp, "rb").read()).hexdigest()
    return info


@pytest.mark.parametrize("env,options,frame_threads", [({"X265_CLI_THREADING": "1"}, ("pools=48", "frame-threads=4"), 4), ({}, (), 1)], ids=["frame-threads", "one-frame-thread"])
def test_lookahead_and_filter_bindings_beside_the_threaded_me_binding(mock, tmp_path, env, options, frame_threads):
    alone = encode(mock, tmp_path, "alone", 0, env, options)
    allof = encode(mock, tmp_path, "all", 1, env, options)
    assert alone["frame_threads"] == frame_threads and alone["gpu_pictures"] == 11 and alone["la_estimates"] == 0 and alone["ff_pictures"] == 0
    assert allof["gpu_pictures"] == 11 and allof["la_estimates"] > 0 and allof["la_cpu_estimates"] == 0
    assert (allof["ff_pictures"], allof["ff_cpu_pictures"]) == (12, 0)
    assert allof["ff_bands"] == 12 if frame_threads == 1 else allof["ff_bands"] > 12      # frame threads: the filters go through their producer in bands of CTU rows
    assert allof["md5"] == alone["md5"]
