"""The host half of the lookahead seam without a GPU: integration/lookahead_adapter.cpp inside the compiled reference encoder (oracle/_ref/x265e2e_8), with
tests/mock_la_producer.cpp standing in for the library's lookahead producer -- answered by the oracle's plain-C lookahead (oracle/x265_oracle_la.c, pinned to the reference's own
in tests/test_lookahead_oracle_vs_ref.py).  What the binding reads out of the encoder's Lowres state, the waves it cuts a CostEstimateGroup::finishBatch queue into, the weighted
copies, the cached list searches and the cuTree step are then all that stands between the plain encoder's bitstream and this one: they must be the same bitstream.
(The device kernels against the same oracle: tests/test_lookahead_gpu.py; the GPU producer inside the encoder: tests/test_e2e_la_gpu.py.)"""
import hashlib
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "x265e2e_8")
ORACLE = os.path.join(ROOT, "oracle", "libx265oracle_me_8.so")

pytestmark = pytest.mark.skipif(not (os.path.exists(EXE) and os.path.exists(ORACLE)), reason="oracle/_ref/x265e2e_8 or the oracle library not built")


@pytest.fixture(scope="module")
def mock(tmp_path_factory):
    """mock(depth) -> the mock library for the encoder of that bit depth (built on first use)"""
    d, built = tmp_path_factory.mktemp("mock_la"), {}

    def for_depth(depth=8):
        if depth not in built:
            built[depth] = str(d / ("libmock_%d.so" % depth))
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DMOCK_DEPTH=%d" % depth, "-o", built[depth], os.path.join(ROOT, "tests", "mock_la_producer.cpp"), "-ldl"], check=True)
        return built[depth]
    return for_depth


def encode(mock, tmp_path, name, la, frames=14, size=(640, 368), env=None, options=(), timeout=240, depth=8):
    outp = str(tmp_path / (name + ".hevc"))
    e = dict(os.environ, X265MOCK_ORACLE_LIB=ORACLE.replace("_8.so", "_%d.so" % depth), X265TME="0", X265TMEGPU="0", X265FFGPU="0", X265LAGPU=str(la), **(env or {}))
    r = subprocess.run([EXE.replace("_8", "_%d" % depth), mock(depth), str(size[0]), str(size[1]), str(frames), "medium", outp] + list(options), capture_output=True, text=True, env=e, timeout=timeout)
    info = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {}
    info["rc"], info["stderr"] = r.returncode, r.stderr
    if r.returncode == 0:
        info["md5"] = hashlib.md5(open(outp, "rb").read()).hexdigest()
    return info


POOL = ("pools=48", "frame-threads=3")


@pytest.mark.parametrize("env,options", [({}, ()),                                                       # one frame thread, no pool: one estimate per call
                                         ({"X265_CLI_THREADING": "1"}, POOL),                            # the batch binding: finishBatch queues in waves
                                         ({"X265_CLI_THREADING": "1", "X265TME_FADE": "1"}, POOL),       # weighted list-0 copies in a batch
                                         ({"X265TME_FADE": "1"}, ()),
                                         ({"X265_CLI_THREADING": "1"}, POOL + ("hme=1",)),
                                         ({"X265_CLI_THREADING": "1"}, POOL + ("bframes=8", "b-adapt=2")),
                                         ({"X265_CLI_THREADING": "1"}, POOL + ("qg-size=8", "lookahead-slices=4")),
                                         ({"X265_CLI_THREADING": "1", "X265LA_BATCH": "0"}, POOL)],      # the one-estimate binding under a pool: concurrent callers
                         ids=["serial", "batches", "batches-weighted", "serial-weighted", "hme", "bframes8", "qg8-slices", "pool-unbatched"])
def test_lookahead_costs_through_the_binding_give_the_plain_encoders_bitstream(mock, tmp_path, env, options):
    plain = encode(mock, tmp_path, "plain", 0, env=env, options=options)
    bound = encode(mock, tmp_path, "bound", 1, env=env, options=options)
    assert plain["rc"] == 0 and bound["rc"] == 0 and "PROTOCOL VIOLATION" not in bound["stderr"], bound["stderr"][-600:]
    assert bound["la_intra_pictures"] == 14 and bound["la_estimates"] > 0 and bound["la_cpu_estimates"] == 0
    if "X265TME_FADE" in env:
        assert bound["la_weighted"] > 0
    if env.get("X265_CLI_THREADING") and "X265LA_BATCH" not in env:
        assert bound["la_batches"] > 0 and bound["la_batch_calls"] >= bound["la_batches"]
    assert plain["la_estimates"] == 0
    assert bound["md5"] == plain["md5"] and bound["bytes"] == plain["bytes"]


def test_a_failing_estimate_call_ends_the_encode_at_once(mock, tmp_path):
    r = encode(mock, tmp_path, "f", 1, env={"X265_CLI_THREADING": "1", "X265MOCK_FAIL_AT": "2"}, options=POOL, timeout=60)
    assert r["rc"] == 3 and "fails on request" in r["stderr"] and "lookahead_adapter" in r["stderr"]


@pytest.mark.skipif(not os.path.exists(EXE.replace("_8", "_10")), reason="the 10-bit encoder is not built")
def test_ten_bit_encoder_batched_and_weighted(mock, tmp_path):
    env, opts = {"X265_CLI_THREADING": "1", "X265TME_FADE": "1"}, POOL
    plain = encode(mock, tmp_path, "plain10", 0, env=env, options=opts, depth=10)
    bound = encode(mock, tmp_path, "bound10", 1, env=env, options=opts, depth=10)
    assert plain["rc"] == 0 and bound["rc"] == 0 and "PROTOCOL VIOLATION" not in bound["stderr"], bound["stderr"][-600:]
    assert bound["la_estimates"] > 0 and bound["la_weighted"] > 0 and bound["la_batches"] > 0 and bound["md5"] == plain["md5"]
