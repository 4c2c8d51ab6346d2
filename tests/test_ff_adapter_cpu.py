"""The host half of the in-loop filter seam without a GPU: integration/filter_adapter.cpp inside the compiled reference encoder (oracle/_ref/x265e2e_8), with
tests/mock_ff_producer.cpp standing in for x265hip_ff_picture -- answered by the oracle's plain-C deblocking filter and SAO statistics (oracle/x265_oracle.c, pinned to the
reference's Deblock / SAO classes in tests/test_deblock_oracle_vs_ref.py, tests/test_sao_oracle_vs_ref.py).  The gather of CUData's arrays, the deferral of a picture's filters to its last row and the replay of
the encoder's row loop behind the call are then all that stands between the plain encoder's bitstream and this one: they must be the same bitstream.
(The device kernels against the same oracle: tests/test_deblock_gpu.py, tests/test_sao_gpu.py, tests/test_ff_host_gpu.py; the GPU producer inside the encoder: tests/test_e2e_ff_gpu.py.)"""
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
    d, built = tmp_path_factory.mktemp("mock_ff"), {}

    def for_depth(depth=8):
        if depth not in built:
            built[depth] = str(d / ("libmock_%d.so" % depth))
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DMOCK_DEPTH=%d" % depth, "-o", built[depth], os.path.join(ROOT, "tests", "mock_ff_producer.cpp"), "-ldl"], check=True)
        return built[depth]
    return for_depth


def encode(mock, tmp_path, name, ff, frames=8, size=(640, 368), env=None, options=(), timeout=240, depth=8):
    outp = str(tmp_path / (name + ".hevc"))
    e = dict(os.environ, X265MOCK_ORACLE_LIB=ORACLE.replace("_8.so", "_%d.so" % depth), X265TME="0", X265TMEGPU="0", X265LAGPU="0", X265FFGPU=str(ff), **(env or {}))
    r = subprocess.run([EXE.replace("_8", "_%d" % depth), mock(depth), str(size[0]), str(size[1]), str(frames), "medium", outp] + list(options), capture_output=True, text=True, env=e, timeout=timeout)
    info = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {}
    info["rc"], info["stderr"] = r.returncode, r.stderr
    if r.returncode == 0:
        info["md5"] = hashlib.md5(open(outp, "rb").read()).hexdigest()
    return info


@pytest.mark.parametrize("size,options", [((640, 368), ()), ((640, 368), ("limit-sao=1",)), ((640, 368), ("sao-non-deblock=1",)), ((640, 368), ("sao=0",)), ((640, 368), ("deblock=2:-2", "bframes=0")),
                                          ((640, 368), ("cu-lossless=1",)), ((640, 368), ("ctu=32",)), ((256, 320), ("slices=4", "wpp=1")), ((320, 384), ("slices=3", "wpp=1", "limit-sao=1"))],
                         ids=lambda v: "+".join(str(x) for x in v) if isinstance(v, tuple) else str(v))
def test_filters_through_the_binding_give_the_plain_encoders_bitstream(mock, tmp_path, size, options):
    plain = encode(mock, tmp_path, "plain", 0, size=size, options=options)
    bound = encode(mock, tmp_path, "bound", 1, size=size, options=options)
    assert plain["rc"] == 0 and bound["rc"] == 0 and "PROTOCOL VIOLATION" not in bound["stderr"], bound["stderr"][-600:]
    assert bound["ff_pictures"] == 8 and bound["ff_cpu_pictures"] == 0 and plain["ff_pictures"] == 0
    assert bound["md5"] == plain["md5"] and bound["bytes"] == plain["bytes"]


@pytest.mark.parametrize("options,band_rows", [(("pools=16", "frame-threads=3"), None), (("pools=16", "frame-threads=4", "wpp=0"), "1"), (("pools=16", "frame-threads=3", "sao=0"), "2"),
                                               (("pools=16", "frame-threads=2", "sao-non-deblock=1"), "3"), (("pools=16", "frame-threads=3", "limit-sao=1", "ctu=32"), None),
                                               (("pools=16", "frame-threads=5", "deblock=0"), "100")],
                         ids=lambda v: "+".join(v) if isinstance(v, tuple) else "rows" + str(v))
def test_frame_threads_filter_in_bands_and_write_the_plain_encoders_bitstream(mock, tmp_path, options, band_rows):
    """Several frame threads (the encoder's default): the next pictures wait for the rows a picture's filters finish (Frame::m_reconRowFlag), so the binding filters in BANDS of CTU
    rows as they arrive (desc.ctuRowFirst / ctuRowCount; the row encoders' early start of a row's deblocking is switched off: FrameFilter::ParallelFilter::processTasks).  The mock
    checks the band protocol (a picture's bands in order, contiguous, every row once; pictures interleave); every picture goes through the binding; the bitstream is the plain
    encoder's under the same threading."""
    env = {"X265_CLI_THREADING": "1"}
    if band_rows:
        env["X265FF_BAND_ROWS"] = band_rows
    size = (640, 704)                                  # eleven CTU rows of 64
    r = encode(mock, tmp_path, "ft", 1, size=size, env=env, options=options)
    p = encode(mock, tmp_path, "ftp", 0, size=size, env={"X265_CLI_THREADING": "1"}, options=options)
    assert r["rc"] == 0 and p["rc"] == 0 and "PROTOCOL VIOLATION" not in r["stderr"], r["stderr"][-600:]
    assert r["frame_threads"] == int([o for o in options if o.startswith("frame-threads=")][0].split("=")[1])
    assert r["ff_pictures"] == 8 and r["ff_cpu_pictures"] == 0
    if band_rows != "100":
        assert r["ff_bands"] > r["ff_pictures"], "the pictures were not cut into bands: %s" % r["ff_bands"]
    else:
        assert r["ff_bands"] == r["ff_pictures"]
    assert r["md5"] == p["md5"] and r["bytes"] == p["bytes"]


def test_bands_with_one_frame_thread_on_request(mock, tmp_path):
    """X265FF_BAND_ROWS with one frame thread forces the band form (default there: the whole picture at its last row): same bitstream, more calls."""
    whole = encode(mock, tmp_path, "w", 1)
    bands = encode(mock, tmp_path, "b", 1, env={"X265FF_BAND_ROWS": "2"})
    wpp = encode(mock, tmp_path, "bw", 1, env={"X265FF_BAND_ROWS": "1"}, options=("wpp=1", "pools=8"))
    plain_wpp = encode(mock, tmp_path, "pw", 0, options=("wpp=1", "pools=8"))
    for r in (whole, bands, wpp):
        assert r["rc"] == 0 and "PROTOCOL VIOLATION" not in r["stderr"], r["stderr"][-600:]
        assert r["ff_pictures"] == 8
    assert whole["ff_bands"] == 8 and bands["ff_bands"] == 8 * 3 and whole["md5"] == bands["md5"]      # 368 lines = six CTU rows: three bands of two
    assert wpp["ff_bands"] == 8 * 6 and wpp["md5"] == plain_wpp["md5"]


@pytest.mark.parametrize("size,options,band_rows", [((256, 320), ("pools=16", "frame-threads=3", "slices=2"), None), ((320, 704), ("pools=24", "frame-threads=4", "slices=3"), "2"),
                                                    ((256, 512), ("pools=16", "frame-threads=2", "slices=4", "limit-sao=1"), "1")],
                         ids=lambda v: "+".join(v) if isinstance(v, tuple) and isinstance(v[0], str) else str(v))
def test_slices_with_frame_threads_filter_in_bands_inside_each_slice(mock, tmp_path, size, options, band_rows):
    """--slices together with frame threads: the slices of a picture finish their rows side by side (each slice a chain of filter rows of its own) while pictures overlap.  A band
    stays inside its slice (a slice's top edge is not filtered: m_cuAbove == NULL), the slices' bands of a picture interleave; the mock checks that order, and the bitstream is the
    plain encoder's"""
    env = {"X265_CLI_THREADING": "1"}
    if band_rows:
        env["X265FF_BAND_ROWS"] = band_rows
    r = encode(mock, tmp_path, "sl", 1, size=size, env=env, options=options)
    p = encode(mock, tmp_path, "slp", 0, size=size, env={"X265_CLI_THREADING": "1"}, options=options)
    assert r["rc"] == 0 and p["rc"] == 0 and "PROTOCOL VIOLATION" not in r["stderr"], r["stderr"][-600:]
    assert r["ff_pictures"] == 8 and r["ff_cpu_pictures"] == 0 and r["ff_bands"] >= 8 * int([o for o in options if o.startswith("slices=")][0].split("=")[1])
    assert r["md5"] == p["md5"] and r["bytes"] == p["bytes"]


def test_a_failing_filter_call_ends_the_encode_at_once(mock, tmp_path):
    r = encode(mock, tmp_path, "f", 1, env={"X265MOCK_FAIL_AT": "2"}, timeout=60)
    assert r["rc"] == 3 and "fails on request" in r["stderr"] and "filter_adapter" in r["stderr"]


@pytest.mark.skipif(not os.path.exists(EXE.replace("_8", "_10")), reason="the 10-bit encoder is not built")
def test_ten_bit_encoder(mock, tmp_path):
    plain = encode(mock, tmp_path, "plain10", 0, options=("limit-sao=1",), depth=10)
    bound = encode(mock, tmp_path, "bound10", 1, options=("limit-sao=1",), depth=10)
    assert plain["rc"] == 0 and bound["rc"] == 0 and "PROTOCOL VIOLATION" not in bound["stderr"], bound["stderr"][-600:]
    assert bound["ff_pictures"] == 8 and bound["md5"] == plain["md5"]


@pytest.mark.parametrize("csp,options,env", [("i422", (), {}), ("i444", (), {}), ("i422", ("pools=16", "frame-threads=3"), {"X265_CLI_THREADING": "1"}),
                                              ("i444", ("pools=16", "frame-threads=3", "limit-sao=1"), {"X265_CLI_THREADING": "1", "X265FF_BAND_ROWS": "2"})],
                         ids=lambda v: v if isinstance(v, str) else "+".join(v) if isinstance(v, tuple) else "env%d" % len(v))
def test_other_chroma_formats_through_the_binding(mock, tmp_path, csp, options, env):
    """4:2:2 and 4:4:4 encodes (Cb / Cr half as wide and as high as luma / full size; chroma edges on their own 8-sample grid, chroma CTUs 32 x 64 / 64 x 64): whole pictures with
    one frame thread, bands under frame threads -- the plain encoder's bitstream, every picture through the producer"""
    e = dict(env, X265_CSP=csp)
    size = (640, 704) if options else (320, 384)          # (one frame thread is one thread: a smaller picture)
    plain = encode(mock, tmp_path, "plain", 0, size=size, env=e, options=options)
    bound = encode(mock, tmp_path, "bound", 1, size=size, env=e, options=options)
    assert plain["rc"] == 0 and bound["rc"] == 0 and "PROTOCOL VIOLATION" not in bound["stderr"], bound["stderr"][-600:]
    assert bound["ff_pictures"] == 8 and bound["ff_cpu_pictures"] == 0
    assert bound["md5"] == plain["md5"] and bound["bytes"] == plain["bytes"]
    other = encode(mock, tmp_path, "other", 0, size=size, options=options, env=env)      # (the format really is another encode)
    assert other["md5"] != plain["md5"]
