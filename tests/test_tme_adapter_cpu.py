"""The host half of the ThreadedME seam without a GPU: integration/tme_adapter.cpp inside the compiled reference encoder (oracle/_ref/x265tmegpu_8), with tests/mock_tme_producer.cpp
standing in for the library's producer.  The mock checks the protocol of include/x265hip_ctx.h on every call (one call at a time, a picture's bands in order / contiguous / once,
the declared final rows of every reference cover what the band's searches reach) and answers with records that are a pure function of what the adapter handed over for each CTU --
the harvested qps, collocated neighbours and medians, the reference tables, the reference planes' rows inside the CTU's window.  So the bitstream must not depend on how a picture
was cut into bands or how the threads interleaved; state handed over before it was final would move it.
(The searches themselves -- x265hip_tme_picture against the encoder's own producers, same bitstream -- are tests/test_e2e_tme_gpu.py, on the GPU.)"""
import hashlib
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "x265tmegpu_8")
REAL = os.path.join(ROOT, "x265-mod-by-patman_amd", "libx265hip_8.so")

pytestmark = pytest.mark.skipif(not (os.path.exists(EXE) and os.path.exists(REAL)), reason="oracle/_ref/x265tmegpu_8 or libx265hip_8.so not built")


@pytest.fixture(scope="module")
def mock(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("mock") / "libmock_tme.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tests", "mock_tme_producer.cpp"), "-ldl"], check=True)
    return out


def encode(mock, tmp_path, name, frames=10, size=(1280, 720), env=None, options=(), timeout=240, preset="medium"):
    outp = str(tmp_path / (name + ".hevc"))
    e = dict(os.environ, X265MOCK_REAL_LIB=REAL, **(env or {}))
    r = subprocess.run([EXE, mock, str(size[0]), str(size[1]), str(frames), preset, outp] + list(options), capture_output=True, text=True, env=e, timeout=timeout)
    info = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {}
    info["rc"], info["stderr"] = r.returncode, r.stderr
    if r.returncode == 0:
        info["md5"] = hashlib.md5(open(outp, "rb").read()).hexdigest()
    return info


THREADS = ("pools=48", "frame-threads=5")     # (the encoder switches --threaded-me off below 32 pool threads)


def test_bands_under_frame_threads_follow_the_protocol_and_do_not_move_the_bitstream(mock, tmp_path):
    """Five frame threads + WPP: every ready row at once, the default (a band waits for half the picture's rows), large bands with a long wait, helpers on the band's
    host passes, and a band harvested while the one before it is in its call (X265TME_AHEAD): the mock sees no violation, every picture goes through it, and the bitstreams are one."""
    runs = {}
    for name, env in (("every_row", {"X265TME_MIN_ROWS": "1", "X265TME_WAIT_US": "0"}), ("default", {}), ("large", {"X265TME_MIN_ROWS": "100", "X265TME_WAIT_US": "30000"}),
                      ("helpers", {"X265TME_HELP": "1"}), ("ahead", {"X265TME_AHEAD": "1"}), ("ahead_every_row_helpers", {"X265TME_AHEAD": "1", "X265TME_MIN_ROWS": "1", "X265TME_WAIT_US": "0", "X265TME_HELP": "1"})):
        r = encode(mock, tmp_path, name, env=dict(env, X265_CLI_THREADING="1"), options=THREADS)
        assert r["rc"] == 0 and "PROTOCOL VIOLATION" not in r["stderr"], r["stderr"][-600:]
        assert r["frame_threads"] == 5 and r["wpp"] == 1 and r["threaded_me"] == 1
        assert r["gpu_pictures"] == 9 and r["gpu_bands"] >= r["gpu_pictures"]          # 10 frames, the first is intra
        runs[name] = r
    assert len({r["md5"] for r in runs.values()}) == 1, {k: (v["md5"], v["gpu_bands"]) for k, v in runs.items()}
    assert runs["every_row"]["gpu_bands"] > runs["large"]["gpu_bands"]               # the policies really cut the pictures differently


def test_one_frame_thread_hands_over_whole_pictures(mock, tmp_path):
    """One frame thread, no WPP (the driver's default threading): a call per picture, complete references (no valid-row counts), the waiting workers help with the host passes."""
    a = encode(mock, tmp_path, "a", frames=6, size=(832, 480))
    b = encode(mock, tmp_path, "b", frames=6, size=(832, 480), env={"X265TME_HELP": "0"})
    for r in (a, b):
        assert r["rc"] == 0 and "PROTOCOL VIOLATION" not in r["stderr"], r["stderr"][-600:]
        assert r["frame_threads"] == 1 and r["gpu_pictures"] == 5 and r["gpu_bands"] == 5
        assert "5 calls, 0 of them bands" in r["stderr"]
    assert a["md5"] == b["md5"]


def test_a_failing_producer_call_ends_the_encode_at_once(mock, tmp_path):
    """The third call fails: the adapter prints the producer's message and leaves with status 3 -- with a pool of running threads, without hanging in exit handlers."""
    r = encode(mock, tmp_path, "f", env={"X265_CLI_THREADING": "1", "X265MOCK_FAIL_AT": "3"}, options=THREADS, timeout=60)
    assert r["rc"] == 3
    assert "fails on request" in r["stderr"] and "x265hip_tme_picture" in r["stderr"]


def test_weighted_references_and_the_rectangular_schedule_under_frame_threads(mock, tmp_path):
    """A fade (weighted prediction: the weighted plane of (picture, list, reference) goes over with the rows MotionReference::applyWeight has finished) and preset slow's
    schedule (255 entries per CTU) under frame threads, each cut into bands two ways -- one of them with a band harvested while the one before it is in its call."""
    fade = [encode(mock, tmp_path, "fade%d" % i, env=dict(e, X265_CLI_THREADING="1", X265TME_FADE="1"), options=("pools=48", "frame-threads=4"))
            for i, e in enumerate(({}, {"X265TME_MIN_ROWS": "1", "X265TME_WAIT_US": "0", "X265TME_AHEAD": "1"}))]
    for r in fade:
        assert r["rc"] == 0 and "PROTOCOL VIOLATION" not in r["stderr"], r["stderr"][-600:]
        assert r["weighted_refs"] > 0 and r["gpu_pictures"] == 9
    assert fade[0]["md5"] == fade[1]["md5"]
    slow = [encode(mock, tmp_path, "slow%d" % i, frames=8, size=(832, 480), env=dict(e, X265_CLI_THREADING="1"), options=("pools=48", "frame-threads=3"), preset="slow")
            for i, e in enumerate(({}, {"X265TME_MIN_ROWS": "1", "X265TME_AHEAD": "1"}))]
    for r in slow:
        assert r["rc"] == 0 and "PROTOCOL VIOLATION" not in r["stderr"], r["stderr"][-600:]
    assert slow[0]["md5"] == slow[1]["md5"]


def test_what_the_producer_does_not_take_under_frame_threads_goes_back_to_the_encoders_own_body(mock, tmp_path):
    """--slices with several frame threads (the reference's ThreadedME workers read Search::m_sliceMinY / m_sliceMaxY, which nothing initialises on THEIR Analysis objects:
    frameencoder.cpp:1624 sets the frame encoder's only -- no defined behaviour to reproduce) and --me sea (the producer keeps no SEA integral planes; under any threading): the
    binding hands every CTU to the encoder's own body, says so once on stderr, and the encode writes what it writes without the binding."""
    for name, opts in (("slices", ("slices=2",)), ("sea", ("me=sea",))):
        size = (416, 240) if name == "sea" else (640, 368)          # (the exhaustive-style SEA search is slow on the CPU)
        with_binding = encode(mock, tmp_path, name + "_b", frames=5, size=size, env={"X265_CLI_THREADING": "1"}, options=THREADS + opts)
        without = encode(mock, tmp_path, name + "_c", frames=5, size=size, env={"X265_CLI_THREADING": "1", "X265TMEGPU": "0"}, options=THREADS + opts)
        assert with_binding["rc"] == 0 and without["rc"] == 0, with_binding["stderr"][-400:]
        assert "the encoder's own ThreadedME producer runs" in with_binding["stderr"] and with_binding["gpu_pictures"] == 0
        assert with_binding["threaded_me"] == 1 and with_binding["frame_threads"] == 5
        assert with_binding["md5"] == without["md5"]
    one = encode(mock, tmp_path, "sea1_b", frames=4, size=(256, 192), options=("me=sea",))
    one_cpu = encode(mock, tmp_path, "sea1_c", frames=4, size=(256, 192), env={"X265TMEGPU": "0"}, options=("me=sea",))
    assert one["rc"] == 0 and one["gpu_pictures"] == 0 and one["frame_threads"] == 1 and "the encoder's own ThreadedME producer runs" in one["stderr"] and one["md5"] == one_cpu["md5"]


def test_intra_refresh_hands_the_window_limit_over(mock, tmp_path):
    """--intra-refresh: P pictures whose first reference has not finished its sweep carry pirStartCol / pirSafeX (Search::setSearchRange, search.cpp:4987-4996); the mock checks
    their range on every call and folds the limit into the records of the CTU columns it applies to.  One and five frame threads, two band policies: no violation, the pictures
    go through the producer (no fallback), one bitstream per threading."""
    one = encode(mock, tmp_path, "ir1", frames=8, size=(832, 480), options=("intra-refresh=1", "keyint=6", "bframes=0"))
    assert one["rc"] == 0 and "PROTOCOL VIOLATION" not in one["stderr"], one["stderr"][-600:]
    assert one["gpu_pictures"] == 7 and "own ThreadedME producer" not in one["stderr"]
    assert "pictures with an intra-refresh window limit" in one["stderr"]
    runs = [encode(mock, tmp_path, "ir5_%d" % i, frames=8, size=(832, 480), env=dict(e, X265_CLI_THREADING="1"), options=("pools=48", "frame-threads=4", "intra-refresh=1", "keyint=6", "bframes=0"))
            for i, e in enumerate(({"X265TME_MIN_ROWS": "1", "X265TME_WAIT_US": "0"}, {"X265TME_AHEAD": "1"}))]
    for r in runs:
        assert r["rc"] == 0 and "PROTOCOL VIOLATION" not in r["stderr"], r["stderr"][-600:]
        assert r["gpu_pictures"] == 7 and "pictures with an intra-refresh window limit" in r["stderr"]
    assert runs[0]["md5"] == runs[1]["md5"]


@pytest.mark.parametrize("options", [("bframes=0",), ("ref=5", "weightb=1"), ("ctu=32",), ("merange=24",), ("merange=120",), ("rect=1", "amp=1"), ("keyint=5", "min-keyint=5", "b-pyramid=0")],
                         ids=lambda o: "+".join(o))
def test_encoder_options_that_change_the_row_lag_the_schedule_or_the_reference_lists(mock, tmp_path, options):
    """Four frame threads; the search range sets how many reference rows a band waits for (FrameEncoder::m_refLagRows), the CTU size the rows themselves, rect / amp the schedule,
    the GOP options which pictures reference which: every ready row at once against bands harvested ahead -- no violation, one bitstream."""
    runs = [encode(mock, tmp_path, "o%d" % i, frames=8, size=(832, 480), env=dict(e, X265_CLI_THREADING="1"), options=("pools=48", "frame-threads=4") + tuple(options))
            for i, e in enumerate(({"X265TME_MIN_ROWS": "1", "X265TME_WAIT_US": "0"}, {"X265TME_AHEAD": "1"}))]
    for r in runs:
        assert r["rc"] == 0 and "PROTOCOL VIOLATION" not in r["stderr"], r["stderr"][-600:]
        assert r["gpu_pictures"] > 0 and r["gpu_bands"] >= r["gpu_pictures"]
    assert runs[0]["md5"] == runs[1]["md5"]


def test_slices_with_one_frame_thread(mock, tmp_path):
    """--slices with one frame thread (the first / last-row flags initCTU gets per slice, threadedme.cpp:298-308): whole pictures, helpers on and off, one bitstream."""
    a = encode(mock, tmp_path, "sa", frames=6, size=(256, 256), options=("slices=2", "wpp=1"))
    b = encode(mock, tmp_path, "sb", frames=6, size=(256, 256), options=("slices=2", "wpp=1"), env={"X265TME_HELP": "0"})
    for r in (a, b):
        assert r["rc"] == 0 and "PROTOCOL VIOLATION" not in r["stderr"], r["stderr"][-600:]
        assert r["gpu_pictures"] == 5
    assert a["md5"] == b["md5"]


@pytest.mark.skipif(not (os.path.exists(EXE.replace("_8", "_10")) and os.path.exists(REAL.replace("_8.so", "_10.so"))), reason="the 10-bit encoder / library is not built")
def test_ten_bit_encoder_under_frame_threads(tmp_path):
    """The 10-bit encoder (two-byte pixels in every plane the binding hands over) with the mock built for it: two band policies, one bitstream."""
    mock10 = str(tmp_path / "libmock_tme_10.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DMOCK_PIXEL_BYTES=2", "-o", mock10, os.path.join(ROOT, "tests", "mock_tme_producer.cpp"), "-ldl"], check=True)
    runs = []
    for i, e in enumerate(({"X265TME_MIN_ROWS": "1", "X265TME_WAIT_US": "0"}, {"X265TME_AHEAD": "1"})):
        outp = str(tmp_path / ("t%d.hevc" % i))
        env = dict(os.environ, X265MOCK_REAL_LIB=REAL.replace("_8.so", "_10.so"), X265_CLI_THREADING="1", X265TME_FADE="1", **e)
        r = subprocess.run([EXE.replace("_8", "_10"), mock10, "832", "480", "8", "medium", outp, "pools=48", "frame-threads=4"], capture_output=True, text=True, env=env, timeout=240)
        assert r.returncode == 0 and "PROTOCOL VIOLATION" not in r.stderr, r.stderr[-600:]
        info = json.loads(r.stdout.strip().splitlines()[-1])
        assert info["gpu_pictures"] == 7 and info["frame_threads"] == 4
        runs.append(hashlib.md5(open(outp, "rb").read()).hexdigest())
    assert runs[0] == runs[1]
