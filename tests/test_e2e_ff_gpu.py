"""SURVEY 8(f4) end to end: the reference encoder whose deblocked pictures and SAO statistics come from libx265hip (the binding integration/filter_adapter.cpp:
FrameFilter::processRow defers the rows of a picture to its last one, one x265hip_ff_picture call deblocks the picture and collects the statistics of every CTU, then the
encoder's own row loop decides and applies SAO with Deblock::deblockCTU a no-op and SAO::calcSaoStatsCTU a table look-up) must write the bitstream it writes with its own
filters: the reconstructed pictures are the references of the following pictures and the SAO parameters are coded syntax, so a differing sample or statistic shows."""
import hashlib
import json
import os
import subprocess

import pytest

import x265hip

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def encode(depth, ff, args, out, la=False, tme=False, defer_only=False, tme_gpu=None, extra_env=None):
    exe = os.path.join(ROOT, "oracle", "_ref", "x265e2e_%d" % depth)
    if not os.path.exists(exe):
        pytest.skip("no oracle/_ref/x265e2e_%d (built where the reference is present)" % depth)
    env = dict(os.environ, X265FFGPU="1" if ff else "0", X265LAGPU="1" if la else "0", X265TME="1" if tme else "0", X265TMEGPU="1" if (tme if tme_gpu is None else tme_gpu) else "0", MALLOC_PERTURB_="85")
    env.pop("X265FF_DEFER_ONLY", None)
    if defer_only:
        env["X265FF_DEFER_ONLY"] = "1"
    env.update(extra_env or {})
    for a in [a for a in args if a.startswith("csp=")]:          # (the driver takes the clip's chroma format from the environment)
        env["X265_CSP"] = a.split("=")[1]
    args = [a for a in args if not a.startswith("csp=")]
    r = subprocess.run([exe, x265hip.lib_path(depth)] + args[:4] + [out] + args[4:], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), hashlib.md5(open(out, "rb").read()).hexdigest()


CONFIGS = [(8, ["256", "192", "10", "medium"]),                                           # P and B pictures, SAO, deblocking: preset defaults
           (10, ["256", "192", "8", "slow"]),
           (8, ["200", "120", "8", "medium", "bframes=2"]),                               # a picture that is no CTU multiple (partial CTUs at the right and the bottom)
           (8, ["256", "192", "8", "medium", "ctu=32", "min-cu-size=8"]),
           (8, ["192", "128", "8", "fast", "ctu=16", "min-cu-size=8"]),
           (8, ["256", "192", "8", "medium", "sao=0"]),                                   # deblocking alone
           (8, ["256", "192", "8", "medium", "deblock=0"]),                               # SAO statistics on the undeblocked picture
           (8, ["256", "192", "8", "medium", "deblock=-2:3", "cbqpoffs=3", "crqpoffs=-4"]),      # slice-level offsets, chroma QP offsets
           (8, ["256", "192", "8", "medium", "sao-non-deblock=1"]),                       # the statistics leave out other border widths; the pre-deblock sums stay with the encoder
           (8, ["256", "192", "6", "medium", "cu-lossless=1"]),                           # lossless CUs are not filtered (tqBypass)
           (8, ["256", "192", "8", "medium", "wpp=1"]),                                   # filter rows as wavefront jobs
           (10, ["320", "192", "8", "veryfast", "qp=40"]),                                # strong filtering
           (8, ["256", "192", "8", "ultrafast", "keyint=1"]),                             # intra pictures only: boundary strength 2 everywhere
           (8, ["256", "192", "6", "medium", "slices=2", "wpp=1"]),                       # --slices: rows finish in any order; no filtering / no SAO neighbours across a slice's top edge
           (10, ["320", "384", "6", "fast", "slices=3", "wpp=1", "bframes=2"]),
           (8, ["256", "320", "6", "medium", "slices=4", "wpp=1"]),                       # (with sao-non-deblock=1 on top the reference's own CPU encode never finishes)
           (8, ["256", "192", "8", "medium", "limit-sao=1"]),                              # --limit-sao: the diagonal classes only where the reference collects them
           (10, ["256", "192", "8", "slow", "limit-sao=1", "bframes=3"]),
           (8, ["256", "192", "8", "medium", "csp=i422"]),                                # 4:2:2: chroma planes half as wide, chroma CTUs 32 x 64
           (10, ["256", "192", "6", "medium", "csp=i444", "bframes=2"]),                  # 4:4:4
           (8, ["200", "120", "6", "fast", "csp=i422", "ctu=32"])]


@pytest.mark.parametrize("depth,args", CONFIGS)
def test_bitstream_identical_with_gpu_filters(depth, args, tmp_path):
    cpu, h_cpu = encode(depth, False, args, str(tmp_path / "cpu.hevc"))
    gpu, h_gpu = encode(depth, True, args, str(tmp_path / "gpu.hevc"))
    n = int(args[2])
    assert gpu["filter_producer"] == "gpu" and gpu["ff_pictures"] == n and gpu["ff_cpu_pictures"] == 0, "the GPU filters did not run: %s" % gpu
    if "deblock=0" not in args:
        assert gpu["ff_deblock_calls_skipped"] > 0
    if "sao=0" not in args and "ultrafast" not in args:
        assert gpu["ff_stats_served"] > 0
    assert cpu["bytes"] == gpu["bytes"] and h_cpu == h_gpu, "bitstreams differ: cpu %s gpu %s" % (cpu, gpu)
    print("e2e ff", depth, args, "per picture: gather %.2f ms, producer %.2f ms, the encoder's row loop behind it %.2f ms" % (
        1e3 * gpu["ff_gather_seconds"] / n, 1e3 * gpu["ff_producer_seconds"] / n, 1e3 * gpu["ff_replay_seconds"] / n))


def test_the_deferral_alone_keeps_the_bitstream(tmp_path):
    """the binding changes two things: WHEN a picture is filtered (all rows at its last one) and WHO deblocks and counts.  The first alone, with the encoder's own bodies:"""
    args = ["256", "192", "8", "medium"]
    cpu, h_cpu = encode(8, False, args, str(tmp_path / "cpu.hevc"))
    dfr, h_dfr = encode(8, True, args, str(tmp_path / "dfr.hevc"), defer_only=True)
    assert dfr["ff_pictures"] == 0 and dfr["ff_cpu_pictures"] == 8 and h_cpu == h_dfr


@pytest.mark.parametrize("depth,args,band_rows", [(8, ["256", "640", "8", "medium", "frame-threads=3", "wpp=1"], None),                 # ten CTU rows, bands of four (the default)
                                                  (8, ["256", "512", "8", "medium", "frame-threads=4", "wpp=0", "bframes=0"], "1"),      # every row its own band
                                                  (10, ["192", "576", "7", "slow", "frame-threads=3", "wpp=1"], "2"),
                                                  (8, ["320", "704", "8", "medium", "frame-threads=2", "wpp=1", "sao-non-deblock=1"], "3"),
                                                  (8, ["256", "640", "8", "medium", "frame-threads=3", "wpp=1", "sao=0"], "2"),           # deblocking alone: the rows' counters come from processPostCu
                                                  (8, ["256", "640", "8", "medium", "frame-threads=3", "wpp=1", "limit-sao=1", "ctu=32"], None),
                                                  (8, ["256", "512", "8", "medium", "frame-threads=3", "wpp=1", "csp=i422"], "2"),          # 4:2:2 in bands
                                                  (10, ["256", "512", "7", "medium", "frame-threads=3", "wpp=1", "csp=i444"], None),
                                                  (8, ["1920", "1080", "6", "medium", "frame-threads=4", "wpp=1"], None)])                 # BASELINE configs[1], threaded as the CLI threads it
def test_bitstream_identical_with_gpu_filters_under_frame_threads(depth, args, band_rows, tmp_path):
    """The encoder's default threading: the next pictures wait for the rows a picture's filters finish (Frame::m_reconRowFlag).  The binding filters in bands of CTU rows as they
    arrive (x265hip_ff_picture_desc.ctuRowFirst / ctuRowCount; the band's top edge changes the last lines of the row above; a band's statistics are taken before the rows below it
    are deblocked) and runs the encoder's row loop over each band: same bitstream as the encoder's own filters under the same threading, every picture through the producer."""
    env = {"X265FF_BAND_ROWS": band_rows} if band_rows else {}
    cpu, h_cpu = encode(depth, False, args, str(tmp_path / "cpu.hevc"))
    gpu, h_gpu = encode(depth, True, args, str(tmp_path / "gpu.hevc"), extra_env=env)
    n = int(args[2])
    want = int([a for a in args if a.startswith("frame-threads=")][0].split("=")[1])
    assert gpu["frame_threads"] == want and cpu["frame_threads"] == want
    assert gpu["ff_pictures"] == n and gpu["ff_cpu_pictures"] == 0 and gpu["ff_bands"] > n, "the pictures did not go through the producer in bands: %s" % gpu
    assert cpu["bytes"] == gpu["bytes"] and h_cpu == h_gpu, "bitstreams differ: cpu %s gpu %s" % (cpu, gpu)
    print("e2e ff frame threads", depth, args, "bands %d for %d pictures; per picture: gather %.2f ms, producer %.2f ms, row loop %.2f ms; fps cpu %.2f gpu %.2f" % (
        gpu["ff_bands"], n, 1e3 * gpu["ff_gather_seconds"] / n, 1e3 * gpu["ff_producer_seconds"] / n, 1e3 * gpu["ff_replay_seconds"] / n, cpu["fps"], gpu["fps"]))


def test_bands_on_request_with_one_frame_thread(tmp_path):
    args = ["256", "320", "8", "medium"]
    cpu, h_cpu = encode(8, False, args, str(tmp_path / "cpu.hevc"))
    gpu, h_gpu = encode(8, True, args, str(tmp_path / "gpu.hevc"), extra_env={"X265FF_BAND_ROWS": "2"})
    assert gpu["ff_pictures"] == 8 and gpu["ff_bands"] == 8 * 3 and h_cpu == h_gpu


@pytest.mark.parametrize("args,band_rows", [(["256", "320", "6", "medium", "frame-threads=2", "wpp=1", "slices=2"], None), (["320", "704", "7", "medium", "frame-threads=3", "wpp=1", "slices=3"], "2")])
def test_slices_under_frame_threads_go_through_the_producer_in_bands_of_their_own(args, band_rows, tmp_path):
    """--slices with frame threads: the slices of a picture finish their rows side by side while pictures overlap -- a band stays inside its slice, the slices' bands interleave"""
    cpu, h_cpu = encode(8, False, args, str(tmp_path / "cpu.hevc"))
    gpu, h_gpu = encode(8, True, args, str(tmp_path / "gpu.hevc"), extra_env={"X265FF_BAND_ROWS": band_rows} if band_rows else None)
    n = int(args[2])
    assert gpu["ff_pictures"] == n and gpu["ff_cpu_pictures"] == 0 and gpu["ff_bands"] >= 2 * n and h_cpu == h_gpu


def test_all_three_seams_together_under_frame_threads(tmp_path):
    """--threaded-me (bands of CTU rows through x265hip_tme_picture), the lookahead and the filters (bands through x265hip_ff_picture) on the GPU with three frame threads + WPP"""
    args = ["256", "640", "9", "medium", "frame-threads=3", "wpp=1"]
    cpu, h_cpu = encode(8, False, args, str(tmp_path / "cpu.hevc"), la=False, tme=True, tme_gpu=False)
    gpu, h_gpu = encode(8, True, args, str(tmp_path / "gpu.hevc"), la=True, tme=True)
    assert gpu["gpu_pictures"] >= 3 and gpu["la_estimates"] > 0 and gpu["ff_pictures"] == 9 and gpu["ff_bands"] > 9
    assert cpu["bytes"] == gpu["bytes"] and h_cpu == h_gpu, "bitstreams differ: cpu %s gpu %s" % (cpu, gpu)


@pytest.mark.parametrize("depth,args", [(8, ["256", "192", "10", "medium"]), (10, ["192", "128", "8", "slow"])])
def test_all_three_seams_together(depth, args, tmp_path):
    """--threaded-me with the GPU producing the MEData tables, the lookahead's costs AND the filtered pictures: the bitstream of the all-CPU run"""
    cpu, h_cpu = encode(depth, False, args, str(tmp_path / "cpu.hevc"), la=False, tme=True, tme_gpu=False)      # --threaded-me with the encoder's own producer
    gpu, h_gpu = encode(depth, True, args, str(tmp_path / "gpu.hevc"), la=True, tme=True)
    assert gpu["gpu_pictures"] >= 3 and gpu["la_estimates"] > 0 and gpu["ff_pictures"] == int(args[2])
    assert cpu["bytes"] == gpu["bytes"] and h_cpu == h_gpu, "bitstreams differ: cpu %s gpu %s" % (cpu, gpu)


def test_1080p_filters(tmp_path):
    args = ["1920", "1080", "3", "medium"]
    cpu, h_cpu = encode(8, False, args, str(tmp_path / "cpu.hevc"))
    gpu, h_gpu = encode(8, True, args, str(tmp_path / "gpu.hevc"))
    assert gpu["ff_pictures"] == 3 and h_cpu == h_gpu
    print("e2e ff 1080p per picture: gather %.2f ms, producer %.2f ms, row loop %.2f ms; fps cpu %.3f gpu %.3f" % (
        1e3 * gpu["ff_gather_seconds"] / 3, 1e3 * gpu["ff_producer_seconds"] / 3, 1e3 * gpu["ff_replay_seconds"] / 3, cpu["fps"], gpu["fps"]))
