"""SURVEY 8(f2) end to end: the reference encoder whose lookahead costs come from libx265hip (the binding integration/lookahead_adapter.cpp: LookaheadTLD::lowresIntraEstimate
and CostEstimateGroup::estimateFrameCost = one x265hip_la_intra / x265hip_la_estimate call each) must take the decisions -- slice types, scene cuts, cuTree offsets, rate
control -- it takes with its own CPU lookahead, i.e. write the same bitstream; alone, and together with the ThreadedME binding."""
import hashlib
import json
import os
import subprocess

import pytest

import x265hip

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def encode(depth, la, tme, tme_gpu, args, out, fade=False, batch=True):
    exe = os.path.join(ROOT, "oracle", "_ref", "x265e2e_%d" % depth)
    if not os.path.exists(exe):
        pytest.skip("no oracle/_ref/x265e2e_%d (built where the reference is present)" % depth)
    env = dict(os.environ, X265LAGPU="1" if la else "0", X265TME="1" if tme else "0", X265TMEGPU="1" if tme_gpu else "0", MALLOC_PERTURB_="85")
    if fade:
        env["X265TME_FADE"] = "1"
    env["X265LA_BATCH"] = "1" if batch else "0"
    r = subprocess.run([exe, x265hip.lib_path(depth)] + args[:4] + [out] + args[4:], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1]), hashlib.md5(open(out, "rb").read()).hexdigest()


@pytest.mark.parametrize("depth,args,fade", [(8, ["256", "192", "12", "medium"], False),                               # preset defaults: b-adapt 2, bframes 4, weightp, cuTree, AQ
                                             (8, ["256", "192", "12", "medium", "b-adapt=1"], False),
                                             (8, ["200", "120", "10", "medium", "bframes=2", "aq-mode=0", "cutree=0"], False),      # lowres pictures that are no block multiple, no AQ factors
                                             (10, ["256", "192", "10", "slow"], False),
                                             (8, ["256", "128", "12", "medium", "weightp=1"], True),                    # a fade: weightsAnalyse weights list 0
                                             (8, ["256", "192", "12", "medium", "lookahead-slices=4"], False),          # the cooperative sweep
                                             (8, ["256", "192", "10", "medium", "qg-size=8"], False),
                                             (8, ["256", "192", "16", "fast", "rc-lookahead=10", "scenecut=40"], False)])
def test_bitstream_identical_with_gpu_lookahead(depth, args, fade, tmp_path):
    cpu, h_cpu = encode(depth, False, False, False, args, str(tmp_path / "cpu.hevc"), fade)
    gpu, h_gpu = encode(depth, True, False, False, args, str(tmp_path / "gpu.hevc"), fade)
    assert gpu["lookahead_producer"] == "gpu" and gpu["la_intra_pictures"] == int(args[2]) and gpu["la_estimates"] > 0, "the GPU lookahead did not run: %s" % gpu
    assert gpu["la_cpu_estimates"] == 0
    assert cpu["bytes"] == gpu["bytes"] and h_cpu == h_gpu, "bitstreams differ: cpu %s gpu %s" % (cpu, gpu)
    if "cutree=0" not in args:
        assert gpu["la_cutree_steps"] > 0 and cpu["la_cutree_steps"] == 0, "cuTree's propagation steps did not go through the producer: %s" % gpu
    if fade:
        assert gpu["la_weighted"] > 0, "the fade did not make the lookahead weight a reference: %s" % gpu
    assert 0 < gpu["la_launches"] <= gpu["la_estimates"]
    print("e2e la", depth, args, "estimates %d in %d launches, %.2f ms each (producer %.2f)" % (gpu["la_estimates"], gpu["la_launches"], 1e3 * gpu["la_estimate_seconds"] / gpu["la_estimates"],
                                                                                  1e3 * gpu["la_producer_seconds"] / (gpu["la_estimates"] + gpu["la_intra_pictures"])))


@pytest.mark.parametrize("depth,args", [(8, ["320", "640", "16", "medium", "frame-threads=3", "wpp=1", "lookahead-slices=4"]),
                                        (10, ["256", "576", "12", "slow", "frame-threads=4", "wpp=1"])])
def test_both_seams_under_frame_threads(depth, args, tmp_path):
    """The encoder threaded the way its CLI threads it (frame threads, WPP, cooperative lookahead slices): the lookahead's workers call the binding concurrently and
    ThreadedME's rows arrive as bands (test_e2e_tme_gpu.py) -- lookahead alone and both seams together write the bitstreams of the CPU producers under the same threading."""
    cpu, h_cpu = encode(depth, False, False, False, args, str(tmp_path / "cpu.hevc"))
    gpu, h_gpu = encode(depth, True, False, False, args, str(tmp_path / "gpu.hevc"))
    assert gpu["lookahead_producer"] == "gpu" and gpu["la_estimates"] > 0 and gpu["frame_threads"] == int(args[4].split("=")[1])
    assert cpu["bytes"] == gpu["bytes"] and h_cpu == h_gpu, "lookahead alone: bitstreams differ: cpu %s gpu %s" % (cpu, gpu)
    cpu2, h_cpu2 = encode(depth, False, True, False, args, str(tmp_path / "cpu2.hevc"))
    gpu2, h_gpu2 = encode(depth, True, True, True, args, str(tmp_path / "gpu2.hevc"))
    assert gpu2["gpu_bands"] >= gpu2["gpu_pictures"] > 0 and gpu2["la_estimates"] > 0
    assert cpu2["bytes"] == gpu2["bytes"] and h_cpu2 == h_gpu2, "both seams: bitstreams differ: cpu %s gpu %s" % (cpu2, gpu2)


@pytest.mark.parametrize("depth,args,fade", [(8, ["320", "192", "20", "medium"], False), (8, ["256", "128", "16", "medium", "weightp=1", "bframes=3"], True),
                                             (10, ["256", "192", "14", "slow", "rc-lookahead=15"], False)])
def test_whole_batches_go_up_at_once(depth, args, fade, tmp_path):
    """CostEstimateGroup::finishBatch bound as a whole (slicetype.cpp:4271-4278: the motion-search batch and the frame-cost batch of b-adapt 2 with a thread pool): the queued
    (p0, b, p1) triples go up in x265hip_la_estimate_batch calls -- far fewer launches than estimates -- and the bitstream is the one of the CPU lookahead and of the
    one-estimate-per-call binding (X265LA_BATCH=0)"""
    # (--threaded-me with the encoder's own CPU producer: it gives the encoder its thread pool, and the lookahead batches its estimates only when it has one -- slicetype.cpp:1140)
    cpu, h_cpu = encode(depth, False, True, False, args, str(tmp_path / "cpu.hevc"), fade)
    one, h_one = encode(depth, True, True, False, args, str(tmp_path / "one.hevc"), fade, batch=False)
    gpu, h_gpu = encode(depth, True, True, False, args, str(tmp_path / "gpu.hevc"), fade)
    assert h_cpu == h_one == h_gpu and cpu["bytes"] == gpu["bytes"], "bitstreams differ: cpu %s one %s batch %s" % (cpu, one, gpu)
    assert one["la_batches"] == 0 and one["la_estimates"] > 0
    assert gpu["la_batches"] > 0 and gpu["la_batch_calls"] >= gpu["la_batches"] and gpu["la_cpu_estimates"] == 0
    assert gpu["la_launches"] < one["la_launches"], (gpu, one)
    assert gpu["la_launches"] * 3 <= gpu["la_estimates"], "batches did not merge: %s" % gpu
    if fade:
        assert gpu["la_weighted"] > 0
    print("e2e la batch", depth, args, "estimates %d: %d launches in %d batches (one per call: %d launches)" % (gpu["la_estimates"], gpu["la_launches"], gpu["la_batches"], one["la_launches"]))


@pytest.mark.parametrize("depth,args", [(8, ["256", "192", "10", "medium"]), (10, ["192", "128", "8", "slow"])])
def test_both_seams_together(depth, args, tmp_path):
    """--threaded-me with the GPU producing the MEData tables AND the lookahead's costs: the bitstream of the all-CPU run"""
    cpu, h_cpu = encode(depth, False, True, False, args, str(tmp_path / "cpu.hevc"))
    gpu, h_gpu = encode(depth, True, True, True, args, str(tmp_path / "gpu.hevc"))
    assert gpu["gpu_pictures"] >= 3 and gpu["la_estimates"] > 0
    assert cpu["bytes"] == gpu["bytes"] and h_cpu == h_gpu, "bitstreams differ: cpu %s gpu %s" % (cpu, gpu)


# --hme: the encoder switches it off below 540 lines (encoder.cpp:4799-4806), and below 720 lines the lookahead has no cooperative slices (slicetype.cpp:1165-1169)
@pytest.mark.parametrize("depth,args", [(8, ["960", "544", "6", "superfast", "hme=1"]),                                   # hme-search hex,umh,umh; hme-range 16,32,48
                                        (8, ["960", "544", "6", "faster", "hme=1", "hme-search=umh,hex,hex", "hme-range=24,24,32", "bframes=3", "b-adapt=2"]),
                                        (10, ["960", "544", "5", "superfast", "hme=1", "hme-search=hex"]),
                                        (8, ["960", "544", "8", "fast", "hme=1", "weightp=1", "bframes=2"]),
                                        (8, ["960", "544", "5", "superfast", "hme=1", "hme-search=dia,full,hex", "hme-range=16,6,32"]),      # diamond / exhaustive levels
                                        (8, ["960", "544", "4", "superfast", "hme=1", "hme-search=star,hex,hex"]),                           # star levels (the raster's window is the picture)
                                        (10, ["960", "544", "4", "superfast", "hme=1", "hme-search=umh,star,hex", "hme-range=16,24,32", "bframes=2"])])       # with a fade: list 0 is searched in the weighted copy on the half-resolution level only
def test_bitstream_identical_with_gpu_hme_lookahead(depth, args, tmp_path):
    """--hme: the quarter-resolution sweep runs on the GPU too (x265hip_la_enable_hme + x265hip_la_estimate_desc.hme); Lowres::lowerResMvs / lowerResMvCosts come back"""
    fade = "weightp=1" in args
    cpu, h_cpu = encode(depth, False, False, False, args, str(tmp_path / "cpu.hevc"), fade)
    gpu, h_gpu = encode(depth, True, False, False, args, str(tmp_path / "gpu.hevc"), fade)
    if fade:
        assert gpu["la_weighted"] > 0, "the fade did not make the lookahead weight a reference: %s" % gpu
    assert gpu["la_estimates"] > 0 and gpu["la_cpu_estimates"] == 0, "the GPU lookahead did not run: %s" % gpu
    assert cpu["bytes"] == gpu["bytes"] and h_cpu == h_gpu, "bitstreams differ: cpu %s gpu %s" % (cpu, gpu)


def test_hme_levels_the_producer_lacks_stay_with_the_encoder(tmp_path):
    """an --hme level the producer does not offer (a range beyond 64; a sea level is not a case: the reference itself dereferences MotionEstimate::integral[] == NULL there,
    tests/test_lookahead_oracle_vs_ref.py::test_reference_cannot_run_a_sea_level_of_hme): the adapter forwards those estimates to the encoder's own body (counted), the intra
    estimates still come from the GPU"""
    args = ["960", "544", "4", "superfast", "hme=1", "hme-search=hex,umh,hex", "hme-range=16,80,96"]
    cpu, h_cpu = encode(8, False, False, False, args, str(tmp_path / "cpu.hevc"))
    gpu, h_gpu = encode(8, True, False, False, args, str(tmp_path / "gpu.hevc"))
    assert gpu["la_estimates"] == 0 and gpu["la_cpu_estimates"] > 0
    assert h_cpu == h_gpu
