#!/usr/bin/env python3
"""bench.py -- ME + DCT + quant throughput of the x265 encoder-primitives hot path on MI355X.

One "step" = one pass of the hot path over one batch of F synthetic (source, reference) frame pairs that are
already resident in HBM: integer + sub-pel motion search for the 2Nx2N PU pyramid of every CTU (85 PUs / CTU64),
then motion compensation -> residual -> DCT -> quant of every 32x32 TU with the MVs just found.
Metric (BASELINE.json): Mpixels/s of luma source pixels through that pipeline, whole job over all ranks.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU.  Under torch.distributed.run (WORLD_SIZE set) this process IS a rank; started plainly with
--gpus N it launches the N ranks itself (x265hip_pkg.sharding.spawn_ranks -> python -m torch.distributed.run).
Ranks process independent frames (weak scaling, no data-path collective; RCCL is used only for the barrier and the
max-over-ranks time).  Rank 0 prints ONE JSON line.

Default workload = BASELINE.json configs[2], the configuration the metric is quoted on that fits one GPU: 4K 10-bit
Main10, preset slow (STAR search, subme 3).  `--workload 1080p8_medium` is configs[1].
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E nominal (guides/MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)
# vector-instruction issue peak: 256 CUs x 4 SIMDs, one wave64 VALU instruction per SIMD every VALU_CYCLES clocks at 2.4 GHz.
# profiles/micro/valu_rate.hip measures it for the search kernels' own instruction (v_sad_u16 / v_sad_u8); see profiles/micro/RESULTS.md
GPU_CLOCK_HZ, N_SIMD, VALU_CYCLES = 2.4e9, 1024, 4
METHODS = {"dia": 0, "hex": 1, "star": 3, "full": 5}
WORKLOADS = {
    # BASELINE.json configs[1]: 1080p 8-bit, preset medium (me hex, subme 2, merange 57 -- param.cpp:188,238-256)
    "1080p8_medium": dict(depth=8, width=1920, height=1088, method="hex", subme=2, merange=57),
    # configs[2]: 2160p 10-bit Main10, preset slow (me star, subme 3)
    "2160p10_slow": dict(depth=10, width=3840, height=2176, method="star", subme=3, merange=57),
    # configs[4]: 8K 10-bit, preset slower (me star, subme 4), --merange 128 (run with --frames 2: 16 phase planes of one
    # 8K10 frame pair are 1.1 GB and the size-specialised kernels address the plane buffer with 32-bit byte offsets)
    "4320p10_slower": dict(depth=10, width=7680, height=4352, method="star", subme=4, merange=128),
    # the exhaustive integer search (VALU-bound by construction), reported next to the pattern searches
    "1080p8_full": dict(depth=8, width=1920, height=1088, method="full", subme=2, merange=16),
    # not a benchmark: a few CTUs, for checking this script's plumbing (the JSON line, the CPU-baseline leg) where a run must be short
    "plumbing_256x128": dict(depth=8, width=256, height=128, method="hex", subme=2, merange=24),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="2160p10_slow", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=8, help="frame pairs per step per GPU")
    ap.add_argument("--qp", type=int, default=28)
    ap.add_argument("--refs", type=int, default=1, help="list-0 reference pictures searched per source picture (preset medium: 3, slow: 4 -- param.cpp:567-587); each is searched down the "
                    "pyramid, x265hip_inter_merge_batch chooses per PU, the TQ stage compensates from the chosen reference.  value stays Mpixels/s of SOURCE pixels")
    ap.add_argument("--rect", action="store_true", help="also search the 2NxN / Nx2N PUs of every CU (bEnableRectInter, preset slow and up): 425 PUs per CTU instead of 85; one reference")
    ap.add_argument("--tu", type=int, default=5, help="log2 TU size of the DCT+quant stage")
    ap.add_argument("--inner", type=int, default=5, help="passes over the resident batch per step (a step of the driver's --steps 20 then lasts long enough for the whole timed region to be >= 0.5 s)")
    ap.add_argument("--no-streams-leg", action="store_true", help="skip the extra leg that steps the same batch as 2 and 3 sub-batches of whole pictures on their own streams (x265hip_batch_desc.streams); reported under \"streams\", not part of value")
    ap.add_argument("--fused", type=int, default=0, help="x265hip_batch_set_mode flags: 0 = the default schedule; 4 = the 64x64 level with its start-stage launch (the form of rounds 1-2); 16 = the phase planes in groups of two pictures (what an 8K batch does by itself).  The fused lower levels (1, 2) and the tiled phase planes (8) were measured losses and left the library in round 5: the library refuses them")
    ap.add_argument("--splits", type=int, default=2, help="cut the batch into this many sub-batches of whole pictures, each on its own HIP stream (independent pictures; the levels of one picture stay in order)")
    ap.add_argument("--no-planes", action="store_true", help="interpolate sub-pel candidates inside the ME kernel instead of using phase planes")
    ap.add_argument("--recon", action="store_true", help="also run S4 (dequant -> IDCT -> recon -> SSE)")
    ap.add_argument("--lookahead", action="store_true", help="also time the lookahead frame-cost batch (lowres init, intra estimate, estimateFrameCost of a 32-picture window); reported under \"lookahead\", not part of value")
    ap.add_argument("--intra", action="store_true", help="also time the intra mode scan (35 sa8d costs per CU, sizes 64..8) over the same frames; reported under \"intra_scan\", not part of value")
    ap.add_argument("--cpu-ctus", type=int, default=4080, help="CTUs in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-dry-run", action="store_true", help="launcher / bookkeeping check without a GPU: gloo ranks, a sleep in place of the step (tests/test_sharding.py); prints a line marked data=dry-run")
    ap.add_argument("--no-tme", action="store_true", help="skip the ThreadedME producer leg (x265hip_tme_picture on synthetic 1080p pictures: medium- and slow-like partition sets); reported under \"tme_producer\", not part of value")
    ap.add_argument("--no-preset-exact", action="store_true", help="skip the preset-exact legs (per BASELINE workload the preset's own reference count, from preset slow on the rectangular PUs, from slower on the asymmetric ones, on 8 pictures); reported under \"preset_exact\", not part of value")
    ap.add_argument("--preset-exact-workloads", default="1080p8_medium,2160p10_slow,4320p10_slower", help="which workloads get a preset-exact leg")
    ap.add_argument("--no-e2e", action="store_true", help="skip the live end-to-end leg (reference encoder, 1920x1088 medium, CPU producer vs GPU producer of the MEData tables, ~20 s); reported under \"e2e_fps\"")
    ap.add_argument("--leg", default=None, help="(internal) run ONE auxiliary leg in this process and print its JSON object: tme_producer | preset_exact:<workload> | streams:<headline s per pass>.  "
                    "The main run starts its GPU-heavy auxiliary legs this way: a fault in one of them then costs that leg, not the line")
    ap.add_argument("--filters", action="store_true", help="also time the in-loop filter chain after reconstruction (deblock, SAO statistics, SAO apply, SSIM, SSD) on 8 coded 1080p pictures; reported under \"filters\", not part of value")
    return ap.parse_args()


def tme_producer_leg(depth):
    """the MEData tables of whole pictures through x265hip_tme_picture (include/x265hip_ctx.h), the call the reference encoder makes per picture in
    place of its CPU producer (oracle/ref_tme_gpu.cpp; bitstream-identical there, tests/test_e2e_tme_gpu.py).  Host planes in, host table out -- PCIe inclusive.
    Synthetic 1920x1080 P pictures (a shifted, noisy copy of the reference; one reference, no temporal neighbours), the partition sets of presets medium and slow."""
    import ctypes as C
    import importlib
    import numpy as np
    import x265hip
    TmeProducer = importlib.import_module("x265-mod-by-patman_amd.tme_host").TmeProducer
    lib = C.CDLL(x265hip.lib_path(depth))
    W, H, margin = 1920, 1080, 96
    stride, rows = W + 2 * margin, ((H + 63) // 64) * 64 + 2 * margin
    rng = np.random.default_rng(7)
    dt = np.uint8 if depth == 8 else np.uint16
    base = rng.integers(0, 1 << depth, (rows // 8 + 2, stride // 8 + 2)).astype(np.int32)
    ref = np.kron(base, np.ones((8, 8), dtype=np.int32))[:rows, :stride]
    ref = np.clip(ref + rng.integers(-6, 7, ref.shape), 0, (1 << depth) - 1).astype(dt)
    cur = np.clip(np.roll(ref, (3, -5), axis=(0, 1)).astype(np.int32) + rng.integers(-4, 5, ref.shape), 0, (1 << depth) - 1).astype(dt)
    ref, cur = np.ascontiguousarray(ref).reshape(-1), np.ascontiguousarray(cur).reshape(-1)
    out = {"unit": "ms per picture", "picture": "%dx%d P picture, 1 reference, host planes in / host table out" % (W, H), "presets": {}}
    # preset medium: 85 entries per CTU, HEX, subme 2; preset slow: rect but no AMP (param.cpp:572-587), STAR, subme 3; preset slower adds AMP and subme 4 (param.cpp:588-608)
    for name, rect, amp, method, subme in (("medium", False, False, 1, 2), ("slow", True, False, 3, 3), ("slower", True, True, 3, 4)):
        prod = TmeProducer(lib, W, H, 64, 8, rect, amp)
        try:
            table = prod.empty_table()
            times = []
            for it in range(5):
                table["ref"] = -1
                t0 = time.perf_counter()
                prod.picture(cur, [[ref], []], stride, margin * stride + margin, table, qp=28, merange=57, method=method, subme=subme)
                times.append(time.perf_counter() - t0)
            used = int((table["ref"][:, 0] >= 0).sum())
            assert used > 0, "the producer wrote no record"
            ms = sorted(times[1:])[len(times[1:]) // 2] * 1e3                                   # the first picture pays for streams and code objects
            counters = {}
            cpath = os.path.join(ROOT, "profiles", "r03_tme_%s_counters.json" % name)
            if os.path.exists(cpath):
                try:
                    cj = json.load(open(cpath))
                    counters = {"source": "profiles/r03_tme_%s_counters.json (%s)" % (name, cj.get("_source", "")),
                                "bound": "valu issue, 1024 SIMDs x 2.4 GHz / 4 = 614.4 G wave-instr/s; the chain kernels are bound by the serial depth of one PU chain (few strands per picture), not by issue or bandwidth",
                                "kernels": {k: {f: v[f] for f in ("avg_us", "calls", "valu_insts", "valu_frac_of_issue_peak", "hbm_read_MB", "hbm_write_MB") if f in v}
                                            for k, v in cj["kernels"].items() if k.startswith("tme_chain_kernel") or k.startswith("diamond") or k.startswith("subpel")}}
                except Exception:
                    counters = {}
            out["presets"][name] = {"ms": round(ms, 2), "pictures_per_s": round(1e3 / ms, 1), "entries_per_ctu": prod.entries, "records_written": used, "roofline_valu": counters or None,
                                    "search": {1: "hex", 3: "star"}[method], "subme": subme, "rect": rect, "amp": amp}
            # the same with the caller's long-lived buffers page-locked once (x265hip_host_register: PicYuv planes and FrameData tables live as long as the encoder)
            pinned = [cur, ref, table]
            for a in pinned:
                prod.pin(a)
            try:
                times = []
                for it in range(5):
                    table["ref"] = -1
                    t0 = time.perf_counter()
                    prod.picture(cur, [[ref], []], stride, margin * stride + margin, table, qp=28, merange=57, method=method, subme=subme)
                    times.append(time.perf_counter() - t0)
                out["presets"][name]["ms_host_buffers_registered"] = round(sorted(times[1:])[len(times[1:]) // 2] * 1e3, 2)
            finally:
                for a in pinned:
                    prod.unpin(a)
        finally:
            prod.close()
        # frame threads: independent pictures in flight, one producer (context, stream, buffers) per host thread, as an encoder's frame encoders would hold them
        import threading
        NT, PER = 4, 4
        prods = [TmeProducer(lib, W, H, 64, 8, rect, amp) for _ in range(NT)]
        tables = [p.empty_table() for p in prods]
        try:
            for p, tb in zip(prods, tables):
                p.picture(cur, [[ref], []], stride, margin * stride + margin, tb, qp=28, merange=57, method=method, subme=subme)       # streams, code objects
            def work(i):
                for _ in range(PER):                         # (the table is not reset between pictures: a numpy pass would hold the interpreter lock of all four threads)
                    prods[i].picture(cur, [[ref], []], stride, margin * stride + margin, tables[i], qp=28, merange=57, method=method, subme=subme)
            th = [threading.Thread(target=work, args=(i,)) for i in range(NT)]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            dt = time.perf_counter() - t0
            out["presets"][name]["pictures_per_s_4_threads"] = round(NT * PER / dt, 1)
        finally:
            for p in prods:
                p.close()
    return out


def e2e_fps_leg(frames=24, seam_frames=8, default_frames=48):
    """BASELINE's M2 measured in THIS run: the reference encoder (oracle/_ref/x265e2e_8 = all of source/common + source/encoder compiled from where they lie, C
    primitives, no asm; it travels with the repository) on BASELINE configs[1] -- 1920x1088 8-bit, preset medium with the preset's own defaults, --threaded-me -- with its
    own CPU producers, with x265hip_tme_picture producing the MEData tables (integration/tme_adapter.cpp), with x265hip_la_intra / x265hip_la_estimate producing the
    lookahead's costs (integration/lookahead_adapter.cpp), with x265hip_ff_picture producing the deblocked picture and the SAO statistics (integration/filter_adapter.cpp),
    and with all three.  Every run must write the same bitstream.
    None when no binary is there (then the committed figure of profiles/e2e_fps.json is reported, labelled as such)."""
    import hashlib, subprocess, tempfile
    import x265hip
    exe = os.path.join(ROOT, "oracle", "_ref", "x265e2e_8")
    both = os.path.exists(exe)
    if not both:
        exe = os.path.join(ROOT, "oracle", "_ref", "x265tmegpu_8")
        if not os.path.exists(exe):
            return None
    runs = {}
    # M2 itself -- the encoder's own producers against the GPU on every bound seam -- over `frames` pictures (at ~1 fps a short clip is a noisy clock: +-3 % between runs of
    # 8 frames); the per-seam attribution runs over `seam_frames`, next to a CPU run of the same length (bitstreams are compared within a length)
    modes = ((("cpu", 0, 0, 0, frames), ("all_gpu", 1, 1, 1, frames), ("cpu_short", 0, 0, 0, seam_frames), ("tme_gpu", 1, 0, 0, seam_frames), ("la_gpu", 0, 1, 0, seam_frames),
              ("ff_gpu", 0, 0, 1, seam_frames), ("cpu_filters_timed", 0, 0, 2, seam_frames)) if both else (("cpu", 0, 0, 0, frames), ("tme_gpu", 1, 0, 0, frames)))
    with tempfile.TemporaryDirectory() as td:
        for name, tme, la, ff, nfr in modes:
            outp = os.path.join(td, name + ".hevc")
            env = dict(os.environ, X265TMEGPU=str(tme), X265LAGPU=str(la), X265FFGPU="1" if ff else "0", MALLOC_PERTURB_="85")
            env.pop("X265FF_DEFER_ONLY", None)
            if ff == 2:
                env["X265FF_DEFER_ONLY"] = "1"             # the binding's deferral with the encoder's own filters: times what the CPU spends on a picture's filters
            if name in ("cpu_short", "tme_gpu"):
                env["X265_FRAME_STATS"] = "1"              # the encoder's own per-frame clocks (csv-log-level 2; same bitstream): what ThreadedME costs and how long rows wait for it
            r = subprocess.run([exe, x265hip.lib_path(8), "1920", "1088", str(nfr), "medium", outp], capture_output=True, text=True, env=env, timeout=900)
            if r.returncode != 0:
                return {"measured": "this run: FAILED", "producer": name, "stderr": r.stderr[-500:]}
            info = json.loads(r.stdout.strip().splitlines()[-1])
            info["md5"] = hashlib.md5(open(outp, "rb").read()).hexdigest()
            info["clip_frames"] = nfr
            runs[name] = info
        # the same encode threaded as the x265 CLI threads it by default (frame threads picked from the core count, WPP, every core: encoder.cpp:285): the encoder alone
        # (no --threaded-me: what anyone runs), with --threaded-me and its own producers, and with the GPU producers.  ThreadedME's binding hands a picture's CTU rows to
        # the producer as bands while the references are still being coded (integration/tme_adapter.cpp); the filter binding keeps the encoder's own filters there
        # (one picture in flight per FrameFilter is what it replays), the lookahead binding is thread-safe as it is
        default_runs = {}
        if both:
            # (256 pool threads under whatever CPU quota the box gives them: the clock of one run moves by +-10 %, so the GPU run is taken three times, the plain CPU run twice)
            for name, tme_on, tme, la, ff, reps in (("cpu_default_threading", 0, 0, 0, 0, 3), ("cpu_default_threading_tme", 1, 0, 0, 0, 3), ("all_gpu_default_threading", 1, 1, 1, 1, 3), ("gpu_tme_lookahead_default_threading", 1, 1, 1, 0, 3),
                                                    ("gpu_lookahead_default_threading", 0, 0, 1, 0, 3)):
                got = []
                for rep in range(reps):
                    outp = os.path.join(td, "%s_%d.hevc" % (name, rep))
                    env = dict(os.environ, X265TME=str(tme_on), X265TMEGPU=str(tme), X265LAGPU=str(la), X265FFGPU=str(ff), X265_CLI_THREADING="1", MALLOC_PERTURB_="85")
                    env.pop("X265FF_DEFER_ONLY", None)
                    r = subprocess.run([exe, x265hip.lib_path(8), "1920", "1088", str(default_frames), "medium", outp], capture_output=True, text=True, env=env, timeout=900)
                    if r.returncode != 0:
                        default_runs[name] = {"failed": r.stderr[-300:]}
                        break
                    info = json.loads(r.stdout.strip().splitlines()[-1])
                    info["md5"] = hashlib.md5(open(outp, "rb").read()).hexdigest()
                    got.append(info)
                else:
                    got.sort(key=lambda i: i["fps"])
                    med = dict(got[len(got) // 2])
                    med["fps_runs"] = [i["fps"] for i in got]
                    med["md5_all_equal"] = all(i["md5"] == got[0]["md5"] for i in got)
                    default_runs[name] = med
            # the encoder's own per-frame clocks for the plain and the GPU run (a run each, not part of the fps figures: the statistics cost the ThreadedME runs some speed)
            for name, tme_on, tme, la in (("cpu_default_threading", 0, 0, 0), ("all_gpu_default_threading", 1, 1, 1)):
                # ... and the encoder's own quality accounting of the same run (x265_stats: bitrate, global PSNR / SSIM): the bitstream is the fps runs' (checked), and a run with
                # --threaded-me does not write the plain encoder's bitstream -- their speeds mean something only next to what each spent and kept
                env = dict(os.environ, X265TME=str(tme_on), X265TMEGPU=str(tme), X265LAGPU=str(la), X265FFGPU="0", X265_CLI_THREADING="1", MALLOC_PERTURB_="85", X265_FRAME_STATS="1", X265_QUALITY="1")
                r = subprocess.run([exe, x265hip.lib_path(8), "1920", "1088", str(default_frames), "medium", os.path.join(td, "stats.hevc")], capture_output=True, text=True, env=env, timeout=900)
                if r.returncode == 0 and name in default_runs and "fps" in default_runs[name]:
                    si = json.loads(r.stdout.strip().splitlines()[-1])
                    default_runs[name]["frame_stats_ms_per_picture"] = si.get("frame_stats_ms_per_picture")
                    if si.get("quality"):
                        q = dict(si["quality"])
                        q["bytes"] = si["bytes"]
                        # (the encoder measures PSNR / SSIM only at log level info, and its option-string SEI then spells "psnr ssim" instead of "no-psnr no-ssim": 6 bytes fewer per SEI)
                        q["bytes_more_than_the_fps_runs"] = si["bytes"] - default_runs[name]["bytes"]
                        default_runs[name]["quality"] = q
    g, c = runs["tme_gpu"], runs["cpu"]
    best = runs.get("all_gpu", g)
    out = {"value": best["fps"], "unit": "frames/s", "measured": "this run",
           "config": "BASELINE configs[1]: 1920x1088 8-bit, preset medium with its own defaults (ref=3, weightp, bframes=4, b-adapt 2), --threaded-me, one frame thread, no WPP (how the three seams were first bound: "
                     "all of them on the GPU, the filter binding needs this threading), %d frames for cpu / all_gpu, %d for the per-seam runs" % (frames, seam_frames),
           "host": "the reference encoder with C primitives (no asm; no nasm on this box) -- its RDO bounds the encode; one frame thread: GPU producers on every seam that is bound (%s)"
                   % ("ThreadedME + lookahead + in-loop filters" if both else "ThreadedME"),
           "fps": {k: v["fps"] for k, v in runs.items()}, "same_encoder_cpu_producer_fps": c["fps"],
           "bitstream_identical": all(v["md5"] == w["md5"] and v["bytes"] == w["bytes"] for v in runs.values() for w in runs.values() if v["clip_frames"] == w["clip_frames"]), "bytes": c["bytes"],
           "tme": {"gpu_pictures": g["gpu_pictures"], "producer_ms_per_picture": round(1e3 * g["gpu_seconds"] / max(1, g["gpu_pictures"]), 2),
                   "producer_ms_per_picture_warm": (round(1e3 * best["gpu_seconds_warm"] / best["gpu_calls_warm"], 2) if best.get("gpu_calls_warm") else None),
                   "producer_note": "producer_ms_per_picture: the %d-picture per-seam run, first launches (code objects) included; _warm: the %d-frame all-GPU run without its first four calls" % (g["clip_frames"], best["clip_frames"]),
                   "adapter_host_ms_per_picture": round(1e3 * (g["adapter_seconds"] - g["gpu_seconds"] - g.get("adapter_create_seconds", 0.0)) / max(1, g["gpu_pictures"]), 2),
                   "adapter_note": "host work around the producer call (qps, collocated neighbours, medians, table conversions), spread over the encoder's ThreadedME workers; creating the producer (%.0f ms, once) not included"
                                   % (1e3 * g.get("adapter_create_seconds", 0.0)),
                   "ctus_harvested_by_helper_workers": int(g["adapter_sections"][2])},
           "encoder_clocks_ms_per_picture": {"one frame thread, its own ThreadedME": runs.get("cpu_short", {}).get("frame_stats_ms_per_picture"), "one frame thread, GPU ThreadedME": g.get("frame_stats_ms_per_picture"),
                                             "note": "threaded_me_tasks = time the ThreadedME workers spend in their tasks, summed over workers (with the GPU producer: mostly waiting for the picture's call); "
                                                     "rows_blocked_on_threaded_me = time the row encoder waits for a CTU's records.  wall - blocked is the frame thread's own work: mode decision, RDO, entropy coding -- "
                                                     "what bounds M2; no encoder-side consumer of the batched TQ / intra arithmetic exists (Quant::transformNxN and estIntraPredQT run per CU inside the RDO loop)"}}
    if default_runs:
        ok = {k: v for k, v in default_runs.items() if "fps" in v}
        # the GPU run of the headline: ThreadedME + lookahead + filters (r06: the filter seam works in bands of CTU rows under frame threads), or the same with the encoder's own
        # filters -- whichever is faster on this host; both are listed, `gpu_run.seams` says which one `value` is
        a = ok.get("cpu_default_threading_tme")
        b3, b2 = ok.get("all_gpu_default_threading"), ok.get("gpu_tme_lookahead_default_threading")
        b = b3 if (b3 and (not b2 or b3["fps"] >= b2["fps"])) else b2
        seams = "ThreadedME + lookahead + in-loop filters" if b is b3 else "ThreadedME + lookahead (the encoder's own filters: faster on this host than the filter seam's bands)"
        dt = {"config": "the same clip and preset, %d frames, threaded as the CLI threads it (frame threads from the core count, WPP, %d cores)" % (default_frames, os.cpu_count() or 0),
              "fps": {k: v["fps"] for k, v in ok.items()}, "fps_runs": {k: v["fps_runs"] for k, v in ok.items()}, "fps_rule": "median of the runs listed in fps_runs",
              "failed": {k: v["failed"] for k, v in default_runs.items() if "failed" in v} or None,
              "frame_threads": b.get("frame_threads") if b else None, "wpp": b.get("wpp") if b else None,
              "bitstream_identical_gpu_vs_cpu_producers": bool(a and b and a["md5"] == b["md5"] and a["bytes"] == b["bytes"] and b["md5_all_equal"]), "host": usable_cores()[1]}
        dt["encoder_clocks_ms_per_picture"] = {k: v.get("frame_stats_ms_per_picture") for k, v in ok.items() if v.get("frame_stats_ms_per_picture")}
        # equal-quality reading of the two bitstreams (the plain encoder's; --threaded-me's, which the GPU producers write too) + the speed per CPU the host grants
        qa, qb = (ok.get("cpu_default_threading") or {}).get("quality"), (b3 or {}).get("quality") or (b2 or {}).get("quality")
        if qa and qb:
            dt["quality"] = {"encoder_alone": qa, "threaded_me_gpu_producers": qb,
                             "threaded_me_vs_encoder_alone": {"kbps_ratio": round(qb["kbps"] / qa["kbps"], 4) if qa["kbps"] else None, "psnr_y_db": round(qb["psnr_y"] - qa["psnr_y"], 4),
                                                              "psnr_global_db": round(qb["psnr_global"] - qa["psnr_global"], 4), "ssim": round(qb["ssim"] - qa["ssim"], 6)},
                             "note": "x265_encoder_get_stats of one extra run per bitstream (PSNR / SSIM accounting on; it reads the reconstruction and changes no decision; the option-string SEI differs by the two words, bytes_more_than_the_fps_runs). "
                                     "--threaded-me replaces the per-CU motion search inside mode decision by searches made ahead of it per CTU: another bitstream than the plain encoder's, compared here at the same rate control"}
        n_cpu = usable_cores()[0]
        dt["fps_per_granted_cpu"] = {k: round(v["fps"] / max(1, n_cpu), 4) for k, v in ok.items()}
        dt["granted_cpus"] = n_cpu
        if b:
            ms = 1e3 * b["gpu_seconds"] / max(1, b["gpu_pictures"])
            dt["producer_ceiling"] = {"tme_producer_ms_per_picture": round(ms, 2), "pictures_per_second_of_the_one_producer": round(1e3 / ms, 1) if ms > 0 else None,
                                      "note": "one ThreadedME producer serves the encode (bands of every picture in flight take turns on it): whatever the host's core count, the encode cannot go faster than this"}
            dt["gpu_run"] = {"seams": seams, "tme_pictures": b["gpu_pictures"], "tme_bands": b.get("gpu_bands"), "tme_producer_ms_per_picture": round(1e3 * b["gpu_seconds"] / max(1, b["gpu_pictures"]), 2),
                             "tme_adapter_seconds": b["adapter_seconds"], "la_estimates": b.get("la_estimates"), "la_producer_seconds": b.get("la_producer_seconds"),
                             "filter_pictures_gpu": b.get("ff_pictures"), "filter_bands": b.get("ff_bands"), "filter_pictures_left_to_the_cpu": b.get("ff_cpu_pictures"),
                             "filter_ms_per_picture": ({"gather": round(1e3 * b["ff_gather_seconds"] / b["ff_pictures"], 2), "x265hip_ff_picture": round(1e3 * b["ff_producer_seconds"] / b["ff_pictures"], 2),
                                                        "row_loop": round(1e3 * b["ff_replay_seconds"] / b["ff_pictures"], 2)} if b.get("ff_pictures") else None), "seconds": b["seconds"]}
        p0, l0 = ok.get("cpu_default_threading"), ok.get("gpu_lookahead_default_threading")
        ck = dt["encoder_clocks_ms_per_picture"]
        if ck.get("cpu_default_threading") and ck.get("all_gpu_default_threading") and ck["cpu_default_threading"].get("ctu_worker_time"):
            # what fraction of the encode the bound seams cover, by the encoder's own clocks (x265 --csv-log-level 2: time its CTU workers spend in compressCTU / encodeCTU per picture, summed
            # over the workers): the plain encoder's figure holds motion search + mode decision + RDO + entropy coding; with the GPU ThreadedME the searches (and AMVP) are gone from it.
            # The transforms and intra prediction of RDO stay on the host (x265hip_tq_batch / x265hip_intra_cost_batch have no caller inside an encode: RDO is serial per CU)
            w0, w1 = ck["cpu_default_threading"]["ctu_worker_time"], ck["all_gpu_default_threading"].get("ctu_worker_time") or 0.0
            dt["seam_coverage"] = {"ctu_worker_ms_per_picture_encoder_alone": w0, "ctu_worker_ms_per_picture_gpu_seams": w1, "share_of_ctu_worker_time_moved_to_the_gpu": round(1.0 - w1 / w0, 3),
                                   "left_on_the_host": "mode decision, RDO (transform / quant / intra prediction through the host table), entropy coding, in-loop filters",
                                   "batched_tq_and_intra_inside_the_encode": "none: x265hip_tq_batch / x265hip_intra_cost_batch are reached by bench and tests only (RDO decides CU by CU)"}
        if p0 and l0:
            dt["bitstream_identical_gpu_lookahead_vs_plain_encoder"] = bool(p0["md5"] == l0["md5"] and l0["md5_all_equal"] and p0["md5_all_equal"])
        if a and b and p0:
            dt["verdict"] = ("with --threaded-me: GPU producers %.2f fps vs the encoder's own producers %.2f fps (%.2fx), same bitstream.  The encoder WITHOUT --threaded-me does %.2f fps: "
                             "the GPU producers are %s than not using ThreadedME at all (%.2fx).  A picture's CTU rows reach the producer in bands as its references' rows become final; a band that "
                             "would hold fewer than half the picture's rows waits up to 16 ms for another row -- a producer call is as long as one CTU's chain of searches whatever it holds and "
                             "costs the host a job set-up and its wake-ups (profiles/r05_min_rows_ab.txt: 7.0 fps without the wait, 8.4 with it; r05_queues_ab.txt: the waiting workers on two queues, no helpers under frame threads: ~9.8).  The lookahead seam alone (no "
                             "--threaded-me, the plain encoder's bitstream) does %s fps; gpu_run.seams says whether the filter seam (bands of CTU rows under frame threads, r06) is part of the value.  RDO and entropy coding on the host bound the encode"
                             % (b["fps"], a["fps"], b["fps"] / a["fps"], p0["fps"], "FASTER" if b["fps"] > p0["fps"] else "SLOWER", b["fps"] / p0["fps"], ("%.2f" % l0["fps"]) if l0 else "n/a"))
        out["default_threading"] = dt
        if b and p0 and dt.get("bitstream_identical_gpu_vs_cpu_producers"):
            # M2 as anyone runs x265: the headline of this object is the default-threaded encode; the one-frame-thread pair (how the seams were first bound) stays beside it
            out["one_frame_thread"] = {"value": out["value"], "same_encoder_cpu_producer_fps": out["same_encoder_cpu_producer_fps"], "config": out["config"]}
            out["value"] = b["fps"]
            out["same_encoder_cpu_producer_fps"] = a["fps"]
            out["encoder_alone_fps"] = p0["fps"]
            out["config"] = ("BASELINE configs[1]: 1920x1088 8-bit, preset medium with its own defaults, the CLI's default threading (%s frame threads, WPP, every core the host grants), %d frames, medians of "
                             "repeated runs: value = --threaded-me with the GPU producers on the seams gpu_run.seams names; same_encoder_cpu_producer_fps = the same "
                             "with the encoder's own producers (same bitstream); encoder_alone_fps = no --threaded-me" % (b.get("frame_threads"), default_frames))
    if both:
        l = runs["la_gpu"]
        out["lookahead"] = {"intra_pictures": l["la_intra_pictures"], "estimates": l["la_estimates"], "device_launches": l.get("la_launches"), "finish_batch_calls_taken_whole": l.get("la_batches"), "estimate_batch_calls": l.get("la_batch_calls"), "estimates_left_to_the_cpu": l["la_cpu_estimates"], "cutree_steps": l.get("la_cutree_steps"), "ms_per_cutree_step": round(1e3 * l["la_cutree_seconds"] / l["la_cutree_steps"], 3) if l.get("la_cutree_steps") else None,
                            "ms_per_estimate": round(1e3 * l["la_estimate_seconds"] / max(1, l["la_estimates"]), 3),
                            "ms_per_intra_picture": round(1e3 * l["la_intra_seconds"] / max(1, l["la_intra_pictures"]), 3),
                            "producer_seconds": l["la_producer_seconds"]}
        f, ct = runs["ff_gpu"], runs["cpu_filters_timed"]
        nf = max(1, f["ff_pictures"])
        out["filters"] = {"pictures": f["ff_pictures"], "pictures_left_to_the_cpu": f["ff_cpu_pictures"],
                          "ms_per_picture": {"gather": round(1e3 * f["ff_gather_seconds"] / nf, 2), "x265hip_ff_picture": round(1e3 * f["ff_producer_seconds"] / nf, 2),
                                             "encoder_row_loop_behind_it": round(1e3 * f["ff_replay_seconds"] / nf, 2)},
                          "encoder_own_filters_ms_per_picture": round(1e3 * ct["ff_replay_seconds"] / max(1, ct["ff_cpu_pictures"]), 2),
                          "note": "x265hip_ff_picture = upload of the reconstructed and the source picture and of CUData's arrays, deblocking, SAO statistics of the three planes, download; the row loop behind it = SAO decision and SAO, border extension, PSNR / SSIM, hashes (the encoder's own code)"}
    return out


def c1_host_leg(frames=40):
    """1080p 8-bit, preset ultrafast, the x265 encoder on the host CPU alone with the threading its CLI uses by default (frame threads, WPP, all cores).
    The binary is the reference encoder compiled here from its sources WITHOUT assembly (no nasm in the image): a C-primitives figure, labelled so."""
    import subprocess, tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "x265e2e_8")
    if not os.path.exists(exe):
        exe = os.path.join(ROOT, "oracle", "_ref", "x265tmegpu_8")
        if not os.path.exists(exe):
            return None
    with tempfile.TemporaryDirectory() as td:
        env = dict(os.environ, X265TMEGPU="0", X265LAGPU="0", X265TME="0", X265_CLI_THREADING="1")
        r = subprocess.run([exe, "none", "1920", "1080", str(frames), "ultrafast", os.path.join(td, "c1.hevc")], capture_output=True, text=True, env=env, timeout=600)
        if r.returncode != 0:
            return {"measured": "this run: FAILED", "stderr": r.stderr[-300:]}
        info = json.loads(r.stdout.strip().splitlines()[-1])
    return {"value": info["fps"], "unit": "frames/s", "measured": "this run", "config": "BASELINE C1: 1920x1080 8-bit synthetic clip, preset ultrafast, %d frames, host CPU only (%d cores), the CLI's default threading" % (frames, os.cpu_count() or 0),
            "asm": False, "note": "the reference encoder built from its sources without assembly (no nasm / yasm on this box: cpu_baseline.asm); the x86 asm build would be faster"}


def filters_leg(depth, steps):
    """In-loop filters and picture statistics after reconstruction: 8 coded 4:2:0 pictures of 1920x1080 resident in HBM ->
    deblocking (in place), SAO statistics of the three planes, SAO of the three planes (out of place), SSIM of the luma plane and the SSD of the
    three planes against the source.  Every stage of picture 0 is compared with the oracle (the CPU restatement pinned to the reference's own
    Deblock / SAO classes and to the encoder's reported SSIM / PSNR); the oracle's clock on that picture is the host number (one core, a port)."""
    import ctypes as C
    import time
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from deblock_util import I8, U8, coded_picture, descriptor, run_oracle
    from oracle_py import Oracle
    from x265hip_pkg.frame import FrameApi
    api, ora = FrameApi(depth), Oracle(depth)
    W, H, ctu, F = 1920, 1080 - 1080 % 8, 64, 8
    pic = coded_picture(depth, W, H, ctu, 21)
    rng = np.random.default_rng(5)
    pm = (1 << depth) - 1
    src = [np.clip(p.astype(np.int64) + rng.integers(-3, 4, p.shape) * (1 << (depth - 8)), 0, pm).astype(p.dtype) for p in pic["planes"]]   # the "source" the statistics compare with
    n_ctu = ((W + ctu - 1) // ctu) * ((H + ctu - 1) // ctu)
    prm = np.zeros((3, n_ctu, 6), np.int32)
    for c in range(3):
        for a in range(n_ctu):
            t = int(rng.integers(-1, 5)) if c != 2 else int(prm[1, a, 0])
            prm[c, a, 0] = t
            if t == 4:
                prm[c, a, 1] = rng.integers(0, 32); prm[c, a, 2:] = rng.integers(-7, 8, 4)
            elif t >= 0:
                prm[c, a, 2:] = (rng.integers(0, 8), rng.integers(0, 8), -rng.integers(0, 8), -rng.integers(0, 8))
    P = lambda x: C.c_void_p(x.data_ptr())
    shapes = [(H, W), (H // 2, W // 2), (H // 2, W // 2)]
    d_pristine = [api.to_device(p.reshape(-1)) for p in pic["planes"]]
    d_src = [api.to_device(p.reshape(-1)) for p in src]
    d_rec = [[torch.empty_like(x) for x in d_pristine] for _ in range(F)]
    d_out = [[torch.empty_like(x) for x in d_pristine] for _ in range(F)]
    arrs = {k: api.to_device(np.ascontiguousarray(pic[k]).reshape(-1)) for k in U8 + I8 + ("mv0", "mv1")}
    desc = descriptor(pic, lambda k: arrs[k].data_ptr())
    d_prm = [api.to_device(prm[c].reshape(-1)) for c in range(3)]
    d_stats = [[torch.zeros(n_ctu * 2 * 5 * 32, dtype=torch.int32, device="cuda") for _ in range(3)] for _ in range(F)]
    api.lib.x265hip_ssim_workspace.restype = C.c_size_t
    d_ws = torch.zeros(api.lib.x265hip_ssim_workspace(W, H) // 4, dtype=torch.float32, device="cuda")
    nrows = (H + ctu - 1) // ctu
    d_rs = [torch.zeros(nrows, dtype=torch.float32, device="cuda") for _ in range(F)]; d_rc = [torch.zeros(nrows, dtype=torch.int32, device="cuda") for _ in range(F)]
    d_fr = [torch.zeros(2, dtype=torch.float64, device="cuda") for _ in range(F)]
    d_ssd = [torch.zeros(3, dtype=torch.int64, device="cuda") for _ in range(F)]
    st, L, esz = api.stream(), api.lib, d_pristine[0].element_size()

    def restore():
        for f in range(F):
            for c in range(3):
                d_rec[f][c].copy_(d_pristine[c])

    def deblock():
        for f in range(F):
            api.h.check(L.x265hip_deblock_frame(st, C.byref(desc), P(d_rec[f][0]), C.c_ssize_t(W), P(d_rec[f][1]), P(d_rec[f][2]), C.c_ssize_t(W // 2), None))

    def sao_stats():
        for f in range(F):
            for c in range(3):
                h, w = shapes[c]
                api.h.check(L.x265hip_sao_stats_frame(st, P(d_src[c]), P(d_rec[f][c]), C.c_ssize_t(w), w, h, ctu if c == 0 else ctu // 2, 0, 0 if c == 0 else 2, P(d_stats[f][c])))

    def sao_apply():
        for f in range(F):
            for c in range(3):
                h, w = shapes[c]
                api.h.check(L.x265hip_sao_apply_frame(st, P(d_rec[f][c]), P(d_out[f][c]), C.c_ssize_t(w), w, h, ctu if c == 0 else ctu // 2, P(d_prm[c])))

    def quality():
        for f in range(F):
            api.h.check(L.x265hip_ssim_frame(st, P(d_out[f][0]), C.c_ssize_t(W), P(d_src[0]), C.c_ssize_t(W), W, H, ctu, P(d_ws), P(d_rs[f]), P(d_rc[f]), P(d_fr[f])))
            for c in range(3):
                h, w = shapes[c]
                api.h.check(L.x265hip_plane_ssd(st, P(d_src[c]), P(d_out[f][c]), C.c_ssize_t(w), w, h, C.c_void_p(d_ssd[f].data_ptr() + 8 * c)))

    def timed(fn, reps, before=None):
        tot = 0.0
        for _ in range(reps + 1):
            if before:
                before()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            tot += e0.elapsed_time(e1) if _ else 0.0
        return tot / reps
    ms = {"deblock": timed(deblock, steps, restore), "sao_stats": timed(sao_stats, steps), "sao_apply": timed(sao_apply, steps), "ssim_ssd": timed(quality, steps)}
    total = sum(ms.values())
    px = F * W * H

    # ... and through the picture-batched entry points: ONE launch per stage (and plane) for all F pictures
    from deblock_util import DeblockPic
    class Job(C.Structure):
        _fields_ = [("pic", DeblockPic), ("Y", C.c_void_p), ("Cb", C.c_void_p), ("Cr", C.c_void_p), ("bsOut", C.c_void_p)]
    elems = [h * w for h, w in shapes]
    s_pristine = [torch.cat([d_pristine[c]] * F) for c in range(3)]
    s_src = [torch.cat([d_src[c]] * F) for c in range(3)]
    s_rec = [torch.empty_like(x) for x in s_pristine]; s_out = [torch.empty_like(x) for x in s_pristine]
    jobs = (Job * F)()
    for f in range(F):
        jobs[f].pic = desc
        jobs[f].Y = s_rec[0].data_ptr() + f * elems[0] * esz; jobs[f].Cb = s_rec[1].data_ptr() + f * elems[1] * esz; jobs[f].Cr = s_rec[2].data_ptr() + f * elems[2] * esz
    d_jobs = api.to_device(np.frombuffer(bytes(jobs), np.uint8).copy())
    s_prm = [torch.cat([d_prm[c]] * F) for c in range(3)]
    s_stats = [torch.zeros(F * n_ctu * 320, dtype=torch.int32, device="cuda") for _ in range(3)]
    s_ws = torch.zeros(F * d_ws.numel(), dtype=torch.float32, device="cuda")
    s_rs = torch.zeros(F * nrows, dtype=torch.float32, device="cuda"); s_rc = torch.zeros(F * nrows, dtype=torch.int32, device="cuda")
    s_fr = torch.zeros(2 * F, dtype=torch.float64, device="cuda"); s_ssd = torch.zeros(3 * F, dtype=torch.int64, device="cuda")

    def restore_stack():
        for c in range(3):
            s_rec[c].copy_(s_pristine[c])

    def batched():
        api.h.check(L.x265hip_deblock_pictures(st, P(d_jobs), jobs, F, C.c_ssize_t(W), C.c_ssize_t(W // 2)))
        for c in range(3):
            h, w = shapes[c]
            cs = ctu if c == 0 else ctu // 2
            api.h.check(L.x265hip_sao_stats_pictures(st, P(s_src[c]), P(s_rec[c]), C.c_ssize_t(w), w, h, cs, 0, 0 if c == 0 else 2, P(s_stats[c]), F, C.c_int64(elems[c])))
            api.h.check(L.x265hip_sao_apply_pictures(st, P(s_rec[c]), P(s_out[c]), C.c_ssize_t(w), w, h, cs, P(s_prm[c]), F, C.c_int64(elems[c])))
            api.h.check(L.x265hip_plane_ssd_pictures(st, P(s_src[c]), P(s_out[c]), C.c_ssize_t(w), w, h, C.c_void_p(s_ssd.data_ptr() + 8 * F * c), F, C.c_int64(elems[c]), C.c_int64(elems[c])))
        api.h.check(L.x265hip_ssim_pictures(st, P(s_out[0]), C.c_ssize_t(W), P(s_src[0]), C.c_ssize_t(W), W, H, ctu, P(s_ws), P(s_rs), P(s_rc), P(s_fr), F, C.c_int64(elems[0]), C.c_int64(elems[0])))
    ms_batched = timed(batched, steps, restore_stack)
    # the same work with every picture's chain on its own HIP stream: the kernels are small (a 1080p plane does not fill 256 CUs), so chains of
    # different pictures overlap instead of queueing behind each other's launch gaps
    streams = [torch.cuda.Stream() for _ in range(F)]

    def chain(f, sh):
        h_ = C.c_void_p(sh)
        api.h.check(L.x265hip_deblock_frame(h_, C.byref(desc), P(d_rec[f][0]), C.c_ssize_t(W), P(d_rec[f][1]), P(d_rec[f][2]), C.c_ssize_t(W // 2), None))
        for c in range(3):
            h, w = shapes[c]
            cs = ctu if c == 0 else ctu // 2
            api.h.check(L.x265hip_sao_stats_frame(h_, P(d_src[c]), P(d_rec[f][c]), C.c_ssize_t(w), w, h, cs, 0, 0 if c == 0 else 2, P(d_stats[f][c])))
            api.h.check(L.x265hip_sao_apply_frame(h_, P(d_rec[f][c]), P(d_out[f][c]), C.c_ssize_t(w), w, h, cs, P(d_prm[c])))
            api.h.check(L.x265hip_plane_ssd(h_, P(d_src[c]), P(d_out[f][c]), C.c_ssize_t(w), w, h, C.c_void_p(d_ssd[f].data_ptr() + 8 * c)))
        api.h.check(L.x265hip_ssim_frame(h_, P(d_out[f][0]), C.c_ssize_t(W), P(d_src[0]), C.c_ssize_t(W), W, H, ctu, P(d_wsf[f]), P(d_rs[f]), P(d_rc[f]), P(d_fr[f])))

    d_wsf = [torch.zeros_like(d_ws) for _ in range(F)]
    def all_chains():
        here = torch.cuda.current_stream()
        for f in range(F):
            streams[f].wait_stream(here)
            chain(f, streams[f].cuda_stream)
        for f in range(F):
            here.wait_stream(streams[f])
    ms_streams = timed(all_chains, steps, restore)
    # ... and captured once as a hipGraph (fork into the per-picture streams, join), replayed: the ~170 launches cost no host time any more
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        all_chains()
    ms_graph = timed(graph.replay, steps, restore)
    nbytes = sum(p.nbytes for p in pic["planes"])
    out = {"what": "deblock (in place) -> SAO statistics -> SAO (out of place) -> SSIM + SSD, 3 planes of %d coded 4:2:0 pictures, one call per picture and plane" % F,
           "frame": "%dx%d" % (W, H), "pictures": F, "ms": {k: round(v, 4) for k, v in ms.items()}, "ms_per_picture": round(total / F, 4),
           "mpixels_per_s": round(px / (total * 1e-3) / 1e6, 1),
           "one_stream_per_picture": {"ms": round(ms_streams, 4), "ms_per_picture": round(ms_streams / F, 4), "mpixels_per_s": round(px / (ms_streams * 1e-3) / 1e6, 1)},
           "hipgraph_of_the_streams": {"ms": round(ms_graph, 4), "ms_per_picture": round(ms_graph / F, 4), "mpixels_per_s": round(px / (ms_graph * 1e-3) / 1e6, 1)},
           "picture_batched_entry_points": {"ms": round(ms_batched, 4), "ms_per_picture": round(ms_batched / F, 4), "mpixels_per_s": round(px / (ms_batched * 1e-3) / 1e6, 1),
                                            "launches": "one per stage and plane for all %d pictures (x265hip_*_pictures)" % F},
           "algorithmic_GBps": {"deblock": round(4 * nbytes * F / (ms["deblock"] * 1e-3) / 1e9, 1), "sao_stats": round(2 * nbytes * F / (ms["sao_stats"] * 1e-3) / 1e9, 1),
                                "sao_apply": round(2 * nbytes * F / (ms["sao_apply"] * 1e-3) / 1e9, 1), "ssim_ssd": round((2 * nbytes + 2 * pic["planes"][0].nbytes) * F / (ms["ssim_ssd"] * 1e-3) / 1e9, 1)}}
    # the same chain through the oracle on one picture: results identical, its clock = the host number
    LO = ora.lib
    N = lambda a: C.c_void_p(a.ctypes.data)
    t0 = time.perf_counter()
    o_rec = run_oracle(ora, pic)
    t1 = time.perf_counter()
    o_stats = []
    for c in range(3):
        h, w = shapes[c]
        o = np.zeros(n_ctu * 2 * 5 * 32, np.int32)
        LO.xo_sao_stats_frame(N(src[c]), N(o_rec[c]), C.c_ssize_t(w), w, h, ctu if c == 0 else ctu // 2, 0, 0 if c == 0 else 2, N(o))
        o_stats.append(o)
    t2 = time.perf_counter()
    o_out = []
    for c in range(3):
        h, w = shapes[c]
        o = np.zeros_like(o_rec[c]); pc = np.ascontiguousarray(prm[c])
        LO.xo_sao_apply_frame(N(o_rec[c]), N(o), C.c_ssize_t(w), w, h, ctu if c == 0 else ctu // 2, N(pc))
        o_out.append(o)
    t3 = time.perf_counter()
    rs, rc = np.zeros(nrows, np.float32), np.zeros(nrows, np.uint32)
    tot, cnt = C.c_double(0), C.c_uint32(0)
    LO.xo_ssim_frame(N(o_out[0]), C.c_ssize_t(W), N(src[0]), C.c_ssize_t(W), W, H, ctu, N(rs), N(rc), C.byref(tot), C.byref(cnt))
    LO.xo_plane_ssd.restype = C.c_uint64
    o_ssd = [int(LO.xo_plane_ssd(N(src[c]), N(o_out[c]), C.c_ssize_t(shapes[c][1]), shapes[c][1], shapes[c][0])) for c in range(3)]
    t4 = time.perf_counter()
    restore(); deblock(); sao_stats(); sao_apply(); quality(); torch.cuda.synchronize()
    for c in range(3):
        assert np.array_equal(d_rec[0][c].cpu().numpy().view(o_rec[c].dtype).reshape(shapes[c]), o_rec[c]), "filters: deblocked plane %d differs from the oracle" % c
        assert np.array_equal(d_stats[0][c].cpu().numpy(), o_stats[c]), "filters: SAO statistics of plane %d differ from the oracle" % c
        assert np.array_equal(d_out[0][c].cpu().numpy().view(o_out[c].dtype).reshape(shapes[c]), o_out[c]), "filters: SAO output plane %d differs from the oracle" % c
    fr = d_fr[0].cpu().numpy()
    assert fr[0] == tot.value and int(fr[1]) == cnt.value, "filters: SSIM differs from the oracle"
    restore_stack(); batched(); torch.cuda.synchronize()                # the batched forms: every picture of the stack = the per-picture results
    for c in range(3):
        for f in (0, F - 1):
            assert torch.equal(s_rec[c][f * elems[c]:(f + 1) * elems[c]], d_rec[0][c]) and torch.equal(s_out[c][f * elems[c]:(f + 1) * elems[c]], d_out[0][c]), "filters: batched plane differs"
        assert torch.equal(s_stats[c][:n_ctu * 320], d_stats[0][c]), "filters: batched statistics differ"
    assert torch.equal(s_fr[:2], d_fr[0]) and [int(v) for v in s_ssd.cpu().numpy()[::F]] == o_ssd, "filters: batched SSIM / SSD differ"
    assert [int(v) for v in d_ssd[0].cpu().numpy()] == o_ssd, "filters: SSD differs from the oracle"
    cpu_ms = {"deblock": (t1 - t0) * 1e3, "sao_stats": (t2 - t1) * 1e3, "sao_apply": (t3 - t2) * 1e3, "ssim_ssd": (t4 - t3) * 1e3}
    out["ssim"] = fr[0] / fr[1]
    out["cpu_port"] = {"kind": "port", "cores": 1, "sample": "the same chain on one of the pictures through the oracle (C, -O2), every plane / statistic / SSIM identical to the GPU's",
                       "ms_per_picture": {k: round(v, 2) for k, v in cpu_ms.items()}, "mpixels_per_s": round(W * H / (sum(cpu_ms.values()) * 1e-3) / 1e6, 1),
                       "gpu_over_one_core": round((px / (min(total, ms_streams, ms_graph, ms_batched) * 1e-3)) / (W * H / (sum(cpu_ms.values()) * 1e-3)), 1)}
    return out


def lookahead_leg(depth, steps):
    """Lookahead frame costs: 32 source pictures of 1920x1080 resident in HBM -> half-resolution planes, intra
    costs and every (p0, b, p1) estimate a bframes=3 lookahead asks about inside that window, in one batch.  Two of the
    estimates are run through the REFERENCE's own Lookahead classes (oracle/_ref/x265la_*, when present) on the same pictures:
    results must be identical, and its clock gives the host number next to ours."""
    import subprocess, tempfile
    import torch
    from x265hip_pkg.lookahead import LookaheadBatch, minigop_estimates, pan_clip
    W, H, N = 1920, 1080, 32
    est = minigop_estimates(N, 3)
    lb = LookaheadBatch(depth, W, H, N, len(est))
    frames = pan_clip(W, H, N, depth, seed=11)
    lb.upload(frames)
    lb.set_estimates(est)

    def timed(fn, reps):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms_low = timed(lb.build_lowres, steps)
    ms_intra = timed(lb.intra, steps)
    ms_cost = timed(lb.costs, steps)
    nb = sum(1 for (p0, b, p1) in est if p1 > b)
    out = {"what": "Lowres::init + lowresIntraEstimate of %d pictures, then estimateFrameCost of %d estimates (%d P, %d B = %d list searches) in one batch"
                   % (N, len(est), len(est) - nb, nb, len(est) + nb),
           "frame": "%dx%d -> lowres %dx%d blocks of 8x8" % (W, H, lb.g.wcu, lb.g.hcu), "pictures": N, "estimates": len(est),
           "lowres_init_ms": round(ms_low, 4), "intra_ms": round(ms_intra, 4), "cost_batch_ms": round(ms_cost, 4),
           "estimates_per_s": round(len(est) / (ms_cost * 1e-3), 1),
           "mpixels_per_s": round(len(est) * W * H / (ms_cost * 1e-3) / 1e6, 1)}
    ref = os.path.join(ROOT, "oracle", "_ref", "x265la_%d" % depth)
    if os.path.exists(ref):
        sample = [next(e for e in est if e[2] == e[1] and e[1] == 16), next(e for e in est if e[2] > e[1] and e[1] == 16)]
        with tempfile.TemporaryDirectory() as td:
            inp, outp = os.path.join(td, "in.raw"), os.path.join(td, "out.bin")
            np.stack(frames).tofile(inp)
            r = subprocess.run([ref, str(W), str(H), str(N), inp, outp, "0"] + ["%d,%d,%d" % e for e in sample], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-1000:]
            d = open(outp, "rb").read()
        recs, off = [], 0
        while off < len(d):
            n = int(np.frombuffer(d, np.int64, 1, off)[0]); off += 8
            recs.append(np.frombuffer(d, np.int32, n, off)); off += 4 * n
        scores = lb.frame_scores()
        lc = lb.d_lc.cpu().numpy().view(np.uint16).reshape(-1, lb.g.ncu)
        ic = lb.d_intra_cost.cpu().numpy().reshape(N, lb.g.ncu)
        for f in range(N):
            assert np.array_equal(ic[f], recs[1 + 9 * f + 4]), "lookahead: intra costs of picture %d differ from the reference" % f
        base = 1 + 9 * N
        for k, e in enumerate(sample):
            i = est.index(e)
            hdr = recs[base + 7 * k]
            assert int(hdr[6]) == int(scores[i]) and np.array_equal(lc[i], recs[base + 7 * k + 5].astype(np.uint16)), "lookahead: estimate %s differs from the reference" % (e,)
        tm = recs[-1].astype(np.int64) & 0xffffffff
        ns_intra, ns_cost = int(tm[0] | (tm[1] << 32)), int(tm[2] | (tm[3] << 32))
        cpu_est_s = len(sample) / (ns_cost * 1e-9)
        out["reference"] = {"kind": "reference", "cores": 1, "sample": "%d estimates (one P, one B) and %d intra pictures through the reference's Lookahead classes, results identical" % (len(sample), N),
                            "estimates_per_s": round(cpu_est_s, 2), "intra_pictures_per_s": round(N / (ns_intra * 1e-9), 2),
                            "gpu_over_one_core": round(len(est) / (ms_cost * 1e-3) / cpu_est_s, 1)}
    return out


def intra_scan_leg(pipe, depth, steps):
    """Intra mode scan (x265hip_intra_cost_batch) of every CU of sizes 64, 32, 16, 8 of the step's source frames.  The neighbour
    arrays are taken from the source picture (like the lookahead, slicetype.cpp:800-831): above row, left column, top-left;
    the filtered arrays come from x265hip_intra_filter_batch.  A sample of CUs is checked against the reference's primitives."""
    import torch
    from refproc import RefProc, ref_available
    api, T = pipe.api, torch
    W, H, F, m, st = pipe.W, pipe.H, pipe.F, pipe.margin, pipe.stride
    cur = pipe.cur_host
    legs, total_ms, checked = {}, 0.0, 0
    rng = np.random.default_rng(3)
    for lg in (6, 5, 4, 3):
        n_ = 1 << lg
        nx, ny = W // n_, H // n_
        f, by, bx = np.meshgrid(np.arange(F), np.arange(ny), np.arange(nx), indexing="ij")
        org = (f * pipe.plane + (m + by * n_) * st + m + bx * n_).reshape(-1).astype(np.int64)
        n = len(org)
        pitch = 4 * n_ + 1
        nb = np.empty((n, pitch), cur.dtype)
        nb[:, 0] = cur[org - st - 1]
        nb[:, 1:2 * n_ + 1] = cur[(org - st)[:, None] + np.arange(2 * n_)[None, :]]
        nb[:, 2 * n_ + 1:] = cur[(org - 1)[:, None] + (np.arange(2 * n_) * st)[None, :]]
        d_nb = api.to_device(nb.reshape(-1))
        d_flt = T.empty_like(d_nb)
        if n_ <= 32:
            import ctypes
            api.h.check(api.lib.x265hip_intra_filter_batch(api.stream(), n_, ctypes.c_void_p(d_nb.data_ptr()), None, ctypes.c_void_p(d_flt.data_ptr()), None, n))
        else:
            d_flt.copy_(d_nb)
        d_off = api.to_device(org.astype(np.int32))
        d_cost = T.zeros(n * 35, dtype=T.int32, device="cuda")
        ws = api.intra_cost_batch(lg, pipe.d_cur, st, d_off, d_nb, d_flt, pitch, n, d_cost)
        T.cuda.synchronize()
        e0, e1 = T.cuda.Event(enable_timing=True), T.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            api.intra_cost_batch(lg, pipe.d_cur, st, d_off, d_nb, d_flt, pitch, n, d_cost, workspace=ws)
        e1.record(); T.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        legs["intra%d" % n_] = round(ms, 4); total_ms += ms
        if ref_available(depth):
            got = d_cost.cpu().numpy().reshape(n, 35)
            flt = d_flt.cpu().numpy().view(cur.dtype).reshape(n, pitch)
            r = RefProc(depth)
            try:
                for i in rng.choice(n, size=min(12, n), replace=False):
                    exp = np.frombuffer(r.call("intra_costs", [n_, st, int(org[i])], [cur, nb[i], flt[i]])[0], np.int32)
                    assert np.array_equal(got[i], exp), "intra scan: CU %d of size %d differs from the reference" % (i, n_)
                    checked += 1
            finally:
                r.close()
    px = pipe.pixels_per_step
    return {"ms_per_step": round(total_ms, 4), "Mpixels/s": round(px / (total_ms * 1e-3) / 1e6, 1), "kernels_ms": legs,
            "what": "sa8d of the 35 intra predictions of every CU of sizes 64/32/16/8 (4 launches; mode bits and RD decision stay on the host)",
            "checked_vs_reference": "%d CUs identical" % checked if checked else "reference binary not present"}


def streams_leg(lib, depth, W, H, wl, args, pairs, headline_s_per_pass):
    """The same batch stepped as S sub-batches of whole pictures on their own streams (x265hip_batch_desc.streams): pictures are independent, so the LDS-bound 64x64 search
    of one sub-batch runs beside the latency-bound 16x16 / 8x8 searches of another; with S = 2 the two streams alternate on the 64x64 level and are not joined between
    passes.  The line's value is the schedule --splits names (default 2); this leg measures the others."""
    import torch
    from x265hip_pkg.host_batch import HostBatch
    res = {"what": "x265hip_batch_step, desc.streams = S: S sub-batches of whole pictures, each on its own stream; same batch, same results (tests/test_host_batch_gpu.py)",
           "headline_streams": args.splits, "headline_ms_per_pass": round(headline_s_per_pass * 1e3, 4)}
    for S in [v for v in (1, 2, 3) if v != args.splits]:
        hb = HostBatch(lib, depth, W, H, args.frames, qp=args.qp, merange=wl["merange"], method=METHODS[wl["method"]], subme=wl["subme"], tu_log2=args.tu, margin=MARGIN,
                       recon=args.recon, use_planes=not args.no_planes, refs=args.refs, rect=args.rect, streams=S, device=torch.cuda.current_device())
        try:
            hb.upload([p[:1 + args.refs] for p in pairs])
            for _ in range(3):
                hb.step()
            hb.sync()
            reps = max(10, 4 * args.inner)
            t0 = time.perf_counter()
            for _ in range(reps):
                hb.step()
            hb.sync()
            dt = (time.perf_counter() - t0) / reps
            res["streams_%d" % S] = {"ms_per_pass": round(dt * 1e3, 4), "Mpixels/s": round(hb.pixels_per_step / dt / 1e6, 1), "headline_vs_this": round(dt / headline_s_per_pass, 4)}
        finally:
            hb.close()
    return res


def preset_exact_leg(name, pairs, args, headline_s_per_search):
    """The search a preset really asks for, in ONE run of the C++ host, on PRESET_F = 8 pictures: every list-0 reference of the preset (each down the pyramid with its own
    predictor chain), the rectangular PUs of every CU from preset slow on, the asymmetric ones from preset slower on (param.cpp:567-608), the per-PU choice among the
    references, TQ from the chosen reference.  `value` of the line stays the SURVEY 8(d) pipeline (85 PUs per CTU, one reference); this leg says what the preset's own
    load costs next to it.  The size-specialised kernels address a reference's 16 phase planes with 32-bit byte offsets; an 8K batch keeps its planes in groups of two
    pictures for them (one batch, one context: `config.batches`)."""
    import torch
    import x265hip
    from x265hip_pkg.host_batch import HostBatch
    wl, pr = WORKLOADS[name], PRESETS[name]
    depth, W, H, refs, rect, amp = wl["depth"], wl["width"], wl["height"], pr["refs"], pr["rect"], pr["amp"]
    lib = x265hip.HipLib(depth, fill_table=False).lib
    F = len(pairs)
    per = F
    # r05: a batch whose 16-slot plane buffer outgrows the kernels' 32-bit byte offsets keeps its planes in groups of pictures by itself (csrc/xh_ctx.cpp, plane groups): ONE batch of
    # 8 pictures also at 8K.  X265HIP_PE_SPLIT_BATCHES=1: round 4's form, separate batches of what fits (for A/B)
    while os.environ.get("X265HIP_PE_SPLIT_BATCHES") == "1" and 16 * per * (W + 2 * MARGIN) * (H + 2 * MARGIN) * (1 if depth == 8 else 2) >= (1 << 32):
        per //= 2
    hbs = []
    try:
        for k in range(0, F, per):
            hb = HostBatch(lib, depth, W, H, per, qp=args.qp, merange=wl["merange"], method=METHODS[wl["method"]], subme=wl["subme"], tu_log2=args.tu, margin=MARGIN,
                           use_planes=True, refs=refs, rect=rect, amp=amp, streams=min(2, per), device=torch.cuda.current_device())
            hbs.append(hb)
            hb.upload([p[:1 + refs] for p in pairs[k:k + per]])
        def run(n):
            for _ in range(n):
                for hb in hbs:
                    hb.step()
            for hb in hbs:
                hb.sync()
        run(1)
        reps = 3
        hbs[0].set_timing(True)
        t0 = time.perf_counter()
        run(reps)
        dt = (time.perf_counter() - t0) / reps
        kms = hbs[0].read_timing()
        pus = 85 + (340 if rect else 0) + (168 if amp else 0)
        searches = F * (W // 64) * (H // 64) * pus * refs
        out = {"config": {"workload": name, "refs": refs, "rect": bool(rect), "amp": bool(amp), "pus_per_ctu": pus, "pictures": F, "batches": len(hbs), "preset_exact": True,
                          "what": "x265hip_batch_step with desc.refs = %d%s%s: the preset's own motion-search load on P pictures" % (refs, ", desc.rect = 1" if rect else "", ", desc.amp = 1" if amp else "")},
               "ms_per_pass": round(dt * 1e3, 3), "ms_per_picture": round(dt * 1e3 / F, 3), "Mpixels/s": round(F * W * H / dt / 1e6, 1),
               "searched_pus_per_pass": searches, "ns_per_searched_pu": round(dt / searches * 1e9, 3),
               "stage_ms_sub_batch_0": {k: round(v, 4) for k, v in kms.items()}}
        if headline_s_per_search:
            out["vs_headline_per_searched_pu"] = round((dt / searches) / headline_s_per_search, 3)
        # vector instructions per pass (SQ_INSTS_VALU summed over every launch of a pass, committed counter pass) over the live time against the issue peak
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "preset_exact_valu.json")))[name]
            if prof.get("pictures") == F:
                out["roofline_valu"] = {"bound": "valu issue, %d SIMDs x %.1f GHz / %d" % (N_SIMD, GPU_CLOCK_HZ / 1e9, VALU_CYCLES), "valu_wave_instr_per_pass": prof["valu_per_pass"],
                                        "achieved": round(prof["valu_per_pass"] / dt / 1e9, 1), "peak": round(N_SIMD * GPU_CLOCK_HZ / VALU_CYCLES / 1e9, 1), "unit": "G wave-instr/s",
                                        "frac": round(prof["valu_per_pass"] / dt / (N_SIMD * GPU_CLOCK_HZ / VALU_CYCLES), 4), "counter_source": prof.get("source")}
        except (OSError, KeyError, ValueError):
            pass
        return out
    finally:
        for hb in hbs:
            hb.close()


def python_pipeline(args, wl, depth, W, H, pairs):
    """the torch-tensor plumbing of the same step (x265hip_pkg.pipeline.FramePipeline): what the optional legs that work on torch tensors use"""
    from x265hip_pkg.frame import FrameApi, mvcost_row
    from x265hip_pkg.pipeline import FramePipeline
    pp = FramePipeline(depth, W, H, args.frames, qp=args.qp, merange=wl["merange"], method=METHODS[wl["method"]], subme=wl["subme"], tu_log2=args.tu, recon=args.recon,
                       cost_row=mvcost_row(depth, args.qp, 1 << 15), api=FrameApi(depth), use_planes=not args.no_planes, refs=1)
    pp.upload([p[:2] for p in pairs])
    return pp


def usable_cores():
    """what this process may run on: the logical CPUs, its affinity mask, and the cgroup's CPU quota where one is set"""
    n = os.cpu_count() or 1
    info = {"cpu_count": n}
    try:
        info["affinity"] = len(os.sched_getaffinity(0)); n = min(n, info["affinity"])
    except (AttributeError, OSError):
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            info["cgroup_quota_cpus"] = round(int(q) / int(per), 2); n = max(1, min(n, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n, info


def cpu_baseline(pipe, depth, n_ctus):
    """T = all host cores: one process per usable core; where that is more than 64, the 64-process figure is measured too (SMT siblings and memory bandwidth can make
    fewer processes the faster job) and the better of the two is the value -- both are listed."""
    from refproc import ref_available
    n, info = usable_cores()
    if not ref_available(depth):
        return cpu_baseline_on(pipe, depth, n_ctus, 1)
    tried = {}
    for c in sorted({n, min(n, 64)}, reverse=True):
        tried[c] = cpu_baseline_on(pipe, depth, n_ctus, c)
    best_c = max(tried, key=lambda c: tried[c]["value"])
    best = dict(tried[best_c])
    best["host"] = info
    best["tried"] = {str(c): {"value": r["value"], "per_core": r["per_core"]} for c, r in tried.items()}
    # the same C table at -O3 for the widest vector ISA of this host (BASELINE.md section 4's stand-in for the asm table: no assembler here or on the GPU box), at the better core
    # count; `value` is the FASTER of the builds, `builds` lists both with their per-core figures
    from refproc import widest_variant
    builds = {"O2": {"value": best["value"], "per_core": best["per_core"], "cores": best["cores"], "flags": "-O2 (no asm)"}}
    var = widest_variant(depth)
    r = None
    if var:
        try:
            r = cpu_baseline_on(pipe, depth, n_ctus, best_c, variant=var)
        except Exception as ex:                                    # (a build that does not run on this host must not cost the line its -O2 baseline)
            builds["O3_x86-64-%s" % var] = {"failed": repr(ex)[:200]}
    if r:
        name = "O3_x86-64-%s" % var
        builds[name] = {"value": r["value"], "per_core": r["per_core"], "cores": r["cores"], "flags": "-O3 -march=x86-64-%s (no asm; auto-vectorised C, %s)" % (var, "AVX-512" if var == "v4" else "AVX2"),
                        "sample": r["sample"]}
        if r["value"] > best["value"]:
            keep = {k: best[k] for k in ("host", "tried")}
            best = dict(r)
            best.update(keep)
            best["build"] = name
    best.setdefault("build", "O2")
    best["builds"] = builds
    best["builds_note"] = "asm not built: no nasm / yasm and no system libx265 on this host (asm_probe); the reference's TestBench asserts asm == C bit-exactly, so the C table is the specification and its -O3 vector build the nearest timed stand-in"
    return best


def cpu_baseline_on(pipe, depth, n_ctus, cores, variant=""):
    """The reference's own C primitives + motionEstimate (oracle/_ref, built from /root/reference sources) on the same
    tasks the GPU just processed, one process per core of `cores`; falls back to the restated oracle when the binary is
    missing.  Also cross-checks the sample's results against the GPU's (parity in the same run)."""
    from refproc import RefProc, ref_available
    from x265hip_pkg.host_batch import LEVELS
    ctus_per_frame = (pipe.W // 64) * (pipe.H // 64)
    n_ctus = min(n_ctus, ctus_per_frame * pipe.F)
    n_frames = (n_ctus + ctus_per_frame - 1) // ctus_per_frame
    elems = n_frames * pipe.plane
    cur, ref = pipe.cur_host[:elems], pipe.ref_host[:elems]
    res = {lv: pipe.results(lv) for lv in LEVELS}
    # work list: (kind, level, task index) limited to the sampled CTUs, split round-robin over processes
    jobs = []
    for lv in LEVELS:
        t = pipe.tasks_host[lv]
        per_frame = (pipe.W // lv) * (pipe.H // lv)
        keep = []
        for f in range(n_frames):
            cnt = per_frame if (f + 1) * ctus_per_frame <= n_ctus else int(per_frame * (n_ctus - f * ctus_per_frame) / ctus_per_frame)
            keep.extend(range(f * per_frame, f * per_frame + cnt))
        keep = np.asarray(keep)
        qmvp = np.zeros((len(keep), 2), np.int64)
        if lv != 64:
            qmvp = res[2 * lv][t["mvpFrom"][keep]]["mv"].astype(np.int64)
        d = pipe.merange << 2
        lo, hi = t["mvmin"][keep].astype(np.int64), t["mvmax"][keep].astype(np.int64)
        mn = np.minimum(hi, np.maximum(lo, qmvp - d)) >> 2
        mx = np.minimum(hi, np.maximum(lo, qmvp + d)) >> 2
        mx[:, 1] = np.maximum(mx[:, 1], mn[:, 1])
        rows = np.concatenate([t["curOff"][keep].astype(np.int64)[:, None], mn, mx, qmvp], axis=1)
        jobs.append((lv, keep, rows))
    n_tu = 1 << pipe.tu_log2
    tu = pipe.tu_host
    tu_per_frame = (pipe.W // n_tu) * (pipe.H // n_tu)
    tu_keep = np.arange(min(len(tu), int(tu_per_frame * n_ctus / ctus_per_frame)))
    tu_rows = np.concatenate([tu["curOff"][tu_keep].astype(np.int64)[:, None],
                              res[pipe.mv_level][tu["mvFrom"][tu_keep]]["mv"].astype(np.int64)], axis=1)
    sample_px = n_ctus * 4096

    if ref_available(depth):
        kind = "reference"
        import tempfile
        procs = [RefProc(depth, variant) for _ in range(cores)]
        t0 = time.time()
        # the sample's two planes go to the processes once, through a file in memory (one process per core: 256 pipes would carry them 5 times each otherwise)
        shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        with tempfile.NamedTemporaryFile(dir=shm, suffix=".planes") as pf:
            pf.write(cur.tobytes()); pf.write(ref.tobytes()); pf.flush()
            import threading as _th
            def _load(p_):
                assert RefProc.i32(p_.call("bench_planes", [cur.nbytes, ref.nbytes], [pf.name.encode()])[0]) == 1, "x265ref could not read the sample planes"
            ths = [_th.Thread(target=_load, args=(p_,)) for p_ in procs]
            for th in ths:
                th.start()
            for th in ths:
                th.join()

        def build(reps):
            pend = []
            for p_i in range(cores):
                reqs = []
                for lv, keep, rows in jobs:
                    sl = rows[p_i::cores]
                    if len(sl):
                        ints = [lv, lv, pipe.stride, pipe.merange, pipe.method, pipe.subme, pipe.qp, len(sl), reps] + sl.reshape(-1).tolist()
                        reqs.append(("bench_me", ints, (lv, keep[p_i::cores])))
                sl = tu_rows[p_i::cores]
                if len(sl):
                    reqs.append(("bench_tq", [pipe.tu_log2, pipe.stride, pipe.qp, 85, len(sl), reps] + sl.reshape(-1).tolist(), ("tq", tu_keep[p_i::cores])))
                pend.append(reqs)
            return pend
        pending = build(1)
        # issue every process its first request before collecting anything, so all cores work concurrently
        import threading
        ns_per_proc = [0] * cores
        mismatches = []
        coeff, numsig = pipe.coeffs()
        coeff = coeff.reshape(-1, n_tu * n_tu)

        def worker(i):
            for op, ints, tag in pending[i]:
                out = procs[i].call(op, ints, [])              # planes: bench_planes above
                ns_per_proc[i] += int(np.frombuffer(out[0], np.int64)[0])
                if op == "bench_me":
                    lv, idx = tag
                    r = np.frombuffer(out[1], np.int32).reshape(-1, 3)
                    g = res[lv][idx]
                    ok = (r[:, 0] == g["mv"][:, 0]) & (r[:, 1] == g["mv"][:, 1]) & (r[:, 2] == g["cost"])
                    if not ok.all():
                        mismatches.append(("me", lv, int((~ok).sum())))
                else:
                    _, idx = tag
                    ns = np.frombuffer(out[1], np.uint32)
                    w = (np.arange(n_tu * n_tu, dtype=np.int64) + 1)
                    cs = int((coeff[idx].astype(np.int64) * w).sum() & 0xFFFFFFFF)
                    if pipe.refs == 1 and not (np.array_equal(ns, numsig[idx].astype(np.uint32)) and cs == int(np.frombuffer(out[2], np.uint32)[0])):
                        mismatches.append(("tq", n_tu, 1))
        def run_all():
            threads = [threading.Thread(target=worker, args=(i,)) for i in range(cores)]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
        run_all()                                   # pass 1: parity cross-check + calibration
        total = sum(ns_per_proc) / 1e9
        reps = int(max(1, min(400, round(max(12.0, 0.25 * cores) / max(total, 1e-3)))))   # aim at ~12 CPU-seconds of reference work, and a quarter second per core at least
        if reps > 1:
            pending = build(reps)
            ns_per_proc = [0] * cores
            run_all()
        wall = time.time() - t0
        for p in procs:
            p.close()
        busy = max(ns_per_proc) / 1e9 / reps
        total_cpu = sum(ns_per_proc) / 1e9
        parity = "identical" if not mismatches else "MISMATCH %s" % mismatches[:3]
    else:
        kind = "port"
        from oracle_py import Oracle
        ora = Oracle(depth)
        cores = 1
        n_ctus = min(n_ctus, 32)
        sample_px = n_ctus * 4096
        t0 = time.time()
        rng = np.random.default_rng(0)
        checked = 0
        for lv, keep, rows in jobs:
            per_ctu = (64 // lv) ** 2
            for j in range(min(len(rows), n_ctus * per_ctu)):
                r = rows[j]
                ora.me(lv, lv, cur, pipe.stride, int(r[0]), ref, pipe.stride, int(r[0]), [int(v) for v in r[1:5]], (int(r[5]), int(r[6])), [],
                       pipe.merange, pipe.method, pipe.subme, pipe.cost_row_host)
                checked += 1
        for j in range(min(len(tu_rows), n_ctus * (64 // n_tu) ** 2)):
            r = tu_rows[j]
            ora.tq_tu(pipe.tu_log2, cur, pipe.stride, int(r[0]), ref, pipe.stride, int(r[0]), (int(r[1]), int(r[2])), pipe.qp, 85)
        wall = busy = time.time() - t0
        total_cpu, reps = busy, 1
        parity = "not cross-checked"
    if pipe.refs > 1:
        busy *= pipe.refs             # one of the references was searched on the CPU (its chain is cross-checked); the others cost the same
    return {"value": round(sample_px / busy / 1e6, 3), "unit": "Mpixels/s", "cores": cores, "kind": kind, "per_core": round(sample_px * (reps if kind == "reference" else 1) / max(total_cpu, 1e-9) / 1e6 / max(1, pipe.refs), 4),
            "sample": "%d CTU64 (%d luma px) of the same batch: ME pyramid (85 PUs/CTU) + %dx%d DCT+quant, %s; %d repetition(s), %.1f CPU-seconds in total, busiest core %.3f s per repetition; "
                      "results vs GPU: %s" % (n_ctus, sample_px, n_tu, n_tu,
                                               "reference C primitives + motionEstimate (no asm%s), one process per core" % (", -O3 -march=x86-64-" + variant if variant else "") if kind == "reference"
                                               else "restated oracle, single thread", reps, total_cpu, busy, parity)}


MARGIN = 96                  # PicYuv-style padding of the synthetic planes (FramePipeline's default)
# the preset's own search: references per list-0 (param.cpp:567-608) and the rectangular PUs (preset slow and up); medium has no rect
# what the presets set for the motion search (param.cpp:567-608): medium ref 3; slow ref 4 + rect; slower ref 5 + rect + amp
PRESETS = {"1080p8_medium": dict(refs=3, rect=False, amp=False), "2160p10_slow": dict(refs=4, rect=True, amp=False), "4320p10_slower": dict(refs=5, rect=True, amp=True)}
PRESET_REFS = {k: v["refs"] for k, v in PRESETS.items()}
PRESET_RECT = {k: v["rect"] for k, v in PRESETS.items()}
PRESET_F = 8                 # pictures of every preset-exact leg


def _make_pair(W, H, depth, seed, refs=1):
    """(source, reference 0 [, further references]): reference r > 0 is reference 0 displaced a little further plus its own noise -- an older picture of the same scene"""
    from x265hip_pkg.synth import frame_pair
    cur, ref, _, _ = frame_pair(W, H, depth, seed=seed, margin=MARGIN, max_shift=24)
    out = [cur, ref]
    rng = np.random.default_rng(77000 + seed)
    for r in range(1, refs):
        o = np.roll(ref, (2 * r, -3 * r), (0, 1)).astype(np.float32) + rng.normal(0, (1.0 + r) * (1 << (depth - 8)), ref.shape).astype(np.float32)
        out.append(np.clip(o, 0, (1 << depth) - 1).astype(ref.dtype))
    return tuple(out)


def asm_probe():
    """north_star asks for the reference's AVX2 / AVX-512 asm path beside ours.  It needs nasm to build and the GPU box receives only this
    repository, so look for what the host offers at run time and say what was found."""
    import ctypes.util
    import shutil
    found = {"nasm": shutil.which("nasm") or shutil.which("yasm"), "x265_cli": shutil.which("x265"), "libx265": ctypes.util.find_library("x265")}
    if found["x265_cli"]:
        note = "a system x265 CLI exists (%s) but exposes no primitive-level entry; its whole-encoder fps is not this metric" % found["x265_cli"]
    elif found["libx265"]:
        note = "a system libx265 exists (%s) but the EncoderPrimitives table is not part of its public API" % found["libx265"]
    elif found["nasm"]:
        note = "nasm is present (%s) but /root/reference is not on this box, so there is nothing to assemble" % found["nasm"]
    else:
        note = "no nasm/yasm, no system x265/libx265 on this host: the x86 asm table cannot be built or found; the C (no-asm) table of the reference is what ran"
    return {"found": found, "asm_baseline": None, "note": note}


def profile_figure(workload, which):
    """Per-launch counter figures that cannot be collected inside an untraced run (HBM bytes, SQ_INSTS_VALU): read from the committed
    rocprofv3 summaries, labelled with where they come from.  {} when no summary exists for the workload."""
    path = os.path.join(ROOT, "profiles", "%s_%s.json" % (which, workload))
    if not os.path.exists(path):
        return {}, None
    try:
        d = json.load(open(path))
    except Exception:
        return {}, None
    src = d.pop("_source", None) if isinstance(d, dict) else None
    return d, "profiles/%s_%s.json (rocprofv3 --pmc pass%s)" % (which, workload, ", " + src if src else "")


def dry_run(args):
    """Launcher and bookkeeping without a GPU (world_size-2 gloo test): the same rank set-up, barrier, MAX-over-ranks time and
    whole-job aggregate as the real run, with a sleep standing in for the step."""
    import torch.distributed as dist
    import x265hip  # noqa: F401
    from x265hip_pkg.sharding import init_ranks, max_over_ranks, whole_job_mpixels_per_s, rank_frame_seeds
    rank, local_rank, world = init_ranks(args.gpus, "gloo")
    wl = WORKLOADS[args.workload]
    px = args.frames * wl["width"] * wl["height"]
    seeds = rank_frame_seeds(rank, args.frames)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (1 + rank))
    if world > 1:
        dist.barrier()
    dt = max_over_ranks(time.perf_counter() - t0, dist if world > 1 else None)
    owned = [seeds]
    if world > 1:
        owned = [None] * world
        dist.all_gather_object(owned, seeds)
    if rank == 0:
        print(json.dumps({"metric": "Mpixels/s ME+DCT+quant on 4K CTU batches", "value": round(whole_job_mpixels_per_s(world, px, args.steps, dt), 2), "unit": "Mpixels/s",
                          "n_gpus": world, "rccl_ranks": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "dry-run",
                          "config": {"workload": args.workload, "backend": "gloo", "frames_owned": owned}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_leg_child(name, argv_extra=(), timeout=900):
    """An auxiliary leg in a process of its own (python bench.py --leg NAME + this run's arguments): returns its JSON object, or {"failed": ...} -- the line survives a leg that dies."""
    import subprocess
    keep = []
    skip = False
    for a in sys.argv[1:]:          # this run's own arguments, minus what belongs to the launcher
        if skip:
            skip = False
            continue
        if a in ("--gpus", "--leg"):
            skip = True
            continue
        keep.append(a)
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env["X265HIP_LEG_DEVICE"] = os.environ.get("LOCAL_RANK", "0")
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + keep + ["--leg", name] + list(argv_extra), capture_output=True, text=True, env=env, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"failed": "leg %s: no result within %d s" % (name, timeout)}
    try:
        if r.returncode == 0:
            return json.loads(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        pass
    return {"failed": "leg %s: exit status %d" % (name, r.returncode), "stderr": r.stderr[-400:]}


def leg_main(args):
    """python bench.py --leg NAME: one auxiliary leg, alone in this process"""
    import x265hip  # noqa: F401
    import multiprocessing
    name, _, arg = args.leg.partition(":")
    dev = int(os.environ.get("X265HIP_LEG_DEVICE", "0"))
    if name == "tme_producer":
        import torch
        torch.cuda.set_device(dev)
        print(json.dumps(tme_producer_leg(WORKLOADS[args.workload]["depth"])))
        return
    if name == "preset_exact":
        w_ = WORKLOADS[arg]
        pool = multiprocessing.get_context("fork").Pool(min(PRESET_F, max(1, os.cpu_count() or 1)))
        pairs = pool.starmap(_make_pair, [(w_["width"], w_["height"], w_["depth"], 5000 + k, PRESETS[arg]["refs"]) for k in range(PRESET_F)])
        pool.close(); pool.join()
        import torch
        torch.cuda.set_device(dev)
        print(json.dumps(preset_exact_leg(arg, pairs, args, None)))
        return
    if name == "streams":
        wl = WORKLOADS[args.workload]
        pool = multiprocessing.get_context("fork").Pool(min(args.frames, max(1, os.cpu_count() or 1)))
        pairs = pool.starmap(_make_pair, [(wl["width"], wl["height"], wl["depth"], sd, args.refs) for sd in range(args.frames)])
        pool.close(); pool.join()
        import torch
        torch.cuda.set_device(dev)
        lib = x265hip.HipLib(wl["depth"], fill_table=False).lib
        print(json.dumps(streams_leg(lib, wl["depth"], wl["width"], wl["height"], wl, args, pairs, float(arg))))
        return
    raise SystemExit("bench.py --leg: unknown leg %r" % args.leg)


def main():
    args = parse()
    if args.leg:
        return leg_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly with --gpus N: become the launcher of N ranks (one per GPU), hand their exit status back
        import x265hip  # noqa: F401  (registers the package under an importable name)
        from x265hip_pkg.sharding import spawn_ranks
        if not args.cpu_dry_run:
            import torch
            have = torch.cuda.device_count() if torch.cuda.is_available() else 0
            if have < args.gpus:
                raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this host (one rank per GPU, no oversubscription)" % (args.gpus, have))
        sys.exit(spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    if args.cpu_dry_run:
        return dry_run(args)
    # the rank's synthetic frames first, in worker processes (a 4K pair takes ~9 s of numpy; forked before torch / HIP are initialised)
    import x265hip  # noqa: F401
    from x265hip_pkg.sharding import rank_frame_seeds
    wl = WORKLOADS[args.workload]
    depth, W, H = wl["depth"], wl["width"], wl["height"]
    seeds = rank_frame_seeds(int(os.environ.get("RANK", "0")), args.frames)          # independent frames per rank, no overlap
    import multiprocessing
    # close() + join(), not the context manager: its terminate() sends SIGTERM to the workers, and under rocprofv3 (whose signal handler is inherited by the forked
    # workers) a worker then never exits and the run hangs in wait4 (seen in the r02 counter passes)
    exact = [w for w in args.preset_exact_workloads.split(",") if w in PRESETS] if (not args.no_preset_exact and int(os.environ.get("RANK", "0")) == 0) else []
    pool = multiprocessing.get_context("fork").Pool(min(len(seeds), max(1, (os.cpu_count() or 1) // max(1, args.gpus))))
    pairs = pool.starmap(_make_pair, [(W, H, depth, sd, args.refs) for sd in seeds])
    pool.close(); pool.join()
    import torch
    import torch.distributed as dist
    import x265hip  # noqa: F401
    from x265hip_pkg.frame import mvcost_row
    from x265hip_pkg.host_batch import HostBatch, LEVELS
    from x265hip_pkg.sharding import init_ranks, max_over_ranks, whole_job_mpixels_per_s

    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible (the HIP path has no CPU fallback)")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))      # before the process group exists: RCCL binds its communicator to the current device
    rank, local_rank, world = init_ranks(args.gpus, "nccl", torch.cuda.device_count())

    # The step is driven by the C++ host of the path (include/x265hip_ctx.h: x265hip_batch_*, csrc/xh_ctx.cpp) -- what a C++ encoder links; Python only calls it.
    # torch is here for the process group (barrier, max over ranks) and the device-wide synchronisation around the timed region.
    lib = x265hip.HipLib(depth, fill_table=False).lib
    pipe = HostBatch(lib, depth, W, H, args.frames, qp=args.qp, merange=wl["merange"], method=METHODS[wl["method"]], subme=wl["subme"], tu_log2=args.tu, margin=MARGIN,
                     recon=args.recon, use_planes=not args.no_planes, refs=args.refs, rect=args.rect, streams=args.splits, device=local_rank)
    pipe.cost_row_host = mvcost_row(depth, args.qp, 1 << 15)
    pipe.set_fused(args.fused)
    pipe.upload([p[:1 + args.refs] for p in pairs])                      # inputs are resident in HBM before the timed region
    assert np.array_equal(pipe.device_plane(1, 0), pairs[0][1].reshape(-1)), "the device's border extension differs from the host-padded plane"

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup * args.inner):
        pipe.step()
    barrier()
    names = pipe.kernel_names()
    # per-stage HIP events (recorded by the host on the stream each stage runs on, x265hip_batch_set_timing) on every 4th step only: an event between two launches
    # keeps the tail of one kernel from overlapping the head of the next (measured: ~5 % of the step time when every launch is bracketed)
    t0 = time.perf_counter()
    serial_stage_times = args.splits > 1       # sub-batches on their own streams overlap: the per-stage times then come from one-stream passes behind the timed region
    for k in range(args.steps):
        pipe.set_timing(k % 4 == 0 and not serial_stage_times)
        for _ in range(args.inner):                 # one step = `inner` passes of the hot path over the resident batch (the timed region of the driver's 20 steps is >= 0.5 s)
            pipe.step()
    barrier()
    dt = time.perf_counter() - t0
    # NOT part of the reported value: the same region (K steps between barriers) four more times -- the spread of the headline's clock over repetitions of 0.5 s each
    repeats = [dt]
    for _ in range(4):
        pipe.set_timing(False)
        barrier()
        tr0 = time.perf_counter()
        for k in range(args.steps * args.inner):
            pipe.step()
        barrier()
        repeats.append(time.perf_counter() - tr0)
    repeats = [max_over_ranks(x, dist if world > 1 else None, device="cuda") for x in repeats]
    t_serial = None
    if serial_stage_times and rank == 0:
        # NOT part of the timed region: the same batch stepped as one sub-batch on one stream (x265hip_batch_step_one_stream, same results), every pass with per-stage events
        pipe.set_timing(False)
        pipe.step_one_stream(); pipe.sync()
        pipe.set_timing(True)
        ts0 = time.perf_counter()
        for _ in range(4):
            pipe.step_one_stream()
        pipe.sync()
        t_serial = (time.perf_counter() - ts0) / 4
        pipe.set_timing(False)
    dt = max_over_ranks(dt, dist if world > 1 else None, device="cuda")
    devices = [torch.cuda.get_device_name(local_rank)]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, "%d:%s" % (local_rank, torch.cuda.get_device_name(local_rank)))

    if rank == 0:
        kms = pipe.read_timing()                    # mean over the sampled passes; stages of sub-batch 0 (one stream: the whole batch)
        bpp = 1 if depth == 8 else 2
        px = pipe.pixels_per_step * args.inner      # pixels of one step
        share = 1.0 if serial_stage_times else pipe.sub_batch_pictures() / args.frames      # the part of the batch one timed launch group covers
        n_tu = 1 << args.tu
        alg = pipe.algorithmic_bytes(whole_batch=serial_stage_times)      # each plane byte once per launch + the records it writes
        # dominant kernel = strictly the longest average launch of the step, whatever it is bound by
        dom = max(kms, key=lambda k: kms[k])
        achieved = alg[dom] / (kms[dom] * 1e-3) / 1e9
        # the strictly longest single KERNEL: where the dominant stage is the 64x64 level of a STAR search it is star64_kernel (the stage = it + the sub-pel launch behind it),
        # bracketed by its own HIP events on the stream it runs on (x265hip_batch_read_kernel_timing; the whole algorithmic traffic of the level is charged to it)
        star_ms = pipe.read_kernel_timing() if hasattr(pipe, "read_kernel_timing") else None
        kernel_name, kernel_ms = dom, kms[dom]
        if dom == "me64" and star_ms and star_ms > 0.5 * kms[dom]:
            kernel_name, kernel_ms = "star64_kernel", star_ms
            achieved = alg[dom] / (kernel_ms * 1e-3) / 1e9
        traffic_all, traffic_src = profile_figure(args.workload, "traffic")
        valu_all, valu_src = profile_figure(args.workload, "valu")
        valu_peak = N_SIMD * GPU_CLOCK_HZ / VALU_CYCLES
        step_bytes = px * ((1 + args.refs) * bpp + 2) + args.inner * sum(len(pipe.tasks_host[lv]) for lv in LEVELS) * 16       # SURVEY 8(d) fused S1-S3: source + reference(s) once, MVs + coefficients out
        step_gbs = step_bytes / (dt / args.steps) / 1e9
        value = whole_job_mpixels_per_s(world, px, args.steps, dt)
        out = {
            "metric": "Mpixels/s ME+DCT+quant on 4K CTU batches (luma source pixels through ME pyramid + MC/DCT/quant)",
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": dist.get_world_size() if world > 1 else 1, "rccl_ranks": world, "devices": devices,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "ms_per_step_spread": {"note": "the timed region and four repetitions of it behind it (same K steps between barriers); value is the FIRST, timed, one", "min": round(min(repeats) / args.steps * 1e3, 4),
                                   "max": round(max(repeats) / args.steps * 1e3, 4), "mean": round(sum(repeats) / len(repeats) / args.steps * 1e3, 4), "regions": len(repeats)},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8" if depth == 8 else "u16", "data": "synthetic",
            "config": {"workload": args.workload, "preset_exact": bool(args.refs == PRESET_REFS.get(args.workload) and args.rect == PRESET_RECT.get(args.workload)),
                       "frame": "%dx%d (CTU-aligned)" % (W, H), "frames_per_step_per_gpu": args.frames * args.inner,
                       "step": "%d passes of the hot path over a resident batch of %d frame pairs" % (args.inner, args.frames), "host": "C++ (x265hip_batch_step, csrc/xh_ctx.cpp)",
                       "streams":
                                  ("2 sub-batches of whole pictures on their own streams, alternating on the 64x64 level, joined at the end of the timed region (x265hip_batch_desc.streams = 2)" if args.splits == 2 else
                                   "%d sub-batches of whole pictures on their own streams" % args.splits if args.splits > 1 else "one stream"),
                       "ctu": 64, "pus_per_ctu": 425 if args.rect else 85, "me": wl["method"], "subme": wl["subme"], "merange": wl["merange"], "refs": args.refs, "qp": args.qp,
                       "tu": "%dx%d" % (n_tu, n_tu), "recon": bool(args.recon), "subpel": "phase planes" if pipe.use_planes else "in-kernel interpolation", "launch": "stage by stage per sub-batch" + ("; the per-stage times of roofline / roofline_valu are NOT from the timed region (its stages overlap): 4 one-stream passes of the same batch behind it, %.4f ms per pass" % (t_serial * 1e3) if serial_stage_times else "; per-stage events on every 4th step (sub-batch 0)"), "sharding": "independent frames per GPU, no collectives"},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "launch_group": {"name": dom, "avg_ms": round(kms[dom], 4), "frac": round(alg[dom] / (kms[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                         "kernel_rule": ("strictly the longest single kernel of a pass, timed by its own HIP events on its stream; it belongs to the longest launch group (all_kernels_ms are groups: me64 = star64_kernel + the sub-pel launch behind it), whose traffic figure is the group's" if kernel_name != dom else "strictly the longest average launch group of a pass") + ("; times of one-stream passes (the whole batch per launch, stages one after the other: they add up to the one-stream pass, %.4f ms, not to ms_per_step / %d = %.4f ms of the overlapped schedule)" % (t_serial * 1e3, args.inner, dt / args.steps / args.inner * 1e3) if serial_stage_times else
                                                                                                      "; a launch covers one of %d sub-batches, stages of different sub-batches run concurrently (their times do not add up to ms_per_step)" % args.splits if args.splits > 1 else ""), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": (traffic_all.get(dom) * share if traffic_all.get(dom) is not None else None), "traffic_source": traffic_src if traffic_all.get(dom) is not None else None,
                         "algorithmic_bytes_per_launch": int(alg[dom]), "avg_launch_ms": round(kernel_ms, 4),
                         "all_kernels_ms": {k: round(v, 4) for k, v in kms.items()},
                         "all_kernels_GBps": {k: round(alg[k] / (v * 1e-3) / 1e9, 2) for k, v in kms.items()},
                         # what the launches actually move through the L2's memory side (committed counter pass: 2 x FETCH_SIZE + WRITE_SIZE, profiles/r03_fetch_calib.txt) over their
                         # live time: the lower levels of the search stream the 16 phase planes at most of the achievable HBM rate
                         "all_kernels_traffic_GBps": ({k: round(traffic_all[k] * share / (v * 1e-3) / 1e9, 1) for k, v in kms.items() if traffic_all.get(k) is not None and v > 0.02} if traffic_all else None),
                         "hbm_achievable_GBps": 6300},
            # the search kernels are vector-issue bound, not bandwidth bound: instructions per launch (SQ_INSTS_VALU of the committed counter pass)
            # over the live launch time, against the chip's wave-instruction issue rate
            "roofline_valu": {"bound": "valu issue", "unit": "G wave-instr/s", "peak": round(valu_peak / 1e9, 1),
                              "peak_rule": "%d SIMDs x %.1f GHz / %d clocks per wave64 instruction" % (N_SIMD, GPU_CLOCK_HZ / 1e9, VALU_CYCLES),
                              "insts_source": valu_src,
                              "kernels": {k: {"insts": int(valu_all[k] * share), "achieved": round(valu_all[k] * share / (kms[k] * 1e-3) / 1e9, 1), "frac": round(valu_all[k] * share / (kms[k] * 1e-3) / valu_peak, 4)}
                                          for k in kms if k in valu_all},
                              "step": ({"insts": int(sum(valu_all[k] for k in kms if k in valu_all) * args.inner), "achieved": round(sum(valu_all[k] for k in kms if k in valu_all) * args.inner / (dt / args.steps) / 1e9, 1),
                                        "frac": round(sum(valu_all[k] for k in kms if k in valu_all) * args.inner / (dt / args.steps) / valu_peak, 4)} if valu_all else None)},
            # the whole step against the HBM bound of SURVEY 8(d): source and reference read once, MVs and coefficients written
            "roofline_step": {"bound": "hbm", "bytes_rule": "pixels x (2*bpp + 2) + 16 B per PU", "bytes": int(step_bytes), "achieved": round(step_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(step_gbs / HBM_PEAK_GBS, 5),
                              "traffic_per_step": (sum(v for k, v in traffic_all.items() if k in kms) * args.inner or None) if traffic_all else None, "traffic_source": traffic_src},
            "e2e_fps": None,
        }
        def leg(name):      # progress on stderr: a leg that takes the process down is then named in the log
            print("[bench] leg %s" % name, file=sys.stderr, flush=True)
        if not args.no_streams_leg and args.frames >= 2:
            leg("streams")
            out["streams"] = run_leg_child("streams:%.9f" % (dt / args.steps / args.inner))
        if exact:
            # one object per BASELINE workload: the preset's own search load on PRESET_F pictures (the headline workload's is measured against the headline's time per search)
            out["preset_exact"] = {}
            for wname in exact:
                per_search = (dt / args.steps / args.inner) / (args.frames * (W // 64) * (H // 64) * 85 * args.refs) if wname == args.workload else None
                leg("preset_exact " + wname)
                r_ = run_leg_child("preset_exact:" + wname, timeout=1200)       # (its own process: own pictures, own contexts)
                if per_search and "ms_per_pass" in r_:
                    r_["vs_headline_per_searched_pu"] = round((r_["ms_per_pass"] * 1e-3 / r_["searched_pus_per_pass"]) / per_search, 3)
                out["preset_exact"][wname] = r_
        e2e_path = os.path.join(ROOT, "profiles", "e2e_fps.json")
        if not args.no_e2e:
            leg("e2e_fps")
            try:
                out["e2e_fps"] = e2e_fps_leg()
            except Exception as ex:                  # the leg runs an external binary: a failure there must not lose the line
                out["e2e_fps"] = {"measured": "this run: FAILED", "error": repr(ex)[:300]}
        if out["e2e_fps"] is None and os.path.exists(e2e_path):
            try:
                out["e2e_fps"] = json.load(open(e2e_path))
            except Exception:
                pass
        if not args.no_e2e:
            try:
                out["c1_host_fps"] = c1_host_leg()
            except Exception as ex:
                out["c1_host_fps"] = {"measured": "this run: FAILED", "error": repr(ex)[:300]}
        if not args.no_tme:
            leg("tme_producer")
            out["tme_producer"] = run_leg_child("tme_producer")
        if args.intra:
            out["intra_scan"] = intra_scan_leg(python_pipeline(args, wl, depth, W, H, pairs), depth, max(2, min(args.steps, 10)))
        if args.lookahead:
            out["lookahead"] = lookahead_leg(depth, max(2, min(args.steps, 10)))
        if args.filters:
            out["filters"] = filters_leg(depth, max(2, min(args.steps, 10)))
        if args.cpu_ctus > 0:
            leg("cpu_baseline")
            out["cpu_baseline"] = cpu_baseline(pipe, depth, args.cpu_ctus)     # rank 0's host cores, after the timed region (the other ranks wait at the barrier below)
            out["cpu_baseline"]["asm"] = asm_probe()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
