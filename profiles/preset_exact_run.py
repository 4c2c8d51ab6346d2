"""python profiles/preset_exact_run.py <workload> <passes>: the preset-exact leg of bench.py (bench.preset_exact_leg's batches: the preset's references, rectangular and
asymmetric PUs on PRESET_F = 8 pictures) stepped <passes> times and nothing else -- the process rocprofv3 wraps for profiles/preset_exact_valu.json
(profiles/collect_preset_exact.sh: SQ_INSTS_VALU summed over every dispatch of a pass)."""
import multiprocessing
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import x265hip  # noqa: E402,F401  (registers the package under an importable name: the pool workers import it)
import bench  # noqa: E402


def main():
    name, passes = sys.argv[1], int(sys.argv[2])
    wl, pr = bench.WORKLOADS[name], bench.PRESETS[name]
    pool = multiprocessing.get_context("fork").Pool(min(bench.PRESET_F, os.cpu_count() or 1))
    pairs = pool.starmap(bench._make_pair, [(wl["width"], wl["height"], wl["depth"], 5000 + k, pr["refs"]) for k in range(bench.PRESET_F)])
    pool.close(); pool.join()
    import torch
    import x265hip
    from x265hip_pkg.host_batch import HostBatch
    torch.cuda.set_device(0)
    lib = x265hip.HipLib(wl["depth"], fill_table=False).lib
    F, per = len(pairs), len(pairs)
    # (one batch, as bench.preset_exact_leg runs it since round 5: a batch too large for the kernels' 32-bit offsets keeps its planes in groups of pictures by itself)
    while os.environ.get("X265HIP_PE_SPLIT_BATCHES") == "1" and 16 * per * (wl["width"] + 2 * bench.MARGIN) * (wl["height"] + 2 * bench.MARGIN) * (1 if wl["depth"] == 8 else 2) >= (1 << 32):
        per //= 2
    hbs = []
    for k in range(0, F, per):
        hb = HostBatch(lib, wl["depth"], wl["width"], wl["height"], per, qp=28, merange=wl["merange"], method=bench.METHODS[wl["method"]], subme=wl["subme"], tu_log2=5, margin=bench.MARGIN,
                       use_planes=True, refs=pr["refs"], rect=pr["rect"], amp=pr["amp"], streams=1, device=0)
        hb.upload([p[:1 + pr["refs"]] for p in pairs[k:k + per]])
        hbs.append(hb)
    for _ in range(passes):
        for hb in hbs:
            hb.step()
        for hb in hbs:
            hb.sync()
    print("passes", passes, "pictures", F, "batches", len(hbs))
    for hb in hbs:
        hb.close()


if __name__ == "__main__":
    main()
