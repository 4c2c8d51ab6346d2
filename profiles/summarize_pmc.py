#!/usr/bin/env python3
"""Mean per-dispatch value of every counter, per kernel, from rocprofv3 `--pmc ... --output-format csv` output.

usage: summarize_pmc.py <dir-with-*_counter_collection.csv> [...]   -> one line per (kernel, pass) on stdout
       summarize_pmc.py --traffic <fetch_dir> <write_dir>            -> JSON {bench kernel name: HBM bytes per launch}

FETCH_SIZE / WRITE_SIZE are reported in KiB.  Calibration on this box: kern_planes writes exactly 15 planes (WRITE_SIZE x 1024 = 324.5 MB vs 324.4 MB computed): writes
need no correction.  FETCH_SIZE reports HALF of the bytes of a coalesced streaming read at EVERY access width these kernels use -- profiles/micro/fetch_calib.hip streams
1 GiB with 4, 8, 12 and 16 bytes per lane: 524,296-524,298 KiB each (profiles/r03_fetch_calib.txt) -- so the read side is doubled (FETCH_CORRECTION), as the MI355X guide
prescribes for gfx950.  (Rounds 1-2 did not double it: tq_kernel's reading 'equal to its two planes' was its line over-fetch of 2x meeting the counter's 1/2.)
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def load(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def bench_name(kernel):
    """me_kernel<G, MAXPIX, ...> / star64_kernel / tq_kernel<N> / subpel_planes_kernel -> the names bench.py uses (one name per launch GROUP of a
    step: me64 = the two halves of the split 64x64 kernel + star64_kernel)."""
    if kernel.startswith("subpel_planes_kernel"):
        return "planes"
    if kernel.startswith("star64_kernel"):
        return "me64"
    if re.match(r"tq_kernel<(\d+)", kernel):
        return "tq"
    m = re.match(r"me_kernel<(\d+), (\d+)", kernel)
    if m:
        return {4096: "me64", 1024: "me32", 256: "me16", 64: "me8"}.get(int(m.group(2)))
    return None


def per_step(d):
    """{bench name: {counter: total per step}}: counter values summed over ALL dispatches of the kernels behind a bench name, divided by the number
    of steps of the run (= dispatches of the phase-plane kernel, launched once per step)."""
    tot, steps = defaultdict(lambda: defaultdict(float)), 0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        first = rows[0]["Counter_Name"] if rows else None
        for r in rows:
            k = short(r["Kernel_Name"])
            b = bench_name(k)
            if b:
                tot[b][r["Counter_Name"]] += float(r["Counter_Value"])
            if k.startswith("subpel_planes_kernel") and r["Counter_Name"] == first:
                steps += 1
    return {b: {c: v / max(steps, 1) for c, v in cs.items()} for b, cs in tot.items()}


FETCH_CORRECTION = 2.0


if __name__ == "__main__":
    if sys.argv[1] == "--traffic":
        fetch, write = per_step(sys.argv[2]), per_step(sys.argv[3])
        out = {"_source": (sys.argv[4] if len(sys.argv) > 4 else "FETCH_SIZE + WRITE_SIZE (KiB) per step") + "; read side = 2 x FETCH_SIZE (profiles/r03_fetch_calib.txt)"}
        for b in fetch:
            if "FETCH_SIZE" in fetch[b] and b in write:
                out[b] = int((FETCH_CORRECTION * fetch[b]["FETCH_SIZE"] + write[b].get("WRITE_SIZE", 0.0)) * 1024)
                out[b + "_read"] = int(FETCH_CORRECTION * fetch[b]["FETCH_SIZE"] * 1024)
                out[b + "_write"] = int(write[b].get("WRITE_SIZE", 0.0) * 1024)
        print(json.dumps(out, indent=1, sort_keys=True))
    elif sys.argv[1] == "--valu":
        sq = per_step(sys.argv[2])
        out = {"_source": sys.argv[3] if len(sys.argv) > 3 else "SQ_INSTS_VALU per step"}
        for b in sq:
            if "SQ_INSTS_VALU" in sq[b]:
                out[b] = int(sq[b]["SQ_INSTS_VALU"])
        print(json.dumps(out, indent=1, sort_keys=True))
    else:
        for d in sys.argv[1:]:
            for k, cs in sorted(load(d).items()):
                print(k, {c: int(v) for c, v in sorted(cs.items())})
