#!/usr/bin/env python3
"""Mean per-dispatch value of every counter, per kernel, from rocprofv3 `--pmc ... --output-format csv` output.

usage: summarize_pmc.py <dir-with-*_counter_collection.csv> [...]   -> one line per (kernel, pass) on stdout
       summarize_pmc.py --traffic <fetch_dir> <write_dir>            -> JSON {bench kernel name: HBM bytes per launch}

FETCH_SIZE / WRITE_SIZE are reported in KiB.  Calibration on this box: kern_planes writes
exactly 15 planes (WRITE_SIZE x 1024 = 324.5 MB vs 324.4 MB computed) and the 4-byte-per-lane loads of tq_kernel
give FETCH_SIZE x 1024 = the two planes it reads, so no x2 correction is applied for these access widths (the
guide's x2 applies to 16-byte-per-lane streaming reads, which these kernels do not issue).
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
    """me_kernel<G, MAXPIX, ...> / tq_kernel<N> / subpel_planes_kernel -> the names bench.py uses."""
    if kernel.startswith("subpel_planes_kernel"):
        return "planes"
    m = re.match(r"tq_kernel<(\d+)", kernel)
    if m:
        return "tq%s" % m.group(1)
    m = re.match(r"me_kernel<(\d+), (\d+)", kernel)
    if m:
        return {4096: "me64", 1024: "me32", 256: "me16", 64: "me8"}.get(int(m.group(2)))
    return None


if __name__ == "__main__":
    if sys.argv[1] == "--traffic":
        fetch, write = load(sys.argv[2]), load(sys.argv[3])
        out = {}
        for k in fetch:
            b = bench_name(k)
            if b and "FETCH_SIZE" in fetch[k] and k in write:
                out[b] = int((fetch[k]["FETCH_SIZE"] + write[k].get("WRITE_SIZE", 0.0)) * 1024)
                out[b + "_read"] = int(fetch[k]["FETCH_SIZE"] * 1024)
                out[b + "_write"] = int(write[k].get("WRITE_SIZE", 0.0) * 1024)
        print(json.dumps(out, indent=1, sort_keys=True))
    else:
        for d in sys.argv[1:]:
            for k, cs in sorted(load(d).items()):
                print(k, {c: int(v) for c, v in sorted(cs.items())})
