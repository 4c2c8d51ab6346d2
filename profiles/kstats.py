#!/usr/bin/env python3
"""kstats.py <kernel_stats.csv> <kernel_trace.csv>: per-kernel averages; for kernels launched several times per step (the halves of a split
search) also the average of the k-th launch of a step."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-100s calls %5s  avg %9.1f us  total %9.2f ms  %5.1f%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
tr = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
seq = collections.defaultdict(list)
for r in tr:
    seq[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
steps = min((len(v) for k, v in seq.items() if "subpel_planes" in k), default=12)
for k, v in seq.items():
    if len(v) > steps and len(v) % steps == 0 and "me_kernel" in k:
        m = len(v) // steps
        print(k[:90], "-- launches per step", m, ["%.1f us" % (sum(v[i::m]) / steps) for i in range(m)])
