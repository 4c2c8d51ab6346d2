#!/bin/bash
# Kernel trace of the GPU TME producer inside the reference encoder (slow preset, 1280x704).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
lib=$(cd $R && python -c "import x265hip; print(x265hip.lib_path(8))")
P=${1:-slow}
X265TMEGPU=1 timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tmetrace_$P -o t -- $R/oracle/_ref/x265tmegpu_8 $lib 1280 704 6 $P /tmp/p.hevc ref=1 > /dev/null 2>&1
f=$(find $R/gpurun_out/tmetrace_$P -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-90s calls %6s total %10.1f us avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
