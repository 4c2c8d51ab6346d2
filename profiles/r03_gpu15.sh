cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof8k -- python bench.py --workload 4320p10_slower --frames 2 --steps 2 --warmup 1 --inner 1 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/bench_8k_prof.json 2> gpurun_out/bench_8k_prof.err
f=$(find gpurun_out/prof8k -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(r["Name"][:110], r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
PY
cp "$f" gpurun_out/r03_8k_kernel_stats.csv
