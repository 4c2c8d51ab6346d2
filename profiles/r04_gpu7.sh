# r04: 8K (configs[4]) per-kernel times: the band re-centred per pass against the round-3 form (general kernel from memory; experiment objects, X265HIP_STAR64_NO_RECENTRE=1)
export TMPDIR=/tmp
for mode in recentre memory; do
  if [ $mode = memory ]; then export X265HIP_STAR64_NO_RECENTRE=1; else unset X265HIP_STAR64_NO_RECENTRE; fi
  X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/exp_rc timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_8k_$mode -- python bench.py --workload 4320p10_slower --frames 2 --steps 6 --warmup 2 --splits 1 --inner 1 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/r04_8k_$mode.json 2> gpurun_out/r04_8k_$mode.err
  f=$(find gpurun_out/r04_8k_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode"; python - "$f" gpurun_out/r04_8k_$mode.json <<'PY'
import csv,sys,json
rows=list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:10]: print("%-90s calls %4s avg_us %9.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3))
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print("pass ms", round(d["ms_per_step"],3), d["roofline"]["all_kernels_ms"])
except Exception as e: print("bench failed", e)
PY
  find gpurun_out/r04_8k_$mode -name "*kernel_trace.csv" -delete
done
