# r04: 8K (configs[4]) raster kernel A/B: chunk width XS_NIC, workgroups per CU XS_RASTER_WGS, window piece prefetched in registers or not (quick_variant builds xw_<nic>_<wgs>_<pf>)
export TMPDIR=/tmp
for v in 16_2_1 16_2_0 16_3_0 12_3_0 8_3_0; do
  X265HIP_LIBDIR=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/xw_$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_8kw_$v -- python bench.py --workload 4320p10_slower --frames 2 --steps 6 --warmup 2 --splits 1 --inner 1 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/r04_8kw_$v.json 2> gpurun_out/r04_8kw_$v.err
  f=$(find gpurun_out/r04_8kw_$v -name "*kernel_stats.csv" | head -1)
  echo "== nic_wgs_prefetch $v"; python - "$f" gpurun_out/r04_8kw_$v.json <<'PY'
import csv,sys,json
rows=list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:3]: print("%-90s calls %4s avg_us %9.1f" % (r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3))
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print("pass ms", round(d["ms_per_step"],3), d["roofline"]["all_kernels_ms"])
except Exception as e: print("bench failed", e)
PY
  find gpurun_out/r04_8kw_$v -name "*kernel_trace.csv" -delete
done
