# r04: chain kernels, references side by side (RP) against one after the other (X265HIP_TME_SERIAL_REFS=1, experiment objects): per-kernel times inside the real encode
export TMPDIR=/tmp
lib=$GRAFT_REPO_ROOT/x265-mod-by-patman_amd/exp_rp/libx265hip_8.so
for mode in rp serial; do
  for preset in medium slow; do
    if [ $mode = serial ]; then export X265HIP_TME_SERIAL_REFS=1; else unset X265HIP_TME_SERIAL_REFS; fi
    X265TME_PROF=1 X265TMEGPU=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04_rp_${mode}_$preset -- oracle/_ref/x265tmegpu_8 $lib 1920 1088 10 $preset /tmp/o_${mode}_$preset.hevc 2>&1 | grep -E "^x265hip_tme:|^\{" | cut -c1-200
    f=$(find gpurun_out/r04_rp_${mode}_$preset -name "*kernel_stats.csv" | head -1)
    echo "== $mode $preset"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", round(tot/1e6,2))
for r in rows[:9]:
    print("%-70s calls %5s avg_us %9.1f total_ms %8.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
    find gpurun_out/r04_rp_${mode}_$preset -name "*kernel_trace.csv" -delete
  done
done
md5sum /tmp/o_*.hevc
