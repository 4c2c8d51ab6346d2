#!/bin/bash
# Counter evidence for the ThreadedME chain kernels (run through gpurun): profiles/collect_tme.sh <tag> <preset>
#   -> gpurun_out/<tag>/{kernel_stats.csv, sq.txt, tme_valu.json}
set -u
tag=$1; preset=${2:-slow}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
cmd="python profiles/tme_prof_run.py $preset 6"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $cmd > $out/stats.log 2> $out/stats.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $out/sq1 -- $cmd > /dev/null 2> $out/sq1.err
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $out/sq2 -- $cmd > /dev/null 2> $out/sq2.err
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $cmd > /dev/null 2> $out/fetch.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- $cmd > /dev/null 2> $out/write.err
python profiles/summarize_pmc.py $out/sq1 $out/sq2 $out/fetch $out/write > $out/sq.txt
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
python - "$out" "$preset" <<'PY'
import csv, json, re, sys
out, preset = sys.argv[1], sys.argv[2]
dur = {}
for r in csv.DictReader(open(out + "/kernel_stats.csv")):
    dur[re.sub(r"\(anonymous namespace\)::|^void ", "", r["Name"]).split("(")[0]] = (float(r["AverageNs"]), int(r["Calls"]), float(r["TotalDurationNs"]))
vals = {}
for line in open(out + "/sq.txt"):
    m = re.match(r"(\S.*?)\s+\|\s+(.*)", line)
sys.path.insert(0, "profiles")
import summarize_pmc as S
sq = S.load(out + "/sq1"); f = S.load(out + "/fetch"); w = S.load(out + "/write")
peak = 1024 * 2.4e9 / 4
res = {"_source": "rocprofv3 --pmc passes of `python profiles/tme_prof_run.py %s 6` (1920x1080, 3 P + 3 B pictures), per dispatch averages" % preset, "kernels": {}}
for k, (avg, calls, tot) in sorted(dur.items(), key=lambda kv: -kv[1][2])[:12]:
    e = {"avg_us": round(avg / 1e3, 1), "calls": calls, "total_ms": round(tot / 1e6, 2)}
    if k in sq and "SQ_INSTS_VALU" in sq[k]:
        e["valu_insts"] = int(sq[k]["SQ_INSTS_VALU"]); e["valu_frac_of_issue_peak"] = round(sq[k]["SQ_INSTS_VALU"] / (avg * 1e-9) / peak, 4)
    if k in f and "FETCH_SIZE" in f[k]:
        e["hbm_read_MB"] = round(f[k]["FETCH_SIZE"] * 1024 / 1e6, 2)
    if k in w and "WRITE_SIZE" in w[k]:
        e["hbm_write_MB"] = round(w[k]["WRITE_SIZE"] * 1024 / 1e6, 2)
    res["kernels"][k] = e
json.dump(res, open(out + "/tme_valu.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete
