#!/bin/bash
# profiles/collect_preset_exact.sh <tag> [workload ...]: SQ_INSTS_VALU per pass of the three preset-exact legs (or of the ones named: the others keep their profiles/preset_exact_valu.json entries) (rocprofv3 --pmc, kernel trace only) -> gpurun_out/<tag>/preset_exact_valu.json
# (copy to profiles/preset_exact_valu.json: bench.py's preset_exact.*.roofline_valu divides it by the live pass time)
set -u
tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
P=2
WL="${*:-1080p8_medium 2160p10_slow 4320p10_slower}"
for w in $WL; do
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $out/pe_$w -- python profiles/preset_exact_run.py $w $P > $out/pe_$w.log 2> $out/pe_$w.err
  tail -1 $out/pe_$w.log
done
python - $out $P "$tag, code $(cat profiles/.commit 2>/dev/null || echo unknown)" <<'PY'
import csv, glob, json, os, re, sys
out, P, src = sys.argv[1], int(sys.argv[2]), sys.argv[3]
res = json.load(open("profiles/preset_exact_valu.json")) if os.path.exists("profiles/preset_exact_valu.json") else {}
for w in ("1080p8_medium", "2160p10_slow", "4320p10_slower"):
    tot, n, per = 0.0, 0, {}
    for f in glob.glob(os.path.join(out, "pe_" + w, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "SQ_INSTS_VALU":
                v = float(r["Counter_Value"]); tot += v; n += 1
                k = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"]).split("(")[0][:70]; per[k] = per.get(k, 0.0) + v
    if n:
        top = sorted(per.items(), key=lambda kv: -kv[1])[:6]
        res[w] = {"valu_per_pass": int(tot / P), "pictures": 8, "dispatches_per_pass": n // P, "source": "rocprofv3 --pmc SQ_INSTS_VALU over profiles/preset_exact_run.py %s %d (%s)" % (w, P, src),
                  "largest_kernels_share": {k: round(v / tot, 3) for k, v in top}}
json.dump(res, open(os.path.join(out, "preset_exact_valu.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete
