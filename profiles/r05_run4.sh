#!/bin/bash
# round 5: the bench line (release), its rocprofv3 kernel stats, the bench under the fence build (both modes), and a soak of the release bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/fence_*.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
tail -c 600 gpurun_out/r05_bench.err
python - <<'P'
import json
try:
    d = json.loads([l for l in open("gpurun_out/r05_bench.json") if l.startswith("{")][-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], d.get("ms_per_step_spread"), "roofline", d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
    print("cpu_baseline", {k: d["cpu_baseline"][k] for k in ("value", "cores", "per_core", "kind")})
    e = d.get("e2e_fps") or {}
    print("e2e fps", e.get("fps"), "identical", e.get("bitstream_identical"))
    print("default threading", json.dumps(e.get("default_threading"))[:1500])
except Exception as ex:
    print("bench line unreadable:", ex)
P
for mode in end start; do
  ( time timeout 900 tools/fence_run.sh $mode python bench.py --steps 5 --warmup 2 --no-e2e --cpu-ctus 0 ) > gpurun_out/r05_bench_fence_$mode.json 2> gpurun_out/r05_bench_fence_$mode.err
  echo "fence $mode bench rc $? : $(tail -c 300 gpurun_out/r05_bench_fence_$mode.err | tr '\n' ' ')"
  rm -f gpurun_out/fence_$mode.log
done
export TMPDIR=/tmp
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof -o r05 -- python $OLDPWD/bench.py --steps 10 --warmup 3 --no-e2e --cpu-ctus 0 --no-tme --no-preset-exact --no-streams-leg > /tmp/prof.log 2>&1 )
find /tmp/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05_2160p10_kernel_stats.csv
head -8 gpurun_out/r05_2160p10_kernel_stats.csv | cut -c1-60,200-330
: > gpurun_out/r05_soak.txt
for i in $(seq 1 ${SOAK:-12}); do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-e2e --cpu-ctus 0 > /tmp/soak.json 2> /tmp/soak.err; rc=$?
  echo "run $i rc $rc $(python -c "import json; d=json.loads([l for l in open('/tmp/soak.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1) $(grep -i -c 'memory access fault' /tmp/soak.err) faults" >> gpurun_out/r05_soak.txt
  [ $rc -ne 0 ] && tail -c 1500 /tmp/soak.err >> gpurun_out/r05_soak.txt
done
cat gpurun_out/r05_soak.txt
for mode in end start; do
  ( time timeout 900 tools/fence_run.sh $mode python -m pytest tests/test_e2e_gpu.py -m gpu -q -p no:cacheprovider --timeout=500 ) > gpurun_out/r05_fence_${mode}_e2e_table.txt 2>&1
  tail -n 3 gpurun_out/r05_fence_${mode}_e2e_table.txt | cut -c1-200
  rm -f gpurun_out/fence_$mode.log
done
