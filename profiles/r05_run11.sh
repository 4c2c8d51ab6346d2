#!/bin/bash
# end of round 5: the ME / batch tests under the fence (final kernels), the release suite, the driver's bench line, the profile passes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/fence_*.log
for mode in end start; do
  ( time timeout 900 tools/fence_run.sh $mode python -m pytest tests/test_host_batch_gpu.py tests/test_me_gpu.py tests/test_pipeline_gpu.py tests/test_tme_producer_gpu.py -m gpu -q -p no:cacheprovider --timeout=800 -k "not every_pu" ) > gpurun_out/r05_fence_${mode}_final_me.txt 2>&1
  grep -E "passed|failed" gpurun_out/r05_fence_${mode}_final_me.txt | tail -1
  rm -f gpurun_out/fence_$mode.log
done
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 ) > gpurun_out/r05_gputest_final.txt 2>&1
tail -n 5 gpurun_out/r05_gputest_final.txt | cut -c1-300
timeout 1500 profiles/collect.sh r05_v2_2160p10 > gpurun_out/r05_collect2.log 2>&1
python - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/r05_v2_2160p10/bench.json") if l.startswith("{")][-1])
print("value", d["value"], "ms/step", d["ms_per_step"], d.get("ms_per_step_spread"), d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "per_core", "kind", "host")})
e = d.get("e2e_fps") or {}
print("e2e", e.get("fps"), e.get("bitstream_identical"), json.dumps((e.get("default_threading") or {}).get("fps_runs")))
print("8K", json.dumps((d.get("preset_exact") or {}).get("4320p10_slower", {}).get("ms_per_picture")), (d.get("preset_exact") or {}).get("4320p10_slower", {}).get("config", {}).get("batches"))
P
tail -n 12 gpurun_out/r05_v2_2160p10/kstats.txt | cut -c1-160
