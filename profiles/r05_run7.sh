#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05_lds
export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d gpurun_out/r05_lds/p -- python bench.py --steps 3 --warmup 1 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg --splits 1 --inner 1 > /dev/null 2> gpurun_out/r05_lds/err.txt
python profiles/summarize_pmc.py gpurun_out/r05_lds/p > gpurun_out/r05_lds_counters.txt 2>&1
find gpurun_out/r05_lds -name "*.csv" -delete
grep -E "star64|me_kernel<8|me_kernel<16" gpurun_out/r05_lds_counters.txt | cut -c1-400 | head -6; tail -c 300 gpurun_out/r05_lds/err.txt
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 ) > gpurun_out/r05_gputest_final.txt 2>&1
tail -n 6 gpurun_out/r05_gputest_final.txt | cut -c1-300
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r05_final_bench.json 2> gpurun_out/r05_final_bench.err
tail -c 400 gpurun_out/r05_final_bench.err
python - <<'P'
import json
d = json.loads([l for l in open("gpurun_out/r05_final_bench.json") if l.startswith("{")][-1])
print("value", d["value"], "ms/step", d["ms_per_step"], d.get("ms_per_step_spread"))
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "per_core", "kind", "host", "tried")})
e = d.get("e2e_fps") or {}
print("e2e fps", e.get("fps"), "identical", e.get("bitstream_identical"))
print("default threading", json.dumps(e.get("default_threading"))[:2500])
print("clocks", json.dumps(e.get("encoder_clocks_ms_per_picture"))[:900])
P
