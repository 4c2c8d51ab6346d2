# r04: merange 128 (BASELINE configs[4]): the pattern passes with the band re-centred per pass (star64_recentre_kernel) -- parity, then the 8K pass; preset-exact counters
python -m pytest tests/test_me_gpu.py tests/test_pipeline_gpu.py tests/test_host_batch_gpu.py -q -x 2>&1 | tail -4
python bench.py --workload 4320p10_slower --frames 2 --steps 10 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/r04_8k_a.json 2> gpurun_out/r04_8k_a.err; echo rc=$?
python bench.py --workload 4320p10_slower --frames 2 --steps 10 --warmup 2 --splits 1 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/r04_8k_b.json 2> gpurun_out/r04_8k_b.err; echo rc=$?
python - <<'PY'
import json
for t in "ab":
    try:
        d=json.loads(open("gpurun_out/r04_8k_%s.json"%t).read().strip().splitlines()[-1]); print(t, d["value"], round(d["ms_per_step"]/5,3), d["roofline"]["all_kernels_ms"])
    except Exception as e: print(t, "failed", e, open("gpurun_out/r04_8k_%s.err"%t).read()[-600:])
PY
bash profiles/collect_preset_exact.sh r04_pe > gpurun_out/r04_pe_collect.log 2>&1; tail -45 gpurun_out/r04_pe_collect.log
