# tiled phase planes (slots 1..15 as 16 x 4-pixel tiles of one 128-byte line; 16-bit library) against row-major ones (bench.py --fused 8): parity tests, then the headline pass
python -m pytest tests/test_host_batch_gpu.py tests/test_pipeline_gpu.py tests/test_me_gpu.py tests/test_tq_gpu.py -x -q 2>&1 | tail -3
run() { name=$1; shift; python bench.py --steps 8 --warmup 2 --no-tme --no-e2e --no-preset-exact --no-streams-leg "$@" > gpurun_out/ti_$name.json 2> gpurun_out/ti_$name.err; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/ti_%s.json"%n).read().strip().splitlines()[-1])
    cb=d.get("cpu_baseline") or {}
    print(n, "ms per pass %.3f" % (d["ms_per_step"]/5), d["roofline"]["all_kernels_ms"], ("| CPU sample: " + cb.get("sample","")[-30:]) if cb else "")
except Exception as e:
    print(n, "failed", e); print(open("gpurun_out/ti_%s.err"%n).read()[-1200:])
PY
}
run tiled_check --fused 8 --splits 2 --cpu-ctus 1020
for rep in a b; do
run rows_1$rep --splits 1 --cpu-ctus 0
run tiled_1$rep --fused 8 --splits 1 --cpu-ctus 0
run rows_2$rep --splits 2 --cpu-ctus 0
run tiled_2$rep --fused 8 --splits 2 --cpu-ctus 0
done
