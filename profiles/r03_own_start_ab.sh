# the 64x64 level without its start-stage launch (star64_kernel measures the zero predictor out of its band): bench.py --fused 4 = with the launch (rounds 1-2), default = without
python -m pytest tests/test_host_batch_gpu.py tests/test_pipeline_gpu.py tests/test_me_gpu.py -x -q 2>&1 | tail -2
run() { name=$1; shift; python bench.py --steps 8 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg "$@" > gpurun_out/os_$name.json 2> gpurun_out/os_$name.err; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/os_%s.json"%n).read().strip().splitlines()[-1])
    print(n, "ms per pass %.3f" % (d["ms_per_step"]/5), d["roofline"]["all_kernels_ms"])
except Exception as e:
    print(n, "failed", e); print(open("gpurun_out/os_%s.err"%n).read()[-800:])
PY
}
for rep in a b; do
run three_1$rep --fused 4 --splits 1
run two_1$rep --splits 1
run three_2$rep --fused 4 --splits 2
run two_2$rep --splits 2
done
