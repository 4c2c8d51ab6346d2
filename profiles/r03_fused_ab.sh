# the three lower levels of the pyramid in one launch, a wavefront per 32x32 quadrant (kern_me_pyr.hip), against a launch per level: parity tests, then the headline pass
python -m pytest tests/test_host_batch_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -3
run() { name=$1; shift; python bench.py --steps 8 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg "$@" > gpurun_out/fu_$name.json 2> gpurun_out/fu_$name.err; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/fu_%s.json"%n).read().strip().splitlines()[-1])
    print(n, "ms per pass %.3f" % (d["ms_per_step"]/5), d["roofline"]["all_kernels_ms"])
except Exception as e:
    print(n, "failed", e); print(open("gpurun_out/fu_%s.err"%n).read()[-800:])
PY
}
for rep in a b; do
run levels_1$rep --splits 1
run fused168_1$rep --fused 1 --splits 1
run fused32168_1$rep --fused 2 --splits 1
run levels_2$rep --splits 2
run fused168_2$rep --fused 1 --splits 2
done
