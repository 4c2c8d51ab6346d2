python -m pytest tests/test_me_gpu.py tests/test_pipeline_gpu.py tests/test_host_batch_gpu.py -x -q 2>&1 | tail -3
for w in 2160p10_slow 1080p8_medium; do for sp in 1 2; do
python bench.py --workload $w --splits $sp --steps 10 --warmup 3 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/pp3_${w}_$sp.json 2> gpurun_out/pp3_${w}_$sp.err
python - $w $sp <<'PY'
import json,sys
w,sp=sys.argv[1:3]
d=json.loads(open("gpurun_out/pp3_%s_%s.json"%(w,sp)).read().strip().splitlines()[-1])
print(w, "splits", sp, "value %.0f Mpx/s, ms per pass %.3f" % (d["value"], d["ms_per_step"]/5), d["roofline"]["all_kernels_ms"], "sum %.3f" % sum(d["roofline"]["all_kernels_ms"].values()))
PY
done; done
python bench.py --workload 4320p10_slower --frames 2 --steps 6 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('8K ms per pass %.3f'%(d['ms_per_step']/5), d['roofline']['all_kernels_ms'])"
