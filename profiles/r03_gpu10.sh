cd $GRAFT_REPO_ROOT
(time python -m pytest tests/test_host_batch_gpu.py tests/test_ctx_gpu.py -x -q -m gpu 2>&1 | tail -5) > gpurun_out/r03_gputest10.txt 2>&1
cat gpurun_out/r03_gputest10.txt
for v in "--splits 2" "--splits 1 --band-rows 4" "--splits 2 --band-rows 4" "--splits 4 --band-rows 4" "--splits 4 --band-rows 2" "--splits 8 --band-rows 2" "--splits 4 --band-rows 8" "--splits 8 --band-rows 4" "--splits 2 --band-rows 17" "--splits 4 --band-rows 17" "--splits 2"; do
  python bench.py --steps 8 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e $v 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], round(d['ms_per_step']/5,4), d['roofline']['all_kernels_ms'])"
done > gpurun_out/r03_band_ab.txt 2>&1
cat gpurun_out/r03_band_ab.txt
