cd $GRAFT_REPO_ROOT
(time python -m pytest tests/test_me_gpu.py tests/test_pipeline_gpu.py tests/test_host_batch_gpu.py tests/test_tme_gpu.py tests/test_tme_producer_gpu.py -x -q -m gpu 2>&1 | tail -6) > gpurun_out/r03_uni_tests.txt 2>&1
cat gpurun_out/r03_uni_tests.txt
for v in "--splits 1" "--splits 2" "--splits 1 --workload 1080p8_medium" "--splits 2 --workload 1080p8_medium" "--splits 2 --workload 4320p10_slower --frames 2"; do
  python bench.py --steps 8 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e $v 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], round(d['ms_per_step']/5,4), d['roofline']['all_kernels_ms'])"
done > gpurun_out/r03_uni_ab.txt 2>&1
cat gpurun_out/r03_uni_ab.txt
