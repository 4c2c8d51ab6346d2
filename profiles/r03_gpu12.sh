cd $GRAFT_REPO_ROOT
(time python -m pytest tests/test_me_gpu.py tests/test_pipeline_gpu.py tests/test_host_batch_gpu.py tests/test_tme_gpu.py tests/test_tme_producer_gpu.py tests/test_e2e_tme_gpu.py -x -q -m gpu 2>&1 | tail -6) > gpurun_out/r03_rect_tests.txt 2>&1
cat gpurun_out/r03_rect_tests.txt
for v in "--refs 4 --rect --frames 4 --splits 2" "--rect --splits 2" "--splits 2"; do
  python bench.py --steps 6 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e $v 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], round(d['ms_per_step']/5,4), d['roofline']['all_kernels_ms'])"
done > gpurun_out/r03_rect_ab.txt 2>&1
cat gpurun_out/r03_rect_ab.txt
