cd $GRAFT_REPO_ROOT
(time python -m pytest tests/test_me_gpu.py tests/test_pipeline_gpu.py tests/test_host_batch_gpu.py tests/test_tme_gpu.py -x -q -m gpu 2>&1 | tail -6) > gpurun_out/r03_tile_tests.txt 2>&1
cat gpurun_out/r03_tile_tests.txt
for v in 0 1024 2048 4096 7168 0; do
  X265HIP_ME_VARIANT=$v python bench.py --steps 8 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --splits 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('variant $v', d['value'], round(d['ms_per_step']/5,4), d['roofline']['all_kernels_ms'])"
done > gpurun_out/r03_tile_ab.txt 2>&1
X265HIP_ME_VARIANT=0 python bench.py --steps 8 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --workload 1080p8_medium 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1080p8 tile', d['value'], round(d['ms_per_step']/5,4), d['roofline']['all_kernels_ms'])" >> gpurun_out/r03_tile_ab.txt 2>&1
X265HIP_ME_VARIANT=7168 python bench.py --steps 8 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --workload 1080p8_medium 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1080p8 old', d['value'], round(d['ms_per_step']/5,4), d['roofline']['all_kernels_ms'])" >> gpurun_out/r03_tile_ab.txt 2>&1
cat gpurun_out/r03_tile_ab.txt
