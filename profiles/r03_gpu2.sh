cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(time python -m pytest tests/test_host_batch_gpu.py tests/test_ctx_gpu.py tests/test_e2e_tme_gpu.py tests/test_tme_producer_gpu.py tests/test_tme_gpu.py -x -q -m gpu 2>&1 | tail -15) > gpurun_out/r03_gputest2.txt 2>&1
for v in "--splits 1" "--splits 2" "--splits 3" "--splits 4"; do
  python bench.py --steps 8 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e $v 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'])"
done > gpurun_out/r03_split_cpp.txt 2>&1
python bench.py --steps 8 --warmup 2 --no-tme --no-e2e --refs 4 --rect --frames 4 --splits 2 > gpurun_out/r03_bench_preset.json 2> gpurun_out/r03_bench_preset.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench2.json 2> gpurun_out/r03_bench2.err
cat gpurun_out/r03_gputest2.txt gpurun_out/r03_split_cpp.txt; tail -c 600 gpurun_out/r03_bench_preset.err; tail -c 600 gpurun_out/r03_bench2.err
python -c "
import json
for f in ('gpurun_out/r03_bench_preset.json','gpurun_out/r03_bench2.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'], (d.get('cpu_baseline') or {}).get('sample','')[-40:])"
