set -x
python -m pytest tests/test_me_gpu.py -x -q -k "merange_128 or star" 2>&1 | tail -3
python -m pytest tests/test_pipeline_gpu.py -x -q -k "8k" 2>&1 | tail -3
python bench.py --workload 4320p10_slower --frames 2 --steps 6 --warmup 2 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg > gpurun_out/bench_8k.json 2> gpurun_out/bench_8k.err
tail -c 3000 gpurun_out/bench_8k.json
