# r04: star64_kernel -- the start SAD inside the first costing batch (exp_s0), and the whole first pattern pass as one batch on top (exp_s1), against the release library
bash profiles/ab.sh r04_s64ab release exp_s0 exp_s1 2>&1 | tee gpurun_out/r04_star64_onebatch_ab.txt
python -m pytest tests/test_e2e_la_gpu.py -q -x -s 2>&1 | grep -E "passed|failed|e2e la batch|Error|assert" | tail -8
python -m pytest tests/test_host_batch_gpu.py tests/test_tme_gpu.py tests/test_e2e_tme_gpu.py -q -x 2>&1 | tail -4
