#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_e2e_tme_gpu.py tests/test_e2e_la_gpu.py tests/test_e2e_ff_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 ) > gpurun_out/r05_e2e_after_queues.txt 2>&1
grep -E "passed|failed|^E " gpurun_out/r05_e2e_after_queues.txt | tail -4 | cut -c1-250
profiles/r05_m2_final_ab.sh > /dev/null 2>&1
sort gpurun_out/r05_m2_final_ab.txt | cut -c1-140
