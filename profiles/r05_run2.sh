#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_tme_producer_gpu.py tests/test_e2e_tme_gpu.py -m gpu -q -s -p no:cacheprovider ) > gpurun_out/r05_tme_bands.txt 2>&1
tail -30 gpurun_out/r05_tme_bands.txt
for mode in end start; do
  ( time timeout 1500 tools/fence_run.sh $mode python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/r05_fence_${mode}_gputest.txt 2>&1
  grep -c "alloc #" gpurun_out/fence_$mode.log > gpurun_out/fence_${mode}_allocs.txt 2>/dev/null
  tail -c 100000 gpurun_out/fence_$mode.log > gpurun_out/fence_${mode}_tail.log; rm -f gpurun_out/fence_$mode.log
  tail -n 12 gpurun_out/r05_fence_${mode}_gputest.txt
done
