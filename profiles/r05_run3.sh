#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/fence_*.log
profiles/r05_ft_debug.sh > /dev/null 2>&1
grep -E "^####|^==|hevc" gpurun_out/r05_ft_debug.txt | cut -c1-150
( time timeout 900 python -m pytest tests/test_tme_producer_gpu.py tests/test_e2e_tme_gpu.py -m gpu -q -s -p no:cacheprovider --timeout=240 -k "frame_threads or bands" ) > gpurun_out/r05_tme_bands.txt 2>&1
tail -n 25 gpurun_out/r05_tme_bands.txt | cut -c1-300
for mode in ${MODES:-end start}; do
  ( time timeout 1500 tools/fence_run.sh $mode python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=400 ) > gpurun_out/r05_fence_${mode}_gputest.txt 2>&1
  grep -c "alloc #" gpurun_out/fence_$mode.log > gpurun_out/fence_${mode}_allocs.txt 2>/dev/null
  tail -c 60000 gpurun_out/fence_$mode.log > gpurun_out/fence_${mode}_tail.log; rm -f gpurun_out/fence_$mode.log
  tail -n 30 gpurun_out/r05_fence_${mode}_gputest.txt | cut -c1-300
done
