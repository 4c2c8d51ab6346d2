#!/bin/bash
# the diamond stage of x265hip_tme_picture as one launch per (CU size, reference) with a cost row per task: parity (recorded reference calls, producer, e2e bitstreams) and its time
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_tme_gpu.py tests/test_tme_producer_gpu.py tests/test_e2e_tme_gpu.py tests/test_e2e_la_gpu.py tests/test_ctx_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 ) > gpurun_out/r05_diamond_rows_tests.txt 2>&1
grep -E "passed|failed|^E " gpurun_out/r05_diamond_rows_tests.txt | tail -4 | cut -c1-250
for thr in one default; do
  [ $thr = default ] && export X265_CLI_THREADING=1 || unset X265_CLI_THREADING
  X265TME_PROF=1 X265TME=1 X265TMEGPU=1 X265LAGPU=0 X265FFGPU=0 MALLOC_PERTURB_=85 timeout 300 oracle/_ref/x265e2e_8 x265-mod-by-patman_amd/libx265hip_8.so 1920 1088 24 medium /tmp/p.hevc 2>&1 | grep -E "x265hip_tme:|fps" | cut -c1-400 | sed "s/^/$thr threading: /" | cut -c1-330
done
