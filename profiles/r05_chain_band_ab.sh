#!/bin/bash
# the small shapes' references side by side in launches of up to 256 CTUs (bands; small pictures): parity, then per-call sections and fps with and without (exp_cb0: -DXH_CHAIN_BAND_CTUS=0)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_tme_gpu.py tests/test_tme_producer_gpu.py tests/test_e2e_tme_gpu.py tests/test_e2e_la_gpu.py tests/test_ctx_gpu.py -m gpu -q -p no:cacheprovider --timeout=600 ) > gpurun_out/r05_chain_band_tests.txt 2>&1
grep -E "passed|failed|^E " gpurun_out/r05_chain_band_tests.txt | tail -4 | cut -c1-250
OUT=gpurun_out/r05_chain_band_ab.txt; : > $OUT
for rep in 1 2 3; do
  for v in new old; do
    [ $v = new ] && LIB=x265-mod-by-patman_amd/libx265hip_8.so || LIB=x265-mod-by-patman_amd/exp_cb0/libx265hip_8.so
    X265_CLI_THREADING=1 X265TME_PROF=1 X265TME=1 X265TMEGPU=1 X265LAGPU=0 X265FFGPU=0 MALLOC_PERTURB_=85 timeout 300 oracle/_ref/x265e2e_8 $LIB 1920 1088 48 medium /tmp/$v.hevc > /tmp/$v.out 2> /tmp/$v.err
    echo "$v $rep: $(grep 'x265hip_tme:' /tmp/$v.err | cut -c1-200) | fps $(tail -1 /tmp/$v.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['fps'], 'bands', d['gpu_bands'], 'producer s', d['gpu_seconds'])") $(md5sum /tmp/$v.hevc | cut -c1-8)" >> $OUT
  done
done
cat $OUT
