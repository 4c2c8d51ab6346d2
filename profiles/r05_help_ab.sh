#!/bin/bash
# helpers of a band's host passes under frame threads: none (the default), at most 2 / 4 beside the leader; and the wait for rows re-tuned.  Same clip and threading as r05_m2_final_ab.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r05_help_ab.txt; : > $OUT
LIB=x265-mod-by-patman_amd/libx265hip_8.so
for rep in 1 2 3 4 5; do
  for cfg in "0 0 16000" "3 0 16000" "5 0 16000" "0 6 10000" "0 12 24000"; do
    set -- $cfg
    X265TME_HELP=$1 X265TME_MIN_ROWS=$2 X265TME_WAIT_US=$3 X265_CLI_THREADING=1 X265TME=1 X265TMEGPU=1 X265LAGPU=1 X265FFGPU=0 MALLOC_PERTURB_=85 timeout 300 oracle/_ref/x265e2e_8 $LIB 1920 1088 48 medium /tmp/h.hevc > /tmp/h.out 2>/dev/null
    echo "help $1 min_rows $2 wait_us $3 run $rep: $(tail -1 /tmp/h.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fps', d['fps'], 'bands', d['gpu_bands'])") $(md5sum /tmp/h.hevc | cut -c1-8)" >> $OUT
  done
done
sort $OUT
