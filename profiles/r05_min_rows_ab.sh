#!/bin/bash
# fewer, larger bands: a band waits up to WAIT_US for MIN_ROWS ready rows.  1920x1088 medium, 48 frames, default threading, GPU ThreadedME; interleaved runs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r05_min_rows_ab.txt; : > $OUT
LIB=x265-mod-by-patman_amd/libx265hip_8.so
for rep in 1 2 3 4 5; do
  for cfg in ${CFGS:-1:0 2:1500 3:3000 4:6000}; do
    set -- ${cfg%%:*} ${cfg##*:}
    X265TME_MIN_ROWS=$1 X265TME_WAIT_US=$2 X265_CLI_THREADING=1 X265TME=1 X265TMEGPU=1 X265LAGPU=${LA:-0} X265FFGPU=0 MALLOC_PERTURB_=85 timeout 300 oracle/_ref/x265e2e_8 $LIB 1920 1088 48 medium /tmp/m.hevc > /tmp/m.out 2>/dev/null
    echo "min_rows $1 wait_us $2 run $rep: $(tail -1 /tmp/m.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fps', d['fps'], 'bands', d['gpu_bands'], 'producer s', d['gpu_seconds'])") $(md5sum /tmp/m.hevc | cut -c1-8)" >> $OUT
  done
  X265_CLI_THREADING=1 X265TME=0 X265TMEGPU=0 X265LAGPU=0 X265FFGPU=0 MALLOC_PERTURB_=85 timeout 300 oracle/_ref/x265e2e_8 $LIB 1920 1088 48 medium /tmp/m.hevc > /tmp/m.out 2>/dev/null
  echo "plain encoder run $rep: $(tail -1 /tmp/m.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fps', d['fps'])")" >> $OUT
done
sort $OUT
