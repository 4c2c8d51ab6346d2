#!/bin/bash
# (X265TME_ONE_QUEUE and X265TME_LANES were switches of integration/tme_adapter.cpp at commit e8e30fb; the measured losers left the binding afterwards)
# the adapter's wake-ups: one condition variable for everybody (first form) against one for the running job's helpers and one for the workers of other bands
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r05_queues_ab.txt; : > $OUT
LIB=x265-mod-by-patman_amd/libx265hip_8.so
for rep in 1 2 3 4 5 6; do
  for q in ${QS:-0 1}; do
    X265TME_HELP=${q#h} X265TME_ONE_QUEUE=0 X265_CLI_THREADING=1 X265TME=1 X265TMEGPU=1 X265LAGPU=1 X265FFGPU=0 MALLOC_PERTURB_=85 timeout 300 oracle/_ref/x265e2e_8 $LIB 1920 1088 48 medium /tmp/q.hevc > /tmp/q.out 2>/dev/null
    echo "help ${q#h} run $rep: $(tail -1 /tmp/q.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fps', d['fps'], 'bands', d['gpu_bands'])") $(md5sum /tmp/q.hevc | cut -c1-8)" >> $OUT
  done
done
sort $OUT
