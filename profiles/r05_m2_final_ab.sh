#!/bin/bash
# M2 under default threading with the adapter's defaults (bands wait for half a picture's rows, 16 ms at most): five interleaved runs each
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r05_m2_final_ab.txt; : > $OUT
LIB=x265-mod-by-patman_amd/libx265hip_8.so
run() { name=$1; shift; ( env "$@" X265_CLI_THREADING=1 MALLOC_PERTURB_=85 timeout 300 oracle/_ref/x265e2e_8 $LIB 1920 1088 48 medium /tmp/$name.hevc > /tmp/$name.out 2>/dev/null )
  echo "$name: $(tail -1 /tmp/$name.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fps', d['fps'], 'bands', d['gpu_bands'], 'producer s', d['gpu_seconds'], 'la s', d.get('la_producer_seconds'))") $(md5sum /tmp/$name.hevc | cut -c1-8)" >> $OUT; }
for rep in 1 2 3 4 5; do
  run plain_encoder_$rep X265TME=0 X265TMEGPU=0 X265LAGPU=0 X265FFGPU=0
  run gpu_tme_$rep X265TME=1 X265TMEGPU=1 X265LAGPU=0 X265FFGPU=0
  run gpu_tme_la_$rep X265TME=1 X265TMEGPU=1 X265LAGPU=1 X265FFGPU=0
  run gpu_la_$rep X265TME=0 X265TMEGPU=0 X265LAGPU=1 X265FFGPU=0
done
run own_tme_1 X265TME=1 X265TMEGPU=0 X265LAGPU=0 X265FFGPU=0
sort $OUT
