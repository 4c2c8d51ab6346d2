#!/bin/bash
# X265TME_AHEAD (a band harvested while the band before it is in its producer call) on the real producer: first the frame-thread bitstream tests with it switched on, then
# default-threaded 1080p medium, four interleaved runs each (plain encoder / GPU ThreadedME + lookahead without and with the switch).  Also the first M2 figures after the e2e driver
# stopped reading the environment per pixel.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r05_ahead_ab.txt; : > $OUT
( X265TME_AHEAD=1 timeout 400 python -m pytest tests/test_e2e_tme_gpu.py -m gpu -q -x -k "frame_threads" --timeout 300 2>&1 | tail -2 ) >> $OUT
LIB=x265-mod-by-patman_amd/libx265hip_8.so
run() { name=$1; shift; ( env "$@" X265_CLI_THREADING=1 MALLOC_PERTURB_=85 timeout 300 oracle/_ref/x265e2e_8 $LIB 1920 1088 48 medium /tmp/$name.hevc > /tmp/$name.out 2>/dev/null )
  echo "$name: $(tail -1 /tmp/$name.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fps', d['fps'], 'bands', d['gpu_bands'], 'producer s', d['gpu_seconds'])") $(md5sum /tmp/$name.hevc | cut -c1-8)" >> $OUT; }
for rep in 1 2 3 4; do
  run plain_encoder_$rep X265TME=0 X265TMEGPU=0 X265LAGPU=0 X265FFGPU=0
  run gpu_serial_$rep X265TME=1 X265TMEGPU=1 X265LAGPU=1 X265FFGPU=0 X265TME_AHEAD=0
  run gpu_ahead_$rep X265TME=1 X265TMEGPU=1 X265LAGPU=1 X265FFGPU=0 X265TME_AHEAD=1
done
sort $OUT
