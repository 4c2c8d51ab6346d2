#!/bin/bash
# M2 under default threading, plain A/B (no frame statistics): the encoder alone vs the GPU producers, three runs each, interleaved
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r05_m2_ab.txt; : > $OUT
LIB=x265-mod-by-patman_amd/libx265hip_8.so
run() { name=$1; shift; ( env "$@" X265_CLI_THREADING=1 MALLOC_PERTURB_=85 timeout 300 oracle/_ref/x265e2e_8 $LIB 1920 1088 ${FRAMES:-48} medium /tmp/$name.hevc ${EXTRA} > /tmp/$name.out 2> /tmp/$name.err )
  echo "== $name: $(tail -1 /tmp/$name.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('fps','frame_threads','gpu_bands','gpu_seconds','adapter_seconds')})" 2>&1) $(md5sum /tmp/$name.hevc | cut -c1-8)" >> $OUT; }
for rep in 1 2 3; do
  run cpu_plain_$rep X265TME=0 X265TMEGPU=0 X265LAGPU=0 X265FFGPU=0
  run gpu_tme_la_$rep X265TME=1 X265TMEGPU=1 X265LAGPU=1 X265FFGPU=1 X265TME_LANES=1
  run gpu_tme_only_$rep X265TME=1 X265TMEGPU=1 X265LAGPU=0 X265FFGPU=0 X265TME_LANES=1
  run gpu_la_only_$rep X265TME=0 X265TMEGPU=0 X265LAGPU=1 X265FFGPU=0
done
sort $OUT
