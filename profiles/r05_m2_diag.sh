#!/bin/bash
# (X265TME_ONE_QUEUE and X265TME_LANES were switches of integration/tme_adapter.cpp at commit e8e30fb; the measured losers left the binding afterwards)
# M2 under default threading: the encoder's own per-frame clocks for the three runs of bench.py's default_threading object
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
OUT=gpurun_out/r05_m2_diag.txt; : > $OUT
LIB=x265-mod-by-patman_amd/libx265hip_8.so
run() { name=$1; shift; ( env "$@" X265_FRAME_STATS=1 X265_CLI_THREADING=1 MALLOC_PERTURB_=85 timeout 300 oracle/_ref/x265e2e_8 $LIB 1920 1088 ${FRAMES:-48} medium /tmp/$name.hevc ${EXTRA} > /tmp/$name.out 2> /tmp/$name.err )
  echo "== $name: $(tail -1 /tmp/$name.out | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('fps','frame_threads','gpu_pictures','gpu_bands','gpu_seconds','adapter_seconds','la_producer_seconds','ff_pictures')})" 2>&1)" >> $OUT
  grep "frame stats" /tmp/$name.err >> $OUT; md5sum /tmp/$name.hevc >> $OUT; }
run cpu_plain X265TME=0 X265TMEGPU=0 X265LAGPU=0 X265FFGPU=0


for L in 1 2 4 8; do for rep in 1 2 3; do run gpu_tme_la_lanes${L}_$rep X265TME=1 X265TMEGPU=1 X265LAGPU=1 X265FFGPU=0 X265TME_LANES=$L; done; done
cat $OUT
