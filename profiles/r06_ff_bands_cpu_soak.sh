#!/bin/bash
# CPU soak of the filter binding's band mode (integration/filter_adapter.cpp) inside the compiled reference encoder, the mock producer answered by the oracle's C filters
# (tests/mock_ff_producer.cpp): N runs per configuration under frame threads + WPP, every run must write the plain encoder's bitstream and no run may report a protocol violation.
cd "$(dirname "$0")/.."
OUT=profiles/r06_ff_bands_cpu_soak.txt; : > $OUT
T=$(mktemp -d)
g++ -O2 -std=c++17 -fPIC -shared -DMOCK_DEPTH=8 -o $T/libmock_ff.so tests/mock_ff_producer.cpp -ldl || exit 1
E="X265MOCK_ORACLE_LIB=$PWD/oracle/libx265oracle_me_8.so X265TME=0 X265TMEGPU=0 X265LAGPU=0 X265_CLI_THREADING=1"
run() { env $E "$@" oracle/_ref/x265e2e_8 $T/libmock_ff.so $SIZE 12 medium $T/o.hevc $OPTS 2>$T/err | tail -1 > $T/json; echo "$(md5sum < $T/o.hevc | cut -c1-12) $(grep -c VIOLATION $T/err) $(python -c "import json; d=json.load(open('$T/json')); print(d['ff_pictures'], d['ff_bands'], d['ff_cpu_pictures'])")"; }
N=${N:-8}
for cfg in "1280 720|pools=24 frame-threads=4|4" "1280 720|pools=24 frame-threads=3 wpp=0|1" "832 480|pools=32 frame-threads=5|2" "640 704|pools=16 frame-threads=3 sao-non-deblock=1|3" "1280 720|pools=24 frame-threads=4 bframes=0 limit-sao=1|4"; do
  SIZE=$(echo "$cfg" | cut -d'|' -f1); OPTS=$(echo "$cfg" | cut -d'|' -f2); ROWS=$(echo "$cfg" | cut -d'|' -f3)
  plain=$(run X265FFGPU=0)
  bad=0; bands=""
  for i in $(seq 1 $N); do r=$(run X265FFGPU=1 X265FF_BAND_ROWS=$ROWS); [ "$(echo $r | cut -d' ' -f1)" = "$(echo $plain | cut -d' ' -f1)" ] && [ "$(echo $r | cut -d' ' -f2)" = "0" ] || bad=$((bad+1)); bands="$bands $(echo $r | cut -d' ' -f4)"; done
  echo "$SIZE | $OPTS | band rows $ROWS: $N runs, $bad differ from the plain encoder's bitstream ($(echo $plain | cut -d' ' -f1)) or report a violation; bands per run:$bands" >> $OUT
done
rm -rf $T
cat $OUT
