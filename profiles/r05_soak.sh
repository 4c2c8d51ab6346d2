#!/bin/bash
# soak: N runs of the release bench with every GPU leg (no e2e encodes, no CPU baseline), each in a fresh process tree
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/r05_soak2.txt
for i in $(seq 1 ${SOAK:-40}); do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-e2e --cpu-ctus 0 > /tmp/soak.json 2> /tmp/soak.err; rc=$?
  echo "run $i rc $rc $(python -c "import json; d=json.loads([l for l in open('/tmp/soak.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], sorted((d.get('preset_exact') or {}).keys()), 'tme_producer' in d, 'streams' in d)" 2>&1 | tail -1) $(grep -i -c 'memory access fault' /tmp/soak.err) faults" >> gpurun_out/r05_soak2.txt
  [ $rc -ne 0 ] && tail -c 1500 /tmp/soak.err >> gpurun_out/r05_soak2.txt
done
tail -n 5 gpurun_out/r05_soak2.txt; grep -c "rc 0" gpurun_out/r05_soak2.txt
