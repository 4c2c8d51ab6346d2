#!/bin/bash
# A/B of ME kernel variants inside ONE box (box-to-box variance is ~15%): profiles/ab.sh "<env assignments>" ...
for v in "$@"; do
  for wl in 1080p8_medium 2160p10_slow; do
    st=20; [ $wl = 2160p10_slow ] && st=5
    env $v python bench.py --workload $wl --steps $st --warmup 2 --cpu-ctus 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '$wl', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'])"
  done
done
