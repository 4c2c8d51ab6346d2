#!/bin/bash
# r03: sub-batches of whole pictures on their own streams (bench.py --splits S --skew K), all variants inside ONE box
for v in "--splits 1" "--splits 2" "--splits 2 --skew 2" "--splits 2 --skew 3" "--splits 4" "--splits 4 --skew 2" "--splits 8" "--splits 2 --skew 0" "--splits 1"; do
  python bench.py --steps 20 --warmup 3 --cpu-ctus 0 --no-tme $v 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['all_kernels_ms'])"
done
python bench.py --steps 20 --warmup 3 --no-tme --splits 2 --cpu-ctus 1020 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('parity', d['value'], d['cpu_baseline']['sample'][-60:])"
