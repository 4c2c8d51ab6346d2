#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_host_batch_gpu.py -m gpu -q -p no:cacheprovider --timeout=1200 ) > gpurun_out/r05_groups_tests.txt 2>&1
grep -E "passed|failed|^E " gpurun_out/r05_groups_tests.txt | tail -5 | cut -c1-300
for split in 1 0; do
  X265HIP_PE_SPLIT_BATCHES=$split timeout 600 python bench.py --leg preset_exact:4320p10_slower 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('8K preset-exact split=$split', d['config']['batches'], 'batches', d['ms_per_pass'], 'ms per pass', d['ms_per_picture'], 'per picture', d.get('stage_ms_sub_batch_0'))"
done
