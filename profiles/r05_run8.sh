#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05_lds2
export TMPDIR=/tmp
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-e2e --cpu-ctus 0 --no-tme --no-preset-exact --no-streams-leg 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('value', d['value'], 'ms/step', d['ms_per_step'], 'star64', d['roofline']['avg_launch_ms'], d['roofline']['all_kernels_ms'])"
done
( time timeout 1200 python -m pytest tests/test_host_batch_gpu.py tests/test_me_gpu.py tests/test_pipeline_gpu.py -m gpu -q -p no:cacheprovider --timeout=900 ) > gpurun_out/r05_bankpairs_tests.txt 2>&1
tail -n 5 gpurun_out/r05_bankpairs_tests.txt | cut -c1-300
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/r05_lds2/p -- python bench.py --steps 3 --warmup 1 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg --splits 1 --inner 1 > /dev/null 2> gpurun_out/r05_lds2/err.txt
python profiles/summarize_pmc.py gpurun_out/r05_lds2/p > gpurun_out/r05_lds_counters2.txt 2>&1
find gpurun_out/r05_lds2 -name "*.csv" -delete
grep -E "star64" gpurun_out/r05_lds_counters2.txt | cut -c1-300
for split in 1 0; do
  X265HIP_PE_SPLIT_BATCHES=$split timeout 600 python bench.py --leg preset_exact:4320p10_slower 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('8K preset-exact split=$split', d['config']['batches'], 'batches', d['ms_per_pass'], 'ms per pass', d['ms_per_picture'], 'per picture', d.get('stage_ms_sub_batch_0'))"
done
