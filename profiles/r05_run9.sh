#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05_lds3
export TMPDIR=/tmp
for v in 1 0 2 1 0 2; do
  [ $v = 1 ] && unset X265HIP_LIBDIR || export X265HIP_LIBDIR=$PWD/x265-mod-by-patman_amd/exp_bp$v
  python bench.py --steps 20 --warmup 5 --no-e2e --cpu-ctus 0 --no-tme --no-preset-exact --no-streams-leg 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('bank pairs $v: value', d['value'], 'ms/step', d['ms_per_step'], 'star64', d['roofline']['avg_launch_ms'], 'me64', d['roofline']['all_kernels_ms']['me64'])"
done
export X265HIP_LIBDIR=$PWD/x265-mod-by-patman_amd/exp_bp2
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/r05_lds3/p -- python bench.py --steps 3 --warmup 1 --cpu-ctus 0 --no-tme --no-e2e --no-preset-exact --no-streams-leg --splits 1 --inner 1 > /dev/null 2> gpurun_out/r05_lds3/err.txt
unset X265HIP_LIBDIR
python profiles/summarize_pmc.py gpurun_out/r05_lds3/p > gpurun_out/r05_lds_counters3.txt 2>&1
find gpurun_out/r05_lds3 -name "*.csv" -delete
grep -E "star64" gpurun_out/r05_lds_counters3.txt | cut -c1-300
( time timeout 1200 python -m pytest tests/test_host_batch_gpu.py -m gpu -q -p no:cacheprovider --timeout=900 ) > gpurun_out/r05_groups_tests.txt 2>&1
tail -n 4 gpurun_out/r05_groups_tests.txt | cut -c1-300
