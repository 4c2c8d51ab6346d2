#!/bin/bash
# Round 6, first GPU action: HEAD of round 5 (masked AMP kernels, band producer, X265TME_AHEAD, refactored queues) under the release suite, the bench and the fence.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=600 2>&1 | tail -8 ) > gpurun_out/r06_gputest_release.txt
cat gpurun_out/r06_gputest_release.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_head_bench.json 2> gpurun_out/r06_head_bench.err
tail -c 400 gpurun_out/r06_head_bench.json
timeout 500 bash profiles/r05_ahead_ab.sh > gpurun_out/r06_ahead_ab.log 2>&1; cp gpurun_out/r05_ahead_ab.txt gpurun_out/r06_ahead_ab.txt; tail -15 gpurun_out/r06_ahead_ab.log
timeout 420 bash profiles/collect_preset_exact.sh r06_pe 4320p10_slower > gpurun_out/r06_pe.log 2>&1
tail -12 gpurun_out/r06_pe.log
rm -f gpurun_out/fence_*.log
for mode in end start; do
( timeout 420 tools/fence_run.sh $mode python -m pytest tests/test_me_gpu.py tests/test_host_batch_gpu.py -m gpu -q -p no:cacheprovider --timeout=400 -k "not every_pu and not beyond_4gb and not whole_4k" > /tmp/part.out 2>&1 )
echo "== fence $mode | test_me_gpu + test_host_batch_gpu: $(grep -E 'passed|failed' /tmp/part.out | tail -1) $(grep -c 'Memory access fault' /tmp/part.out) faults" >> gpurun_out/r06_amp_fence.txt
tail -5 /tmp/part.out >> gpurun_out/r06_amp_fence.txt
done
rm -f gpurun_out/fence_*.log
cat gpurun_out/r06_amp_fence.txt
