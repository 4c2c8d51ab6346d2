#!/bin/bash
# the fence without its launch serialisation (X265HIP_FENCE_SYNC=0): streams and sub-batches run concurrently as in the release build, every block still ends at an unmapped page
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/fence_*.log
export X265HIP_FENCE_SYNC=0
# (the test with a 4.6 GB tensor in a process of its own: profiles/r05_fence_flake.txt)
( time timeout 1500 tools/fence_run.sh end python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 -k "not beyond_4gb" ) > gpurun_out/r05_fence_nosync_gputest.txt 2>&1
grep -E "passed|failed" gpurun_out/r05_fence_nosync_gputest.txt | tail -1
rm -f gpurun_out/fence_end.log
( timeout 600 tools/fence_run.sh end python -m pytest tests/test_me_gpu.py -m gpu -q -p no:cacheprovider -k "beyond_4gb" ) >> gpurun_out/r05_fence_nosync_gputest.txt 2>&1
grep -E "passed|failed" gpurun_out/r05_fence_nosync_gputest.txt | tail -1
rm -f gpurun_out/fence_end.log
: > gpurun_out/r05_fence_nosync_bench.txt
for i in 1 2 3 4; do
  timeout 600 tools/fence_run.sh end python bench.py --steps 20 --warmup 5 --no-e2e --cpu-ctus 0 > /tmp/b.json 2> /tmp/b.err; rc=$?
  echo "run $i rc $rc $(python -c "import json; d=json.loads([l for l in open('/tmp/b.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])" 2>&1 | tail -1) $(grep -i -c 'memory access fault' /tmp/b.err) faults" >> gpurun_out/r05_fence_nosync_bench.txt
  [ $rc -ne 0 ] && tail -c 1200 /tmp/b.err >> gpurun_out/r05_fence_nosync_bench.txt
  rm -f gpurun_out/fence_end.log
done
cat gpurun_out/r05_fence_nosync_bench.txt
