#!/bin/bash
# end of round 5: the batch-host / search / pipeline / producer tests on the FINAL kernels under the fence, both modes.  The two tests that hold buffers beyond 4 GB run in processes
# of their own: behind each other in one process the fence allocator's own hipMemset of the second 4.6 GB block faults now and then (profiles/r05_fence_flake.txt) -- the runtime,
# not a kernel of the library (the address lies in the block being filled, no launch of the library is in flight)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/fence_*.log
OUT=gpurun_out/r05_fence_final.txt; : > $OUT
for mode in end start; do
  for part in "not every_pu and not beyond_4gb and not 8k_batch" "8k_batch" "beyond_4gb"; do
    ( timeout 900 tools/fence_run.sh $mode python -m pytest tests/test_host_batch_gpu.py tests/test_me_gpu.py tests/test_pipeline_gpu.py tests/test_tme_producer_gpu.py -m gpu -q -p no:cacheprovider --timeout=800 -k "$part" > /tmp/part.out 2>&1 )
    echo "== $mode | $part: $(grep -E 'passed|failed' /tmp/part.out | tail -1) $(grep -c 'Memory access fault' /tmp/part.out) faults" >> $OUT
    rm -f gpurun_out/fence_$mode.log
  done
done
cat $OUT
