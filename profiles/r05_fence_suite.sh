#!/bin/bash
# round 5: the -m gpu suite on the release build, then on the fence build (csrc/xh_fence.h) in both modes; the selftest shows the fence catching a one-line overrun
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export X265HIP_FENCE_LOG=$PWD/gpurun_out/fence_selftest.log
rm -f gpurun_out/fence_*.log
( timeout 120 tools/fence_run.sh end python tools/fence_selftest.py > gpurun_out/fence_selftest.out 2> gpurun_out/fence_selftest.err; echo "selftest rc $?" >> gpurun_out/fence_selftest.out )
python tools/fence_report.py gpurun_out/fence_selftest.log gpurun_out/fence_selftest.err > gpurun_out/fence_selftest_report.txt 2>&1
unset X265HIP_FENCE_LOG
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r05_gputest_release.txt 2>&1
for mode in ${MODES:-end start}; do
  ( time timeout ${FENCE_TIMEOUT:-1800} tools/fence_run.sh $mode python -m pytest tests -m gpu -q -p no:cacheprovider ) > gpurun_out/r05_fence_${mode}_gputest.txt 2>&1
  grep -c "alloc #" gpurun_out/fence_$mode.log > gpurun_out/fence_${mode}_allocs.txt 2>/dev/null
  tail -c 200000 gpurun_out/fence_$mode.log > gpurun_out/fence_${mode}_tail.log; rm -f gpurun_out/fence_$mode.log
done
tail -5 gpurun_out/fence_selftest_report.txt gpurun_out/r05_gputest_release.txt gpurun_out/r05_fence_*_gputest.txt
