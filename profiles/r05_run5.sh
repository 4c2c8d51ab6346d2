#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; rm -f gpurun_out/fence_*.log
{ echo "nproc $(nproc)"; python -c "import os; print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))"; echo "cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; lscpu | grep -E "Model name|Socket|Core|Thread|^CPU\(s\)"; free -g | head -2; } > gpurun_out/r05_box.txt 2>&1
cat gpurun_out/r05_box.txt
for mode in end start; do
  ( time timeout 900 tools/fence_run.sh $mode python -m pytest tests/test_e2e_gpu.py -m gpu -q -p no:cacheprovider --timeout=500 ) > gpurun_out/r05_fence_${mode}_e2e_table.txt 2>&1
  grep -E "passed|failed" gpurun_out/r05_fence_${mode}_e2e_table.txt | tail -1
  rm -f gpurun_out/fence_$mode.log
done
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout=900 ) > gpurun_out/r05_gputest_release2.txt 2>&1
tail -n 6 gpurun_out/r05_gputest_release2.txt | cut -c1-300
git rev-parse --short HEAD > profiles/.commit 2>/dev/null
timeout 1500 profiles/collect.sh r05_v1_2160p10 > gpurun_out/r05_collect.log 2>&1
tail -c 1500 gpurun_out/r05_collect.log
