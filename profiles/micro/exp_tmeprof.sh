#!/bin/bash
# Where a picture of the GPU TME producer spends its time (X265HIP_TME_PROF sections), medium and slow.
lib=$(python -c "import x265hip; print(x265hip.lib_path(8))")
for cfg in "1280 704 10 medium" "1280 704 8 slow" "1920 1088 6 medium"; do
  echo "== $cfg"
  X265HIP_TME_PROF=1 X265TMEGPU=1 oracle/_ref/x265tmegpu_8 $lib $cfg /tmp/p.hevc ref=1 2>&1 | grep -E "x265hip_tme:|producer" | cut -c1-300
done
