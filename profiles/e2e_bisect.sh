#!/bin/bash
# usage: bisect.sh depth w h frames preset opts...   -> compares c vs hip for several X265ENC_HIP_RANGE values
d=$1; w=$2; h=$3; f=$4; p=$5; shift 5
export LD_LIBRARY_PATH=$(python -c "import torch,os; print(os.path.join(os.path.dirname(torch.__file__),'lib'))"):/opt/rocm/lib:$LD_LIBRARY_PATH
E=oracle/_ref/x265enc_$d; L=x265-mod-by-patman_amd/libx265hip_$d.so
$E c $L $w $h $f $p /tmp/c.hevc "$@" > /dev/null 2>&1
for r in "$@"; do :; done
for range in 0:6888 0:6928 0:6944 0:6968 0:7104 0:7200 0:18240 6888:6928 6944:6968 7104:7200; do
  X265ENC_HIP_RANGE=$range $E hip $L $w $h $f $p /tmp/h.hevc "$@" > /dev/null 2>&1
  if cmp -s /tmp/c.hevc /tmp/h.hevc; then echo "range $range: identical"; else echo "range $range: DIFFERENT"; fi
done
