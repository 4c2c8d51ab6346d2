#!/bin/bash
# End-to-end figures of the ThreadedME seam (run through gpurun): the reference encoder (C primitives, no asm) with its own CPU producer and with the GPU producer,
# same binary (oracle/_ref/x265tmegpu_8), same bitstream.   profiles/e2e_tme.sh [width height frames preset [key=value ...]]   (the preset's own defaults: ref, weightp, bframes)
W=${1:-1280}; H=${2:-704}; F=${3:-16}; P=${4:-medium}; shift 4
lib=$(python -c "import x265hip; print(x265hip.lib_path(8))")
for prod in 0 1 1; do
  X265TME_PROF=1 X265TMEGPU=$prod timeout -k 10 600 oracle/_ref/x265tmegpu_8 $lib $W $H $F $P /tmp/e2e_$prod.hevc "$@" 2>&1 | grep -E "^x265hip_tme:|^\{" 
done
md5sum /tmp/e2e_0.hevc /tmp/e2e_1.hevc
