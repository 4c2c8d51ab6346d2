#!/bin/bash
# End-to-end figures of the ThreadedME seam (run through gpurun): the reference encoder (C primitives, no asm) with its own CPU producer and with the GPU producer,
# same binary (oracle/_ref/x265tmegpu_8), same bitstream.   profiles/e2e_tme.sh [width height frames preset]
W=${1:-1280}; H=${2:-704}; F=${3:-16}; P=${4:-medium}
lib=$(python -c "import x265hip; print(x265hip.lib_path(8))")
for prod in 0 1 1; do
  X265TMEGPU=$prod oracle/_ref/x265tmegpu_8 $lib $W $H $F $P /tmp/e2e_$prod.hevc ref=1 weightp=0 weightb=0 2>/dev/null | tail -1
done
md5sum /tmp/e2e_0.hevc /tmp/e2e_1.hevc
