#!/bin/bash
# quick_variant.sh <dir name under x265-mod-by-patman_amd/> "<extra flags>" <kernel files...>: an A/B library for profiles/*_ab.sh without a full rebuild --
# the release objects of build/$DEPTH with the named translation units recompiled (depth 10 only; the 8-bit library is the release one).  X265HIP_LIBDIR points at it.
set -e
P=$(dirname "$(readlink -f "$0")")/../x265-mod-by-patman_amd
D=$P/$1; EXTRA=$2; shift 2; DEPTH=${DEPTH:-10}; OTHER=$((18-DEPTH))
O=/tmp/xv/$1/obj/$DEPTH; mkdir -p $D $O
cp -u $P/build/$DEPTH/*.o $O/
for f in "$@"; do
  b=$(basename $f); o=${b%.*}.o
  x=""; case $b in *.cpp) x="-x hip";; esac
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical $EXTRA $x -DX265_DEPTH=$DEPTH -c $P/csrc/$b -o $O/$o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libx265hip_$DEPTH.so $O/*.o
cp -u $P/libx265hip_$OTHER.so $D/
ls -la $D/*.so
