#!/bin/bash
# kres.sh <object or shared library> ...: registers / LDS / scratch / spills of every kernel in the HIP code objects it carries (llvm-readelf --notes through profiles/kres.py)
B=/opt/rocm/lib/llvm/bin; T=$(mktemp -d)
for f in "$@"; do
  $B/llvm-objcopy --dump-section=.hip_fatbin=$T/fat.bin "$f" 2>/dev/null || continue
  $B/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co 2>/dev/null || continue
  $B/llvm-readelf --notes $T/k.co > $T/notes.txt 2>/dev/null
  python "$(dirname "$0")/kres.py" $T/notes.txt
done
rm -rf $T
