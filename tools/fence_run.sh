#!/bin/bash
# Run a command on the fence build (csrc/xh_fence.h): every device block of the library -- and, through torch's pluggable allocator, every device tensor of the tests --
# ends (X265HIP_FENCE=end, default) or starts (=start) at an unmapped page; a kernel that steps outside dies with a page fault whose last "[fence] launch" line names it.
#   tools/fence_run.sh build                   compile x265-mod-by-patman_amd/fence/libx265hip_{8,10}.so + libxh_fence_torch.so (cross-compiles without a GPU)
#   tools/fence_run.sh [end|start] <command>   e.g. tools/fence_run.sh end python -m pytest tests -m gpu -x -q
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG=$ROOT/x265-mod-by-patman_amd
if [ "$1" = build ]; then
    make -s -j"${JOBS:-16}" -C "$PKG" FENCE=1 OUT="$PKG/fence/" OBJ="$PKG/fence/obj" "$PKG/fence/libx265hip_8.so" "$PKG/fence/libx265hip_10.so"
    /opt/rocm/bin/hipcc -O1 -std=c++17 -fPIC -shared -x hip --offload-arch=gfx950 -o "$PKG/fence/libxh_fence_torch.so" "$ROOT/tools/fence_torch.cpp"
    exit 0
fi
MODE=end
case "$1" in end|start) MODE=$1; shift;; esac
export X265HIP_FENCE=$MODE X265HIP_LIBDIR=$PKG/fence X265HIP_FENCE_TORCH=$PKG/fence/libxh_fence_torch.so
mkdir -p "$ROOT/gpurun_out"
export X265HIP_FENCE_LOG=${X265HIP_FENCE_LOG:-$ROOT/gpurun_out/fence_$MODE.log}
exec "$@"
