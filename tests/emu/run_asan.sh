#!/bin/bash
# The -m gpu tests against the emulated library UNDER AddressSanitizer (make -C tests/emu ASAN=1): device memory is the host heap and LDS arrays are statics there, so an access
# one element outside a device block, a plane, a task / result array or an LDS array is a report with the source line.  What the fence build shows on the GPU for device blocks --
# and, for LDS, what nothing shows on the GPU.
#   tests/emu/run_asan.sh [pytest node ids / options]     ->  profiles/r06_emu_asan.txt (summary), /tmp/emu_asan.log
cd "$(dirname "$0")/../.."
make -s -j8 -C tests/emu ASAN=1 all || exit 1
ASANLIB=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
SKIP='not full_size and not 8k and not whole_4k and not every_pu and not beyond_4gb and not 1920 and not 1080 and not 4k and not baseline_workloads and not merange_128 and not soak and not e2e and not multi_gpu'
ARGS=("$@"); [ ${#ARGS[@]} -eq 0 ] && ARGS=(tests)
LD_PRELOAD=$ASANLIB ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1:abort_on_error=1 X265HIP_EMU=1 X265HIP_LIBDIR=$PWD/tests/emu/_build_asan \
  python -m pytest -m gpu -q -p no:cacheprovider -n ${JOBS:-7} --timeout=${TIMEOUT:-1800} --timeout-method=thread -k "$SKIP" "${ARGS[@]}" > /tmp/emu_asan.log 2>&1
{ echo "The -m gpu tests against the emulated library under AddressSanitizer (tests/emu/README.md), code at $(git rev-parse --short HEAD)$(git diff --quiet || echo +), $(date -u +%F)"; echo "selection: ${ARGS[*]}; deselected: $SKIP (the e2e tests load the library into the reference encoder's own process)"; echo; grep -E "^(FAILED|ERROR)|passed|failed|AddressSanitizer|SUMMARY" /tmp/emu_asan.log | cut -c1-220 | sort | uniq -c | sort -rn | head -40; } > profiles/r06_emu_asan.txt
tail -5 profiles/r06_emu_asan.txt
