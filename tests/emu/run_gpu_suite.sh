#!/bin/bash
# The `-m gpu` tests against the EMULATED library (tests/emu: the kernel sources compiled for the host, a workgroup as fibers) -- what the CPU can say about the kernels' logic
# where no GPU is available.  Full-size workloads (4K / 8K pictures, 1080p encodes) are left out: the emulation runs a work-item at a time.
#   tests/emu/run_gpu_suite.sh [pytest arguments]      ->  profiles/r06_emu_gpu_suite.txt (summary), /tmp/emu_suite.log (everything)
cd "$(dirname "$0")/../.."
make -s -j8 -C tests/emu all || exit 1
SKIP='not full_size and not 8k and not whole_4k and not every_pu and not beyond_4gb and not 1920 and not 1080 and not 4k and not baseline_workloads and not merange_128 and not soak'
X265HIP_EMU=1 X265HIP_LIBDIR=$PWD/tests/emu/_build python -m pytest tests -m gpu -q -p no:cacheprovider -n ${JOBS:-7} --timeout=${TIMEOUT:-900} --timeout-method=thread -k "$SKIP" "$@" > /tmp/emu_suite.log 2>&1
{ echo "The -m gpu tests against the emulated library (tests/emu/README.md), code at $(git rev-parse --short HEAD)$(git diff --quiet || echo +), $(date -u +%F)"; echo "deselected: $SKIP"; echo; grep -E "^(FAILED|ERROR)|passed|failed" /tmp/emu_suite.log | cut -c1-220; } > profiles/r06_emu_gpu_suite.txt
tail -3 profiles/r06_emu_gpu_suite.txt
