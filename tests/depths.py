"""Bit depths the depth-parametrised tests run at: libx265hip_8 / _10 by default; X265HIP_TEST_DEPTHS=8,10,12 adds the 12-bit library (`make lib12` in the package,
`make ref12 oracle12` in oracle/ -- reference MAIN12, source/CMakeLists.txt:790-792)."""
import os

DEPTHS = [int(v) for v in os.environ.get("X265HIP_TEST_DEPTHS", "8,10").split(",") if v.strip()]
# committed reference fixtures (tests/golden/) and the survey's known answers exist for the two depths of the BASELINE configurations
GOLDEN_DEPTHS = [d for d in DEPTHS if d in (8, 10)]
