"""x265 encoder-primitives hot path on MI355X (gfx950): Python-side loader of the C-ABI libraries.

The product is the pair of shared libraries built from csrc/ (libx265hip_8.so / libx265hip_10.so,
C ABI in include/x265hip.h).  This package only locates/builds/loads them for tests, bench.py and
tooling; it contains no arithmetic and no CPU fallback.
"""
from .binding import HipLib, build_libraries, lib_path  # noqa: F401
from . import synth  # noqa: F401
