// kern_star64_wide.hip -- the raster refinement of 64x64 PUs whose window does not fit the band (merange 58..160; 128 at BASELINE configs[4]): star64_body.inc in its
// raster-only mode.  A chunk is 26 vertical x 16 (16 bit) / 24 (8 bit) horizontal placements.  The 52 x 52 placements of merange 128 are 2 chunks down with no idle
// accumulator (23-row chunks left a 6-row remainder: 3588 placements computed for 2704) and, at 16 bit, 16 + 16 + 16 + 4 columns across = 4, 4, 4 and 1 column groups for
// the two wavefront pairs of a workgroup to share: 2 + 2 + 2 + 1 walks on the longer side (20-column chunks: 5, 5 and 3 groups = 3 + 3 + 2; 8K pass 10.8 -> 10.3 ms).
// r04 (profiles/r04_8k_raster_ab.txt), 16 bit: 8-column chunks (2 column groups = one walk per wavefront pair, 43 KB of LDS) without the next chunk's window piece in flight
// in registers (56 VGPRs less: 156) put three workgroups on a CU instead of two: star64_raster_kernel 4.64 -> 4.27 ms per 2-picture 8K launch.
#define XS_WIDE 1
#define XS_NJ 26
#ifndef XS_NIC
#if X265_DEPTH == 8
#define XS_NIC 24
#else
#define XS_NIC 8
#define XS_RASTER_WGS 3
#define XS_RASTER_PREFETCH 0
#endif
#endif
#include "star64_body.inc"
