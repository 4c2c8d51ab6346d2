// kern_star64_wide.hip -- the raster refinement of 64x64 PUs whose window does not fit the band (merange 58..160; 128 at BASELINE configs[4]): star64_body.inc in its
// raster-only mode.  A chunk is 26 vertical x 20 (16 bit) / 24 (8 bit) horizontal placements: the 52 x 52 placements of merange 128 are 2 x 3 chunks with no idle
// accumulator (23 x 24 chunks left a 6-row and a 4-column remainder and computed 3588 placements for 2704).
#define XS_WIDE 1
#define XS_NJ 26
#if X265_DEPTH == 8
#define XS_NIC 24
#else
#define XS_NIC 20
#endif
#include "star64_body.inc"
