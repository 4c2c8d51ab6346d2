// kern_me_star_tiled.hip -- the STAR kernels of me_body.inc for TILED phase planes (slots 1..15 of the plane buffer as 16 x 4-pixel tiles of one 128-byte line,
// xh_subpel_planes_tiled; 16-bit library).  The batch host (csrc/xh_ctx.cpp) uses them for its square pyramid: a sub-pel candidate of an 8x8 PU then touches 4-6 lines
// instead of 9-18 -- the lower levels are bound by what they move through the memory side of the L2.
#include "xh_common.h"
#if X265_DEPTH != 8
#define XH_TILED 1
#define XH_LWIN 1
#include "me_body.inc"
#include "xh_internal.h"

int xh_me_star_tiled(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                     const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
                     int merange, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
                     const void* tiledPlanes, int64_t planeElems, bool ownStart64)
{
    if (w != h || (w != 64 && w != 32 && w != 16 && w != 8) || !tiledPlanes || (uint64_t)planeElems * 16u * sizeof(pixel) >= (1ull << 32) || (w == 64 && !xh_star64_ok(refStride, merange)))
    { set_error("me_star_tiled: the square pyramid sizes with a plane buffer below 4 GB only"); return X265HIP_EARG; }
    return dispatch_me<1>(stream, w, h, curPlane, curStride, refPlane, refStride, tasks, n, costRow, costHalfRange, merange, X265HIP_ME_STAR, subpelRefine, results, mvpSource, tiledPlanes, planeElems,
                          nullptr, 0, ownStart64);
}
#else
#include "xh_internal.h"
int xh_me_star_tiled(void*, int, int, const void*, intptr_t, const void*, intptr_t, const x265hip_me_task*, int, const uint16_t*, int, int, int, x265hip_me_result*, const x265hip_me_result*,
                     const void*, int64_t, bool)
{
    xh::set_error("me_star_tiled: 16-bit library only"); return X265HIP_EARG;
}
#endif
