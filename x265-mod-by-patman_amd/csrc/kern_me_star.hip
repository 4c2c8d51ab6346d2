// kern_me_star.hip -- the STAR-search instantiations of me_body.inc (reference motion.cpp:387-629, 1328-1436)
#ifndef XH_LWIN
#define XH_LWIN 1   // the square PUs of the CU pyramid cost their full-pel candidates out of a per-wavefront LDS window (me_body.inc)
#endif
#include "me_body.inc"

int xh_me_star(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
               const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
               int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
               const void* subpelPlanes, int64_t planeElems)
{
    return dispatch_me<1>(stream, w, h, curPlane, curStride, refPlane, refStride, tasks, n, costRow, costHalfRange, merange, method, subpelRefine, results, mvpSource, subpelPlanes, planeElems);
}

// The 64x64 level of the batch host (csrc/xh_ctx.cpp): its tasks have a zero predictor and no candidates, so the start stage is one SAD at the co-located block and
// star64_kernel takes it out of its band -- two launches (full-pel search, sub-pel stage) instead of three.  Tasks that are not of that kind are still searched correctly
// (the second launch then runs their whole search).
int xh_me_star_own64(void* stream, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                     const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
                     int merange, int subpelRefine, x265hip_me_result* results, const void* subpelPlanes, int64_t planeElems)
{
    return dispatch_me<1>(stream, 64, 64, curPlane, curStride, refPlane, refStride, tasks, n, costRow, costHalfRange, merange, X265HIP_ME_STAR, subpelRefine, results, nullptr, subpelPlanes, planeElems,
                          nullptr, 0, true);
}
