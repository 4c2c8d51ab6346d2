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
