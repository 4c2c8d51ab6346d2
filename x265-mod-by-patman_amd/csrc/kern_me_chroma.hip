// kern_me_chroma.hip -- the DIA / HEX / FULL instantiations of me_body.inc with the chroma SATD terms of subpelCompare (motion.cpp:1805-1865)
#include "me_body.inc"

int xh_me_chroma(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                 const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
                 int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
                 const void* subpelPlanes, int64_t planeElems, const x265hip_me_chroma* ch)
{
    const ChromaArgs ca = { (const pixel*)ch->curCb, (const pixel*)ch->curCr, ch->curStrideC, (const pixel*)ch->refCb, (const pixel*)ch->refCr, ch->refStrideC, ch->curOffC, ch->refOffC };
    return dispatch_me_chroma<0>(stream, w, h, curPlane, curStride, refPlane, refStride, tasks, n, costRow, costHalfRange, merange, method, subpelRefine, results, mvpSource,
                                 subpelPlanes, planeElems, &ca);
}
