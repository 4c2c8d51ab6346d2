// kern_me.hip -- x265hip_me_batch: DIA / HEX / FULL kernels live in this translation unit, the STAR and UMH kernels (whose
// pattern code is large and kept out of line) in kern_me_star.hip / kern_me_umh.hip; all are generated from me_body.inc.
#include "me_body.inc"

int xh_me_star(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
               const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
               int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
               const void* subpelPlanes, int64_t planeElems);

int xh_me_umh(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
               const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
               int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
               const void* subpelPlanes, int64_t planeElems);

int xh_me_sea(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
              const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
              int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
              const void* subpelPlanes, int64_t planeElems, const uint32_t* integral, int64_t integralElems);

int xh_me_chroma(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                 const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
                 int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
                 const void* subpelPlanes, int64_t planeElems, const x265hip_me_chroma* ch);
int xh_me_star_chroma(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                      const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
                      int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
                      const void* subpelPlanes, int64_t planeElems, const x265hip_me_chroma* ch);

// the search of Search::predInterSearch: chroma SATD terms in every sub-pel cost (4:2:0)
extern "C" int x265hip_me_batch_chroma(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                                       const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
                                       int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
                                       const void* subpelPlanes, int64_t planeElems, const x265hip_me_chroma* chroma)
{
    if (n <= 0) return X265HIP_OK;
    if (w < 4 || h < 4 || w > 64 || h > 64 || ((w | h) & 3) || !tasks || !results || !costRow || costHalfRange < 1)
    { set_error("me_batch_chroma: bad arguments"); return X265HIP_EARG; }
    if (!chroma || !chroma->curCb || !chroma->curCr || !chroma->refCb || !chroma->refCr || !chroma->curOffC || !chroma->refOffC)
    { set_error("me_batch_chroma: chroma planes and per-task chroma offsets are required"); return X265HIP_EARG; }
    if (method != X265HIP_ME_DIA && method != X265HIP_ME_HEX && method != X265HIP_ME_STAR && method != X265HIP_ME_FULL)
    { set_error("me_batch_chroma: search method %d is not offered with chroma terms (DIA / HEX / STAR / FULL are)", method); return X265HIP_EARG; }
    if (subpelRefine < 0 || subpelRefine > 7 || merange < 1) { set_error("me_batch_chroma: bad subme/merange"); return X265HIP_EARG; }
    if (!subpelPlanes || planeElems <= 0 || ((uintptr_t)subpelPlanes & 7)) { set_error("me_batch_chroma: the phase planes of the reference picture are required"); return X265HIP_EARG; }
    if (method == X265HIP_ME_STAR)
        return xh_me_star_chroma(stream, w, h, curPlane, curStride, refPlane, refStride, tasks, n, costRow, costHalfRange, merange, method, subpelRefine, results, mvpSource, subpelPlanes, planeElems, chroma);
    return xh_me_chroma(stream, w, h, curPlane, curStride, refPlane, refStride, tasks, n, costRow, costHalfRange, merange, method, subpelRefine, results, mvpSource, subpelPlanes, planeElems, chroma);
}

// SEA: the search needs the 12 integral planes of the reference picture (x265hip_sea_integral_planes), laid out like the reference plane
extern "C" int x265hip_me_batch_sea(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                                    const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
                                    int merange, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
                                    const void* subpelPlanes, int64_t planeElems, const uint32_t* integralPlanes, int64_t integralPlaneElems)
{
    if (n <= 0) return X265HIP_OK;
    if (w < 4 || h < 4 || w > 64 || h > 64 || ((w | h) & 3) || !tasks || !results || !costRow || costHalfRange < 1 || !integralPlanes || integralPlaneElems <= 0)
    { set_error("me_batch_sea: bad arguments"); return X265HIP_EARG; }
    if ((w == 8 && h == 4) || (w == 4 && h == 8) || (w == 8 && h == 32) || (w == 32 && h == 8))
    {   // motion.cpp:1467-1468,1504-1510: for these shapes the reference's DC terms read fenc rows / columns outside the PU, i.e. whatever
        // earlier PUs left in MotionEstimate::fencPUYuv -- its result is not a function of the inputs, so there is nothing to reproduce
        set_error("me_batch_sea: the reference's SEA result for %dx%d PUs depends on stale buffer contents; not offloaded", w, h); return X265HIP_EARG;
    }
    if (subpelRefine < 0 || subpelRefine > 7 || merange < 1) { set_error("me_batch_sea: bad subme/merange"); return X265HIP_EARG; }
    if (subpelPlanes && (planeElems <= 0 || ((uintptr_t)subpelPlanes & 7))) { set_error("me_batch_sea: bad subpel planes"); return X265HIP_EARG; }
    return xh_me_sea(stream, w, h, curPlane, curStride, refPlane, refStride, tasks, n, costRow, costHalfRange, merange, X265HIP_ME_SEA, subpelRefine, results, mvpSource,
                     subpelPlanes, planeElems, integralPlanes, integralPlaneElems);
}

extern "C" int x265hip_me_batch(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                                const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
                                int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
               const void* subpelPlanes, int64_t planeElems)
{
    if (n <= 0) return X265HIP_OK;
    if (w < 4 || h < 4 || w > 64 || h > 64 || ((w | h) & 3) || !tasks || !results || !costRow || costHalfRange < 1)
    { set_error("me_batch: bad arguments"); return X265HIP_EARG; }
    if (method != X265HIP_ME_DIA && method != X265HIP_ME_HEX && method != X265HIP_ME_UMH && method != X265HIP_ME_STAR && method != X265HIP_ME_FULL)
    { set_error("me_batch: search method %d is not offloaded here (DIA/HEX/UMH/STAR/FULL are; SEA needs its integral planes: x265hip_me_batch_sea)", method); return X265HIP_EARG; }
    if (subpelRefine < 0 || subpelRefine > 7 || merange < 1) { set_error("me_batch: bad subme/merange"); return X265HIP_EARG; }
    if (subpelPlanes && (planeElems <= 0 || ((uintptr_t)subpelPlanes & 7))) { set_error("me_batch: bad subpel planes"); return X265HIP_EARG; }
    if (method == X265HIP_ME_STAR)
        return xh_me_star(stream, w, h, curPlane, curStride, refPlane, refStride, tasks, n, costRow, costHalfRange, merange, method, subpelRefine, results, mvpSource, subpelPlanes, planeElems);
    if (method == X265HIP_ME_UMH)
        return xh_me_umh(stream, w, h, curPlane, curStride, refPlane, refStride, tasks, n, costRow, costHalfRange, merange, method, subpelRefine, results, mvpSource, subpelPlanes, planeElems);
    return dispatch_me<0>(stream, w, h, curPlane, curStride, refPlane, refStride, tasks, n, costRow, costHalfRange, merange, method, subpelRefine, results, mvpSource, subpelPlanes, planeElems);
}

// ... the same search for tasks of different qp in one launch (Analysis::setLambdaFromQP runs per CU): costRows is a TABLE of rows, 2 * costHalfRange + 1 entries each, and
// a task with X265HIP_ME_ROWS in its flags prices its MVDs with row (flags >> 8) & 0xFF.  Kernels built for it (kern_me*_rows.hip): the cost lookups of such tasks go to
// memory, which the plain kernels do not pay for.
int xh_me_rows(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride, const x265hip_me_task* tasks, int n, const uint16_t* costRow,
               int costHalfRange, int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource, const void* subpelPlanes, int64_t planeElems);
int xh_me_star_rows(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride, const x265hip_me_task* tasks, int n, const uint16_t* costRow,
                    int costHalfRange, int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource, const void* subpelPlanes, int64_t planeElems);
int xh_me_umh_rows(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride, const x265hip_me_task* tasks, int n, const uint16_t* costRow,
                   int costHalfRange, int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource, const void* subpelPlanes, int64_t planeElems);
extern "C" int x265hip_me_batch_rows(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                                     const x265hip_me_task* tasks, int n, const uint16_t* costRows, int costHalfRange,
                                     int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
                                     const void* subpelPlanes, int64_t planeElems)
{
    if (n <= 0) return X265HIP_OK;
    if (w < 4 || h < 4 || w > 64 || h > 64 || ((w | h) & 3) || !tasks || !results || !costRows || costHalfRange < 1)
    { set_error("me_batch_rows: bad arguments"); return X265HIP_EARG; }
    if (method != X265HIP_ME_DIA && method != X265HIP_ME_HEX && method != X265HIP_ME_UMH && method != X265HIP_ME_STAR && method != X265HIP_ME_FULL)
    { set_error("me_batch_rows: search method %d is not offloaded here (DIA/HEX/UMH/STAR/FULL are)", method); return X265HIP_EARG; }
    if (subpelRefine < 0 || subpelRefine > 7 || merange < 1) { set_error("me_batch_rows: bad subme/merange"); return X265HIP_EARG; }
    if (subpelPlanes && (planeElems <= 0 || ((uintptr_t)subpelPlanes & 7))) { set_error("me_batch_rows: bad subpel planes"); return X265HIP_EARG; }
    if (method == X265HIP_ME_STAR)
        return xh_me_star_rows(stream, w, h, curPlane, curStride, refPlane, refStride, tasks, n, costRows, costHalfRange, merange, method, subpelRefine, results, mvpSource, subpelPlanes, planeElems);
    if (method == X265HIP_ME_UMH)
        return xh_me_umh_rows(stream, w, h, curPlane, curStride, refPlane, refStride, tasks, n, costRows, costHalfRange, merange, method, subpelRefine, results, mvpSource, subpelPlanes, planeElems);
    return xh_me_rows(stream, w, h, curPlane, curStride, refPlane, refStride, tasks, n, costRows, costHalfRange, merange, method, subpelRefine, results, mvpSource, subpelPlanes, planeElems);
}
