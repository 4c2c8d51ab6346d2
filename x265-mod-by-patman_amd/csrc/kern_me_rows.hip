// kern_me_rows.hip -- me_body.inc built for tasks that carry their own MVD cost row (X265HIP_ME_ROWS; x265hip_me_batch_rows): DIA / HEX / FULL
#define XH_ME_ROWS 1
#include "me_body.inc"

int xh_me_rows(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
               const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
               int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
               const void* subpelPlanes, int64_t planeElems)
{
    return dispatch_me<0>(stream, w, h, curPlane, curStride, refPlane, refStride, tasks, n, costRow, costHalfRange, merange, method, subpelRefine, results, mvpSource, subpelPlanes, planeElems);
}
