/* x265_oracle_me.h -- TEST INFRASTRUCTURE ONLY: restated motion-search driver (see x265_oracle_me.c). */
#ifndef X265_ORACLE_ME_H
#define X265_ORACLE_ME_H
#include "x265_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif
/* search methods, numbered like X265_*_SEARCH in the reference's x265.h */
enum { XO_ME_DIA = 0, XO_ME_HEX = 1, XO_ME_UMH = 2, XO_ME_STAR = 3, XO_ME_SEA = 4, XO_ME_FULL = 5 };

double xo_lambda(int qp);                                        /* constants.cpp x265_lambda_tab */
void xo_mvcost_row(int qp, int halfRange, uint16_t* out);        /* out[halfRange + d] = cost of MVD component d (qpel) */

/* One PU, one reference.  fencPlane points at the PU's top-left source pixel, fref at the co-located
 * reference pixel (planes must have margins covering the search window + 4).  Returns the cost; the
 * chosen quarter-pel MV goes to outQMv[0..1]. */
int xo_motion_estimate(const xo_pixel* fencPlane, intptr_t fencStride, int w, int h,
                       const xo_pixel* fref, intptr_t refStride, const int32_t* bounds,
                       int qmvpx, int qmvpy, int numCand, const int32_t* mvc,
                       int merange, int method, int subme, const uint16_t* costRowCentre, int32_t* outQMv);
#ifdef __cplusplus
}
#endif
#endif
