/* x265_oracle_la.h -- TEST INFRASTRUCTURE ONLY: restated lookahead frame-cost path (see x265_oracle_la.c). */
#ifndef X265_ORACLE_LA_H
#define X265_ORACLE_LA_H
#include "x265_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif
int xo_lookahead_qp(void);                                       /* X265_LOOKAHEAD_QP of this bit depth */

/* Intra cost of every 8x8 block of a lowres picture (slicetype.cpp:755-864).  plane0 = pixel (0,0) of the padded full-pel
 * lowres plane; invQscale (8.8 fixed point per block) may be NULL.  lowresCosts = the uint16 packed value widened to int32;
 * sums = { costEst, costEstAq }. */
void xo_lowres_intra_estimate(const xo_pixel* plane0, intptr_t stride, int widthInCU, int heightInCU, const int32_t* invQscale,
                              int32_t* intraCost, int32_t* intraMode, int32_t* lowresCosts, int32_t* rowSatds, int64_t* sums);

/* Frame cost of one (p0, b, p1) choice (slicetype.cpp:4365-4640, serial loops, no HME / weightp / slices).
 * ref0 / ref1 = the four half-pel planes (pixel (0,0) each) of frames p0 / p1; ref1 = NULL for a P estimate (b == p1);
 * ref0w = the weighted copy of p0's planes list 0 is searched in (LookaheadTLD::weightsAnalyse, slicetype.cpp:919-1020) or NULL.
 * mvsN (x,y pairs, quarter-pel) and mvCostsN are in/out: read when doSearchN == 0 (results cached by an earlier estimate
 * with the same reference distance), written otherwise.  sums = { costEst (before the B normalisation :4456-4457),
 * costEstAq, intraMbs }. */
void xo_lowres_frame_cost(const xo_pixel* fencPlane0, const xo_pixel* const* ref0, const xo_pixel* const* ref1, const xo_pixel* const* ref0w, intptr_t stride,
                          int widthInCU, int heightInCU, const int32_t* intraCost, const int32_t* invQscale, const uint16_t* costRowCentre,
                          int doSearch0, int doSearch1, int rowsPerSlice /* 0 = one slice; Lookahead::m_numRowsPerSlice */, int32_t* mvs0, int32_t* mvCosts0, int32_t* mvs1, int32_t* mvCosts1,
                          int32_t* lowresCosts, int32_t* rowSatds, int64_t* sums);
/* The same with --hme (slicetype.cpp:4430-4439, 4483-4575): the searches run on the quarter-resolution pictures first (Lowres::lowerResPlane, rows stride apart; the
 * Lookahead::m_4x4Width x m_4x4Height grid; hmeRange[0], hmeSearchMethod[0]) into mvs / mvCosts of the searched lists (Lowres::lowerResMvs / lowerResMvCosts), then on the
 * half-resolution ones (hmeRange[1], hmeSearchMethod[1]) with twice the quarter-resolution MV of block (cuX / 2) + (cuY / 2) * widthInCU / 2 as a fifth predictor candidate.
 * method: XO_ME_HEX or XO_ME_UMH (x265_oracle_me.h).  hme == NULL: xo_lowres_frame_cost. */
typedef struct xo_la_hme
{
    const xo_pixel* fenc;                         /* lowerResPlane[0] of picture b, pixel (0,0) */
    const xo_pixel* const* ref0; const xo_pixel* const* ref1;     /* the four lowerResPlane of p0 / p1 (ref1 unused in a P estimate) */
    intptr_t stride; int wcu, hcu;
    int method[2], range[2];
    int32_t* mvs[2]; int32_t* mvCosts[2];
} xo_la_hme;
void xo_lowres_frame_cost_hme(const xo_pixel* fencPlane0, const xo_pixel* const* ref0, const xo_pixel* const* ref1, const xo_pixel* const* ref0w, intptr_t stride,
                              int widthInCU, int heightInCU, const int32_t* intraCost, const int32_t* invQscale, const uint16_t* costRowCentre,
                              int doSearch0, int doSearch1, int rowsPerSlice, int32_t* mvs0, int32_t* mvCosts0, int32_t* mvs1, int32_t* mvCosts1,
                              int32_t* lowresCosts, int32_t* rowSatds, int64_t* sums, const xo_la_hme* hme);
/* cuTree (slicetype.cpp:3850-3953): propagate the cost of picture b into its references; see x265_oracle_la.c */
void xo_cu_propagate_cost(int32_t* dst, const uint16_t* propagateIn, const int32_t* intraCosts, const uint16_t* interCosts,
                          const int32_t* invQscales, double fpsFactor, int len);
void xo_estimate_cu_propagate(int widthInCU, int heightInCU, int distP0, int distP1, int weightedBiPred, double fpsFactor, int referenced,
                              const int32_t* intraCost, const uint16_t* lowresCosts, const int32_t* invQscale,
                              const int32_t* mvs0, const int32_t* mvs1, uint16_t* propB, uint16_t* prop0, uint16_t* prop1);
void xo_cutree_finish(int ncu, const int32_t* intraCost, const int32_t* invQscale, const uint16_t* propagateCost, const double* qpAqOffset, int fpsFactor,
                      double weightdelta, double strength, double* qpCuTreeOffset);
#ifdef __cplusplus
}
#endif
#endif
