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
/* the same with the chroma SATD terms of subpelCompare (the Yuv overload of setSourcePU with bChroma, 4:2:0): fencCb / fencCr point at the PU's
 * chroma source blocks, refCb / refCr at the co-located reference chroma pixels.  Not reentrant (test infrastructure). */
int xo_motion_estimate_chroma(const xo_pixel* fencPlane, intptr_t fencStride, int w, int h, const xo_pixel* fref, intptr_t refStride, const int32_t* bounds,
                              int qmvpx, int qmvpy, int numCand, const int32_t* mvc, int merange, int method, int subme, const uint16_t* costRowCentre, int32_t* outQMv,
                              const xo_pixel* fencCb, const xo_pixel* fencCr, intptr_t fencStrideC, const xo_pixel* refCb, const xo_pixel* refCr, intptr_t refStrideC);
/* SEA (XO_ME_SEA): the 12 integral planes of the reference picture (framefilter.cpp:740-833; order 32x32, 32x24, 32x8, 24x32, 16x16,
 * 16x12, 16x4, 12x16, 8x32, 8x8, 4x16, 4x4) and the search with them; integral[k] points at the PU's co-located position. */
void xo_sea_integral_planes(const xo_pixel* pic, intptr_t stride, int maxHeight, int padX, int padY, uint32_t* const* planes);
int xo_motion_estimate_sea(const xo_pixel* fencPlane, intptr_t fencStride, int w, int h,
                           const xo_pixel* fref, intptr_t refStride, const int32_t* bounds,
                           int qmvpx, int qmvpy, int numCand, const int32_t* mvc,
                           int merange, int method, int subme, const uint16_t* costRowCentre, int32_t* outQMv, const uint32_t* const* integral);
#ifdef __cplusplus
}
#endif
#endif

#ifdef __cplusplus
extern "C" {
#endif
/* One luma TU of the inter residual path, restating the caller chain
 *   Predict::predInterLumaPixel (predict.cpp:279-300) -> sub_ps -> Quant::transformNxN (quant.cpp:397-480, rdoq off)
 *   [-> Quant::invtransformNxN (quant.cpp:543-605, flat lists) -> add_ps -> sse_pp   when recon != NULL]
 * cur / fref point at the TU's top-left source pixel and the co-located reference pixel.
 * quantCoeff may be NULL (flat s_quantScales[qp%6], scalinglist.cpp:129).  Returns numSig. */
uint32_t xo_tq_tu(int log2TrSize, const xo_pixel* cur, intptr_t curStride, const xo_pixel* fref, intptr_t refStride,
                  int qmvx, int qmvy, int qp, int addNumerator /*171 or 85*/, const int32_t* quantCoeff,
                  int16_t* coeff /*N*N*/, int32_t* deltaU /*N*N or NULL*/,
                  xo_pixel* recon /*or NULL*/, intptr_t reconStride, uint64_t* sse);
uint32_t xo_mv_bitcost(const float* bitsCentre, int mvx, int mvy, int px, int py);   /* BitCost::bitcost(mv, mvp), bitcost.h:66-70 */
/* Search::selectMVP / checkBestMVP / updateMVP (search.cpp:2347-2382, 4947-4967); see x265_oracle_me.c */
int xo_select_mvp(int w, int h, const xo_pixel* fenc, intptr_t fencStride, const xo_pixel* fref, intptr_t refStride, const int32_t* amvp, const int32_t* clip, int32_t* costs);
void xo_check_best_mvp(const float* bitsCentre, uint64_t lambda, const int32_t* amvp, int mvx, int mvy, uint32_t* io);
void xo_update_mvp(const float* bitsCentre, uint64_t lambda, int amvpx, int amvpy, int mvx, int mvy, int alterx, int altery, uint32_t* io);
/* CUData::getPMV (cudata.cpp:1806-1990); see x265_oracle_me.c */
int xo_get_pmv(const int32_t* nb, int list, int refIdx, int curPOC, int temporalEnabled, const int32_t* refPOC, int colPOC, int colRefPOC, int32_t* amvp, int32_t* mvc);
/* MotionEstimate::diamondSearch (motion.cpp:631-773); bounds and outMv in full pels; qmvp = the MVD origin of mvcost (setMVP) */
int xo_diamond_search(const xo_pixel* fencPlane, intptr_t fencStride, int w, int h, const xo_pixel* fref, intptr_t refStride, const int32_t* bounds,
                      int qmvpx, int qmvpy, const uint16_t* costRowCentre, int32_t* outMv);
uint32_t xo_tq_tu_bi(int log2TrSize, const xo_pixel* cur, intptr_t curStride, const xo_pixel* fref0, const xo_pixel* fref1, intptr_t refStride,
                     int qmv0x, int qmv0y, int qmv1x, int qmv1y, int qp, int addNumerator, const int32_t* quantCoeff,
                     int16_t* coeff, int32_t* deltaU, xo_pixel* recon /*or NULL*/, intptr_t reconStride, uint64_t* sse);
uint32_t xo_tq_tu_dst4(const xo_pixel* cur, intptr_t curStride, const xo_pixel* predPlane, intptr_t predStride, int qp, int addNumerator,
                       int16_t* coeff, int32_t* deltaU, xo_pixel* recon /*or NULL*/, intptr_t reconStride, uint64_t* sse);
uint32_t xo_tq_tu_chroma(int log2TrSize, const xo_pixel* cur, intptr_t curStride, const xo_pixel* fref, intptr_t refStride,
                         int qmvx, int qmvy, int qp, int addNumerator, const int32_t* quantCoeff,
                         int16_t* coeff, int32_t* deltaU, xo_pixel* recon, intptr_t reconStride, uint64_t* sse);
#ifdef __cplusplus
}
#endif

#ifdef __cplusplus
extern "C" {
#endif
/* the per-PU choice among references and the bidirectional candidate (tail of Search::puMotionEstimation, search.cpp:258-556; see x265_oracle_me.c) */
void xo_mvbits_row(int halfRange, float* out);
uint64_t xo_rd_lambda(int qp);
int xo_bidir_satd(int w, int h, const xo_pixel* fenc, intptr_t fencStride, const xo_pixel* ref0, intptr_t stride0, int mv0x, int mv0y,
                  const xo_pixel* ref1, intptr_t stride1, int mv1x, int mv1y);
void xo_inter_merge(int w, int h, const int32_t* numRef, const int32_t* mv, const int32_t* mvp, const int32_t* cost, const int32_t* mvcost,
                    const float* bitsCentre, uint64_t lambda, int bidir, int sourceMaxDim, const int32_t* clip,
                    const xo_pixel* fenc, intptr_t fencStride, const xo_pixel* const* refs, intptr_t refStride, int32_t* out, uint32_t* mvCostOut);
#ifdef __cplusplus
}
#endif
