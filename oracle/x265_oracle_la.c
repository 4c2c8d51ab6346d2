/*
 * x265_oracle_la.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's LOOKAHEAD frame-cost path on half-resolution ("lowres") pictures:
 *   - LookaheadTLD::lowresIntraEstimate              (encoder/slicetype.cpp:755-864)
 *   - CostEstimateGroup::estimateFrameCost, serial   (encoder/slicetype.cpp:4365-4463; weightsAnalyse itself stays outside, its weighted planes come in as ref0w;
 *                                                      with --hme the quarter-resolution sweep of :4430-4439 first, whose MVs seed the half-resolution one, :4532-4535)
 *   - CostEstimateGroup::estimateCUCost              (encoder/slicetype.cpp:4467-4640)
 *   - MotionEstimate::motionEstimate, ref->isLowres  (encoder/motion.cpp:923-1140 HEX, :1142-1326 UMH as an --hme level runs it, :1644-1773 with the lowres branch :1667-1699)
 *   - ReferencePlanes::lowresMC / lowresQPelCost     (common/lowres.h:75-124)
 *   - Lookahead::estimateCUPropagate + propagateCost  (encoder/slicetype.cpp:3850-3953, common/pixel.cpp:906-931)
 * built on the primitive restatements of x265_oracle.c.  Pinned against the REAL reference classes
 * (oracle/_ref/x265la_*, oracle/ref_lookahead.cpp) by tests/test_lookahead_oracle_vs_ref.py.
 */
#include "x265_oracle_la.h"
#include "x265_oracle_me.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CU 8                                   /* X265_LOWRES_CU_SIZE (common.h) */
#define COST_MAX (1 << 28)                     /* MotionEstimate::COST_MAX (motion.h) */
#define LOWRES_COST_MASK ((1 << 14) - 1)       /* slicetype.h:42-43 */
#define LOWRES_COST_SHIFT 14

typedef struct { int x, y; } mv_t;

int xo_lookahead_qp(void) { return 12 + 6 * (X265_DEPTH - 8); }          /* X265_LOOKAHEAD_QP, common.h:223 */

static const unsigned char k_filterFlags[35] = {                           /* constants.cpp:561 g_intraFilterFlags */
    0x38, 0x00, 0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x38 };

static int frame_score_cu(int cuX, int cuY, int wcu, int hcu)
{   /* edge blocks are left out of the frame totals (slicetype.cpp:843-844, 4607-4608) */
    return (cuX > 0 && cuX < wcu - 1 && cuY > 0 && cuY < hcu - 1) || wcu <= 2 || hcu <= 2;
}

/* slicetype.cpp:755-864 */
void xo_lowres_intra_estimate(const xo_pixel* plane0, intptr_t stride, int wcu, int hcu, const int32_t* invQscale,
                              int32_t* intraCost, int32_t* intraMode, int32_t* lowresCosts, int32_t* rowSatds, int64_t* sums)
{
    const int lambda = (int)xo_lambda(xo_lookahead_qp());
    const int intraPenalty = 5 * lambda, lowresPenalty = 4;
    int64_t costEst = 0, costEstAq = 0;
    for (int cuY = 0; cuY < hcu; cuY++)
    {
        rowSatds[cuY] = 0;
        for (int cuX = 0; cuX < wcu; cuX++)
        {
            const int cuXY = cuX + cuY * wcu;
            const xo_pixel* cur = plane0 + CU * cuX + (intptr_t)CU * cuY * stride;
            xo_pixel fenc[CU * CU], pred[CU * CU], nb[2][4 * CU + 1];
            for (int y = 0; y < CU; y++) memcpy(fenc + CU * y, cur + y * stride, CU * sizeof(xo_pixel));
            const xo_pixel* tl = cur - stride - 1;
            memcpy(nb[0], tl, (2 * CU + 1) * sizeof(xo_pixel));                                   /* top-left, top, top-right */
            for (int i = 1; i <= 2 * CU; i++) nb[0][2 * CU + i] = tl[i * stride];                /* left, below-left */
            xo_intra_filter(CU, nb[0], nb[1]);
            int icost = COST_MAX, ilow = 0, cost;
            xo_intra_pred(CU, pred, CU, nb[0], 1, 1);                                             /* DC, edge filter on (cuSize <= 16) */
            cost = xo_satd(CU, CU, fenc, CU, pred, CU);
            if (cost < icost) { icost = cost; ilow = 1; }
            xo_intra_pred(CU, pred, CU, nb[1], 0, 0);                                             /* planar on the filtered neighbours (cuSize >= 8) */
            cost = xo_satd(CU, CU, fenc, CU, pred, CU);
            if (cost < icost) { icost = cost; ilow = 0; }
            int acost = COST_MAX, alow = 4;
            for (int mode = 5; mode < 35; mode += 5)
            {
                xo_intra_pred(CU, pred, CU, nb[!!(k_filterFlags[mode] & CU)], mode, 1);
                cost = xo_satd(CU, CU, fenc, CU, pred, CU);
                if (cost < acost) { acost = cost; alow = mode; }
            }
            for (int dist = 2; dist >= 1; dist--)
            {
                const int two[2] = { alow - dist, alow + dist };
                for (int k = 0; k < 2; k++)
                {
                    const int mode = two[k];
                    xo_intra_pred(CU, pred, CU, nb[!!(k_filterFlags[mode] & CU)], mode, 1);
                    cost = xo_satd(CU, CU, fenc, CU, pred, CU);
                    if (cost < acost) { acost = cost; alow = mode; }
                }
            }
            if (acost < icost) { icost = acost; ilow = alow; }
            icost += intraPenalty + lowresPenalty;
            lowresCosts[cuXY] = (uint16_t)(icost < LOWRES_COST_MASK ? icost : LOWRES_COST_MASK);
            intraCost[cuXY] = icost; intraMode[cuXY] = ilow;
            const int score = frame_score_cu(cuX, cuY, wcu, hcu);
            const int icostAq = (score && invQscale) ? ((icost * invQscale[cuXY] + 128) >> 8) : icost;
            if (score) { costEst += icost; costEstAq += icostAq; }
            rowSatds[cuY] += icostAq;
        }
    }
    sums[0] = costEst; sums[1] = costEstAq;
}

/* ---- lowres reference access (lowres.h:75-124) ---- */
typedef struct
{
    const xo_pixel* plane[4]; intptr_t stride;    /* block origin inside the four half-pel planes (0 = full, 1 = H, 2 = V, 3 = HV) */
    xo_pixel fenc[CU * CU];
    const uint16_t* cost; mv_t mvp;
    /* the final zero-MV check (motion.cpp:1763-1768) goes through subpelCompare, which reads ReferencePlanes::fpelPlane[0] with lumaStride whatever the level: on the
       quarter-resolution level of --hme that is the HALF-resolution plane at the quarter-resolution block offset.  zero == NULL: plane[0] / stride. */
    const xo_pixel* zero; intptr_t zeroStride;
} la_t;

static const xo_pixel* lowres_mc(const la_t* m, int qx, int qy, xo_pixel* buf, intptr_t* outStride)
{
    if ((qx | qy) & 1)
    {
        const int hpelA = (qy & 2) | ((qx & 2) >> 1);
        const xo_pixel* a = m->plane[hpelA] + (qx >> 2) + (qy >> 2) * m->stride;
        const int qx2 = qx + (qx & 1), qy2 = qy + (qy & 1);
        const int hpelB = (qy2 & 2) | ((qx2 & 2) >> 1);
        const xo_pixel* b = m->plane[hpelB] + (qx2 >> 2) + (qy2 >> 2) * m->stride;
        xo_pixelavg_pp(CU, CU, buf, CU, a, m->stride, b, m->stride);
        *outStride = CU;
        return buf;
    }
    *outStride = m->stride;
    return m->plane[(qy & 2) | ((qx & 2) >> 1)] + (qx >> 2) + (qy >> 2) * m->stride;
}
static int qpel_cost(const la_t* m, int qx, int qy, int useSatd)
{
    xo_pixel buf[CU * CU]; intptr_t s;
    const xo_pixel* p = lowres_mc(m, qx, qy, buf, &s);
    return useSatd ? xo_satd(CU, CU, m->fenc, CU, p, s) : xo_sad(CU, CU, m->fenc, CU, p, s);
}
static inline int mvcost(const la_t* m, int qx, int qy) { return (uint16_t)(m->cost[qx - m->mvp.x] + m->cost[qy - m->mvp.y]); }
static inline int sad_at(const la_t* m, int mx, int my) { return xo_sad(CU, CU, m->fenc, CU, m->plane[0] + mx + my * m->stride, m->stride); }

static const mv_t hex2[8] = { {-1,-2}, {-2,0}, {-1,2}, {1,2}, {2,0}, {1,-2}, {-1,-2}, {-2,0} };
static const unsigned char mod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };
static const mv_t square1[9] = { {0,0}, {0,-1}, {0,1}, {-1,0}, {1,0}, {-1,-1}, {-1,1}, {1,-1}, {1,1} };

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int in_range(mv_t v, mv_t mn, mv_t mx) { return v.x >= mn.x && v.x <= mx.x && v.y >= mn.y && v.y <= mx.y; }

/* X265_UMH_SEARCH (motion.cpp:1142-1326) on an 8x8 lowres block: no motion candidates, so the range is never adapted (:1186).  Returns 0 when the search ends inside
 * (the early terminations :1170-1183, or a best point out of range :1321-1322), 1 when it goes on into the hexagon search (`goto me_hex2`).
 * The statement order, the strict `<` and the vertical-only range test of COST_MV_X4 follow x265_oracle_me.c's restatement of the same code. */
static int lowres_umh(const la_t* m, mv_t mvmin, mv_t mvmax, mv_t pmv /* full-pel */, int merange, mv_t* bmvIO, int* bcostIO)
{
    static const mv_t hex4[16] = { {0,-4}, {0,4}, {-2,-3}, {2,-3}, {-4,-2}, {4,-2}, {-4,-1}, {4,-1},
                                   {-4,0}, {4,0}, {-4,1}, {4,1}, {-4,2}, {4,2}, {-2,3}, {2,3} };       /* motion.cpp:67-73 */
    const int scale = (CU * CU) >> 4;                                                              /* sizeScale[LUMA_8x8], :60-61,123-153 */
    mv_t bmv = *bmvIO; int bcost = *bcostIO;
#define SAD_THRESH(v) (bcost < (((v) >> 4) * scale))
#define COST_MV(mx_, my_) do { const int c_ = sad_at(m, (mx_), (my_)) + mvcost(m, (mx_) * 4, (my_) * 4); if (c_ < bcost) { bcost = c_; bmv.x = (mx_); bmv.y = (my_); } } while (0)
#define X4(ax, ay, bx_, by_, cx, cy, dx, dy) do { const mv_t d_[4] = { {ax, ay}, {bx_, by_}, {cx, cy}, {dx, dy} }; \
        for (int k_ = 0; k_ < 4; k_++) { \
            const int c_ = sad_at(m, omv.x + d_[k_].x, omv.y + d_[k_].y) + mvcost(m, (omv.x + d_[k_].x) * 4, (omv.y + d_[k_].y) * 4); \
            if ((omv.y + d_[k_].y >= mvmin.y) & (omv.y + d_[k_].y <= mvmax.y)) \
                if (c_ < bcost) { bcost = c_; bmv.x = omv.x + d_[k_].x; bmv.y = omv.y + d_[k_].y; } } } while (0)
#define DIA1(mx_, my_) do { omv.x = (mx_); omv.y = (my_); X4(0, -1, 0, 1, -1, 0, 1, 0); } while (0)
#define CROSS(start, x_max, y_max) do { int i_ = (start); \
        if ((x_max) <= imin(mvmax.x - omv.x, omv.x - mvmin.x)) for (; i_ < (x_max) - 2; i_ += 4) X4(i_, 0, -i_, 0, i_ + 2, 0, -i_ - 2, 0); \
        for (; i_ < (x_max); i_ += 2) { if (omv.x + i_ <= mvmax.x) COST_MV(omv.x + i_, omv.y); if (omv.x - i_ >= mvmin.x) COST_MV(omv.x - i_, omv.y); } \
        i_ = (start); \
        if ((y_max) <= imin(mvmax.y - omv.y, omv.y - mvmin.y)) for (; i_ < (y_max) - 2; i_ += 4) X4(0, i_, 0, -i_, 0, i_ + 2, 0, -i_ - 2); \
        for (; i_ < (y_max); i_ += 2) { if (omv.y + i_ <= mvmax.y) COST_MV(omv.x, omv.y + i_); if (omv.y - i_ >= mvmin.y) COST_MV(omv.x, omv.y - i_); } } while (0)
    mv_t omv = bmv;
    int ucost1 = bcost, ucost2, cross_start = 1, done = 0, hexToo = 0;
    DIA1(pmv.x, pmv.y);
    if (pmv.x | pmv.y) DIA1(0, 0);
    ucost2 = bcost;
    if ((bmv.x | bmv.y) && !(bmv.x == pmv.x && bmv.y == pmv.y)) DIA1(bmv.x, bmv.y);
    if (bcost == ucost2) cross_start = 3;
    omv = bmv;
    if (bcost == ucost2 && SAD_THRESH(2000))
    {
        X4(0, -2, -1, -1, 1, -1, -2, 0);
        X4(2, 0, -1, 1, 1, 1, 0, 2);
        if (bcost == ucost1 && SAD_THRESH(500)) done = 1;
        else if (bcost == ucost2)
        {
            const int range = (int16_t)((merange >> 1) | 1);
            CROSS(3, range, range);
            X4(-1, -2, 1, -2, -2, -1, 2, -1);
            X4(-2, 1, 2, 1, -1, 2, 1, 2);
            if (bcost == ucost2) done = 1;
            cross_start = range + 2;
        }
    }
    if (!done)
    {
        CROSS(cross_start, merange, merange >> 1);
        X4(-2, -2, -2, 2, 2, -2, 2, 2);
        omv = bmv;
        int i = 1;
        do
            for (int j = 0; j < 16; j++)
            {
                const mv_t mv = { omv.x + hex4[j].x * i, omv.y + hex4[j].y * i };
                if (in_range(mv, mvmin, mvmax)) COST_MV(mv.x, mv.y);
            }
        while (++i <= merange >> 2);
        hexToo = in_range(bmv, mvmin, mvmax);
    }
#undef SAD_THRESH
#undef COST_MV
#undef X4
#undef DIA1
#undef CROSS
    *bmvIO = bmv; *bcostIO = bcost;
    return hexToo;
}

/* X265_STAR_SEARCH (motion.cpp:1328-1436) with StarPatternSearch (:387-629) on an 8x8 lowres block: the point order, the strict `<`, the per-point window tests of the
 * guarded form (the x4 form visits the same points in the same order), the rounds counter and the raster's fourth mv cost taken at tmv << 3 (:1392) follow
 * x265_oracle_me.c's restatement of the same code; what differs is where the pixels come from (the level's own full-pel plane: `hme ? fpelLowerResPlane : fpelPlane`,
 * :400-401, 940-941). */
typedef struct { mv_t bmv; int bcost, bPointNr, bDistance; } la_star_t;
long xo_la_star_rasters;       /* how many blocks went through the raster refinement (the tests ask that their clips reach it) */
#define PT_DIST(mx_, my_, point, dist) do { const int c_ = sad_at(m, (mx_), (my_)) + mvcost(m, (mx_) * 4, (my_) * 4); \
    if (c_ < s->bcost) { s->bcost = c_; s->bmv.x = (mx_); s->bmv.y = (my_); s->bPointNr = (point); s->bDistance = (dist); } } while (0)
static void lowres_star_pattern(const la_t* m, mv_t mvmin, mv_t mvmax, la_star_t* s, int earlyExitIters, int merange)
{
    const mv_t omv = s->bmv;
    int saved = s->bcost, rounds = 0;
    {
        const int top = omv.y - 1, bottom = omv.y + 1, left = omv.x - 1, right = omv.x + 1;
        if (top >= mvmin.y) PT_DIST(omv.x, top, 2, 1);
        if (left >= mvmin.x) PT_DIST(left, omv.y, 4, 1);
        if (right <= mvmax.x) PT_DIST(right, omv.y, 5, 1);
        if (bottom <= mvmax.y) PT_DIST(omv.x, bottom, 7, 1);
        if (s->bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
    for (int dist = 2; dist <= 8; dist <<= 1)
    {
        const int top = omv.y - dist, bottom = omv.y + dist, left = omv.x - dist, right = omv.x + dist;
        const int top2 = omv.y - (dist >> 1), bottom2 = omv.y + (dist >> 1), left2 = omv.x - (dist >> 1), right2 = omv.x + (dist >> 1);
        saved = s->bcost;
        if (top >= mvmin.y && left >= mvmin.x && right <= mvmax.x && bottom <= mvmax.y)
        {   /* x4 order: (2, 1, 3, 4) then (5, 6, 8, 7) */
            PT_DIST(omv.x, top, 2, dist); PT_DIST(left2, top2, 1, dist >> 1); PT_DIST(right2, top2, 3, dist >> 1); PT_DIST(left, omv.y, 4, dist);
            PT_DIST(right, omv.y, 5, dist); PT_DIST(left2, bottom2, 6, dist >> 1); PT_DIST(right2, bottom2, 8, dist >> 1); PT_DIST(omv.x, bottom, 7, dist);
        }
        else
        {
            if (top >= mvmin.y) PT_DIST(omv.x, top, 2, dist);
            if (top2 >= mvmin.y)
            {
                if (left2 >= mvmin.x) PT_DIST(left2, top2, 1, dist >> 1);
                if (right2 <= mvmax.x) PT_DIST(right2, top2, 3, dist >> 1);
            }
            if (left >= mvmin.x) PT_DIST(left, omv.y, 4, dist);
            if (right <= mvmax.x) PT_DIST(right, omv.y, 5, dist);
            if (bottom2 <= mvmax.y)
            {
                if (left2 >= mvmin.x) PT_DIST(left2, bottom2, 6, dist >> 1);
                if (right2 <= mvmax.x) PT_DIST(right2, bottom2, 8, dist >> 1);
            }
            if (bottom <= mvmax.y) PT_DIST(omv.x, bottom, 7, dist);
        }
        if (s->bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
    for (int dist = 16; dist <= (int16_t)merange; dist <<= 1)
    {
        const int top = omv.y - dist, bottom = omv.y + dist, left = omv.x - dist, right = omv.x + dist;
        saved = s->bcost;
        const int all = top >= mvmin.y && left >= mvmin.x && right <= mvmax.x && bottom <= mvmax.y;
        if (all || top >= mvmin.y) PT_DIST(omv.x, top, 0, dist);
        if (all || left >= mvmin.x) PT_DIST(left, omv.y, 0, dist);
        if (all || right <= mvmax.x) PT_DIST(right, omv.y, 0, dist);
        if (all || bottom <= mvmax.y) PT_DIST(omv.x, bottom, 0, dist);
        for (int index = 1; index < 4; index++)
        {
            const int posYT = top + ((dist >> 2) * index), posYB = bottom - ((dist >> 2) * index);
            const int posXL = omv.x - ((dist >> 2) * index), posXR = omv.x + ((dist >> 2) * index);
            if (all || posYT >= mvmin.y)
            {
                if (all || posXL >= mvmin.x) PT_DIST(posXL, posYT, 0, dist);
                if (all || posXR <= mvmax.x) PT_DIST(posXR, posYT, 0, dist);
            }
            if (all || posYB <= mvmax.y)
            {
                if (all || posXL >= mvmin.x) PT_DIST(posXL, posYB, 0, dist);
                if (all || posXR <= mvmax.x) PT_DIST(posXR, posYB, 0, dist);
            }
        }
        if (s->bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
}
#undef PT_DIST
static void lowres_star(const la_t* m, mv_t mvmin, mv_t mvmax, int merange, mv_t* bmvIO, int* bcostIO)
{
    static const mv_t offsets[16] = { {-1,0}, {0,-1}, {-1,-1}, {1,-1}, {-1,0}, {1,0}, {-1,1}, {-1,-1},
                                      {1,-1}, {1,1}, {-1,0}, {0,1}, {-1,1}, {1,1}, {1,0}, {0,1} };       /* motion.cpp:75-85 */
    mv_t bmv = *bmvIO; int bcost = *bcostIO;
#define COST_MV(mx_, my_) do { const int c_ = sad_at(m, (mx_), (my_)) + mvcost(m, (mx_) * 4, (my_) * 4); if (c_ < bcost) { bcost = c_; bmv.x = (mx_); bmv.y = (my_); } } while (0)
#define TWO_POINTS(nr) do { const mv_t a_ = { bmv.x + offsets[((nr) - 1) * 2].x, bmv.y + offsets[((nr) - 1) * 2].y }, b_ = { bmv.x + offsets[((nr) - 1) * 2 + 1].x, bmv.y + offsets[((nr) - 1) * 2 + 1].y }; \
        if (in_range(a_, mvmin, mvmax)) COST_MV(a_.x, a_.y); if (in_range(b_, mvmin, mvmax)) COST_MV(b_.x, b_.y); } while (0)
    la_star_t st = { bmv, bcost, 0, 0 };
    lowres_star_pattern(m, mvmin, mvmax, &st, 3, merange);
    bmv = st.bmv; bcost = st.bcost;
    int done = 0;
    if (st.bDistance == 1)
    {
        if (st.bPointNr)
        {
            const int saved = bcost;
            TWO_POINTS(st.bPointNr);
            if (bcost == saved) done = 1;
        }
        else done = 1;
    }
    if (!done)
    {
        const int RasterDistance = 5;
        if (st.bDistance > RasterDistance)
        {
            mv_t t;
            xo_la_star_rasters++;
            for (t.y = mvmin.y; t.y <= mvmax.y; t.y += RasterDistance)
                for (t.x = mvmin.x; t.x <= mvmax.x; t.x += RasterDistance)
                {
                    if (t.x + RasterDistance * 3 <= mvmax.x)
                    {
                        int c4[4];
                        for (int k = 0; k < 4; k++) c4[k] = sad_at(m, t.x + RasterDistance * k, t.y);
                        for (int k = 0; k < 4; k++)
                        {
                            if (k) t.x += RasterDistance;
                            const int c = c4[k] + (k == 3 ? mvcost(m, t.x * 8, t.y * 8) : mvcost(m, t.x * 4, t.y * 4));     /* :1392: tmv << 3 */
                            if (c < bcost) { bcost = c; bmv = t; }
                        }
                    }
                    else
                        COST_MV(t.x, t.y);
                }
        }
        int bDistance = st.bDistance;
        while (bDistance > 0)
        {
            st.bmv = bmv; st.bcost = bcost; st.bPointNr = 0; st.bDistance = 0;
            lowres_star_pattern(m, mvmin, mvmax, &st, 32, merange);
            bmv = st.bmv; bcost = st.bcost; bDistance = st.bDistance;
            if (bDistance == 1)
            {
                if (st.bPointNr) TWO_POINTS(st.bPointNr);
                break;
            }
        }
    }
#undef COST_MV
#undef TWO_POINTS
    *bmvIO = bmv; *bcostIO = bcost;
}

/* motionEstimate for a lowres reference: no candidates, subme 1 (slicetype.cpp:4483-4486 setSourcePU); the hexagon search, or -- an --hme level whose
 * hmeSearchMethod says so (motion.cpp:1013) -- the uneven multi-hexagon search in front of it */
static int lowres_me(la_t* m, mv_t mvmin, mv_t mvmax, mv_t qmvp, int merange, int method, mv_t* out)
{
    const int umh = method == XO_ME_UMH;
    m->mvp = qmvp;
    const mv_t qmin = { mvmin.x * 4, mvmin.y * 4 }, qmax = { mvmax.x * 4, mvmax.y * 4 };
    mv_t pmv = { qmvp.x < qmin.x ? qmin.x : qmvp.x > qmax.x ? qmax.x : qmvp.x, qmvp.y < qmin.y ? qmin.y : qmvp.y > qmax.y ? qmax.y : qmvp.y };
    const mv_t bestpre = pmv;
    int bprecost = qpel_cost(m, pmv.x, pmv.y, 0);                                             /* motion.cpp:967-968 */
    mv_t bmv = { (pmv.x + 2) >> 2, (pmv.y + 2) >> 2 };
    int bcost = bprecost, costs[3];
    if ((pmv.x | pmv.y) & 3) bcost = sad_at(m, bmv.x, bmv.y) + mvcost(m, bmv.x * 4, bmv.y * 4);
    if (pmv.x | pmv.y)
    {
        const int cost = sad_at(m, 0, 0) + mvcost(m, 0, 0);
        if (cost < bcost) { bcost = cost; bmv.x = 0; int t = 0 < mvmax.y ? 0 : mvmax.y; bmv.y = t > mvmin.y ? t : mvmin.y; }
    }
    if (bcost == 0) { out->x = bmv.x * 4; out->y = bmv.y * 4; return mvcost(m, out->x, out->y); }
    if (method == XO_ME_DIA)
    {   /* diamond search, radius 1 (motion.cpp:1016-1039): the four neighbours costed together (sad_x4: whether or not their row is inside the window), the row test only
           decides which may win; the step is decoded from the low bits of the packed cost */
        static const mv_t d4[4] = { {0,-1}, {0,1}, {-1,0}, {1,0} };
        int i = merange;
        bcost <<= 4;
        do
        {
            int c4[4];
            for (int k = 0; k < 4; k++) c4[k] = sad_at(m, bmv.x + d4[k].x, bmv.y + d4[k].y) + mvcost(m, (bmv.x + d4[k].x) * 4, (bmv.y + d4[k].y) * 4);
            if ((bmv.y - 1 >= mvmin.y) & (bmv.y - 1 <= mvmax.y)) { if ((c4[0] << 4) + 1 < bcost) bcost = (c4[0] << 4) + 1; }
            if ((bmv.y + 1 >= mvmin.y) & (bmv.y + 1 <= mvmax.y)) { if ((c4[1] << 4) + 3 < bcost) bcost = (c4[1] << 4) + 3; }
            if ((c4[2] << 4) + 4 < bcost) bcost = (c4[2] << 4) + 4;
            if ((c4[3] << 4) + 12 < bcost) bcost = (c4[3] << 4) + 12;
            if (!(bcost & 15)) break;
            bmv.x -= (int32_t)((uint32_t)bcost << 28) >> 30;
            bmv.y -= (int32_t)((uint32_t)bcost << 30) >> 30;
            bcost &= ~15;
        }
        while (--i && bmv.x >= mvmin.x && bmv.x <= mvmax.x && bmv.y >= mvmin.y && bmv.y <= mvmax.y);
        bcost >>= 4;
        goto refine;
    }
    if (method == XO_ME_FULL)
    {   /* exhaustive search (motion.cpp:1593-1632).  For an --hme reference (ReferencePlanes::isHMELowres, set for every Lowres of an --hme encode: lowres.cpp:96) the
           window is cut to +-merange around the ZERO vector, not around the predictor (:1598-1605); rows of four placements at a time, then the rest one by one: the order
           of the strict `<` is plain raster either way */
        const int r = merange < 0 ? -merange : merange;
        const int y0 = mvmin.y > -r ? mvmin.y : -r, x0 = mvmin.x > -r ? mvmin.x : -r, y1 = mvmax.y < r ? mvmax.y : r, x1 = mvmax.x < r ? mvmax.x : r;
        for (int ty = y0; ty <= y1; ty++)
            for (int tx = x0; tx <= x1; tx++)
            {
                const int cost = sad_at(m, tx, ty) + mvcost(m, tx * 4, ty * 4);
                if (cost < bcost) { bcost = cost; bmv.x = tx; bmv.y = ty; }
            }
        goto refine;
    }
    if (method == XO_ME_STAR) { lowres_star(m, mvmin, mvmax, merange, &bmv, &bcost); goto refine; }
    {
        const mv_t fpmv = { (pmv.x + 2) >> 2, (pmv.y + 2) >> 2 };                              /* motion.cpp:1005 */
        if (umh && !lowres_umh(m, mvmin, mvmax, fpmv, merange, &bmv, &bcost)) goto refine;
    }
    /* hexagon search, motion.cpp:1041-1140 */
#define INY(v) (((v) >= mvmin.y) & ((v) <= mvmax.y))
#define X3(a, b, c) do { const mv_t d_[3] = { a, b, c }; for (int k_ = 0; k_ < 3; k_++) \
        costs[k_] = sad_at(m, bmv.x + d_[k_].x, bmv.y + d_[k_].y) + mvcost(m, (bmv.x + d_[k_].x) * 4, (bmv.y + d_[k_].y) * 4); } while (0)
#define LT(x, y) do { if ((y) < (x)) (x) = (y); } while (0)
    { const mv_t a = {-2,0}, b = {-1,2}, c = {1,2}; X3(a, b, c); }
    bcost <<= 3;
    if (INY(bmv.y)) LT(bcost, (costs[0] << 3) + 2);
    if (INY(bmv.y + 2)) { LT(bcost, (costs[1] << 3) + 3); LT(bcost, (costs[2] << 3) + 4); }
    { const mv_t a = {2,0}, b = {1,-2}, c = {-1,-2}; X3(a, b, c); }
    if (INY(bmv.y)) LT(bcost, (costs[0] << 3) + 5);
    if (INY(bmv.y - 2)) { LT(bcost, (costs[1] << 3) + 6); LT(bcost, (costs[2] << 3) + 7); }
    if (bcost & 7)
    {
        int dir = (bcost & 7) - 2;
        if (INY(bmv.y + hex2[dir + 1].y))
        {
            bmv.x += hex2[dir + 1].x; bmv.y += hex2[dir + 1].y;
            for (int i = (merange >> 1) - 1; i > 0 && bmv.x >= mvmin.x && bmv.x <= mvmax.x && INY(bmv.y); i--)
            {
                X3(hex2[dir + 0], hex2[dir + 1], hex2[dir + 2]);
                bcost &= ~7;
                if (INY(bmv.y + hex2[dir + 0].y)) LT(bcost, (costs[0] << 3) + 1);
                if (INY(bmv.y + hex2[dir + 1].y)) LT(bcost, (costs[1] << 3) + 2);
                if (INY(bmv.y + hex2[dir + 2].y)) LT(bcost, (costs[2] << 3) + 3);
                if (!(bcost & 7)) break;
                dir += (bcost & 7) - 2;
                dir = mod6m1[dir + 1];
                bmv.x += hex2[dir + 1].x; bmv.y += hex2[dir + 1].y;
            }
        }
    }
    bcost >>= 3;
    {
        int dir = 0, c8[8];
        for (int k = 0; k < 8; k++) c8[k] = sad_at(m, bmv.x + square1[k + 1].x, bmv.y + square1[k + 1].y) + mvcost(m, (bmv.x + square1[k + 1].x) * 4, (bmv.y + square1[k + 1].y) * 4);
        const int upOk = INY(bmv.y - 1), dnOk = INY(bmv.y + 1);
        if (upOk && c8[0] < bcost) { bcost = c8[0]; dir = 1; }
        if (dnOk && c8[1] < bcost) { bcost = c8[1]; dir = 2; }
        if (c8[2] < bcost) { bcost = c8[2]; dir = 3; }
        if (c8[3] < bcost) { bcost = c8[3]; dir = 4; }
        if (upOk && c8[4] < bcost) { bcost = c8[4]; dir = 5; }
        if (dnOk && c8[5] < bcost) { bcost = c8[5]; dir = 6; }
        if (upOk && c8[6] < bcost) { bcost = c8[6]; dir = 7; }
        if (dnOk && c8[7] < bcost) { bcost = c8[7]; dir = 8; }
        bmv.x += square1[dir].x; bmv.y += square1[dir].y;
    }
#undef X3
#undef LT
refine:
    /* motion.cpp:1644-1699 */
    if (bprecost < bcost) { bmv = bestpre; bcost = bprecost; }
    else { bmv.x *= 4; bmv.y *= 4; }
    if (!bcost)
        bcost = mvcost(m, bmv.x, bmv.y);
    else
    {   /* the lowres branch: 4 half-pel directions at SAD, re-measure at SATD, 4 quarter-pel directions at SATD (workload[1]) */
        int bdir = 0;
        for (int i = 1; i <= 4; i++)
        {
            const int qx = bmv.x + square1[i].x * 2, qy = bmv.y + square1[i].y * 2;
            if ((qy < qmin.y) | (qy > qmax.y)) continue;
            const int cost = qpel_cost(m, qx, qy, 0) + mvcost(m, qx, qy);
            if (cost < bcost) { bcost = cost; bdir = i; }
        }
        bmv.x += square1[bdir].x * 2; bmv.y += square1[bdir].y * 2;
        bcost = qpel_cost(m, bmv.x, bmv.y, 1) + mvcost(m, bmv.x, bmv.y);
        bdir = 0;
        for (int i = 1; i <= 4; i++)
        {
            const int qx = bmv.x + square1[i].x, qy = bmv.y + square1[i].y;
            if ((qy < qmin.y) | (qy > qmax.y)) continue;
            const int cost = qpel_cost(m, qx, qy, 1) + mvcost(m, qx, qy);
            if (cost < bcost) { bcost = cost; bdir = i; }
        }
        bmv.x += square1[bdir].x; bmv.y += square1[bdir].y;
    }
    if (bmv.x | bmv.y)
    {   /* motion.cpp:1763-1768: subpelCompare at MV 0 = SATD against the full-pel plane */
        const int cost = (m->zero ? xo_satd(CU, CU, m->fenc, CU, m->zero, m->zeroStride) : xo_satd(CU, CU, m->fenc, CU, m->plane[0], m->stride)) + mvcost(m, 0, 0);
        if (cost <= bcost) { bmv.x = 0; bmv.y = 0; }
    }
#undef INY
    *out = bmv;
    return bcost;
}

/* the motion search of one list of one block (slicetype.cpp:4504-4573): reverse-order MV prediction, the candidate with the lowest SATD as MVP, motionEstimate, the
 * zero-MV skip rule of B estimates.  fencMV / fencCost address the block inside the list's arrays (widthInCU entries per row); extra = the fifth candidate or NULL. */
static void list_search(la_t* me, int cuX, int wcu, int lastRow, int bBidir, mv_t mvmin, mv_t mvmax, const mv_t* extra, int merange, int method,
                        int32_t* fencMV, int32_t* fencCost)
{
    mv_t mvc[5], mvp = { 0, 0 }; int numc = 0, skipCost = INT_MAX;
#define MVC(o) do { mvc[numc].x = fencMV[2 * (o)]; mvc[numc].y = fencMV[2 * (o) + 1]; numc++; } while (0)
    if (cuX < wcu - 1) MVC(1);
    if (!lastRow)
    {
        MVC(wcu);
        if (cuX > 0) MVC(wcu - 1);
        if (cuX < wcu - 1) MVC(wcu + 1);
    }
#undef MVC
    if (extra) mvc[numc++] = *extra;
    if (numc)
    {   /* :4541-4557 */
        int mvpcost = COST_MAX;
        for (int idx = 0; idx < numc; idx++)
        {
            const int cost = qpel_cost(me, mvc[idx].x, mvc[idx].y, 1);
            if (cost < mvpcost) { mvpcost = cost; mvp = mvc[idx]; }
            if (!(mvp.x | mvp.y) && bBidir) skipCost = cost;
        }
    }
    mv_t out;
    *fencCost = lowres_me(me, mvmin, mvmax, mvp, merange, method, &out);
    fencMV[0] = out.x; fencMV[1] = out.y;
    if (skipCost < 64 && skipCost < *fencCost && bBidir) { *fencCost = skipCost; fencMV[0] = 0; fencMV[1] = 0; }
}

/* slicetype.cpp:4365-4463 (serial branch) + :4467-4640 */
void xo_lowres_frame_cost_hme(const xo_pixel* fencPlane0, const xo_pixel* const* ref0, const xo_pixel* const* ref1, const xo_pixel* const* ref0w, intptr_t stride,
                              int wcu, int hcu, const int32_t* intraCost, const int32_t* invQscale, const uint16_t* costRowCentre,
                              int doSearch0, int doSearch1, int rowsPerSlice, int32_t* mvs0, int32_t* mvCosts0, int32_t* mvs1, int32_t* mvCosts1,
                              int32_t* lowresCosts, int32_t* rowSatds, int64_t* sums, const xo_la_hme* hme)
{
    const int bBidir = ref1 != NULL;
    const int doSearch[2] = { doSearch0, doSearch1 };
    int32_t* const mvs[2] = { mvs0, mvs1 };
    int32_t* const mvCosts[2] = { mvCosts0, mvCosts1 };
    /* list 0 is SEARCHED in the weighted copy of p0 when weightsAnalyse made one (wfref0, :4473; never on the quarter-resolution level); the bidirectional average uses p0 itself */
    const xo_pixel* const* const refs[2] = { ref0w ? ref0w : ref0, ref1 };
    int64_t costEst = 0, costEstAq = 0, intraMbs = 0;
    const int lowresPenalty = 4, merange = hme ? hme->range[1] : 16;                             /* slicetype.h:337 s_merange; :4560 */
    if (hme)
    {   /* :4430-4439, estimateCUCost(..., hme = 1): the searches only, on the quarter-resolution pictures, into Lowres::lowerResMvs / lowerResMvCosts */
        const xo_pixel* const* const refs4[2] = { hme->ref0, hme->ref1 };
        for (int cuY = hme->hcu - 1; cuY >= 0; cuY--)
            for (int cuX = hme->wcu - 1; cuX >= 0; cuX--)
            {
                const int cuXY = cuX + cuY * hme->wcu;
                const intptr_t pel = CU * cuX + (intptr_t)CU * cuY * hme->stride;
                la_t me;
                me.stride = hme->stride; me.cost = costRowCentre;
                for (int y = 0; y < CU; y++) memcpy(me.fenc + CU * y, hme->fenc + pel + y * hme->stride, CU * sizeof(xo_pixel));
                const mv_t mvmin = { -cuX * CU - 8, -cuY * CU - 8 }, mvmax = { (hme->wcu - cuX - 1) * CU + 8, (hme->hcu - cuY - 1) * CU + 8 };
                for (int i = 0; i < 1 + bBidir; i++)
                {
                    if (!doSearch[i]) continue;
                    for (int k = 0; k < 4; k++) me.plane[k] = refs4[i][k] + pel;
                    me.zero = (i ? ref1 : ref0)[0] + pel; me.zeroStride = stride;
                    list_search(&me, cuX, hme->wcu, cuY == hme->hcu - 1, bBidir, mvmin, mvmax, NULL, hme->range[0], hme->method[0],
                                &hme->mvs[i][2 * cuXY], &hme->mvCosts[i][cuXY]);
                }
            }
    }
    /* cooperative slices (--lookahead-slices, :1173-1176, 4347-4357): slice i covers rowsPerSlice block rows (the last one the remainder too) and
       starts its reverse sweep with lastRow = true, i.e. no predictor crosses its lower edge; the frame totals are the sums over the slices */
    const int rps = rowsPerSlice > 0 ? rowsPerSlice : hcu, nslices = hcu / rps;
    for (int cuY = hcu - 1; cuY >= 0; cuY--)
    {
        const int sl = cuY / rps < nslices - 1 ? cuY / rps : nslices - 1;
        const int lastRow = cuY == (sl == nslices - 1 ? hcu - 1 : rps * (sl + 1) - 1);
        rowSatds[cuY] = 0;
        for (int cuX = wcu - 1; cuX >= 0; cuX--)
        {
            const int cuXY = cuX + cuY * wcu;
            const int cuXY_4x4 = (cuX / 2) + (cuY / 2) * wcu / 2;                                /* :4479, as written there (not a position on the quarter-resolution grid) */
            const intptr_t pel = CU * cuX + (intptr_t)CU * cuY * stride;
            la_t me;
            me.stride = stride; me.cost = costRowCentre; me.zero = NULL; me.zeroStride = 0;
            for (int y = 0; y < CU; y++) memcpy(me.fenc + CU * y, fencPlane0 + pel + y * stride, CU * sizeof(xo_pixel));
            const mv_t mvmin = { -cuX * CU - 8, -cuY * CU - 8 }, mvmax = { (wcu - cuX - 1) * CU + 8, (hcu - cuY - 1) * CU + 8 };
            int bcost = COST_MAX, listused = 0;
            for (int i = 0; i < 1 + bBidir; i++)
            {
                int32_t* fencCost = &mvCosts[i][cuXY];
                int32_t* fencMV = &mvs[i][2 * cuXY];
                if (!doSearch[i])
                {
                    if (*fencCost < bcost) { bcost = *fencCost; listused = i + 1; }
                    continue;
                }
                for (int k = 0; k < 4; k++) me.plane[k] = refs[i][k] + pel;
                mv_t extra; int haveExtra = 0;
                if (hme && hme->mvCosts[i][cuXY_4x4] > 0)                                        /* :4532-4535: twice the quarter-resolution MV */
                { extra.x = hme->mvs[i][2 * cuXY_4x4] * 2; extra.y = hme->mvs[i][2 * cuXY_4x4 + 1] * 2; haveExtra = 1; }
                list_search(&me, cuX, wcu, lastRow, bBidir, mvmin, mvmax, haveExtra ? &extra : NULL, merange, hme ? hme->method[1] : XO_ME_HEX, fencMV, fencCost);
                if (*fencCost < bcost) { bcost = *fencCost; listused = i + 1; }
            }
            if (bBidir)
            {   /* :4577-4597 */
                xo_pixel b0[CU * CU], b1[CU * CU], avg[CU * CU]; intptr_t s0, s1;
                la_t r0 = me, r1 = me;
                for (int k = 0; k < 4; k++) { r0.plane[k] = ref0[k] + pel; r1.plane[k] = ref1[k] + pel; }
                const xo_pixel* src0 = lowres_mc(&r0, mvs0[2 * cuXY], mvs0[2 * cuXY + 1], b0, &s0);
                const xo_pixel* src1 = lowres_mc(&r1, mvs1[2 * cuXY], mvs1[2 * cuXY + 1], b1, &s1);
                xo_pixelavg_pp(CU, CU, avg, CU, src0, s0, src1, s1);
                int bicost = xo_satd(CU, CU, me.fenc, CU, avg, CU);
                if (bicost < bcost) { bcost = bicost; listused = 3; }
                xo_pixelavg_pp(CU, CU, avg, CU, ref0[0] + pel, stride, ref1[0] + pel, stride);
                bicost = xo_satd(CU, CU, me.fenc, CU, avg, CU);
                if (bicost < bcost) { bcost = bicost; listused = 3; }
                bcost += lowresPenalty;
            }
            else
            {
                bcost += lowresPenalty;
                if (intraCost[cuXY] < bcost) { bcost = intraCost[cuXY]; listused = 0; }
            }
            const int score = frame_score_cu(cuX, cuY, wcu, hcu);
            const int bcostAq = (score && invQscale) ? ((bcost * invQscale[cuXY] + 128) >> 8) : bcost;
            if (score)
            {
                costEst += bcost; costEstAq += bcostAq;
                if (!listused && !bBidir) intraMbs++;
            }
            rowSatds[cuY] += bcostAq;
            lowresCosts[cuXY] = (uint16_t)((bcost < LOWRES_COST_MASK ? bcost : LOWRES_COST_MASK) | (listused << LOWRES_COST_SHIFT));
        }
    }
    sums[0] = costEst; sums[1] = costEstAq; sums[2] = intraMbs;
}
void xo_lowres_frame_cost(const xo_pixel* fencPlane0, const xo_pixel* const* ref0, const xo_pixel* const* ref1, const xo_pixel* const* ref0w, intptr_t stride,
                          int wcu, int hcu, const int32_t* intraCost, const int32_t* invQscale, const uint16_t* costRowCentre,
                          int doSearch0, int doSearch1, int rowsPerSlice, int32_t* mvs0, int32_t* mvCosts0, int32_t* mvs1, int32_t* mvCosts1,
                          int32_t* lowresCosts, int32_t* rowSatds, int64_t* sums)
{
    xo_lowres_frame_cost_hme(fencPlane0, ref0, ref1, ref0w, stride, wcu, hcu, intraCost, invQscale, costRowCentre, doSearch0, doSearch1, rowsPerSlice, mvs0, mvCosts0, mvs1, mvCosts1,
                             lowresCosts, rowSatds, sums, NULL);
}

/* ---- cuTree cost propagation of one picture (slicetype.cpp:3850-3953 estimateCUPropagate; pixel.cpp:906-931 propagateCost) ----
 * fpsFactor = CLIP_DURATION(frame duration) / CLIP_DURATION(average duration) as the caller computes it (:3863); the per-block
 * arithmetic is the reference's double arithmetic, operation by operation.  prop0 / prop1 / propB are the pictures'
 * Lowres::propagateCost arrays (uint16, saturating adds). */
void xo_cu_propagate_cost(int32_t* dst, const uint16_t* propagateIn, const int32_t* intraCosts, const uint16_t* interCosts,
                          const int32_t* invQscales, double fpsFactor, int len)
{
    const double fps = fpsFactor / 256;
    for (int i = 0; i < len; i++)
    {
        const int intraCost = intraCosts[i];
        const int inter = interCosts[i] & LOWRES_COST_MASK;
        const int interCost = intraCost < inter ? intraCost : inter;
        const double propagateIntra = intraCost * invQscales[i];
        const double propagateAmount = (double)propagateIn[i] + propagateIntra * fps;
        const double propagateNum = (double)(intraCost - interCost);
        const double propagateDenom = (double)intraCost;
        dst[i] = (int)(propagateAmount * propagateNum / propagateDenom + 0.5);
    }
}

void xo_estimate_cu_propagate(int wcu, int hcu, int distP0 /* b - p0 */, int distP1 /* p1 - b */, int weightedBiPred, double fpsFactor, int referenced,
                              const int32_t* intraCost, const uint16_t* lowresCosts, const int32_t* invQscale,
                              const int32_t* mvs0, const int32_t* mvs1, uint16_t* propB, uint16_t* prop0, uint16_t* prop1)
{
    uint16_t* refCosts[2] = { prop0, prop1 };
    const int32_t* mvsL[2] = { mvs0, mvs1 };
    const int span = distP0 + distP1;
    const int distScaleFactor = ((distP0 << 8) + (span >> 1)) / span;
    const int bipredWeight = weightedBiPred ? 64 - (distScaleFactor >> 2) : 32;
    const int bipredWeights[2] = { bipredWeight, 64 - bipredWeight };
    int32_t* scratch = (int32_t*)calloc((size_t)wcu, sizeof(int32_t));
    const uint16_t* propagateIn = propB;
    if (!referenced) memset(propB, 0, wcu * sizeof(uint16_t));
    for (int blocky = 0; blocky < hcu; blocky++)
    {
        int cuIndex = blocky * wcu;
        xo_cu_propagate_cost(scratch, propagateIn, intraCost + cuIndex, lowresCosts + cuIndex, invQscale + cuIndex, fpsFactor, wcu);
        if (referenced) propagateIn += wcu;
        for (int blockx = 0; blockx < wcu; blockx++, cuIndex++)
        {
            const int amount = scratch[blockx];
            if (amount <= 0) continue;                                               /* intra blocks do not propagate */
            const int listsUsed = lowresCosts[cuIndex] >> LOWRES_COST_SHIFT;
            for (int list = 0; list < 2; list++)
            {
                if (!((listsUsed >> list) & 1)) continue;
#define CLIP_ADD(s, x) (s) = (uint16_t)((s) + (x) < (1 << 16) - 1 ? (s) + (x) : (1 << 16) - 1)
                int listamount = amount;
                if (listsUsed == 3) listamount = (listamount * bipredWeights[list] + 32) >> 6;
                int x = mvsL[list][2 * cuIndex], y = mvsL[list][2 * cuIndex + 1];
                uint16_t* rc = refCosts[list];
                if (!(x | y)) { CLIP_ADD(rc[cuIndex], listamount); continue; }
                const int cux = (x >> 5) + blockx, cuy = (y >> 5) + blocky;
                const int idx0 = cux + cuy * wcu, idx1 = idx0 + 1, idx2 = idx0 + wcu, idx3 = idx0 + wcu + 1;
                x &= 31; y &= 31;
                const int w0 = (32 - y) * (32 - x), w1 = (32 - y) * x, w2 = y * (32 - x), w3 = y * x;
                if (cux < wcu - 1 && cuy < hcu - 1 && cux >= 0 && cuy >= 0)
                {
                    CLIP_ADD(rc[idx0], (listamount * w0 + 512) >> 10); CLIP_ADD(rc[idx1], (listamount * w1 + 512) >> 10);
                    CLIP_ADD(rc[idx2], (listamount * w2 + 512) >> 10); CLIP_ADD(rc[idx3], (listamount * w3 + 512) >> 10);
                }
                else
                {   /* offsets checked individually: blocks outside the picture receive nothing */
                    if (cux < wcu && cuy < hcu && cux >= 0 && cuy >= 0) CLIP_ADD(rc[idx0], (listamount * w0 + 512) >> 10);
                    if (cux + 1 < wcu && cuy < hcu && cux + 1 >= 0 && cuy >= 0) CLIP_ADD(rc[idx1], (listamount * w1 + 512) >> 10);
                    if (cux < wcu && cuy + 1 < hcu && cux >= 0 && cuy + 1 >= 0) CLIP_ADD(rc[idx2], (listamount * w2 + 512) >> 10);
                    if (cux + 1 < wcu && cuy + 1 < hcu && cux + 1 >= 0 && cuy + 1 >= 0) CLIP_ADD(rc[idx3], (listamount * w3 + 512) >> 10);
                }
#undef CLIP_ADD
            }
        }
    }
    free(scratch);
}

/* Lookahead::cuTreeFinish (slicetype.cpp:4098-4150), the branch of the default configuration (no hevc-aq, qgSize != 8): per block of the half-resolution picture
 * the qp offset that follows from how much later pictures inherit from it.  fpsFactor = (int)(CLIP_DURATION(averageDuration) / CLIP_DURATION(frame duration) * 256);
 * weightdelta = 1 - weightedCostDelta[ref0Distance - 1] when that is > 0, else 0.  Blocks with no intra cost keep their previous qpCuTreeOffset. */
void xo_cutree_finish(int ncu, const int32_t* intraCost, const int32_t* invQscale, const uint16_t* propagateCost, const double* qpAqOffset, int fpsFactor,
                      double weightdelta, double strength, double* qpCuTreeOffset)
{
    for (int i = 0; i < ncu; i++)
    {
        const int intracost = (intraCost[i] * invQscale[i] + 128) >> 8;
        if (intracost)
        {
            const int propagate = (propagateCost[i] * fpsFactor + 128) >> 8;
            const double log2_ratio = log2((double)(intracost + propagate)) - log2((double)intracost) + weightdelta;
            qpCuTreeOffset[i] = qpAqOffset[i] - strength * log2_ratio;
        }
    }
}
