/*
 * x265_oracle_me.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's motion-search DRIVER for one PU and one reference:
 *   - BitCost::setQP / CalculateLogs  (encoder/bitcost.cpp:30-105): lambda-scaled MVD cost row
 *   - MotionEstimate::motionEstimate  (encoder/motion.cpp:923-1773): start-point selection,
 *     DIA / HEX / UMH / STAR / SEA / FULL integer search, sub-pel refinement per workload[subme], final zero-MV check
 *   - MotionEstimate::subpelCompare   (encoder/motion.cpp:1775-1803, luma part)
 * built on the primitive restatements of x265_oracle.c.  Pinned against the REAL reference driver
 * (oracle/_ref, op "me" / "mvcost_row") by tests/test_me_oracle_vs_ref.py.
 */
#include "x265_oracle_me.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int x, y; } mv_t;

/* constants.cpp:28-116: x265_lambda_tab[q] = pow(2, q/6 - 2) * (1 << (depth-8)), tabulated to 4 decimals */
double xo_lambda(int qp)
{
    double v = pow(2.0, (double)qp / 6.0 - 2.0) * (double)(1 << (X265_DEPTH - 8));
    return floor(v * 10000.0 + 0.5) / 10000.0;
}

/* bitcost.cpp:87-100 CalculateLogs (float storage, double log) + :51-56 the uint16 cost row */
void xo_mvcost_row(int qp, int halfRange, uint16_t* out)
{
    double lambda = xo_lambda(qp);
    float log2_2 = (float)(2.0f / log(2.0f));
    for (int i = 0; i <= halfRange; i++)
    {
        float bits = i ? (float)(log((double)(float)(i + 1)) * log2_2 + 1.718f) : 0.718f;
        double c = bits * lambda + 0.5f;
        if (c > (double)((1 << 15) - 1)) c = (double)((1 << 15) - 1);
        out[halfRange + i] = out[halfRange - i] = (uint16_t)c;
    }
}

typedef struct
{
    const xo_pixel* fref; intptr_t stride;       /* co-located block origin in the reference plane */
    xo_pixel fenc[64 * 64];                        /* PU cached at FENC_STRIDE (motion.cpp:223-229) */
    int w, h;
    const uint16_t* cost;                          /* centred cost row */
    mv_t mvp;
    /* chroma SATD terms of subpelCompare (motion.cpp:218-247, 1805-1865), 4:2:0: the PU's Cb / Cr source blocks and the co-located reference pixels */
    int chroma, cw, ch;
    xo_pixel fencC[2][32 * 32];                    /* cached at stride 32 */
    const xo_pixel* frefC[2]; intptr_t strideC;
} me_t;

static inline int mvcost(const me_t* m, int qx, int qy)
{   /* bitcost.h:57: uint16_t sum of the two table entries */
    return (uint16_t)(m->cost[qx - m->mvp.x] + m->cost[qy - m->mvp.y]);
}
static inline int sad_at(const me_t* m, int mx, int my)
{
    return xo_sad(m->w, m->h, m->fenc, 64, m->fref + mx + my * m->stride, m->stride);
}
/* motion.cpp:1775-1803 */
static int subpel_compare(const me_t* m, int qx, int qy, int useSatd)
{
    const xo_pixel* fref = m->fref + (qx >> 2) + (qy >> 2) * m->stride;
    int xf = qx & 3, yf = qy & 3;
    xo_pixel buf[64 * 64];
    const xo_pixel* p = fref; intptr_t ps = m->stride;
    if (xf | yf)
    {
        if (!yf) xo_interp_hpp(8, m->w, m->h, fref, m->stride, buf, m->w, xf);
        else if (!xf) xo_interp_vpp(8, m->w, m->h, fref, m->stride, buf, m->w, yf);
        else xo_interp_hvpp(8, m->w, m->h, fref, m->stride, buf, m->w, xf, yf);
        p = buf; ps = m->w;
    }
    int cost = useSatd ? xo_satd(m->w, m->h, m->fenc, 64, p, ps) : xo_sad(m->w, m->h, m->fenc, 64, p, ps);
    if (m->chroma)
    {   /* motion.cpp:1805-1865 with hshift = vshift = 1: the quarter-pel luma MV is the eighth-pel chroma MV; 4-tap filters, SATD whatever `cmp` is */
        const int mvx = qx, mvy = qy, cxf = mvx & 7, cyf = mvy & 7;
        const intptr_t off = (mvx >> 3) + (mvy >> 3) * m->strideC;
        for (int c = 0; c < 2; c++)
        {
            const xo_pixel* r = m->frefC[c] + off;
            xo_pixel cb[32 * 32];
            const xo_pixel* q = r; intptr_t qs = m->strideC;
            if (cxf | cyf)
            {
                if (!cyf) xo_interp_hpp(4, m->cw, m->ch, r, m->strideC, cb, m->cw, cxf);
                else if (!cxf) xo_interp_vpp(4, m->cw, m->ch, r, m->strideC, cb, m->cw, cyf);
                else
                {
                    int16_t immed[32 * (32 + 3)];
                    xo_interp_hps(4, m->cw, m->ch, r, m->strideC, immed, m->cw, cxf, 1);
                    xo_interp_vsp(4, m->cw, m->ch, immed + m->cw, m->cw, cb, m->cw, cyf);
                }
                q = cb; qs = m->cw;
            }
            cost += xo_satd(m->cw, m->ch, m->fencC[c], 32, q, qs);
        }
    }
    return cost;
}

static const mv_t hex2[8] = { {-1,-2}, {-2,0}, {-1,2}, {1,2}, {2,0}, {1,-2}, {-1,-2}, {-2,0} };
static const uint8_t mod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };
static const mv_t square1[9] = { {0,0}, {0,-1}, {0,1}, {-1,0}, {1,0}, {-1,-1}, {-1,1}, {1,-1}, {1,1} };
static const mv_t offsets2[16] = { {-1,0}, {0,-1}, {-1,-1}, {1,-1}, {-1,0}, {1,0}, {-1,1}, {-1,-1},
                                   {1,-1}, {1,1}, {-1,0}, {0,1}, {-1,1}, {1,1}, {1,0}, {0,1} };
/* motion.cpp:48-58 workload[]: hpel_iters, hpel_dirs, qpel_iters, qpel_dirs, hpel_satd */
static const int workload[8][5] = { {1,4,0,4,0}, {1,4,1,4,0}, {1,4,1,4,1}, {2,4,1,4,1}, {2,4,2,4,1}, {1,8,1,8,1}, {2,8,1,8,1}, {2,8,2,8,1} };

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int in_range(mv_t v, mv_t mn, mv_t mx) { return v.x >= mn.x && v.x <= mx.x && v.y >= mn.y && v.y <= mx.y; }

typedef struct { mv_t bmv; int bcost, bPointNr, bDistance; } star_t;

#define PT_DIST(mx_, my_, point, dist) do { int c_ = sad_at(m, (mx_), (my_)) + mvcost(m, (mx_) * 4, (my_) * 4); \
    if (c_ < s->bcost) { s->bcost = c_; s->bmv.x = (mx_); s->bmv.y = (my_); s->bPointNr = (point); s->bDistance = (dist); } } while (0)

/* motion.cpp:387-629 */
static void star_pattern(const me_t* m, mv_t mvmin, mv_t mvmax, star_t* s, int earlyExitIters, int merange)
{
    mv_t omv = s->bmv;
    int saved = s->bcost, rounds = 0;
    {
        int dist = 1;
        int top = omv.y - dist, bottom = omv.y + dist, left = omv.x - dist, right = omv.x + dist;
        /* the x4 form and the guarded scalar form visit the same points in the same order */
        if (top >= mvmin.y) PT_DIST(omv.x, top, 2, dist);
        if (left >= mvmin.x) PT_DIST(left, omv.y, 4, dist);
        if (right <= mvmax.x) PT_DIST(right, omv.y, 5, dist);
        if (bottom <= mvmax.y) PT_DIST(omv.x, bottom, 7, dist);
        if (s->bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
    for (int dist = 2; dist <= 8; dist <<= 1)
    {
        int top = omv.y - dist, bottom = omv.y + dist, left = omv.x - dist, right = omv.x + dist;
        int top2 = omv.y - (dist >> 1), bottom2 = omv.y + (dist >> 1), left2 = omv.x - (dist >> 1), right2 = omv.x + (dist >> 1);
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
        int top = omv.y - dist, bottom = omv.y + dist, left = omv.x - dist, right = omv.x + dist;
        saved = s->bcost;
        int all = top >= mvmin.y && left >= mvmin.x && right <= mvmax.x && bottom <= mvmax.y;
        if (all || top >= mvmin.y) PT_DIST(omv.x, top, 0, dist);
        if (all || left >= mvmin.x) PT_DIST(left, omv.y, 0, dist);
        if (all || right <= mvmax.x) PT_DIST(right, omv.y, 0, dist);
        if (all || bottom <= mvmax.y) PT_DIST(omv.x, bottom, 0, dist);
        for (int index = 1; index < 4; index++)
        {
            int posYT = top + ((dist >> 2) * index), posYB = bottom - ((dist >> 2) * index);
            int posXL = omv.x - ((dist >> 2) * index), posXR = omv.x + ((dist >> 2) * index);
            if (all)
            {
                PT_DIST(posXL, posYT, 0, dist); PT_DIST(posXR, posYT, 0, dist); PT_DIST(posXL, posYB, 0, dist); PT_DIST(posXR, posYB, 0, dist);
            }
            else
            {
                if (posYT >= mvmin.y)
                {
                    if (posXL >= mvmin.x) PT_DIST(posXL, posYT, 0, dist);
                    if (posXR <= mvmax.x) PT_DIST(posXR, posYT, 0, dist);
                }
                if (posYB <= mvmax.y)
                {
                    if (posXL >= mvmin.x) PT_DIST(posXL, posYB, 0, dist);
                    if (posXR <= mvmax.x) PT_DIST(posXR, posYB, 0, dist);
                }
            }
        }
        if (s->bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
}

#define COST_MV(mx_, my_) do { int c_ = sad_at(m, (mx_), (my_)) + mvcost(m, (mx_) * 4, (my_) * 4); \
    if (c_ < bcost) { bcost = c_; bmv.x = (mx_); bmv.y = (my_); } } while (0)

/* ---- CUData::getPMV (common/cudata.cpp:1806-1990, the build without multiview / SCC): the two AMVP candidates and the motion-candidate list of a PU ----
 * nb = 6 neighbours in MVP_DIR order (cudata.h:67-75: LEFT, ABOVE, ABOVE_RIGHT, BELOW_LEFT, ABOVE_LEFT, COLLOCATED) x { mv[0].x, .y, mv[1].x, .y, refIdx[0], refIdx[1],
 * cuAddr[0], cuAddr[1], isAvailable }; refPOC = m_refPOCList[2][16]; colPOC / colRefPOC = the POCs :1962-1967 looks up for the temporal candidate.
 * amvp = { c0.x, c0.y, c1.x, c1.y }; returns numMvc, mvc = x, y pairs. */
static void scale_mv_poc(const int32_t* in, int curPOC, int curRefPOC, int colPOC, int colRefPOC, int32_t* out)
{   /* scaleMvByPOCDist (cudata.cpp:2229-2244), scaleMv (:104-110) */
    const int diffPocD = colPOC - colRefPOC, diffPocB = curPOC - curRefPOC;
    if (diffPocD == diffPocB) { out[0] = in[0]; out[1] = in[1]; return; }
    const int tdb = diffPocB < -128 ? -128 : diffPocB > 127 ? 127 : diffPocB, tdd = diffPocD < -128 ? -128 : diffPocD > 127 ? 127 : diffPocD;
    const int x = (0x4000 + abs(tdd / 2)) / tdd;
    int scale = (tdb * x + 32) >> 6;
    scale = scale < -4096 ? -4096 : scale > 4095 ? 4095 : scale;
    for (int k = 0; k < 2; k++)
    {
        int v = (scale * in[k] + 127 + (scale * in[k] < 0)) >> 8;
        out[k] = v < -32768 ? -32768 : v > 32767 ? 32767 : v;
    }
}
int xo_get_pmv(const int32_t* nb, int list, int refIdx, int curPOC, int temporalEnabled, const int32_t* refPOC, int colPOC, int colRefPOC, int32_t* amvp, int32_t* mvc)
{
    enum { LEFT, ABOVE, ABOVE_RIGHT, BELOW_LEFT, ABOVE_LEFT, COLLOCATED };
    int32_t direct[5][2], indirect[5][2];
    int validD[5], validI[5];
    const int curRefPOC = refPOC[list * 16 + refIdx];
    for (int d = 0; d < 5; d++)
    {
        const int32_t* n = nb + 9 * d;
        validD[d] = validI[d] = 0;
        /* getDirectPMV (:2110-2123): same reference picture, either list, the asked-for list first */
        for (int i = 0, l = list; i < 2; i++, l = !l)
        {
            const int r = n[4 + l];
            if (r >= 0 && curRefPOC == refPOC[l * 16 + r]) { direct[d][0] = n[2 * l]; direct[d][1] = n[2 * l + 1]; validD[d] = 1; break; }
        }
        /* getIndirectPMV (:2126-2156): any reference, scaled by the POC distances (the neighbour belongs to the current picture) */
        for (int i = 0, l = list; i < 2; i++, l = !l)
        {
            const int r = n[4 + l];
            if (r >= 0) { scale_mv_poc(n + 2 * l, curPOC, curRefPOC, curPOC, refPOC[l * 16 + r], indirect[d]); validI[d] = 1; break; }
        }
    }
    int num = 0;
#define PUT(v) do { amvp[2 * num] = (v)[0]; amvp[2 * num + 1] = (v)[1]; num++; } while (0)
    if (validD[BELOW_LEFT]) PUT(direct[BELOW_LEFT]);
    else if (validD[LEFT]) PUT(direct[LEFT]);
    else if (validI[BELOW_LEFT]) PUT(indirect[BELOW_LEFT]);
    else if (validI[LEFT]) PUT(indirect[LEFT]);
    const int bAddedSmvp = num > 0;
    if (validD[ABOVE_RIGHT]) PUT(direct[ABOVE_RIGHT]);
    else if (validD[ABOVE]) PUT(direct[ABOVE]);
    else if (validD[ABOVE_LEFT]) PUT(direct[ABOVE_LEFT]);
    if (!bAddedSmvp)
    {
        if (validI[ABOVE_RIGHT]) PUT(indirect[ABOVE_RIGHT]);
        else if (validI[ABOVE]) PUT(indirect[ABOVE]);
        else if (validI[ABOVE_LEFT]) PUT(indirect[ABOVE_LEFT]);
    }
    /* (at most two entries so far: left group <= 1; above group: a direct one, and an indirect one only when the left group gave none) */
    int numMvc = 0;
    for (int d = LEFT; d <= ABOVE_LEFT; d++)
    {
        if (validD[d] && (direct[d][0] | direct[d][1])) { mvc[2 * numMvc] = direct[d][0]; mvc[2 * numMvc + 1] = direct[d][1]; numMvc++; }
        if (validI[d] && (indirect[d][0] | indirect[d][1])) { mvc[2 * numMvc] = indirect[d][0]; mvc[2 * numMvc + 1] = indirect[d][1]; numMvc++; }
    }
    if (num == 2) num -= (amvp[0] == amvp[2] && amvp[1] == amvp[3]);
    if (temporalEnabled && num < 2)
    {
        const int32_t* n = nb + 9 * COLLOCATED;
        const int tempRefIdx = n[4 + list];
        if (tempRefIdx != -1)
        {
            int32_t t[2];
            scale_mv_poc(n + 2 * list, curPOC, curRefPOC, colPOC, colRefPOC, t);
            mvc[2 * numMvc] = t[0]; mvc[2 * numMvc + 1] = t[1]; numMvc++;
            PUT(t);
        }
    }
#undef PUT
    while (num < 2) { amvp[2 * num] = 0; amvp[2 * num + 1] = 0; num++; }
    return numMvc;
}

/* MotionEstimate::diamondSearch (motion.cpp:631-773): the full-pel predictor search of ThreadedME's first stage (search.cpp:355-363, range 32, MVP (0,0)).
 * It starts from bmv = (0,0) with bcost = INT_MAX WITHOUT costing (0,0); a first loop of distances 1, 2, 4 always around omv = (0,0) (omv is
 * not moved inside it), then distances 8 .. 64 around the best so far, each ending when the best did not move.  Away from the window's edge the
 * points go through COST_MV_X4 (:307-328), whose arguments are OFFSETS from omv -- but diamondSearch passes absolute coordinates (omv.x, top ...),
 * so in the second loop, where omv != 0, the measured positions are omv + (absolute coordinate): the restatement keeps that, and the macro's
 * vertical range test `omv.y + my in [mvmin.y, mvmax.y]` with no horizontal one.  Returns bcost, outMv = full-pel MV. */
int xo_diamond_search(const xo_pixel* fencPlane, intptr_t fencStride, int w, int h, const xo_pixel* fref, intptr_t refStride, const int32_t* bounds,
                      int qmvpx, int qmvpy, const uint16_t* costRowCentre, int32_t* outMv)
{
    me_t mm; me_t* m = &mm;
    m->fref = fref; m->stride = refStride; m->w = w; m->h = h; m->cost = costRowCentre; m->mvp.x = qmvpx; m->mvp.y = qmvpy; m->chroma = 0;
    for (int y = 0; y < h; y++) memcpy(m->fenc + y * 64, fencPlane + y * fencStride, w * sizeof(xo_pixel));
    const mv_t mvmin = { bounds[0], bounds[1] }, mvmax = { bounds[2], bounds[3] };
    int bcost = INT_MAX;
    mv_t bmv = { 0, 0 }, omv = bmv;
#define DX4(a0, a1, b0, b1, c0, c1, d0, d1) do { const int o_[4][2] = { { a0, a1 }, { b0, b1 }, { c0, c1 }, { d0, d1 } }; int c_[4]; \
        for (int k_ = 0; k_ < 4; k_++) c_[k_] = sad_at(m, omv.x + o_[k_][0], omv.y + o_[k_][1]) + mvcost(m, (omv.x + o_[k_][0]) * 4, (omv.y + o_[k_][1]) * 4); \
        for (int k_ = 0; k_ < 4; k_++) if (omv.y + o_[k_][1] >= mvmin.y && omv.y + o_[k_][1] <= mvmax.y && c_[k_] < bcost) \
            { bcost = c_[k_]; bmv.x = omv.x + o_[k_][0]; bmv.y = omv.y + o_[k_][1]; } } while (0)
    for (int dist = 1; dist <= 4; dist <<= 1)
    {
        const mv_t bmv0 = bmv;
        const int top = omv.y - dist, bottom = omv.y + dist, left = omv.x - dist, right = omv.x + dist;
        const int top2 = omv.y - (dist >> 1), bottom2 = omv.y + (dist >> 1), left2 = omv.x - (dist >> 1), right2 = omv.x + (dist >> 1);
        if (top >= mvmin.y && left >= mvmin.x && right <= mvmax.x && bottom <= mvmax.y)
        {
            DX4(omv.x, top, omv.x, bottom, left, omv.y, right, omv.y);
            DX4(left2, top2, right2, top2, left2, bottom2, right2, bottom2);
        }
        else
        {
            if (top >= mvmin.y) COST_MV(omv.x, top);
            if (top2 >= mvmin.y)
            {
                if (left2 >= mvmin.x) COST_MV(left2, top2);
                if (right2 <= mvmax.x) COST_MV(right2, top2);
            }
            if (left >= mvmin.x) COST_MV(left, omv.y);
            if (right <= mvmax.x) COST_MV(right, omv.y);
            if (bottom2 <= mvmax.y)
            {
                if (left2 >= mvmin.x) COST_MV(left2, bottom2);
                if (right2 <= mvmax.x) COST_MV(right2, bottom2);
            }
            if (bottom <= mvmax.y) COST_MV(omv.x, bottom);
        }
        if (bmv.x == bmv0.x && bmv.y == bmv0.y) break;
    }
    omv = bmv;
    for (int dist = 8; dist <= 64; dist += 8)
    {
        const mv_t bmv0 = bmv;
        const int top = omv.y - dist, bottom = omv.y + dist, left = omv.x - dist, right = omv.x + dist;
        if (top >= mvmin.y && left >= mvmin.x && right <= mvmax.x && bottom <= mvmax.y)
        {
            DX4(omv.x, top, left, omv.y, right, omv.y, omv.x, bottom);
            for (int index = 1; index < 4; index++)
            {
                const int posYT = top + ((dist >> 2) * index), posYB = bottom - ((dist >> 2) * index);
                const int posXL = omv.x - ((dist >> 2) * index), posXR = omv.x + ((dist >> 2) * index);
                DX4(posXL, posYT, posXR, posYT, posXL, posYB, posXR, posYB);
            }
        }
        else
        {
            if (top >= mvmin.y) COST_MV(omv.x, top);
            if (left >= mvmin.x) COST_MV(left, omv.y);
            if (right <= mvmax.x) COST_MV(right, omv.y);
            if (bottom <= mvmax.y) COST_MV(omv.x, bottom);
            for (int index = 1; index < 4; index++)
            {
                const int posYT = top + ((dist >> 2) * index), posYB = bottom - ((dist >> 2) * index);
                const int posXL = omv.x - ((dist >> 2) * index), posXR = omv.x + ((dist >> 2) * index);
                if (posYT >= mvmin.y)
                {
                    if (posXL >= mvmin.x) COST_MV(posXL, posYT);
                    if (posXR <= mvmax.x) COST_MV(posXR, posYT);
                }
                if (posYB <= mvmax.y)
                {
                    if (posXL >= mvmin.x) COST_MV(posXL, posYB);
                    if (posXR <= mvmax.x) COST_MV(posXR, posYB);
                }
            }
        }
        if (bmv.x == bmv0.x && bmv.y == bmv0.y) break;
        omv = bmv;
    }
#undef DX4
    outMv[0] = bmv.x; outMv[1] = bmv.y;
    return bcost;
}

/* encoder/framefilter.cpp:38-139 (integral_init{4,8,12,16,24,32}{h,v}_c) driven the way FrameFilter::processPostRow drives them
 * (:740-833) over a whole padded picture: plane k holds, at every position whose box lies inside the padded picture, the sum of
 * the W x H pixels whose top-left corner is that position.  planes[k] and pic point at pixel (0,0); rows -padY .. maxHeight+padY-1. */
static const int k_seaW[12] = { 32, 32, 32, 24, 16, 16, 16, 12, 8, 8, 4, 4 }, k_seaH[12] = { 32, 24, 8, 32, 16, 12, 4, 16, 32, 8, 16, 4 };
void xo_sea_integral_planes(const xo_pixel* pic, intptr_t stride, int maxHeight, int padX, int padY, uint32_t* const* planes)
{
    for (int k = 0; k < 12; k++)
    {
        const int W = k_seaW[k], H = k_seaH[k];
        uint32_t* I = planes[k];
        memset(I - padY * stride - padX, 0, stride * sizeof(uint32_t));                              /* :768-769 */
        for (int y = -padY; y < maxHeight + padY - 1; y++)
        {
            const xo_pixel* pix = pic + y * stride - padX;
            uint32_t* sum = I + (y + 1) * stride - padX;
            int32_t v = 0;
            for (int i = 0; i < W; i++) v += pix[i];
            for (int x = 0; x < stride - W; x++) { sum[x] = v + sum[x - stride]; v += pix[x + W] - pix[x]; }   /* integral_initNh_c */
            if (y >= H - padY)
            {
                uint32_t* s2 = sum - H * stride;
                for (int x = 0; x < stride; x++) s2[x] = s2[x + H * stride] - s2[x];                /* integral_initNv_c */
            }
        }
    }
}

static int ads_parts(int w, int h)
{   /* which ads variant a PU's slot holds (pixel.cpp:1122-1146) */
    static const int x1[6][2] = { {4,4}, {8,8}, {16,12}, {12,16}, {16,4}, {4,16} };
    static const int x2[8][2] = { {8,4}, {4,8}, {16,8}, {8,16}, {32,16}, {16,32}, {64,32}, {32,64} };
    for (int i = 0; i < 6; i++) if (x1[i][0] == w && x1[i][1] == h) return 1;
    for (int i = 0; i < 8; i++) if (x2[i][0] == w && x2[i][1] == h) return 2;
    return 4;
}
static int is_pu(int w, int h, const int (*set)[2], int n) { for (int i = 0; i < n; i++) if (set[i][0] == w && set[i][1] == h) return 1; return 0; }

int xo_motion_estimate(const xo_pixel* fencPlane, intptr_t fencStride, int w, int h,
                       const xo_pixel* fref, intptr_t refStride,
                       const int32_t* bounds /* mvmin.x, mvmin.y, mvmax.x, mvmax.y (full-pel) */,
                       int qmvpx, int qmvpy, int numCand, const int32_t* mvc /* qpel x,y pairs */,
                       int merange, int method, int subme, const uint16_t* costRowCentre, int32_t* outQMv)
{
    return xo_motion_estimate_sea(fencPlane, fencStride, w, h, fref, refStride, bounds, qmvpx, qmvpy, numCand, mvc, merange, method, subme, costRowCentre, outQMv, NULL);
}

/* the search of Search::predInterSearch (search.cpp:2582): setSourcePU's Yuv overload with bChroma -- at subme >= 3, and when the 4:2:0 chroma block
 * is a multiple of 4x4, every subpelCompare adds the SATD of the Cb and Cr predictions (motion.cpp:236-238, 1805-1865) */
static const xo_pixel* g_fencC[2]; static const xo_pixel* g_refC[2]; static intptr_t g_fencStrideC, g_refStrideC;
int xo_motion_estimate_chroma(const xo_pixel* fencPlane, intptr_t fencStride, int w, int h, const xo_pixel* fref, intptr_t refStride, const int32_t* bounds,
                              int qmvpx, int qmvpy, int numCand, const int32_t* mvc, int merange, int method, int subme, const uint16_t* costRowCentre, int32_t* outQMv,
                              const xo_pixel* fencCb, const xo_pixel* fencCr, intptr_t fencStrideC, const xo_pixel* refCb, const xo_pixel* refCr, intptr_t refStrideC)
{
    g_fencC[0] = fencCb; g_fencC[1] = fencCr; g_refC[0] = refCb; g_refC[1] = refCr; g_fencStrideC = fencStrideC; g_refStrideC = refStrideC;
    const int r = xo_motion_estimate_sea(fencPlane, fencStride, w, h, fref, refStride, bounds, qmvpx, qmvpy, numCand, mvc, merange, method, subme, costRowCentre, outQMv, NULL);
    g_fencC[0] = g_fencC[1] = NULL;
    return r;
}

/* the same with the 12 SEA integral planes of the reference picture (pointers at the PU's co-located position), needed by XO_ME_SEA */
int xo_motion_estimate_sea(const xo_pixel* fencPlane, intptr_t fencStride, int w, int h,
                           const xo_pixel* fref, intptr_t refStride, const int32_t* bounds,
                           int qmvpx, int qmvpy, int numCand, const int32_t* mvc,
                           int merange, int method, int subme, const uint16_t* costRowCentre, int32_t* outQMv, const uint32_t* const* integral)
{
    me_t me, *m = &me;
    m->fref = fref; m->stride = refStride; m->w = w; m->h = h; m->cost = costRowCentre;
    m->mvp.x = qmvpx; m->mvp.y = qmvpy;
    for (int y = 0; y < h; y++) memcpy(m->fenc + 64 * y, fencPlane + y * fencStride, w * sizeof(xo_pixel));
    m->chroma = 0;
    if (g_fencC[0] && subme > 2 && !((w >> 1) & 3) && !((h >> 1) & 3))
    {   /* bChromaSATD (motion.cpp:236-238): chromaSatd exists for 4:2:0 blocks that are multiples of 4x4 (primitives.cpp:213-234) */
        m->chroma = 1; m->cw = w >> 1; m->ch = h >> 1; m->strideC = g_refStrideC;
        for (int c = 0; c < 2; c++)
        {
            m->frefC[c] = g_refC[c];
            for (int y = 0; y < m->ch; y++) memcpy(m->fencC[c] + 32 * y, g_fencC[c] + y * g_fencStrideC, m->cw * sizeof(xo_pixel));
        }
    }
    const mv_t mvmin = { bounds[0], bounds[1] }, mvmax = { bounds[2], bounds[3] };
    const mv_t qmvmin = { mvmin.x * 4, mvmin.y * 4 }, qmvmax = { mvmax.x * 4, mvmax.y * 4 };
    int costs[4];

    /* motion.cpp:955-1012: start point */
    mv_t pmv = { qmvpx > qmvmax.x ? qmvmax.x : qmvpx, qmvpy > qmvmax.y ? qmvmax.y : qmvpy };
    if (pmv.x < qmvmin.x) pmv.x = qmvmin.x;
    if (pmv.y < qmvmin.y) pmv.y = qmvmin.y;
    mv_t bestpre = pmv;
    int bprecost = subpel_compare(m, pmv.x, pmv.y, 0);           /* no mvcost on the MVP itself (:970) */
    mv_t bmv = { (pmv.x + 2) >> 2, (pmv.y + 2) >> 2 };
    int bcost = bprecost;
    if ((pmv.x | pmv.y) & 3)
        bcost = sad_at(m, bmv.x, bmv.y) + mvcost(m, bmv.x * 4, bmv.y * 4);
    if (pmv.x | pmv.y)
    {
        int cost = sad_at(m, 0, 0) + mvcost(m, 0, 0);
        if (cost < bcost)
        {
            bcost = cost; bmv.x = 0;
            int t = 0 < mvmax.y ? 0 : mvmax.y;
            bmv.y = t > mvmin.y ? t : mvmin.y;
        }
    }
    for (int i = 0; i < numCand; i++)
    {
        mv_t c = { mvc[2 * i], mvc[2 * i + 1] };
        if (c.x > qmvmax.x) c.x = qmvmax.x; if (c.y > qmvmax.y) c.y = qmvmax.y;
        if (c.x < qmvmin.x) c.x = qmvmin.x; if (c.y < qmvmin.y) c.y = qmvmin.y;
        if ((c.x | c.y) && !(c.x == pmv.x && c.y == pmv.y) && !(c.x == bestpre.x && c.y == bestpre.y))
        {
            int cost = subpel_compare(m, c.x, c.y, 0) + mvcost(m, c.x, c.y);
            if (cost < bprecost) { bprecost = cost; bestpre = c; }
        }
    }
    pmv.x = (pmv.x + 2) >> 2; pmv.y = (pmv.y + 2) >> 2;
    if (bcost == 0)
    {
        outQMv[0] = bmv.x * 4; outQMv[1] = bmv.y * 4;
        return mvcost(m, bmv.x * 4, bmv.y * 4);
    }

    switch (method)
    {
    case XO_ME_DIA:
    {   /* motion.cpp:1016-1039 */
        unsigned ubcost = (unsigned)bcost << 4;
        int i = merange;
        do
        {
            static const mv_t d[4] = { {0,-1}, {0,1}, {-1,0}, {1,0} };
            for (int k = 0; k < 4; k++) costs[k] = sad_at(m, bmv.x + d[k].x, bmv.y + d[k].y) + mvcost(m, (bmv.x + d[k].x) * 4, (bmv.y + d[k].y) * 4);
            int bc = (int)ubcost;
            if ((bmv.y - 1 >= mvmin.y) & (bmv.y - 1 <= mvmax.y)) { if ((costs[0] << 4) + 1 < bc) bc = (costs[0] << 4) + 1; }
            if ((bmv.y + 1 >= mvmin.y) & (bmv.y + 1 <= mvmax.y)) { if ((costs[1] << 4) + 3 < bc) bc = (costs[1] << 4) + 3; }
            if ((costs[2] << 4) + 4 < bc) bc = (costs[2] << 4) + 4;
            if ((costs[3] << 4) + 12 < bc) bc = (costs[3] << 4) + 12;
            ubcost = (unsigned)bc;
            if (!(bc & 15)) break;
            bmv.x -= (int32_t)((uint32_t)bc << 28) >> 30;
            bmv.y -= (int32_t)((uint32_t)bc << 30) >> 30;
            ubcost &= ~15u;
        }
        while (--i && in_range(bmv, mvmin, mvmax));
        bcost = (int)ubcost >> 4;
        break;
    }
    case XO_ME_UMH:
    {   /* motion.cpp:1142-1326 (uneven multi-hexagon); candidates relative to omv, strict `<` in the order written there */
        static const mv_t hex4[16] = { {0,-4}, {0,4}, {-2,-3}, {2,-3}, {-4,-2}, {4,-2}, {-4,-1}, {4,-1},
                                       {-4,0}, {4,0}, {-4,1}, {4,1}, {-4,2}, {4,2}, {-2,3}, {2,3} };   /* motion.cpp:67-73 */
        const int scale = (h * h) >> 4;                                                                /* sizeScale, :60-61,123-153 */
#define SAD_THRESH(v) (bcost < (((v) >> 4) * scale))
        /* COST_MV_X4 (:296-317): all four measured, only the vertical range is tested */
#define X4(ax, ay, bx_, by_, cx, cy, dx, dy) do { const mv_t d_[4] = { {ax, ay}, {bx_, by_}, {cx, cy}, {dx, dy} }; \
        for (int k_ = 0; k_ < 4; k_++) { \
            int c_ = sad_at(m, omv.x + d_[k_].x, omv.y + d_[k_].y) + mvcost(m, (omv.x + d_[k_].x) * 4, (omv.y + d_[k_].y) * 4); \
            if ((omv.y + d_[k_].y >= mvmin.y) & (omv.y + d_[k_].y <= mvmax.y)) \
                if (c_ < bcost) { bcost = c_; bmv.x = omv.x + d_[k_].x; bmv.y = omv.y + d_[k_].y; } } } while (0)
#define DIA1(mx_, my_) do { omv.x = (mx_); omv.y = (my_); X4(0, -1, 0, 1, -1, 0, 1, 0); } while (0)
        /* CROSS (:361-385) */
#define CROSS(start, x_max, y_max) do { int i_ = (start); \
        if ((x_max) <= imin(mvmax.x - omv.x, omv.x - mvmin.x)) for (; i_ < (x_max) - 2; i_ += 4) X4(i_, 0, -i_, 0, i_ + 2, 0, -i_ - 2, 0); \
        for (; i_ < (x_max); i_ += 2) { if (omv.x + i_ <= mvmax.x) COST_MV(omv.x + i_, omv.y); if (omv.x - i_ >= mvmin.x) COST_MV(omv.x - i_, omv.y); } \
        i_ = (start); \
        if ((y_max) <= imin(mvmax.y - omv.y, omv.y - mvmin.y)) for (; i_ < (y_max) - 2; i_ += 4) X4(0, i_, 0, -i_, 0, i_ + 2, 0, -i_ - 2); \
        for (; i_ < (y_max); i_ += 2) { if (omv.y + i_ <= mvmax.y) COST_MV(omv.x, omv.y + i_); if (omv.y - i_ >= mvmin.y) COST_MV(omv.x, omv.y - i_); } } while (0)
        mv_t omv = bmv;
        int ucost1 = bcost, ucost2, cross_start = 1, done = 0;
        DIA1(pmv.x, pmv.y);
        if (pmv.x | pmv.y) DIA1(0, 0);
        ucost2 = bcost;
        if ((bmv.x | bmv.y) && !(bmv.x == pmv.x && bmv.y == pmv.y)) DIA1(bmv.x, bmv.y);
        if (bcost == ucost2) cross_start = 3;
        omv = bmv;
        if (bcost == ucost2 && SAD_THRESH(2000))
        {   /* early termination (:1161-1180) */
            X4(0, -2, -1, -1, 1, -1, -2, 0);
            X4(2, 0, -1, 1, 1, 1, 0, 2);
            if (bcost == ucost1 && SAD_THRESH(500)) done = 1;
            else if (bcost == ucost2)
            {
                int range = (int16_t)((merange >> 1) | 1);
                CROSS(3, range, range);
                X4(-1, -2, 1, -2, -2, -1, 2, -1);
                X4(-2, 1, 2, 1, -1, 2, 1, 2);
                if (bcost == ucost2) done = 1;
                cross_start = range + 2;
            }
        }
        if (done) break;
        if (numCand)
        {   /* adaptive range from the agreement of the predictors (:1186-1236) */
            static const uint8_t range_mul[4][4] = { {3,3,4,4}, {3,4,4,4}, {4,4,4,5}, {4,4,5,6} };
            const int is64 = (w == 64 && h == 64);
            int mvd, denom = 1;
            if (numCand == 1)
                mvd = is64 ? 25 : abs(qmvpx - mvc[0]) + abs(qmvpy - mvc[1]);
            else
            {
                denom = numCand - 1; mvd = 0;
                if (!is64) { mvd = abs(qmvpx - mvc[0]) + abs(qmvpy - mvc[1]); denom++; }
                for (int i = 0; i < numCand - 1; i++)                       /* predictorDifference (:87-98) */
                    mvd += abs(mvc[2 * i] - mvc[2 * i + 2]) + abs(mvc[2 * i + 1] - mvc[2 * i + 3]);
            }
            int sad_ctx = SAD_THRESH(1000) ? 0 : SAD_THRESH(2000) ? 1 : SAD_THRESH(4000) ? 2 : 3;
            int mvd_ctx = mvd < 10 * denom ? 0 : mvd < 20 * denom ? 1 : mvd < 40 * denom ? 2 : 3;
            merange = (merange * range_mul[mvd_ctx][sad_ctx]) >> 2;
        }
        CROSS(cross_start, merange, merange >> 1);
        X4(-2, -2, -2, 2, 2, -2, 2, 2);
        /* hexagon grid (:1243-1320): rings of 16 points at radius i */
        omv = bmv;
        int i = 1;
        do
        {
            for (int j = 0; j < 16; j++)
            {
                mv_t mv = { omv.x + hex4[j].x * i, omv.y + hex4[j].y * i };
                if (in_range(mv, mvmin, mvmax)) COST_MV(mv.x, mv.y);       /* the sad_x4 branch has every point in range */
            }
        }
        while (++i <= merange >> 2);
        if (!in_range(bmv, mvmin, mvmax)) break;
#undef SAD_THRESH
#undef X4
#undef DIA1
#undef CROSS
    }
    /* fall through: `goto me_hex2` (motion.cpp:1323-1324) with the adapted merange */
    case XO_ME_HEX:
    {   /* motion.cpp:1041-1140 */
#define X3_DIR(a, b, c) do { const mv_t d_[3] = { a, b, c }; for (int k_ = 0; k_ < 3; k_++) \
        costs[k_] = sad_at(m, bmv.x + d_[k_].x, bmv.y + d_[k_].y) + mvcost(m, (bmv.x + d_[k_].x) * 4, (bmv.y + d_[k_].y) * 4); } while (0)
#define LT1(x, y) do { if ((y) < (x)) (x) = (y); } while (0)
        { const mv_t a = {-2,0}, b = {-1,2}, c = {1,2}; X3_DIR(a, b, c); }
        bcost <<= 3;
        if ((bmv.y >= mvmin.y) & (bmv.y <= mvmax.y)) LT1(bcost, (costs[0] << 3) + 2);
        if ((bmv.y + 2 >= mvmin.y) & (bmv.y + 2 <= mvmax.y)) { LT1(bcost, (costs[1] << 3) + 3); LT1(bcost, (costs[2] << 3) + 4); }
        { const mv_t a = {2,0}, b = {1,-2}, c = {-1,-2}; X3_DIR(a, b, c); }
        if ((bmv.y >= mvmin.y) & (bmv.y <= mvmax.y)) LT1(bcost, (costs[0] << 3) + 5);
        if ((bmv.y - 2 >= mvmin.y) & (bmv.y - 2 <= mvmax.y)) { LT1(bcost, (costs[1] << 3) + 6); LT1(bcost, (costs[2] << 3) + 7); }
        if (bcost & 7)
        {
            int dir = (bcost & 7) - 2;
            if ((bmv.y + hex2[dir + 1].y >= mvmin.y) & (bmv.y + hex2[dir + 1].y <= mvmax.y))
            {
                bmv.x += hex2[dir + 1].x; bmv.y += hex2[dir + 1].y;
                for (int i = (merange >> 1) - 1; i > 0 && in_range(bmv, mvmin, mvmax); i--)
                {
                    X3_DIR(hex2[dir + 0], hex2[dir + 1], hex2[dir + 2]);
                    bcost &= ~7;
                    if ((bmv.y + hex2[dir + 0].y >= mvmin.y) & (bmv.y + hex2[dir + 0].y <= mvmax.y)) LT1(bcost, (costs[0] << 3) + 1);
                    if ((bmv.y + hex2[dir + 1].y >= mvmin.y) & (bmv.y + hex2[dir + 1].y <= mvmax.y)) LT1(bcost, (costs[1] << 3) + 2);
                    if ((bmv.y + hex2[dir + 2].y >= mvmin.y) & (bmv.y + hex2[dir + 2].y <= mvmax.y)) LT1(bcost, (costs[2] << 3) + 3);
                    if (!(bcost & 7)) break;
                    dir += (bcost & 7) - 2;
                    dir = mod6m1[dir + 1];
                    bmv.x += hex2[dir + 1].x; bmv.y += hex2[dir + 1].y;
                }
            }
        }
        bcost >>= 3;
        /* square refine */
        int dir = 0;
        static const mv_t sq[8] = { {0,-1}, {0,1}, {-1,0}, {1,0}, {-1,-1}, {-1,1}, {1,-1}, {1,1} };
        int c8[8];
        for (int k = 0; k < 8; k++) c8[k] = sad_at(m, bmv.x + sq[k].x, bmv.y + sq[k].y) + mvcost(m, (bmv.x + sq[k].x) * 4, (bmv.y + sq[k].y) * 4);
        int upOk = (bmv.y - 1 >= mvmin.y) & (bmv.y - 1 <= mvmax.y), dnOk = (bmv.y + 1 >= mvmin.y) & (bmv.y + 1 <= mvmax.y);
        if (upOk && c8[0] < bcost) { bcost = c8[0]; dir = 1; }
        if (dnOk && c8[1] < bcost) { bcost = c8[1]; dir = 2; }
        if (c8[2] < bcost) { bcost = c8[2]; dir = 3; }
        if (c8[3] < bcost) { bcost = c8[3]; dir = 4; }
        if (upOk && c8[4] < bcost) { bcost = c8[4]; dir = 5; }
        if (dnOk && c8[5] < bcost) { bcost = c8[5]; dir = 6; }
        if (upOk && c8[6] < bcost) { bcost = c8[6]; dir = 7; }
        if (dnOk && c8[7] < bcost) { bcost = c8[7]; dir = 8; }
        bmv.x += square1[dir].x; bmv.y += square1[dir].y;
        break;
    }
    case XO_ME_STAR:
    {   /* motion.cpp:1328-1436 */
        star_t st = { bmv, bcost, 0, 0 }, *s = &st;
        star_pattern(m, mvmin, mvmax, s, 3, merange);
        bmv = s->bmv; bcost = s->bcost;
        int done = 0;
        if (s->bDistance == 1)
        {
            if (s->bPointNr)
            {
                int saved = bcost;
                mv_t mv1 = { bmv.x + offsets2[(s->bPointNr - 1) * 2].x, bmv.y + offsets2[(s->bPointNr - 1) * 2].y };
                mv_t mv2 = { bmv.x + offsets2[(s->bPointNr - 1) * 2 + 1].x, bmv.y + offsets2[(s->bPointNr - 1) * 2 + 1].y };
                if (in_range(mv1, mvmin, mvmax)) COST_MV(mv1.x, mv1.y);
                if (in_range(mv2, mvmin, mvmax)) COST_MV(mv2.x, mv2.y);
                if (bcost == saved) done = 1;
            }
            else done = 1;
        }
        if (done) break;
        const int RasterDistance = 5;
        if (s->bDistance > RasterDistance)
        {
            mv_t t;
            for (t.y = mvmin.y; t.y <= mvmax.y; t.y += RasterDistance)
                for (t.x = mvmin.x; t.x <= mvmax.x; t.x += RasterDistance)
                {
                    if (t.x + RasterDistance * 3 <= mvmax.x)
                    {
                        for (int k = 0; k < 4; k++) costs[k] = sad_at(m, t.x + RasterDistance * k, t.y);
                        for (int k = 0; k < 4; k++)
                        {
                            if (k) t.x += RasterDistance;
                            /* reference quirk (:1392): the 4th candidate's mv cost is taken at tmv << 3 */
                            int c = costs[k] + (k == 3 ? mvcost(m, t.x * 8, t.y * 8) : mvcost(m, t.x * 4, t.y * 4));
                            if (c < bcost) { bcost = c; bmv = t; }
                        }
                    }
                    else
                        COST_MV(t.x, t.y);
                }
        }
        int bDistance = s->bDistance;
        while (bDistance > 0)
        {
            st.bmv = bmv; st.bcost = bcost; st.bPointNr = 0; st.bDistance = 0;
            star_pattern(m, mvmin, mvmax, s, 32, merange);
            bmv = s->bmv; bcost = s->bcost; bDistance = s->bDistance;
            if (bDistance == 1)
            {
                if (!s->bPointNr) break;
                mv_t mv1 = { bmv.x + offsets2[(s->bPointNr - 1) * 2].x, bmv.y + offsets2[(s->bPointNr - 1) * 2].y };
                mv_t mv2 = { bmv.x + offsets2[(s->bPointNr - 1) * 2 + 1].x, bmv.y + offsets2[(s->bPointNr - 1) * 2 + 1].y };
                if (in_range(mv1, mvmin, mvmax)) COST_MV(mv1.x, mv1.y);
                if (in_range(mv2, mvmin, mvmax)) COST_MV(mv2.x, mv2.y);
                break;
            }
        }
        break;
    }
    case XO_ME_SEA:
    {   /* motion.cpp:1438-1591: successive elimination -- the ads pre-filter on the integral planes decides which positions of
           each row get a SAD.  Costs are restated literally, including the doubled-MVP indexing of p_cost_mvx / p_cost_mvy. */
        if (!integral) return -1;
        const mv_t omv = bmv;
        const int minX = omv.x - merange > mvmin.x ? omv.x - merange : mvmin.x, minY = omv.y - merange > mvmin.y ? omv.y - merange : mvmin.y;
        const int maxX = omv.x + merange < mvmax.x ? omv.x + merange : mvmax.x, maxY = omv.y + merange < mvmax.y ? omv.y + merange : mvmax.y;
        const uint16_t* p_cost_mvx = m->cost - 2 * qmvpx;       /* m_cost_mvx - qmvp.x, with m_cost_mvx = m_cost - mvp.x (bitcost.h:49-52) */
        const uint16_t* p_cost_mvy = m->cost - 2 * qmvpy;
        const int meRangeWidth = (maxX - minX + 3) & ~3;
        int16_t* scratch = (int16_t*)calloc((size_t)(merange * 2 + 4 > meRangeWidth ? merange * 2 + 4 : meRangeWidth) + 4, sizeof(int16_t));
        uint16_t* costMvX = (uint16_t*)malloc((size_t)((meRangeWidth > 0 ? meRangeWidth : 0) + 4) * sizeof(uint16_t));      /* (an empty window -- maxX < minX -- makes the width negative: no position is costed then) */
        for (int i = 0; i < meRangeWidth; i++) costMvX[i] = m->cost[4 * (minX + i) - qmvpx];     /* m_fpelMvCosts[-qmvp.x & 3] + (-qmvp.x >> 2) + minX (bitcost.cpp:57-80) */
        int deltaX = w <= 8 ? w : w >> 1, deltaY = h <= 8 ? h : h >> 1;
        static const int smallRect[5][2] = { {4,4}, {16,12}, {12,16}, {16,4}, {4,16} };
        static const int vertRect[4][2] = { {32,64}, {16,32}, {8,16}, {4,8} }, horRect[4][2] = { {64,32}, {32,16}, {16,8}, {8,4} };
        static const int asymV[6][2] = { {12,16}, {4,16}, {24,32}, {8,32}, {48,64}, {16,64} }, asymH[6][2] = { {16,12}, {16,4}, {32,24}, {32,8}, {64,48}, {64,16} };
        static const int mulStride[13][2] = { {64,64}, {32,32}, {16,16}, {32,64}, {16,32}, {8,16}, {4,8}, {12,16}, {4,16}, {24,32}, {8,32}, {48,64}, {16,64} };
        const int verticalRect = is_pu(w, h, vertRect, 4), horizontalRect = is_pu(w, h, horRect, 4);
        int tw, th;                                              /* the sub-block whose DC the pre-filter compares (:1485-1502) */
        if (verticalRect) { tw = w; th = h >> 1; }
        else if (horizontalRect) { tw = w >> 1; th = h; }
        else if (is_pu(w, h, asymV, 6) || is_pu(w, h, asymH, 6)) { if (is_pu(w, h, smallRect, 5)) { tw = w; th = h; } else { tw = w >> 1; th = h >> 1; } }
        else if (w <= 8) { tw = w; th = h; }
        else { tw = w >> 1; th = h >> 1; }
        int encDC[4];
        {
            const xo_pixel* f4[4] = { m->fenc, m->fenc + deltaX, m->fenc + deltaY * 64, m->fenc + deltaX + deltaY * 64 };
            for (int k = 0; k < 4; k++)
            {
                int sum = 0;
                for (int y = 0; y < th; y++) for (int x = 0; x < tw; x++) sum += f4[k][y * 64 + x];
                encDC[k] = sum;
            }
        }
        int plane;
        switch (deltaX)
        {
        case 32: plane = deltaY % 24 == 0 ? 1 : deltaY == 8 ? 2 : 0; break;
        case 24: plane = 3; break;
        case 16: plane = deltaY % 12 == 0 ? 5 : deltaY == 4 ? 6 : 4; break;
        case 12: plane = 7; break;
        case 8: plane = deltaY == 32 ? 8 : 9; break;
        case 4: plane = deltaY == 16 ? 10 : 11; break;
        default: plane = 11; break;
        }
        const uint32_t* sumsBase = integral[plane];
        if (is_pu(w, h, mulStride, 13)) deltaY *= (int)refStride;
        if (verticalRect) encDC[1] = encDC[2];
        if (horizontalRect) deltaY = deltaX;
        const int parts = ads_parts(w, h);
        for (int ty = minY; ty <= maxY; ty++)
        {
            const int ycost = p_cost_mvy[ty] << 2;
            if (bcost <= ycost) continue;
            bcost -= ycost;
            const int xn = xo_ads(parts, w, encDC, sumsBase + minX + ty * refStride, deltaY, costMvX, scratch, meRangeWidth, bcost);
            int i;
            for (i = 0; i < xn - 2; i += 3)
                for (int k = 0; k < 3; k++)
                {   /* COST_MV_X3_ABS (:319-332) */
                    const int mx = minX + scratch[i + k];
                    const int c = sad_at(m, mx, ty) + p_cost_mvx[mx * 4];
                    if (c < bcost) { bcost = c; bmv.x = mx; bmv.y = ty; }
                }
            bcost += ycost;
            for (; i < xn; i++) COST_MV(minX + scratch[i], ty);
        }
        free(scratch); free(costMvX);
        break;
    }
    case XO_ME_FULL:
    {   /* motion.cpp:1593-1637 (visiting order == plain raster) */
        mv_t t;
        for (t.y = mvmin.y; t.y <= mvmax.y; t.y++)
            for (t.x = mvmin.x; t.x <= mvmax.x; t.x++)
                COST_MV(t.x, t.y);
        break;
    }
    default:
        return -1;
    }

    /* motion.cpp:1644-1768 */
    if (bprecost < bcost) { bmv = bestpre; bcost = bprecost; }
    else { bmv.x *= 4; bmv.y *= 4; }
    const int* wl = workload[subme];
    if (!bcost)
        bcost = mvcost(m, bmv.x, bmv.y);
    else
    {
        int hpelSatd = wl[4];
        if (hpelSatd) bcost = subpel_compare(m, bmv.x, bmv.y, 1) + mvcost(m, bmv.x, bmv.y);
        for (int iter = 0; iter < wl[0]; iter++)
        {
            int bdir = 0;
            for (int i = 1; i <= wl[1]; i++)
            {
                int qx = bmv.x + square1[i].x * 2, qy = bmv.y + square1[i].y * 2;
                if ((qy < qmvmin.y) | (qy > qmvmax.y)) continue;
                int cost = subpel_compare(m, qx, qy, hpelSatd) + mvcost(m, qx, qy);
                if (cost < bcost) { bcost = cost; bdir = i; }
            }
            if (bdir) { bmv.x += square1[bdir].x * 2; bmv.y += square1[bdir].y * 2; }
            else break;
        }
        if (!hpelSatd) bcost = subpel_compare(m, bmv.x, bmv.y, 1) + mvcost(m, bmv.x, bmv.y);
        for (int iter = 0; iter < wl[2]; iter++)
        {
            int bdir = 0;
            for (int i = 1; i <= wl[3]; i++)
            {
                int qx = bmv.x + square1[i].x, qy = bmv.y + square1[i].y;
                if ((qy < qmvmin.y) | (qy > qmvmax.y)) continue;
                int cost = subpel_compare(m, qx, qy, 1) + mvcost(m, qx, qy);
                if (cost < bcost) { bcost = cost; bdir = i; }
            }
            if (bdir) { bmv.x += square1[bdir].x; bmv.y += square1[bdir].y; }
            else break;
        }
    }
    if (bmv.x | bmv.y)
    {
        int cost = subpel_compare(m, 0, 0, 1) + mvcost(m, 0, 0);
        if (cost <= bcost) { bmv.x = 0; bmv.y = 0; }
    }
    outQMv[0] = bmv.x; outQMv[1] = bmv.y;
    return bcost;
}

/* ------------------------------------------------------------------------------------------ */
/* inter TU pipeline                                                                           */
/* ------------------------------------------------------------------------------------------ */
static int g_tqChroma;
static const xo_pixel* g_tqRef1; static int g_tqMv1x, g_tqMv1y;
static int g_tqDst4;
/* the chain of an intra luma 4x4 TU (quant.cpp:429-432, 585-603): DST-VII pair, no DC shortcut; the prediction is whatever `fref` holds at MV (0,0) */
uint32_t xo_tq_tu_dst4(const xo_pixel* cur, intptr_t curStride, const xo_pixel* predPlane, intptr_t predStride, int qp, int addNumerator,
                       int16_t* coeff, int32_t* deltaU, xo_pixel* recon, intptr_t reconStride, uint64_t* sse)
{
    g_tqDst4 = 1;
    const uint32_t r = xo_tq_tu(2, cur, curStride, predPlane, predStride, 0, 0, qp, addNumerator, NULL, coeff, deltaU, recon, reconStride, sse);
    g_tqDst4 = 0;
    return r;
}
/* Predict::predInterLumaShort (predict.cpp:302-338) */
static void mc_luma_short(const xo_pixel* fref, intptr_t stride, int N, int qx, int qy, int16_t* dst)
{
    const xo_pixel* src = fref + (qx >> 2) + (qy >> 2) * stride;
    const int xf = qx & 3, yf = qy & 3;
    if (!(xf | yf)) xo_p2s(N, N, src, stride, dst, N);
    else if (!yf) xo_interp_hps(8, N, N, src, stride, dst, N, xf, 0);
    else if (!xf) xo_interp_vps(8, N, N, src, stride, dst, N, yf);
    else
    {
        int16_t immed[32 * (32 + 7)];
        xo_interp_hps(8, N, N, src, stride, immed, N, xf, 1);
        xo_interp_vss(8, N, N, immed + 3 * N, N, dst, N, yf);
    }
}
/* the same chain for a bi-directionally predicted TU: the B-slice branch of Predict::motionCompensation (predict.cpp:186-211) -- two 14-bit predictions, addAvg */
uint32_t xo_tq_tu_bi(int log2TrSize, const xo_pixel* cur, intptr_t curStride, const xo_pixel* fref0, const xo_pixel* fref1, intptr_t refStride,
                     int qmv0x, int qmv0y, int qmv1x, int qmv1y, int qp, int addNumerator, const int32_t* quantCoeff,
                     int16_t* coeff, int32_t* deltaU, xo_pixel* recon, intptr_t reconStride, uint64_t* sse)
{
    g_tqRef1 = fref1; g_tqMv1x = qmv1x; g_tqMv1y = qmv1y;
    const uint32_t r = xo_tq_tu(log2TrSize, cur, curStride, fref0, refStride, qmv0x, qmv0y, qp, addNumerator, quantCoeff, coeff, deltaU, recon, reconStride, sse);
    g_tqRef1 = NULL;
    return r;
}
/* the same chain for a chroma TU of a 4:2:0 picture: Predict::predInterChromaPixel (predict.cpp:340-380) in place of the luma motion compensation; qp = the plane's chroma qp */
uint32_t xo_tq_tu_chroma(int log2TrSize, const xo_pixel* cur, intptr_t curStride, const xo_pixel* fref, intptr_t refStride,
                         int qmvx, int qmvy, int qp, int addNumerator, const int32_t* quantCoeff,
                         int16_t* coeff, int32_t* deltaU, xo_pixel* recon, intptr_t reconStride, uint64_t* sse)
{
    g_tqChroma = 1;
    const uint32_t r = xo_tq_tu(log2TrSize, cur, curStride, fref, refStride, qmvx, qmvy, qp, addNumerator, quantCoeff, coeff, deltaU, recon, reconStride, sse);
    g_tqChroma = 0;
    return r;
}
uint32_t xo_tq_tu(int log2TrSize, const xo_pixel* cur, intptr_t curStride, const xo_pixel* fref, intptr_t refStride,
                  int qmvx, int qmvy, int qp, int addNumerator, const int32_t* quantCoeff,
                  int16_t* coeff, int32_t* deltaU, xo_pixel* recon, intptr_t reconStride, uint64_t* sse)
{
    static const int quantScales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };   /* scalinglist.cpp:129 */
    static const int invQuantScales[6] = { 40, 45, 51, 57, 64, 72 };                  /* scalinglist.cpp:130 */
    const int N = 1 << log2TrSize, num = N * N;
    xo_pixel pred[32 * 32];
    int16_t resi[32 * 32], dct[32 * 32];
    int32_t du[32 * 32], flat[32 * 32];

    /* predict.cpp:279-300: copy_pp | luma_hpp | luma_vpp | luma_hvpp by MV fraction */
    if (g_tqRef1)
    {
        int16_t s0[32 * 32], s1[32 * 32];
        mc_luma_short(fref, refStride, N, qmvx, qmvy, s0);
        mc_luma_short(g_tqRef1, refStride, N, g_tqMv1x, g_tqMv1y, s1);
        xo_addAvg(N, N, s0, s1, pred, N, N, N);
    }
    else if (g_tqChroma)
    {   /* predict.cpp:340-380 (4:2:0: mvx = mv.x, mvy = mv.y in eighth-pels): copy | filter_hpp | filter_vpp | filter_hps (row-extended) + filter_vsp */
        const xo_pixel* csrc = fref + (qmvx >> 3) + (qmvy >> 3) * refStride;
        const int cxf = qmvx & 7, cyf = qmvy & 7;
        if (!(cxf | cyf)) xo_copy_pp(N, N, pred, N, csrc, refStride);
        else if (!cyf) xo_interp_hpp(4, N, N, csrc, refStride, pred, N, cxf);
        else if (!cxf) xo_interp_vpp(4, N, N, csrc, refStride, pred, N, cyf);
        else
        {
            int16_t immed[32 * (32 + 3)];
            xo_interp_hps(4, N, N, csrc, refStride, immed, N, cxf, 1);
            xo_interp_vsp(4, N, N, immed + N, N, pred, N, cyf);
        }
    }
    const xo_pixel* src = fref + (qmvx >> 2) + (qmvy >> 2) * refStride;
    int xf = qmvx & 3, yf = qmvy & 3;
    if (g_tqChroma || g_tqRef1) { }
    else if (!(xf | yf)) xo_copy_pp(N, N, pred, N, src, refStride);
    else if (!yf) xo_interp_hpp(8, N, N, src, refStride, pred, N, xf);
    else if (!xf) xo_interp_vpp(8, N, N, src, refStride, pred, N, yf);
    else xo_interp_hvpp(8, N, N, src, refStride, pred, N, xf, yf);

    xo_sub_ps(N, N, resi, N, cur, pred, curStride, N);
    if (g_tqDst4 && N == 4) xo_dst4(resi, dct, N); else xo_dct(N, resi, dct, N);

    /* quant.cpp:458-469 */
    const int per = qp / 6, rem = qp % 6;
    const int transformShift = 15 - X265_DEPTH - log2TrSize;
    const int qbits = 14 + per + transformShift;
    const int add = addNumerator << (qbits - 9);
    if (!quantCoeff) { for (int i = 0; i < num; i++) flat[i] = quantScales[rem]; quantCoeff = flat; }
    uint32_t numSig = xo_quant(dct, quantCoeff, deltaU ? deltaU : du, coeff, qbits, add, num);

    if (recon)
    {
        int16_t deq[32 * 32], res2[32 * 32];
        if (!numSig)
            memset(res2, 0, sizeof(res2));              /* search.cpp:5630 blockfill_s(curResi, 0) */
        else
        {
            /* quant.cpp:555-568 */
            const int shift = 20 - 14 - transformShift;
            xo_dequant_normal(coeff, deq, num, invQuantScales[rem] << per, shift);
            if (numSig == 1 && coeff[0] != 0 && !(g_tqDst4 && N == 4))
            {   /* DC shortcut, quant.cpp:588-597 */
                const int shift_1st = 7 - 6, add_1st = 1 << (shift_1st - 1);
                const int shift_2nd = 12 - (X265_DEPTH - 8) - 3, add_2nd = 1 << (shift_2nd - 1);
                int dc_val = (((deq[0] * (64 >> 6) + add_1st) >> shift_1st) * (64 >> 3) + add_2nd) >> shift_2nd;
                xo_blockfill_s(N, res2, N, (int16_t)dc_val);
            }
            else if (g_tqDst4 && N == 4)
                xo_idst4(deq, res2, N);
            else
                xo_idct(N, deq, res2, N);
        }
        xo_add_ps(N, N, recon, reconStride, pred, res2, N, N);
        if (sse) *sse = xo_sse_pp(N, N, cur, curStride, recon, reconStride);
    }
    return numSig;
}

/* ---------------------------------------------------------------------------------------------------------------------------------------
 * The tail of Search::puMotionEstimation / predInterSearch for one 2Nx2N PU after its per-reference searches (encoder/search.cpp:258-556):
 * bits and cost of every (list, reference), best reference per list, the bidirectional candidate (average of the two best predictions at
 * SATD, and the same with both MVs zero), final choice.  No AMVP list (the search predictor is the MVP): checkBestMVP / updateMVP do nothing.
 * --------------------------------------------------------------------------------------------------------------------------------------- */
void xo_mvbits_row(int halfRange, float* out)
{   /* BitCost::CalculateLogs (bitcost.cpp:72-86): s_bitsizes, symmetric; out[halfRange + d] */
    float log2_2 = (float)(2.0f / log(2.0f));
    for (int i = 0; i <= halfRange; i++)
        out[halfRange + i] = out[halfRange - i] = i ? (float)(log((double)(float)(i + 1)) * log2_2 + 1.718f) : 0.718f;
}
static uint32_t mv_bitcost(const float* bitsCentre, int mvx, int mvy, int px, int py)
{   /* bitcost.h:60-70 */
    return (uint32_t)(bitsCentre[mvx - px] + bitsCentre[mvy - py] + 0.5f);
}
uint32_t xo_mv_bitcost(const float* bitsCentre, int mvx, int mvy, int px, int py) { return mv_bitcost(bitsCentre, mvx, mvy, px, py); }
uint64_t xo_rd_lambda(int qp) { return (uint64_t)floor(256.0 * xo_lambda(qp)); }          /* RDCost::setLambda, rdcost.h:88-92 */
static uint32_t rd_getcost(uint64_t lambda, uint32_t bits) { return (uint32_t)((bits * lambda + 128) >> 8); }   /* rdcost.h:164-169 */

/* luma motion compensation of one block (Predict::predInterLumaPixel, predict.cpp:279-300) */
static void mc_luma(const xo_pixel* fref, intptr_t stride, int w, int h, int qx, int qy, xo_pixel* dst)
{
    const xo_pixel* src = fref + (qx >> 2) + (qy >> 2) * stride;
    const int xf = qx & 3, yf = qy & 3;
    if (!(xf | yf)) xo_copy_pp(w, h, dst, w, src, stride);
    else if (!yf) xo_interp_hpp(8, w, h, src, stride, dst, w, xf);
    else if (!xf) xo_interp_vpp(8, w, h, src, stride, dst, w, yf);
    else xo_interp_hvpp(8, w, h, src, stride, dst, w, xf, yf);
}
/* Search::selectMVP (search.cpp:2347-2382; m_bFrameParallel off): which of the two AMVP candidates predicts the PU better -- each is clipped like CUData::clipMv
 * (clip = xmin, ymin, xmax, ymax in quarter-pels), the block is motion compensated (predInterLumaPixel) and compared at SAD; ties go to candidate 0.
 * costs (optional) receives the two SADs. */
int xo_select_mvp(int w, int h, const xo_pixel* fenc, intptr_t fencStride, const xo_pixel* fref, intptr_t refStride, const int32_t* amvp, const int32_t* clip, int32_t* costs)
{
    if (amvp[0] == amvp[2] && amvp[1] == amvp[3]) return 0;
    xo_pixel pred[64 * 64];
    int c[2];
    for (int i = 0; i < 2; i++)
    {
        int mx = amvp[2 * i], my = amvp[2 * i + 1];
        mx = mx < clip[0] ? clip[0] : mx > clip[2] ? clip[2] : mx;            /* X265_MIN(xmax, X265_MAX(xmin, mv.x)), cudata.cpp:2105-2106 */
        my = my < clip[1] ? clip[1] : my > clip[3] ? clip[3] : my;
        mc_luma(fref, refStride, w, h, mx, my, pred);
        c[i] = xo_sad(w, h, fenc, fencStride, pred, w);
        if (costs) costs[i] = c[i];
    }
    return c[0] <= c[1] ? 0 : 1;
}
/* Search::checkBestMVP (search.cpp:4947-4958): would the other AMVP candidate code this MV in fewer bits?  io = { mvpIdx, bits, cost } */
void xo_check_best_mvp(const float* bitsCentre, uint64_t lambda, const int32_t* amvp, int mvx, int mvy, uint32_t* io)
{
    const int idx = (int)io[0];
    const int diffBits = (int)mv_bitcost(bitsCentre, mvx, mvy, amvp[2 * !idx], amvp[2 * !idx + 1]) - (int)mv_bitcost(bitsCentre, mvx, mvy, amvp[2 * idx], amvp[2 * idx + 1]);
    if (diffBits < 0)
    {
        const uint32_t orig = io[1];
        io[0] = !idx;
        io[1] = orig + diffBits;
        io[2] = (io[2] - rd_getcost(lambda, orig)) + rd_getcost(lambda, io[1]);
    }
}
/* Search::updateMVP (search.cpp:4961-4967): bits / cost of the MV against `amvp` when they were counted against `alter`; io = { bits, cost } */
void xo_update_mvp(const float* bitsCentre, uint64_t lambda, int amvpx, int amvpy, int mvx, int mvy, int alterx, int altery, uint32_t* io)
{
    const int diffBits = (int)mv_bitcost(bitsCentre, mvx, mvy, amvpx, amvpy) - (int)mv_bitcost(bitsCentre, mvx, mvy, alterx, altery);
    const uint32_t orig = io[0];
    io[0] = orig + diffBits;
    io[1] = (io[1] - rd_getcost(lambda, orig)) + rd_getcost(lambda, io[0]);
}
int xo_bidir_satd(int w, int h, const xo_pixel* fenc, intptr_t fencStride, const xo_pixel* ref0, intptr_t stride0, int mv0x, int mv0y,
                  const xo_pixel* ref1, intptr_t stride1, int mv1x, int mv1y)
{   /* search.cpp:436-446: predInterLumaPixel twice, pixelavg_pp, SATD */
    xo_pixel p0[64 * 64], p1[64 * 64], avg[64 * 64];
    mc_luma(ref0, stride0, w, h, mv0x, mv0y, p0); mc_luma(ref1, stride1, w, h, mv1x, mv1y, p1);
    xo_pixelavg_pp(w, h, avg, w, p0, w, p1, w);
    return xo_satd(w, h, fenc, fencStride, avg, w);
}

/* numRef[l] references per list (numRef[1] = 0: P slice).  Per (list, ref) k = l * 4 + r: mv / mvp (quarter-pel), cost (motionEstimate's return value) and
 * mvcost.  clip[4] = the PU's quarter-pel MV limits (CUData::clipMv: xmin, ymin, xmax, ymax).  fenc / refs[k] point at the PU in the source and co-located in
 * each reference.  out[12] = { mv0x, mv0y, mv1x, mv1y, mvp0x, mvp0y, mvp1x, mvp1y, ref0, ref1 (-1 = list unused), bits, cost }, mvCostOut[2]. */
void xo_inter_merge(int w, int h, const int32_t* numRef, const int32_t* mv, const int32_t* mvp, const int32_t* cost, const int32_t* mvcost,
                    const float* bitsCentre, uint64_t lambda, int bidir, int sourceMaxDim, const int32_t* clip,
                    const xo_pixel* fenc, intptr_t fencStride, const xo_pixel* const* refs, intptr_t refStride, int32_t* out, uint32_t* mvCostOut)
{
    const int isP = numRef[1] == 0;
    const uint32_t listSelBits[3] = { isP ? 1u : 3u, 3u, 5u };                    /* getBlkBits, SIZE_2Nx2N (search.cpp:4896-4901) */
    struct { int mvx, mvy, px, py, ref; uint32_t cost, bits, mvCost; } best[2];
    best[0].cost = best[1].cost = 0xFFFFFFFFu; best[0].ref = best[1].ref = -1;
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < numRef[l]; r++)
        {
            const int k = l * 4 + r;
            uint32_t bits = listSelBits[l] + 1 /* MVP_IDX_BITS */ + (uint32_t)(r + (r < numRef[l] - 1));      /* getTUBits, search.h:252-255 */
            bits += mv_bitcost(bitsCentre, mv[2 * k], mv[2 * k + 1], mvp[2 * k], mvp[2 * k + 1]);
            const uint32_t c = (uint32_t)(cost[k] - mvcost[k]) + rd_getcost(lambda, bits);                    /* search.cpp:372-374 */
            if (c < best[l].cost)
            {
                best[l].cost = c; best[l].bits = bits; best[l].mvCost = (uint32_t)mvcost[k]; best[l].ref = r;
                best[l].mvx = mv[2 * k]; best[l].mvy = mv[2 * k + 1]; best[l].px = mvp[2 * k]; best[l].py = mvp[2 * k + 1];
            }
        }
    uint32_t bidirCost = 0xFFFFFFFFu; int bidirBits = 0;
    int b0x = 0, b0y = 0, b1x = 0, b1y = 0;
    if (!isP && bidir && best[0].cost != 0xFFFFFFFFu && best[1].cost != 0xFFFFFFFFu)
    {   /* search.cpp:420-503 */
        b0x = best[0].mvx; b0y = best[0].mvy; b1x = best[1].mvx; b1y = best[1].mvy;
        int satd = xo_bidir_satd(w, h, fenc, fencStride, refs[best[0].ref], refStride, b0x, b0y, refs[4 + best[1].ref], refStride, b1x, b1y);
        bidirBits = (int)(best[0].bits + best[1].bits + listSelBits[2] - (listSelBits[0] + listSelBits[1]));
        bidirCost = (uint32_t)satd + rd_getcost(lambda, (uint32_t)bidirBits);
        int tryZero = (b0x | b0y | b1x | b1y) != 0;
        if (tryZero)
        {   /* setSearchRange(cu, mvzero, max(sourceWidth, sourceHeight)) (search.cpp:4969-5021), mvmax.y += 2, << 2; both MVPs inside */
            int mnx = -(sourceMaxDim << 2), mny = mnx, mxx = sourceMaxDim << 2, mxy = mxx;
            mnx = mnx < clip[0] ? clip[0] : mnx > clip[2] ? clip[2] : mnx; mxx = mxx < clip[0] ? clip[0] : mxx > clip[2] ? clip[2] : mxx;
            mny = mny < clip[1] ? clip[1] : mny > clip[3] ? clip[3] : mny; mxy = mxy < clip[1] ? clip[1] : mxy > clip[3] ? clip[3] : mxy;
            mnx >>= 2; mny >>= 2; mxx >>= 2; mxy >>= 2;
            if (mxy < mny) mxy = mny;
            mxy += 2;
            mnx <<= 2; mny <<= 2; mxx <<= 2; mxy <<= 2;
            for (int l = 0; l < 2; l++)
                tryZero &= best[l].px >= mnx && best[l].px <= mxx && best[l].py >= mny && best[l].py <= mxy;
        }
        if (tryZero)
        {
            satd = xo_bidir_satd(w, h, fenc, fencStride, refs[best[0].ref], refStride, 0, 0, refs[4 + best[1].ref], refStride, 0, 0);
            const uint32_t bits0 = best[0].bits - mv_bitcost(bitsCentre, best[0].mvx, best[0].mvy, best[0].px, best[0].py) + mv_bitcost(bitsCentre, 0, 0, best[0].px, best[0].py);
            const uint32_t bits1 = best[1].bits - mv_bitcost(bitsCentre, best[1].mvx, best[1].mvy, best[1].px, best[1].py) + mv_bitcost(bitsCentre, 0, 0, best[1].px, best[1].py);
            const uint32_t c = (uint32_t)satd + rd_getcost(lambda, bits0) + rd_getcost(lambda, bits1);
            if (c < bidirCost)
            {
                b0x = b0y = b1x = b1y = 0; bidirCost = c;
                bidirBits = (int)(bits0 + bits1 + listSelBits[2] - (listSelBits[0] + listSelBits[1]));
            }
        }
    }
    /* search.cpp:504-555 */
    for (int i = 0; i < 12; i++) out[i] = 0;
    out[8] = out[9] = -1; mvCostOut[0] = mvCostOut[1] = 0;
    if (bidirCost < best[0].cost && bidirCost < best[1].cost)
    {
        out[0] = b0x; out[1] = b0y; out[2] = b1x; out[3] = b1y;
        out[4] = best[0].px; out[5] = best[0].py; out[6] = best[1].px; out[7] = best[1].py;
        out[8] = best[0].ref; out[9] = best[1].ref; out[10] = bidirBits; out[11] = (int32_t)bidirCost;
        mvCostOut[0] = best[0].mvCost; mvCostOut[1] = best[1].mvCost;
    }
    else
    {
        const int l = best[0].cost <= best[1].cost ? 0 : 1;
        out[2 * l] = best[l].mvx; out[2 * l + 1] = best[l].mvy; out[4 + 2 * l] = best[l].px; out[5 + 2 * l] = best[l].py;
        out[8 + l] = best[l].ref; out[10] = (int32_t)best[l].bits; out[11] = (int32_t)best[l].cost; mvCostOut[l] = best[l].mvCost;
    }
}
