/*
 * ref_lookahead.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * The reference's LOOKAHEAD frame-cost path, compiled from its own sources (oracle/Makefile target "la", whole
 * source/common + source/encoder, no asm), driven directly on its own classes:
 *
 *     PicYuv (full-resolution source, borders replicated with the reference's extendPicBorder)
 *       -> Lowres::create / Lowres::init              (lowres.cpp:79-250, 349-407: frameInitLowres + 4x extendPicBorder)
 *       -> LookaheadTLD::lowresIntraEstimate          (slicetype.cpp:755-870)
 *       -> CostEstimateGroup::singleCost(p0, p1, b)   (slicetype.cpp:4230-4234 -> estimateFrameCost :4365-4463
 *                                                      -> estimateCUCost :4467-4640 -> MotionEstimate::motionEstimate, lowres mode)
 *
 * It pins oracle/x265_oracle_la.c (and through it the HIP batch of x265hip_lookahead_cost_batch).  No thread pool: the serial loops of estimateFrameCost
 * (the cooperative slices through slicedCost below).  Environment X265LA_HME=method0,method1,range0,range1 turns --hme on (param->bEnableHME, hmeSearchMethod[0..1],
 * hmeRange[0..1]; X265_HEX_SEARCH = 1, X265_UMH_SEARCH = 2): Lowres then carries the quarter-resolution planes and estimateFrameCost sweeps them first (slicetype.cpp:4430-4439).
 *
 * usage: x265la_<depth> <width> <height> <nframes> <in.raw> <out.bin> <aq 0|1> [p0,b,p1[,keep] | prop:p0,b,p1,referenced,seed ...]
 *   in.raw  : nframes luma planes, width x height pixels each (u8 / u16), no padding
 *   p0,b,p1,keep,rows : a fifth value > 0 = cooperative lookahead slices of that many block rows (--lookahead-slices, slicetype.cpp:1173-1176)
 *   triples : indices into the frame list, p0 <= b <= p1; "keep" = 1 leaves the MV caches of frame b as the previous
 *             triples left them (bDoSearch then follows the reference's own rule, slicetype.cpp:4376-4377), 0 resets them
 *   prop:   : the estimate (caches reset), then Lookahead::estimateCUPropagate(frames, 0.05, p0, p1, b, referenced) (slicetype.cpp:3850-3953)
 *             on propagateCost arrays of the three pictures pre-filled from `seed`; needs aq = 1 (the AQ factor array must exist)
 *   with X265LA_HME: the header record grows by { m_4x4Width, m_4x4Height }, every frame by its four lowerResBuffer planes (planesize / 2 pixels each, after the lowres planes),
 *             every estimate by lowerResMvs / lowerResMvCosts of both lists (after rowSatds)
 *   out.bin : records of [int64 count][count x int32], in the order written below; the last record holds the time spent in
 *             lowresIntraEstimate (all frames) and in singleCost (all estimates), nanoseconds as (lo, hi) int32 pairs
 */
#include "common.h"
#include "primitives.h"
#include "picyuv.h"
#include "lowres.h"
#include "slicetype.h"
#include "motion.h"
#include "ratecontrol.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace X265_NS;

static FILE* g_out;
static void rec(const std::vector<int32_t>& v)
{
    int64_t n = (int64_t)v.size();
    fwrite(&n, 8, 1, g_out);
    if (n) fwrite(v.data(), 4, (size_t)n, g_out);
}

/* the cuTree propagation step is a protected member */
struct LA : public Lookahead
{
    LA(x265_param* p) : Lookahead(p, NULL) {}
    using Lookahead::estimateCUPropagate;
    using Lookahead::cuTreeFinish;
};

/* estimateFrameCost is reached through the public singleCost; the group only needs the frame list */
struct Group : public CostEstimateGroup
{
    Group(Lookahead& l, Lowres** f) : CostEstimateGroup(l, f) {}
    /* The cooperative-slices branch of estimateFrameCost (slicetype.cpp:4394-4426) with its per-slice row loops (processTasks, :4347-4357) run in
     * this thread: the reference distributes the slices over pool workers, which this harness does not have; every block goes through the
     * reference's own estimateCUCost.  rowsPerSlice = Lookahead::m_numRowsPerSlice (:1173-1176). */
    int64_t slicedCost(LookaheadTLD& tld, int p0, int p1, int b, int rowsPerSlice)
    {
        Lowres* fenc = m_frames[b];
        bool bDoSearch[2] = { fenc->lowresMvs[0][b - p0][0].x == 0x7FFF, p1 > b && fenc->lowresMvs[1][p1 - b][0].x == 0x7FFF };
        fenc->weightedRef[b - p0].isWeighted = false;
        fenc->costEst[b - p0][p1 - b] = 0; fenc->costEstAq[b - p0][p1 - b] = 0;
        const int H = m_lookahead.m_8x8Height, W = m_lookahead.m_8x8Width, nslices = H / rowsPerSlice;
        memset(&m_slice, 0, sizeof(Slice) * nslices);
        for (int i = 0; i < nslices; i++)
        {
            const int firstY = rowsPerSlice * i, lastY = (i == nslices - 1) ? H - 1 : rowsPerSlice * (i + 1) - 1;
            bool lastRow = true;
            for (int cuY = lastY; cuY >= firstY; cuY--)
            {
                fenc->rowSatds[b - p0][p1 - b][cuY] = 0;
                for (int cuX = W - 1; cuX >= 0; cuX--) estimateCUCost(tld, cuX, cuY, p0, p1, b, bDoSearch, lastRow, i, 0);
                lastRow = false;
            }
        }
        for (int i = 0; i < nslices; i++)
        {
            fenc->costEst[b - p0][p1 - b] += m_slice[i].costEst; fenc->costEstAq[b - p0][p1 - b] += m_slice[i].costEstAq;
            if (p1 == b) fenc->intraMbs[b - p0] += m_slice[i].intraMbs;
        }
        int64_t score = fenc->costEst[b - p0][p1 - b];
        if (b != p1) score = score * 100 / (130 + m_lookahead.m_param->bFrameBias);
        fenc->costEst[b - p0][p1 - b] = score;
        return score;
    }
};

static void resetCaches(Lowres& f, int bframes)
{   /* the per-frame cache state Lowres::init establishes (lowres.cpp:359-376) */
    memset(f.costEst, -1, sizeof(f.costEst));
    if (f.qpAqOffset && f.invQscaleFactor) memset(f.costEstAq, -1, sizeof(f.costEstAq));
    for (int y = 0; y < bframes + 2; y++)
        for (int x = 0; x < bframes + 2; x++)
            f.rowSatds[y][x][0] = -1;
    for (int i = 0; i < bframes + 2; i++)
    {
        f.lowresMvs[0][i][0].x = 0x7FFF;
        f.lowresMvs[1][i][0].x = 0x7FFF;
        f.intraMbs[i] = 0;
    }
}

int main(int argc, char** argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s width height nframes in.raw out.bin aq [p0,b,p1[,keep] ...]\n", argv[0]); return 2; }
    const int W = atoi(argv[1]), H = atoi(argv[2]), N = atoi(argv[3]), aq = atoi(argv[6]);
    x265_param* p = x265_param_alloc();
    x265_param_default_preset(p, "medium", NULL);
    p->sourceWidth = W; p->sourceHeight = H; p->internalCsp = X265_CSP_I400;
    const bool weightp = (aq & 2) != 0;            /* aq argument: bit 0 = AQ factors, bit 1 = weighted prediction analysis (weightsAnalyse) */
    p->bEnableWeightedPred = weightp; p->bEnableWeightedBiPred = 0; p->bEnableHME = 0; p->lookaheadSlices = 0;
    const char* hmeEnv = getenv("X265LA_HME");
    const bool hme = hmeEnv && *hmeEnv;
    if (hme)
    {
        if (sscanf(hmeEnv, "%d,%d,%d,%d", &p->hmeSearchMethod[0], &p->hmeSearchMethod[1], &p->hmeRange[0], &p->hmeRange[1]) != 4) { fprintf(stderr, "bad X265LA_HME\n"); return 2; }
        p->bEnableHME = 1;
    }
    p->rc.aqMode = (aq & 1) ? X265_AQ_VARIANCE : X265_AQ_NONE; p->rc.cuTree = 0; p->bEnableTemporalFilter = 0;
    p->bHistBasedSceneCut = 0; p->bAQMotion = 0;
    x265_setup_primitives(p);
    MotionEstimate::initScales();

    FILE* in = fopen(argv[4], "rb");
    g_out = fopen(argv[5], "wb");
    if (!in || !g_out) { fprintf(stderr, "cannot open files\n"); return 2; }

    LA la(p);
    if (!la.create()) { fprintf(stderr, "Lookahead::create failed\n"); return 2; }
    LookaheadTLD& tld = la.m_tld[0];

    int64_t nsIntra = 0, nsCost = 0;
    std::vector<PicYuv*> pics(N);
    std::vector<Lowres*> low(N);
    std::vector<pixel> row(W);
    for (int f = 0; f < N; f++)
    {
        PicYuv* pic = pics[f] = new PicYuv;
        if (!pic->create(p, true, NULL)) { fprintf(stderr, "PicYuv::create failed\n"); return 2; }
        /* the allocation is larger than the picture when the size is no multiple of the CTU: clear it, then fill and replicate */
        const uint32_t ctuRows = (H + p->maxCUSize - 1) / p->maxCUSize;
        memset(pic->m_picBuf[0], 0, sizeof(pixel) * pic->m_stride * (ctuRows * p->maxCUSize + 2 * pic->m_lumaMarginY));
        for (int y = 0; y < H; y++)
        {
            if (fread(row.data(), sizeof(pixel), W, in) != (size_t)W) { fprintf(stderr, "short input\n"); return 2; }
            memcpy(pic->m_picOrg[0] + (intptr_t)y * pic->m_stride, row.data(), W * sizeof(pixel));
        }
        extendPicBorder(pic->m_picOrg[0], pic->m_stride, W, H, pic->m_lumaMarginX, pic->m_lumaMarginY);
        Lowres* l = low[f] = new Lowres;
        memset((void*)l, 0, sizeof(Lowres));
        if (!l->create(p, pic, p->rc.qgSize)) { fprintf(stderr, "Lowres::create failed\n"); return 2; }
        l->init(pic, f, false);
        const int ncu = l->maxBlocksInRow * l->maxBlocksInCol;
        {   /* what calcAdaptiveQuantFrame leaves for weightsAnalyse (slicetype.cpp:49-57, 727-734): pixel sum and the sum of squares about the mean */
            uint64_t sum = 0, ssd = 0;
            for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { const uint64_t v = pic->m_picOrg[0][(intptr_t)y * pic->m_stride + x]; sum += v; ssd += v * v; }
            l->wp_sum[0] = sum; l->wp_ssd[0] = ssd - (sum * sum + ((uint64_t)W * H) / 2) / ((uint64_t)W * H);
            /* weightsAnalyse divides the FULL-resolution sum by the LOWRES area (slicetype.cpp:951-952), which makes its offsets four times too
               large and weights a rarity; aq bit 2 hands it a quarter of the sum instead so that the weighted search path gets exercised */
            if (aq & 4) l->wp_sum[0] = sum / 4;
        }
        if ((aq & 1) && l->invQscaleFactor)
        {   /* synthetic AQ factors (the AQ analysis itself is floating-point host code outside the path): 8.8 fixed point around 1.0 */
            const int nfull = (p->rc.qgSize > 8) ? ncu : ncu << 2;
            for (int i = 0; i < nfull; i++) l->invQscaleFactor[i] = 160 + ((i * 37 + f * 11) % 200);
            if (l->invQscaleFactor8x8) for (int i = 0; i < ncu; i++) l->invQscaleFactor8x8[i] = 160 + ((i * 37 + f * 11) % 200);
        }
        { const auto t0 = std::chrono::steady_clock::now(); tld.lowresIntraEstimate(*l, p->rc.qgSize); nsIntra += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
    }
    Lowres* L0 = low[0];
    const int wcu = L0->maxBlocksInRow, hcu = L0->maxBlocksInCol, ncu = wcu * hcu;
    const int marginX = pics[0]->m_lumaMarginX, marginY = pics[0]->m_lumaMarginY;
    const int ncu4 = la.m_4x4Width * la.m_4x4Height;
    if (hme) rec({ W, H, N, (int32_t)L0->lumaStride, L0->width, L0->lines, wcu, hcu, marginX, marginY, X265_DEPTH, (int32_t)p->rc.qgSize, p->bframes, la.m_4x4Width, la.m_4x4Height });
    else rec({ W, H, N, (int32_t)L0->lumaStride, L0->width, L0->lines, wcu, hcu, marginX, marginY, X265_DEPTH, (int32_t)p->rc.qgSize, p->bframes });
    for (int f = 0; f < N; f++)
    {   /* the four padded lowres planes, then the intra results */
        Lowres* l = low[f];
        const size_t planesize = (size_t)l->lumaStride * (l->lines + 2 * marginY);
        for (int k = 0; k < 4; k++)
        {
            const pixel* base = l->lowresPlane[k] - ((intptr_t)l->lumaStride * marginY + marginX);
            std::vector<int32_t> v(planesize);
            for (size_t i = 0; i < planesize; i++) v[i] = base[i];
            rec(v);
        }
        if (hme)
            for (int k = 0; k < 4; k++)
            {   /* lowres.cpp:171-188: planesize / 2 pixels per plane, rows lumaStride / 2 apart, pixel (0,0) at padoffset / 2 */
                std::vector<int32_t> v(planesize / 2);
                for (size_t i = 0; i < planesize / 2; i++) v[i] = l->lowerResBuffer[k][i];
                rec(v);
            }
        std::vector<int32_t> ic(l->intraCost, l->intraCost + ncu), im(ncu), rs(hcu), lc(ncu), q(ncu);
        for (int i = 0; i < ncu; i++) { im[i] = l->intraMode[i]; lc[i] = l->lowresCosts[0][0][i]; }
        for (int i = 0; i < hcu; i++) rs[i] = l->rowSatds[0][0][i];
        for (int i = 0; i < ncu; i++) q[i] = !l->invQscaleFactor ? -1 /* no AQ array: costs are not scaled */ : (p->rc.qgSize == 8 ? l->invQscaleFactor8x8[i] : l->invQscaleFactor[i]);
        rec(ic); rec(im); rec(lc); rec(rs); rec(q);
    }
    for (int a = 7; a < argc; a++)
    {
        int p0, b, p1, keep = 0, referenced = 0, seed = 0, sliceRows = 0;
        const bool prop = !strncmp(argv[a], "prop:", 5);
        if (prop) { if (sscanf(argv[a] + 5, "%d,%d,%d,%d,%d", &p0, &b, &p1, &referenced, &seed) < 5) { fprintf(stderr, "bad prop %s\n", argv[a]); return 2; } }
        else if (sscanf(argv[a], "%d,%d,%d,%d,%d", &p0, &b, &p1, &keep, &sliceRows) < 3) { fprintf(stderr, "bad triple %s\n", argv[a]); return 2; }
        Lowres* fenc = low[b];
        if (!keep) resetCaches(*fenc, p->bframes);
        const int doSearch0 = fenc->lowresMvs[0][b - p0][0].x == 0x7FFF, doSearch1 = p1 > b && fenc->lowresMvs[1][p1 - b][0].x == 0x7FFF;
        Group g(la, low.data());
        const auto t0 = std::chrono::steady_clock::now();
        const int64_t score = sliceRows > 0 ? g.slicedCost(tld, p0, p1, b, sliceRows) : g.singleCost(p0, p1, b, false);
        nsCost += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
        rec({ p0, b, p1, keep, doSearch0, doSearch1, (int32_t)score, (int32_t)fenc->costEst[b - p0][p1 - b], (int32_t)fenc->costEstAq[b - p0][p1 - b],
              fenc->intraMbs[b - p0] });
        for (int l = 0; l < 2; l++)
        {
            const int d = l ? p1 - b : b - p0;
            std::vector<int32_t> mv(2 * (size_t)ncu, 0), mc(ncu, 0);
            if (l == 0 || p1 > b)
                for (int i = 0; i < ncu; i++) { mv[2 * i] = fenc->lowresMvs[l][d][i].x; mv[2 * i + 1] = fenc->lowresMvs[l][d][i].y; mc[i] = fenc->lowresMvCosts[l][d][i]; }
            rec(mv); rec(mc);
        }
        std::vector<int32_t> lc(ncu), rs(hcu);
        for (int i = 0; i < ncu; i++) lc[i] = fenc->lowresCosts[b - p0][p1 - b][i];
        for (int i = 0; i < hcu; i++) rs[i] = fenc->rowSatds[b - p0][p1 - b][i];
        rec(lc); rec(rs);
        if (hme)
            for (int l = 0; l < 2; l++)
            {
                const int d = l ? p1 - b : b - p0;
                std::vector<int32_t> mv(2 * (size_t)ncu4, 0), mc(ncu4, 0);
                if (l == 0 || p1 > b)
                    for (int i = 0; i < ncu4; i++) { mv[2 * i] = fenc->lowerResMvs[l][d][i].x; mv[2 * i + 1] = fenc->lowerResMvs[l][d][i].y; mc[i] = fenc->lowerResMvCosts[l][d][i]; }
                rec(mv); rec(mc);
            }
        if (weightp)
        {   /* did weightsAnalyse (slicetype.cpp:919-1020) weight the list-0 reference, and with which planes */
            ReferencePlanes& wr = fenc->weightedRef[b - p0];
            const int isW = wr.isWeighted ? 1 : 0;
            rec({ isW });
            if (isW)
            {
                const size_t planesize = (size_t)fenc->lumaStride * (fenc->lines + 2 * marginY);
                for (int k = 0; k < 4; k++)
                {
                    const pixel* base = wr.lowresPlane[k] - ((intptr_t)fenc->lumaStride * marginY + marginX);
                    std::vector<int32_t> v(planesize);
                    for (size_t i = 0; i < planesize; i++) v[i] = base[i];
                    rec(v);
                }
            }
        }
        if (prop)
        {
            uint32_t st = 2463534242u + 7919u * (uint32_t)seed;
            Lowres* three[3] = { fenc, low[p0], low[p1] };
            std::vector<int32_t> before[3], after[3];
            for (int k = 0; k < 3; k++)
            {
                if (k == 2 && p1 == b) { before[k] = before[0]; continue; }
                if (k == 1 && p0 == b) { before[k] = before[0]; continue; }
                for (int i = 0; i < ncu; i++)
                {   /* xorshift32; mostly moderate values, some close to saturation */
                    st ^= st << 13; st ^= st >> 17; st ^= st << 5;
                    three[k]->propagateCost[i] = (uint16_t)((st & 15) == 0 ? 65000 + (st >> 8) % 536 : (st >> 8) % 6000);
                }
                before[k].assign(three[k]->propagateCost, three[k]->propagateCost + ncu);
            }
            la.estimateCUPropagate(low.data(), 0.05, p0, p1, b, referenced);
            for (int k = 0; k < 3; k++) after[k].assign(three[k]->propagateCost, three[k]->propagateCost + ncu);
            const double fpsFactor = CLIP_DURATION((double)p->fpsDenom / p->fpsNum) / CLIP_DURATION(0.05);
            int32_t bits[2]; memcpy(bits, &fpsFactor, 8);
            rec({ referenced, seed, bits[0], bits[1], p->bEnableWeightedBiPred });
            for (int k = 0; k < 3; k++) { rec(before[k]); rec(after[k]); }
            /* ... then Lookahead::cuTreeFinish on picture b (slicetype.cpp:4098-4150, the qgSize != 8, non-hevc-aq branch): qp offsets from the propagated costs.
               Recorded as int32 pairs of the doubles: { strength, weightedCostDelta used, ref0Distance }, qpAqOffset (input), qpCuTreeOffset (output). */
            const int ref0Distance = b == p1 ? b - p0 : 0;
            if (ref0Distance) fenc->weightedCostDelta[ref0Distance - 1] = (seed & 1) ? 0.25 + 0.01 * (seed % 7) : 0.0;     /* both branches of weightdelta */
            la.cuTreeFinish(fenc, 0.05, ref0Distance);
            auto dbl = [&](const double* v, int n) { std::vector<int32_t> o(2 * n); memcpy(o.data(), v, 8 * (size_t)n); rec(o); };
            const double hdr3[3] = { la.m_cuTreeStrength, ref0Distance ? fenc->weightedCostDelta[ref0Distance - 1] : 0.0, (double)ref0Distance };
            dbl(hdr3, 3); dbl(fenc->qpAqOffset, ncu); dbl(fenc->qpCuTreeOffset, ncu);
        }
    }
    rec({ (int32_t)(nsIntra & 0xffffffff), (int32_t)(nsIntra >> 32), (int32_t)(nsCost & 0xffffffff), (int32_t)(nsCost >> 32) });
    fclose(g_out); fclose(in);
    return 0;
}
