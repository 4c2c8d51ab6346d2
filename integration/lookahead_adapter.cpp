/*
 * lookahead_adapter.cpp -- LookaheadTLD::lowresIntraEstimate and CostEstimateGroup::estimateFrameCost with the GPU as the lookahead's cost producer
 * (see lookahead_adapter.h).
 *
 * Per call the adapter hands over what the encoder's state holds for the pictures involved -- the Lowres planes (buffer[0]: four half-pel planes back to back), the AQ
 * factors, for a list that was searched before its MVs and costs (the encoder's own bDoSearch caching) -- and writes the results where the encoder's body writes them.
 * LookaheadTLD::weightsAnalyse stays host code (it is a handful of SATD sums); when it weights the list-0 reference its weighted copy (wbuffer) goes along.
 */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <vector>
#include "x265.h"
#include "common.h"
#include "primitives.h"
#include "lowres.h"
#include "mv.h"
#include "slicetype.h"
#include "../include/x265hip_ctx.h"
#include "lookahead_adapter.h"

using namespace X265_NS;

namespace {
struct Api
{
    int (*ctx_create)(int, x265hip_ctx**);
    void (*ctx_destroy)(x265hip_ctx*);
    int (*la_create)(x265hip_ctx*, int, int, intptr_t, int64_t, int64_t, int, x265hip_la**);
    void (*la_destroy)(x265hip_la*);
    int (*la_enable_hme)(x265hip_la*, int, int, intptr_t, int64_t, int64_t);
    int (*la_intra)(x265hip_la*, uint64_t, const void*, const int32_t*, int32_t*, uint8_t*, uint16_t*, int32_t*, int64_t*);
    int (*la_estimate)(x265hip_la*, const x265hip_la_estimate_desc*);
    int (*la_batch_stats)(const x265hip_la*, int64_t*, int64_t*);
    const char* (*last_error)();
} g_api;
void* g_lib;
x265hip_ctx* g_ctx;
x265hip_la* g_la;
int g_on, g_device;
int g_hme;                               /* 0 not tried yet, 1 the producer holds the quarter-resolution pictures, -1 it cannot: --hme estimates stay with the encoder */
std::mutex g_lock;                       /* creation of the producer; the producer serialises its own calls */
x265hip_la_adapter_stats g_stats;
std::mutex g_statLock;

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
uint64_t key_of(const Lowres& f) { return (uint64_t)(int64_t)f.frameNum + 2; }      /* frameNum starts at 0; 0 is "no key" */

/* one producer per encoder: the geometry of the first picture seen (all Lowres of an encoder share it) */
x265hip_la* producer(const Lowres& f, int widthInCU, int heightInCU)
{
    std::lock_guard<std::mutex> guard(g_lock);
    if (g_la) return g_la;
    const int64_t planeElems = f.buffer[1] - f.buffer[0], origin = f.lowresPlane[0] - f.buffer[0];
    /* pictures kept on the device: the lookahead's depth and then some; the producer addresses its lowres buffer with 32-bit element offsets (16 more places hold the
       weighted copies of a launch), which bounds the count for very large pictures -- fewer places only mean more uploads */
    int64_t keep = (((int64_t)1 << 31) - 1) / (4 * planeElems) - 17;          /* (the quarter-resolution pictures of --hme are a quarter of that again: never the bound) */
    if (keep > 96) keep = 96;
    if (keep < 8 || g_api.ctx_create(g_device, &g_ctx) || g_api.la_create(g_ctx, widthInCU, heightInCU, f.lumaStride, planeElems, origin, (int)keep, &g_la))
    {
        fprintf(stderr, "lookahead_adapter: x265hip_la_create: %s -- the encoder's own lookahead runs\n", g_api.last_error());
        g_on = 0; g_la = nullptr;
    }
    return g_la;
}
/* --hme: the quarter-resolution level on the producer too (Lookahead::m_4x4Width x m_4x4Height blocks of Lowres::lowerResPlane, rows lumaStride / 2 apart, lowres.cpp:170-188).
   Hexagon and uneven multi-hexagon levels (the default --hme-search hex,umh,umh) are offered; anything else keeps the encoder's own estimate. */
bool hme_ready(x265hip_la* la, const Lowres& f, const x265_param& p, int w4, int h4)
{
    for (int l = 0; l < 2; l++)
        if ((p.hmeSearchMethod[l] != X265_HEX_SEARCH && p.hmeSearchMethod[l] != X265_UMH_SEARCH) || p.hmeRange[l] < 1 || p.hmeRange[l] > 64) return false;
    std::lock_guard<std::mutex> guard(g_lock);
    if (!g_hme)
    {
        const int rc = g_api.la_enable_hme(la, w4, h4, f.lumaStride / 2, f.lowerResBuffer[1] - f.lowerResBuffer[0], f.lowerResPlane[0] - f.lowerResBuffer[0]);
        if (rc) fprintf(stderr, "lookahead_adapter: x265hip_la_enable_hme: %s -- the encoder's own estimate runs with --hme\n", g_api.last_error());
        g_hme = rc ? -1 : 1;
    }
    return g_hme > 0;
}
}

int64_t estimateFrameCost_cpu(CostEstimateGroup* self, LookaheadTLD& tld, int p0, int p1, int b, bool bIntraPenalty) __asm__("xla_estimateFrameCost_cpu");
void lowresIntraEstimate_cpu(LookaheadTLD* self, Lowres& fenc, uint32_t qgSize) __asm__("xla_lowresIntraEstimate_cpu");

namespace X265_NS {

void LookaheadTLD::lowresIntraEstimate(Lowres& fenc, uint32_t qgSize)
{
    x265hip_la* la = g_on ? producer(fenc, widthInCU, heightInCU) : nullptr;
    if (!la) { ::lowresIntraEstimate_cpu(this, fenc, qgSize); return; }
    const double t0 = now();
    /* slicetype.cpp:755-864: per block intraCost / intraMode / lowresCosts[0][0], per row rowSatds[0][0], the frame's costEst[0][0] / costEstAq[0][0].  The AQ factors
       the estimate weighs with are invQscaleFactor8x8 with qgSize 8, invQscaleFactor otherwise, and none at all when invQscaleFactor is NULL (:851-854) */
    const int32_t* invq = fenc.invQscaleFactor ? (qgSize == 8 ? fenc.invQscaleFactor8x8 : fenc.invQscaleFactor) : nullptr;
    int64_t sums[2] = { 0, 0 };
    const double t1 = now();
    const int rc = g_api.la_intra(la, key_of(fenc), fenc.buffer[0], invq, fenc.intraCost, fenc.intraMode, fenc.lowresCosts[0][0], fenc.rowSatds[0][0], sums);
    const double t2 = now();
    if (rc) { fprintf(stderr, "lookahead_adapter: x265hip_la_intra (frame %d): %d %s\n", fenc.frameNum, rc, g_api.last_error()); exit(3); }
    fenc.costEst[0][0] = sums[0];
    fenc.costEstAq[0][0] = sums[1];
    std::lock_guard<std::mutex> guard(g_statLock);
    g_stats.intraPictures++; g_stats.intraSeconds += now() - t0; g_stats.producerSeconds += t2 - t1;
}

int64_t CostEstimateGroup::estimateFrameCost(LookaheadTLD& tld, int p0, int p1, int b, bool bIntraPenalty)
{
    Lowres* fenc = m_frames[b];
    x265_param* param = m_lookahead.m_param;
    x265hip_la* la = g_on ? producer(*fenc, m_lookahead.m_8x8Width, m_lookahead.m_8x8Height) : nullptr;
    /* with --hme the cooperative sweep reads quarter-resolution results of other slices while they are being written (slicetype.cpp:4332-4358, :4532): the encoder keeps it */
    const bool hme = param->bEnableHME;
    if (la && hme && ((m_lookahead.m_numCoopSlices > 1 && !m_batchMode) || !hme_ready(la, *fenc, *param, m_lookahead.m_4x4Width, m_lookahead.m_4x4Height))) la = nullptr;
    if (!la)
    {
        if (g_on) { std::lock_guard<std::mutex> guard(g_statLock); g_stats.cpuEstimates++; }
        return ::estimateFrameCost_cpu(this, tld, p0, p1, b, bIntraPenalty);
    }
    int64_t score = 0;
    if (fenc->costEst[b - p0][p1 - b] >= 0 && fenc->rowSatds[b - p0][p1 - b][0] != -1)      /* estimated before (slicetype.cpp:4372-4373) */
        score = fenc->costEst[b - p0][p1 - b];
    else
    {
        const double t0 = now();
        const int ncu = m_lookahead.m_8x8Width * m_lookahead.m_8x8Height;
        bool bDoSearch[2];
        bDoSearch[0] = fenc->lowresMvs[0][b - p0][0].x == 0x7FFF;
        bDoSearch[1] = p1 > b && fenc->lowresMvs[1][p1 - b][0].x == 0x7FFF;
        fenc->weightedRef[b - p0].isWeighted = false;
        if (param->bEnableWeightedPred && bDoSearch[0])
            tld.weightsAnalyse(*m_frames[b], *m_frames[p0]);
        x265hip_la_estimate_desc d;
        memset(&d, 0, sizeof(d));
        Lowres* fr[3] = { m_frames[p0], fenc, m_frames[p1] };
        for (int k = 0; k < 3; k++) { d.key[k] = key_of(*fr[k]); d.planes[k] = fr[k]->buffer[0]; }
        d.invQscale = fenc->invQscaleFactor ? (param->rc.qgSize == 8 ? fenc->invQscaleFactor8x8 : fenc->invQscaleFactor) : nullptr;
        d.intraCost = fenc->intraCost;
        if (fenc->weightedRef[b - p0].isWeighted) d.weightedPlanes = tld.wbuffer[0];
        d.doSearch[0] = bDoSearch[0]; d.doSearch[1] = bDoSearch[1];
        /* the cooperative sweep (a slice of block rows per worker, slicetype.cpp:4394-4426) gives other MV predictors at the slice borders than the serial one: same rule here */
        const bool coop = !m_batchMode && m_lookahead.m_numCoopSlices > 1 && ((p1 > b) || bDoSearch[0] || bDoSearch[1]);
        d.rowsPerSlice = coop ? m_lookahead.m_numRowsPerSlice : 0;
        /* MVs travel as int16 pairs; the encoder keeps int32 pairs (MV) */
        std::vector<int16_t> mv[2]; 
        const int nl = p1 > b ? 2 : 1;
        const int dist[2] = { b - p0, p1 - b };
        for (int l = 0; l < nl; l++)
        {
            mv[l].resize((size_t)ncu * 2);
            if (!bDoSearch[l]) { const MV* src = fenc->lowresMvs[l][dist[l]]; for (int i = 0; i < ncu; i++) { mv[l][2 * i] = (int16_t)src[i].x; mv[l][2 * i + 1] = (int16_t)src[i].y; } }
            d.mvs[l] = mv[l].data(); d.mvCosts[l] = fenc->lowresMvCosts[l][dist[l]];
        }
        std::vector<int16_t> mv4[2];
        const int ncu4 = m_lookahead.m_4x4Width * m_lookahead.m_4x4Height;
        if (hme)
        {
            d.hme = 1;
            for (int k = 0; k < 3; k++) d.lowerPlanes[k] = fr[k]->lowerResBuffer[0];
            for (int l = 0; l < 2; l++) { d.hmeMethod[l] = param->hmeSearchMethod[l]; d.hmeRange[l] = param->hmeRange[l]; }
            for (int l = 0; l < nl; l++)
                if (bDoSearch[l]) { mv4[l].resize((size_t)ncu4 * 2); d.lowerMvs[l] = mv4[l].data(); d.lowerMvCosts[l] = fenc->lowerResMvCosts[l][dist[l]]; }
        }
        int64_t sums[3] = { 0, 0, 0 };
        d.lowresCosts = fenc->lowresCosts[b - p0][p1 - b]; d.rowSatds = fenc->rowSatds[b - p0][p1 - b]; d.sums = sums;
        const double t1 = now();
        const int rc = g_api.la_estimate(la, &d);
        const double t2 = now();
        if (rc) { fprintf(stderr, "lookahead_adapter: x265hip_la_estimate (%d, %d, %d): %d %s\n", p0, b, p1, rc, g_api.last_error()); exit(3); }
        for (int l = 0; l < nl; l++)
            if (bDoSearch[l])
            {
                MV* dst = fenc->lowresMvs[l][dist[l]]; for (int i = 0; i < ncu; i++) { dst[i].x = mv[l][2 * i]; dst[i].y = mv[l][2 * i + 1]; }
                if (hme) { MV* d4 = fenc->lowerResMvs[l][dist[l]]; for (int i = 0; i < ncu4; i++) { d4[i].x = mv4[l][2 * i]; d4[i].y = mv4[l][2 * i + 1]; } }
            }
        fenc->costEstAq[b - p0][p1 - b] = sums[1];
        if (p1 == b) fenc->intraMbs[b - p0] += (int)sums[2];
        score = sums[0];
        if (b != p1)
            score = score * 100 / (130 + param->bFrameBias);
        fenc->costEst[b - p0][p1 - b] = score;
        std::lock_guard<std::mutex> guard(g_statLock);
        g_stats.estimates++; g_stats.estimateSeconds += now() - t0; g_stats.producerSeconds += t2 - t1; g_stats.weighted += d.weightedPlanes != nullptr;
    }
    if (bIntraPenalty)
        // arbitrary penalty for I-blocks after B-frames
        score += score * fenc->intraMbs[b - p0] / (tld.ncu * 8);
    return score;
}

}

extern "C" int x265hip_la_adapter_load(const char* libraryPath, int device)
{
    g_lib = dlopen(libraryPath, RTLD_NOW | RTLD_LOCAL);
    if (!g_lib) { fprintf(stderr, "lookahead_adapter: dlopen: %s\n", dlerror()); return -1; }
#define SYM(field, name) *(void**)&g_api.field = dlsym(g_lib, name); if (!g_api.field) { fprintf(stderr, "lookahead_adapter: %s lacks %s\n", libraryPath, name); return -1; }
    SYM(ctx_create, "x265hip_ctx_create") SYM(ctx_destroy, "x265hip_ctx_destroy") SYM(la_create, "x265hip_la_create") SYM(la_destroy, "x265hip_la_destroy") SYM(la_enable_hme, "x265hip_la_enable_hme")
    SYM(la_intra, "x265hip_la_intra") SYM(la_estimate, "x265hip_la_estimate") SYM(la_batch_stats, "x265hip_la_batch_stats") SYM(last_error, "x265hip_last_error")
#undef SYM
    g_device = device; g_on = 1;
    return 0;
}
extern "C" void x265hip_la_adapter_enable(int on) { g_on = on && g_lib; }
extern "C" void x265hip_la_adapter_close(void)
{
    std::lock_guard<std::mutex> guard(g_lock);
    if (g_la)
    {
        int64_t l = 0, e = 0;
        g_api.la_batch_stats(g_la, &l, &e);
        { std::lock_guard<std::mutex> sg(g_statLock); g_stats.launches = (int)l; }
        g_api.la_destroy(g_la); g_la = nullptr;
    }
    if (g_ctx) { g_api.ctx_destroy(g_ctx); g_ctx = nullptr; }
    g_on = 0; g_hme = 0;
}
extern "C" void x265hip_la_adapter_get_stats(x265hip_la_adapter_stats* o) { std::lock_guard<std::mutex> guard(g_statLock); *o = g_stats; }
