/*
 * lookahead_adapter.cpp -- LookaheadTLD::lowresIntraEstimate and CostEstimateGroup::estimateFrameCost with the GPU as the lookahead's cost producer
 * (see lookahead_adapter.h).
 *
 * Per call the adapter hands over what the encoder's state holds for the pictures involved -- the Lowres planes (buffer[0]: four half-pel planes back to back), the AQ
 * factors, for a list that was searched before its MVs and costs (the encoder's own bDoSearch caching) -- and writes the results where the encoder's body writes them.
 * LookaheadTLD::weightsAnalyse stays host code (it is a handful of SATD sums); when it weights the list-0 reference its weighted copy (wbuffer) goes along.
 *
 * Two bindings of the frame-cost estimate: CostEstimateGroup::estimateFrameCost (one estimate; concurrent callers are merged by the producer) and
 * CostEstimateGroup::finishBatch (slicetype.cpp:4271-4278) -- the whole queue of (p0, b, p1) triples slicetypeAnalyse / slicetypePath / cuTree built with add() goes up as
 * x265hip_la_estimate_batch calls: the batch kernel at its best (a launch costs little more for 32 estimates than for one).
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
#include "ratecontrol.h"                     /* CLIP_DURATION */
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
    int (*la_estimate_batch)(x265hip_la*, const x265hip_la_estimate_desc*, int);
    int (*la_batch_stats)(const x265hip_la*, int64_t*, int64_t*);
    int (*la_cutree_propagate)(x265hip_la*, const x265hip_la_cutree_desc*);
    const char* (*last_error)();
} g_api;
void* g_lib;
x265hip_ctx* g_ctx;
x265hip_la* g_la;
int g_on, g_device, g_batchBinding = 1;      /* X265LA_BATCH=0: estimates one call at a time (the round-3 binding), for A/B */
int g_hme;                               /* 0 not tried yet, 1 the producer holds the quarter-resolution pictures, -1 it cannot: --hme estimates stay with the encoder */
std::mutex g_lock;                       /* creation of the producer; the producer serialises its own calls */
x265hip_la_adapter_stats g_stats;
std::mutex g_statLock;

/* a producer call failed: the encode must not go on quietly.  The encoder's pool threads are running -- exit() would run the static destructors under them (a hang, seen in
   the ThreadedME binding): leave at once, the message is on stderr */
[[noreturn]] void die() { fflush(stdout); fflush(stderr); _Exit(3); }
double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
uint64_t key_of(const Lowres& f) { return (uint64_t)(int64_t)f.frameNum + 2; }      /* frameNum starts at 0; 0 is "no key" */

/* one producer per encoder: the geometry of the first picture seen (all Lowres of an encoder share it) */
x265hip_la* producer(const Lowres& f, int widthInCU, int heightInCU)
{
    std::lock_guard<std::mutex> guard(g_lock);
    if (g_la) return g_la;
    const int64_t planeElems = f.buffer[1] - f.buffer[0], origin = f.lowresPlane[0] - f.buffer[0];
    /* pictures kept on the device: the lookahead's depth and then some; the producer addresses its lowres buffer with 32-bit element offsets (X265HIP_LA_MAX_BATCH more places hold the
       weighted copies of a launch), which bounds the count for very large pictures -- fewer places only mean more uploads */
    int64_t keep = (((int64_t)1 << 31) - 1) / (4 * planeElems) - (X265HIP_LA_MAX_BATCH + 1);          /* (the quarter-resolution pictures of --hme are a quarter of that again: never the bound) */
    if (keep > 96) keep = 96;
    if (keep < 8 || g_api.ctx_create(g_device, &g_ctx) || g_api.la_create(g_ctx, widthInCU, heightInCU, f.lumaStride, planeElems, origin, (int)keep, &g_la))
    {
        fprintf(stderr, "lookahead_adapter: x265hip_la_create: %s -- the encoder's own lookahead runs\n", g_api.last_error());
        g_on = 0; g_la = nullptr;
    }
    return g_la;
}
/* --hme: the quarter-resolution level on the producer too (Lookahead::m_4x4Width x m_4x4Height blocks of Lowres::lowerResPlane, rows lumaStride / 2 apart, lowres.cpp:170-188).
   Diamond, hexagon, uneven multi-hexagon, star and exhaustive levels are offered (the default is --hme-search hex,umh,umh).  A sea level keeps the encoder's own estimate:
   the lookahead's MotionEstimate has no integral planes (MotionEstimate::integral[] stays NULL, motion.cpp:115, 1509-1541), the reference itself cannot run it. */
bool hme_ready(x265hip_la* la, const Lowres& f, const x265_param& p, int w4, int h4)
{
    for (int l = 0; l < 2; l++)
        if ((p.hmeSearchMethod[l] != X265_DIA_SEARCH && p.hmeSearchMethod[l] != X265_HEX_SEARCH && p.hmeSearchMethod[l] != X265_UMH_SEARCH && p.hmeSearchMethod[l] != X265_STAR_SEARCH && p.hmeSearchMethod[l] != X265_FULL_SEARCH) ||
            p.hmeRange[l] < 1 || p.hmeRange[l] > 64) return false;
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
void finishBatch_cpu(CostEstimateGroup* self) __asm__("xla_finishBatch_cpu");
void lowresIntraEstimate_cpu(LookaheadTLD* self, Lowres& fenc, uint32_t qgSize) __asm__("xla_lowresIntraEstimate_cpu");
void estimateCUPropagate_cpu(Lookahead* self, Lowres** frames, double averageDuration, int p0, int p1, int b, int referenced) __asm__("xla_estimateCUPropagate_cpu");

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
    if (rc) { fprintf(stderr, "lookahead_adapter: x265hip_la_intra (frame %d): %d %s\n", fenc.frameNum, rc, g_api.last_error()); die(); }
    fenc.costEst[0][0] = sums[0];
    fenc.costEstAq[0][0] = sums[1];
    std::lock_guard<std::mutex> guard(g_statLock);
    g_stats.intraPictures++; g_stats.intraSeconds += now() - t0; g_stats.producerSeconds += t2 - t1;
}

/* One estimate between the encoder's state and the producer: prepare() is the prologue of estimateFrameCost (slicetype.cpp:4376-4390) + the hand-over of what the estimate
   reads; finish() writes the results where the encoder's body writes them (:4440-4457).  Local to the two member functions below (they reach the group's private state). */
struct XlaEstimate
{
    int p0, p1, b, nl, ncu, ncu4; bool doSearch[2], hme;
    x265hip_la_estimate_desc d;
    std::vector<int16_t> mv[2], mv4[2]; std::vector<pixel> weighted;
    int64_t sums[3];
};

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
    /* prepare / finish as local lambdas: they touch private members of the group and of the lookahead */
    auto prepare = [&](XlaEstimate& e, LookaheadTLD& t, bool keepWeighted)
    {
        Lowres* f = m_frames[e.b];
        e.ncu = m_lookahead.m_8x8Width * m_lookahead.m_8x8Height; e.ncu4 = m_lookahead.m_4x4Width * m_lookahead.m_4x4Height; e.hme = hme;
        e.doSearch[0] = f->lowresMvs[0][e.b - e.p0][0].x == 0x7FFF;
        e.doSearch[1] = e.p1 > e.b && f->lowresMvs[1][e.p1 - e.b][0].x == 0x7FFF;
        f->weightedRef[e.b - e.p0].isWeighted = false;
        if (param->bEnableWeightedPred && e.doSearch[0])
            t.weightsAnalyse(*m_frames[e.b], *m_frames[e.p0]);
        x265hip_la_estimate_desc& d = e.d;
        memset(&d, 0, sizeof(d));
        Lowres* fr[3] = { m_frames[e.p0], f, m_frames[e.p1] };
        for (int k = 0; k < 3; k++) { d.key[k] = key_of(*fr[k]); d.planes[k] = fr[k]->buffer[0]; }
        d.invQscale = f->invQscaleFactor ? (param->rc.qgSize == 8 ? f->invQscaleFactor8x8 : f->invQscaleFactor) : nullptr;
        d.intraCost = f->intraCost;
        if (f->weightedRef[e.b - e.p0].isWeighted)
        {
            if (keepWeighted)
            {   /* the thread's wbuffer is rewritten by the next weightsAnalyse of the batch: this estimate keeps its own copy of the four planes */
                const size_t n = 4 * (size_t)(f->buffer[1] - f->buffer[0]);
                e.weighted.assign(t.wbuffer[0], t.wbuffer[0] + n);
                d.weightedPlanes = e.weighted.data();
            }
            else d.weightedPlanes = t.wbuffer[0];
        }
        d.doSearch[0] = e.doSearch[0]; d.doSearch[1] = e.doSearch[1];
        /* the cooperative sweep (a slice of block rows per worker, slicetype.cpp:4394-4426) gives other MV predictors at the slice borders than the serial one: same rule here */
        const bool coop = !m_batchMode && m_lookahead.m_numCoopSlices > 1 && ((e.p1 > e.b) || e.doSearch[0] || e.doSearch[1]);
        d.rowsPerSlice = coop ? m_lookahead.m_numRowsPerSlice : 0;
        /* MVs travel as int16 pairs; the encoder keeps int32 pairs (MV) */
        e.nl = e.p1 > e.b ? 2 : 1;
        const int dist[2] = { e.b - e.p0, e.p1 - e.b };
        for (int l = 0; l < e.nl; l++)
        {
            e.mv[l].resize((size_t)e.ncu * 2);
            if (!e.doSearch[l]) { const MV* src = f->lowresMvs[l][dist[l]]; for (int i = 0; i < e.ncu; i++) { e.mv[l][2 * i] = (int16_t)src[i].x; e.mv[l][2 * i + 1] = (int16_t)src[i].y; } }
            d.mvs[l] = e.mv[l].data(); d.mvCosts[l] = f->lowresMvCosts[l][dist[l]];
        }
        if (hme)
        {
            d.hme = 1;
            for (int k = 0; k < 3; k++) d.lowerPlanes[k] = fr[k]->lowerResBuffer[0];
            for (int l = 0; l < 2; l++) { d.hmeMethod[l] = param->hmeSearchMethod[l]; d.hmeRange[l] = param->hmeRange[l]; }
            for (int l = 0; l < e.nl; l++)
                if (e.doSearch[l]) { e.mv4[l].resize((size_t)e.ncu4 * 2); d.lowerMvs[l] = e.mv4[l].data(); d.lowerMvCosts[l] = f->lowerResMvCosts[l][dist[l]]; }
        }
        e.sums[0] = e.sums[1] = e.sums[2] = 0;
        d.lowresCosts = f->lowresCosts[e.b - e.p0][e.p1 - e.b]; d.rowSatds = f->rowSatds[e.b - e.p0][e.p1 - e.b]; d.sums = e.sums;
    };
    auto finish = [&](XlaEstimate& e) -> int64_t
    {
        Lowres* f = m_frames[e.b];
        const int dist[2] = { e.b - e.p0, e.p1 - e.b };
        for (int l = 0; l < e.nl; l++)
            if (e.doSearch[l])
            {
                MV* dst = f->lowresMvs[l][dist[l]]; for (int i = 0; i < e.ncu; i++) { dst[i].x = e.mv[l][2 * i]; dst[i].y = e.mv[l][2 * i + 1]; }
                if (e.hme) { MV* d4 = f->lowerResMvs[l][dist[l]]; for (int i = 0; i < e.ncu4; i++) { d4[i].x = e.mv4[l][2 * i]; d4[i].y = e.mv4[l][2 * i + 1]; } }
            }
        f->costEstAq[e.b - e.p0][e.p1 - e.b] = e.sums[1];
        if (e.p1 == e.b) f->intraMbs[e.b - e.p0] += (int)e.sums[2];
        int64_t sc = e.sums[0];
        if (e.b != e.p1)
            sc = sc * 100 / (130 + param->bFrameBias);
        f->costEst[e.b - e.p0][e.p1 - e.b] = sc;
        return sc;
    };
    if (p0 == -1)
    {   /* finishBatch (below) borrows the two helpers: the queue of the group, in waves */
        const double t0 = now();
        std::vector<XlaEstimate> wave; wave.reserve((size_t)m_jobTotal);
        std::vector<char> done((size_t)m_jobTotal, 0);
        int left = m_jobTotal, launches = 0;
        while (left > 0)
        {   /* a wave: every pending estimate whose list searches are cached or not claimed by an earlier estimate of the same wave (the reference's workers would race for
               a shared search and both find the same MVs; here the later estimate waits for the next wave and reuses them, the way the serial encoder does) */
            wave.clear();
            std::vector<int> claimed;          /* (b, list, distance) keys of this wave's searches */
            std::vector<int> members;
            for (int i = 0; i < m_jobTotal && (int)members.size() < X265HIP_LA_MAX_BATCH; i++)      /* a launch takes X265HIP_LA_MAX_BATCH estimates: a wave holds no more (bounds the weighted-plane copies) */
            {
                if (done[(size_t)i]) continue;
                const Estimate& q = m_estimates[i];
                Lowres* f = m_frames[q.b];
                if (f->costEst[q.b - q.p0][q.p1 - q.b] >= 0 && f->rowSatds[q.b - q.p0][q.p1 - q.b][0] != -1) { done[(size_t)i] = 1; left--; continue; }      /* estimated before (:4372) */
                bool dup = false;
                for (int m : members) dup |= m_estimates[m].p0 == q.p0 && m_estimates[m].p1 == q.p1 && m_estimates[m].b == q.b;
                const bool s0 = f->lowresMvs[0][q.b - q.p0][0].x == 0x7FFF, s1 = q.p1 > q.b && f->lowresMvs[1][q.p1 - q.b][0].x == 0x7FFF;
                const int k0 = (q.b << 12) | ((q.b - q.p0) << 1), k1 = (q.b << 12) | ((q.p1 - q.b) << 1) | 1;
                bool wait = dup;
                for (int k : claimed) wait |= (s0 && k == k0) || (s1 && k == k1);
                if (wait) continue;
                if (s0) claimed.push_back(k0);
                if (s1) claimed.push_back(k1);
                members.push_back(i);
            }
            if (members.empty()) break;
            std::vector<x265hip_la_estimate_desc> descs;
            for (int i : members)
            {
                wave.emplace_back();
                XlaEstimate& e = wave.back();
                e.p0 = m_estimates[i].p0; e.p1 = m_estimates[i].p1; e.b = m_estimates[i].b;
            }
            for (XlaEstimate& e : wave) prepare(e, tld, true);
            for (XlaEstimate& e : wave) { e.d.sums = e.sums; descs.push_back(e.d); }
            const double t1 = now();
            const int rc = g_api.la_estimate_batch(la, descs.data(), (int)descs.size());
            const double t2 = now();
            if (rc) { fprintf(stderr, "lookahead_adapter: x265hip_la_estimate_batch (%d estimates): %d %s\n", (int)descs.size(), rc, g_api.last_error()); die(); }
            for (XlaEstimate& e : wave) (void)finish(e);
            for (int i : members) { done[(size_t)i] = 1; left--; }
            launches++;
            std::lock_guard<std::mutex> guard(g_statLock);
            g_stats.estimates += (int)wave.size(); g_stats.producerSeconds += t2 - t1; g_stats.batchCalls++;
            for (XlaEstimate& e : wave) g_stats.weighted += e.d.weightedPlanes != nullptr;
        }
        std::lock_guard<std::mutex> guard(g_statLock);
        g_stats.estimateSeconds += now() - t0; g_stats.batches++;
        return 0;
    }
    int64_t score = 0;
    if (fenc->costEst[b - p0][p1 - b] >= 0 && fenc->rowSatds[b - p0][p1 - b][0] != -1)      /* estimated before (slicetype.cpp:4372-4373) */
        score = fenc->costEst[b - p0][p1 - b];
    else
    {
        const double t0 = now();
        XlaEstimate e; e.p0 = p0; e.p1 = p1; e.b = b;
        prepare(e, tld, false);
        const double t1 = now();
        const int rc = g_api.la_estimate(la, &e.d);
        const double t2 = now();
        if (rc) { fprintf(stderr, "lookahead_adapter: x265hip_la_estimate (%d, %d, %d): %d %s\n", p0, b, p1, rc, g_api.last_error()); die(); }
        score = finish(e);
        std::lock_guard<std::mutex> guard(g_statLock);
        g_stats.estimates++; g_stats.estimateSeconds += now() - t0; g_stats.producerSeconds += t2 - t1; g_stats.weighted += e.d.weightedPlanes != nullptr;
    }
    if (bIntraPenalty)
        // arbitrary penalty for I-blocks after B-frames
        score += score * fenc->intraMbs[b - p0] / (tld.ncu * 8);
    return score;
}

/* slicetype.cpp:4271-4278: the queue add() built.  Frame-cost triples go up in batches; row jobs of the temporal filter (addRow: e.frame set) and anything the producer does
   not offer keep the encoder's own body (its workers then call estimateFrameCost above one estimate at a time). */
void CostEstimateGroup::finishBatch()
{
    bool mine = g_on && g_batchBinding && m_jobTotal > 0 && m_jobAcquired == 0;
    for (int i = 0; mine && i < m_jobTotal; i++) mine = m_estimates[i].frame == NULL && m_estimates[i].blockRow == -1;
    if (mine)
    {
        Lowres* fenc = m_frames[m_estimates[0].b];
        x265hip_la* la = producer(*fenc, m_lookahead.m_8x8Width, m_lookahead.m_8x8Height);
        const x265_param* param = m_lookahead.m_param;
        if (!la || (param->bEnableHME && !hme_ready(la, *fenc, *param, m_lookahead.m_4x4Width, m_lookahead.m_4x4Height))) mine = false;
    }
    if (!mine) { ::finishBatch_cpu(this); return; }
    ThreadPool* pool = m_lookahead.m_pool;
    LookaheadTLD& tld = m_lookahead.m_tld[pool ? pool->m_numWorkers : 0];      /* the caller's own thread-local data (processTasks(-1), :4286-4288) */
    (void)estimateFrameCost(tld, -1, 0, 0, false);                                /* p0 == -1: the batch form above */
    m_jobTotal = m_jobAcquired = 0;
}

/* cuTree, one propagation step (slicetype.cpp:3850-3956): the arrays of picture b and of its two references go through x265hip_la_cutree_propagate; what the reference does
   around the loop -- the zeroed first row of an unreferenced picture's costs, cuTreeFinish under VBV -- stays here.  (cuTreeFinish itself is left to the encoder: its two
   log2 of doubles are libm's, and a device library's last bit is not guaranteed to be the same.) */
void Lookahead::estimateCUPropagate(Lowres** frames, double averageDuration, int p0, int p1, int b, int referenced)
{
    x265hip_la* la = g_on ? producer(*frames[b], m_8x8Width, m_8x8Height) : nullptr;
    if (!la || b <= p0 || p1 < b) { ::estimateCUPropagate_cpu(this, frames, averageDuration, p0, p1, b, referenced); return; }
    const double t0 = now();
    Lowres* fb = frames[b];
    const int ncu = m_8x8Width * m_8x8Height;
    static thread_local std::vector<int16_t> mv16[2];
    x265hip_la_cutree_desc d;
    memset(&d, 0, sizeof(d));
    d.distP0 = b - p0; d.distP1 = p1 - b; d.weightedBiPred = m_param->bEnableWeightedBiPred; d.referenced = referenced;
    x265_emms();
    d.fpsFactor = CLIP_DURATION((double)m_param->fpsDenom / m_param->fpsNum) / CLIP_DURATION(averageDuration);
    if (!referenced) memset(fb->propagateCost, 0, m_8x8Width * sizeof(uint16_t));                      /* :3866-3867 */
    d.intraCost = fb->intraCost; d.lowresCosts = fb->lowresCosts[b - p0][p1 - b];
    d.invQscale = m_param->rc.qgSize == 8 ? fb->invQscaleFactor8x8 : fb->invQscaleFactor;
    for (int l = 0; l < (p1 > b ? 2 : 1); l++)
    {
        const MV* m = fb->lowresMvs[l][l ? p1 - b : b - p0];
        mv16[l].resize(2 * (size_t)ncu);
        for (int i = 0; i < ncu; i++) { mv16[l][2 * i] = (int16_t)m[i].x; mv16[l][2 * i + 1] = (int16_t)m[i].y; }
    }
    d.mvs0 = mv16[0].data(); d.mvs1 = p1 > b ? mv16[1].data() : NULL;
    d.propB = fb->propagateCost; d.prop0 = frames[p0]->propagateCost; d.prop1 = p1 > b ? frames[p1]->propagateCost : NULL;
    const int rc = g_api.la_cutree_propagate(la, &d);
    if (rc) { fprintf(stderr, "lookahead_adapter: x265hip_la_cutree_propagate: %d %s\n", rc, g_api.last_error()); die(); }
    { std::lock_guard<std::mutex> sg(g_statLock); g_stats.cutreeSteps++; g_stats.cutreeSeconds += now() - t0; }
    if (m_param->rc.vbvBufferSize && m_param->lookaheadDepth && referenced)
        cuTreeFinish(fb, averageDuration, b == p1 ? b - p0 : 0);
}

}

extern "C" int x265hip_la_adapter_load(const char* libraryPath, int device)
{
    g_lib = dlopen(libraryPath, RTLD_NOW | RTLD_LOCAL);
    if (!g_lib) { fprintf(stderr, "lookahead_adapter: dlopen: %s\n", dlerror()); return -1; }
#define SYM(field, name) *(void**)&g_api.field = dlsym(g_lib, name); if (!g_api.field) { fprintf(stderr, "lookahead_adapter: %s lacks %s\n", libraryPath, name); return -1; }
    SYM(ctx_create, "x265hip_ctx_create") SYM(ctx_destroy, "x265hip_ctx_destroy") SYM(la_create, "x265hip_la_create") SYM(la_destroy, "x265hip_la_destroy") SYM(la_enable_hme, "x265hip_la_enable_hme")
    SYM(la_intra, "x265hip_la_intra") SYM(la_estimate, "x265hip_la_estimate") SYM(la_estimate_batch, "x265hip_la_estimate_batch") SYM(la_batch_stats, "x265hip_la_batch_stats") SYM(la_cutree_propagate, "x265hip_la_cutree_propagate") SYM(last_error, "x265hip_last_error")
#undef SYM
    g_device = device; g_on = 1;
    g_batchBinding = !(getenv("X265LA_BATCH") && !atoi(getenv("X265LA_BATCH")));
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
