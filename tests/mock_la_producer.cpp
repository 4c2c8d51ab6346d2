/*
 * mock_la_producer.cpp -- TEST INFRASTRUCTURE ONLY (tests/test_la_adapter_cpu.py builds it with g++): the lookahead producer's entry points (include/x265hip_ctx.h:
 * x265hip_la_create / _intra / _estimate / _estimate_batch / _cutree_propagate ...) answered WITHOUT a GPU by the oracle's plain-C restatement of the lookahead
 * (oracle/x265_oracle_la.c in oracle/libx265oracle_me_8.so, named by X265MOCK_ORACLE_LIB), so that the host half of the seam -- integration/lookahead_adapter.cpp: what it reads
 * out of the encoder's Lowres state, the waves it cuts a finishBatch queue into, the weighted copies, the cached list searches, the cuTree step -- can be driven by the compiled
 * reference encoder (oracle/_ref/x265e2e_8) on the CPU.  The oracle is pinned to the reference's own lookahead (tests/test_lookahead_oracle_vs_ref.py), so an encode whose
 * lookahead costs come through this mock must write the bitstream of the plain encoder.
 * X265MOCK_FAIL_AT=n: the n-th estimate call fails (the binding must end the encode at once, loudly).
 * It also checks what the header promises on every call: plane pointers there, keys consistent, an estimate of one batch call never reusing a list search another estimate of
 * the SAME call makes (include/x265hip_ctx.h, x265hip_la_estimate_batch).
 */
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <set>
#include <tuple>
#include <vector>
#include "../include/x265hip_ctx.h"

#if defined(MOCK_DEPTH) && MOCK_DEPTH > 8
typedef uint16_t xo_pixel;     /* -DMOCK_DEPTH=10: the 10-bit encoder with the 10-bit oracle */
#else
typedef uint8_t xo_pixel;      /* the 8-bit encoder */
#endif
struct xo_la_hme { const xo_pixel* fenc; const xo_pixel* const* ref0; const xo_pixel* const* ref1; intptr_t stride; int wcu, hcu; int method[2], range[2]; int32_t* mvs[2]; int32_t* mvCosts[2]; };

struct x265hip_ctx { int device; };
struct x265hip_la
{
    int wcu, hcu; intptr_t stride; int64_t planeElems, origin;
    int wcu4 = 0, hcu4 = 0; intptr_t stride4 = 0; int64_t planeElems4 = 0, origin4 = 0;
    std::vector<uint16_t> row; int half;
    int64_t launches = 0, estimates = 0;
    std::mutex lock;
};

namespace {
char g_err[512] = "";
int fail(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    fprintf(stderr, "mock_la_producer: PROTOCOL VIOLATION: %s\n", g_err);
    return X265HIP_EARG;
}
struct Oracle
{
    int (*lookahead_qp)();
    void (*mvcost_row)(int, int, uint16_t*);
    void (*intra)(const xo_pixel*, intptr_t, int, int, const int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, int64_t*);
    void (*frame_cost_hme)(const xo_pixel*, const xo_pixel* const*, const xo_pixel* const*, const xo_pixel* const*, intptr_t, int, int, const int32_t*, const int32_t*, const uint16_t*,
                           int, int, int, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, int32_t*, int64_t*, const xo_la_hme*);
    void (*propagate)(int, int, int, int, int, double, int, const int32_t*, const uint16_t*, const int32_t*, const int32_t*, const int32_t*, uint16_t*, uint16_t*, uint16_t*);
} g_o;
bool load_oracle()
{
    static bool tried = false, ok = false;
    if (tried) return ok;
    tried = true;
    const char* path = getenv("X265MOCK_ORACLE_LIB");
    void* lib = path ? dlopen(path, RTLD_NOW | RTLD_LOCAL) : nullptr;
    if (!lib) { fail("X265MOCK_ORACLE_LIB (%s) does not load: %s", path ? path : "unset", dlerror()); return false; }
    *(void**)&g_o.lookahead_qp = dlsym(lib, "xo_lookahead_qp"); *(void**)&g_o.mvcost_row = dlsym(lib, "xo_mvcost_row"); *(void**)&g_o.intra = dlsym(lib, "xo_lowres_intra_estimate");
    *(void**)&g_o.frame_cost_hme = dlsym(lib, "xo_lowres_frame_cost_hme"); *(void**)&g_o.propagate = dlsym(lib, "xo_estimate_cu_propagate");
    ok = g_o.lookahead_qp && g_o.mvcost_row && g_o.intra && g_o.frame_cost_hme && g_o.propagate;
    if (!ok) fail("%s lacks the oracle's lookahead functions", path);
    return ok;
}

int one_estimate(x265hip_la* la, const x265hip_la_estimate_desc* d)
{
    const int ncu = la->wcu * la->hcu;
    const bool isP = d->key[2] == d->key[1];
    if (!d->key[0] || !d->key[1] || !d->key[2]) return fail("estimate: a picture without a key");
    if (!d->planes[0] || !d->planes[1] || (!isP && !d->planes[2])) return fail("estimate: planes missing (the mock keeps no pictures: the binding always hands them over)");
    if (!d->intraCost || !d->lowresCosts || !d->rowSatds || !d->sums || !d->mvs[0] || !d->mvCosts[0] || (!isP && (!d->mvs[1] || !d->mvCosts[1]))) return fail("estimate: output arrays missing");
    if (isP && d->doSearch[1]) return fail("estimate: a P estimate that searches list 1");
    const xo_pixel* pl[3] = { (const xo_pixel*)d->planes[0], (const xo_pixel*)d->planes[1], (const xo_pixel*)d->planes[2] };
    const xo_pixel *r0[4], *r1[4], *rw[4];
    for (int k = 0; k < 4; k++)
    {
        r0[k] = pl[0] + k * la->planeElems + la->origin;
        r1[k] = isP ? nullptr : pl[2] + k * la->planeElems + la->origin;
        rw[k] = d->weightedPlanes ? (const xo_pixel*)d->weightedPlanes + k * la->planeElems + la->origin : nullptr;
    }
    /* the oracle keeps MVs as int32 pairs and the packed costs widened to int32 */
    std::vector<int32_t> mv[2], lc((size_t)ncu, 0), mv4[2], mc4[2];
    for (int l = 0; l < 2; l++)
    {
        mv[l].assign((size_t)ncu * 2, 0);
        if (l < (isP ? 1 : 2) && !d->doSearch[l]) for (int i = 0; i < 2 * ncu; i++) mv[l][(size_t)i] = d->mvs[l][i];
    }
    std::vector<int32_t> dummyCost((size_t)ncu, 0);
    xo_la_hme h; const xo_la_hme* hp = nullptr;
    const xo_pixel *q0[4], *q1[4];
    if (d->hme)
    {
        if (!la->wcu4) return fail("estimate: desc.hme without x265hip_la_enable_hme");
        if (!d->lowerPlanes[0] || !d->lowerPlanes[1] || (!isP && !d->lowerPlanes[2])) return fail("estimate: quarter-resolution planes missing");
        const int ncu4 = la->wcu4 * la->hcu4;
        for (int k = 0; k < 4; k++) { q0[k] = (const xo_pixel*)d->lowerPlanes[0] + k * la->planeElems4 + la->origin4; q1[k] = isP ? nullptr : (const xo_pixel*)d->lowerPlanes[2] + k * la->planeElems4 + la->origin4; }
        h.fenc = (const xo_pixel*)d->lowerPlanes[1] + la->origin4; h.ref0 = q0; h.ref1 = isP ? nullptr : q1; h.stride = la->stride4; h.wcu = la->wcu4; h.hcu = la->hcu4;
        for (int l = 0; l < 2; l++) { h.method[l] = d->hmeMethod[l]; h.range[l] = d->hmeRange[l]; mv4[l].assign((size_t)ncu4 * 2, 0); mc4[l].assign((size_t)ncu4, 0); h.mvs[l] = mv4[l].data(); h.mvCosts[l] = mc4[l].data(); }
        hp = &h;
    }
    g_o.frame_cost_hme(pl[1] + la->origin, r0, isP ? nullptr : r1, d->weightedPlanes ? rw : nullptr, la->stride, la->wcu, la->hcu, d->intraCost, d->invQscale, la->row.data() + la->half,
                       d->doSearch[0], d->doSearch[1], d->rowsPerSlice, mv[0].data(), d->mvCosts[0], mv[1].data(), isP ? dummyCost.data() : d->mvCosts[1], lc.data(), d->rowSatds, d->sums, hp);
    for (int l = 0; l < (isP ? 1 : 2); l++)
        if (d->doSearch[l])
        {
            for (int i = 0; i < 2 * ncu; i++) d->mvs[l][i] = (int16_t)mv[l][(size_t)i];
            if (d->hme)
            {
                const int ncu4 = la->wcu4 * la->hcu4;
                if (d->lowerMvs[l]) for (int i = 0; i < 2 * ncu4; i++) d->lowerMvs[l][i] = (int16_t)mv4[l][(size_t)i];
                if (d->lowerMvCosts[l]) memcpy(d->lowerMvCosts[l], mc4[l].data(), (size_t)ncu4 * sizeof(int32_t));
            }
        }
    for (int i = 0; i < ncu; i++) d->lowresCosts[i] = (uint16_t)lc[(size_t)i];
    { std::lock_guard<std::mutex> g(la->lock); la->estimates++; }
    return X265HIP_OK;
}
} // namespace

extern "C" {
#ifndef MOCK_NO_COMMON      /* (the three mocks in one library -- tests/test_all_adapters_cpu.py -- keep one copy of the context functions: the ThreadedME mock's) */
const char* x265hip_last_error(void) { return g_err; }
int x265hip_ctx_create(int device, x265hip_ctx** out) { *out = new x265hip_ctx{ device }; return X265HIP_OK; }
void x265hip_ctx_destroy(x265hip_ctx* c) { delete c; }
#endif

int x265hip_la_create(x265hip_ctx* ctx, int widthInCU, int heightInCU, intptr_t stride, int64_t planeElems, int64_t origin, int maxPictures, x265hip_la** out)
{
    if (!ctx || !out || widthInCU < 1 || heightInCU < 1 || maxPictures < 8) return fail("la_create: bad arguments");
    if (!load_oracle()) return X265HIP_EARG;
    x265hip_la* la = new x265hip_la();
    la->wcu = widthInCU; la->hcu = heightInCU; la->stride = stride; la->planeElems = planeElems; la->origin = origin;
    la->half = 1 << 13; la->row.resize(2 * (size_t)la->half + 1);
    g_o.mvcost_row(g_o.lookahead_qp(), la->half, la->row.data());
    *out = la;
    return X265HIP_OK;
}
void x265hip_la_destroy(x265hip_la* la)
{
    if (!la) return;
    fprintf(stderr, "mock_la_producer: %lld estimates in %lld calls\n", (long long)la->estimates, (long long)la->launches);
    delete la;
}
int x265hip_la_enable_hme(x265hip_la* la, int w4, int h4, intptr_t stride4, int64_t planeElems4, int64_t origin4)
{
    if (!la || w4 < 1 || h4 < 1) return fail("la_enable_hme: bad arguments");
    std::lock_guard<std::mutex> g(la->lock);
    la->wcu4 = w4; la->hcu4 = h4; la->stride4 = stride4; la->planeElems4 = planeElems4; la->origin4 = origin4;
    return X265HIP_OK;
}
int x265hip_la_intra(x265hip_la* la, uint64_t key, const void* planes4, const int32_t* invQscale, int32_t* intraCost, uint8_t* intraMode, uint16_t* lowresCosts, int32_t* rowSatds, int64_t* sums2)
{
    if (!la || !key || !planes4 || !intraCost || !intraMode || !lowresCosts || !rowSatds || !sums2) return fail("la_intra: bad arguments");
    const int ncu = la->wcu * la->hcu;
    std::vector<int32_t> im((size_t)ncu), lc((size_t)ncu);
    g_o.intra((const xo_pixel*)planes4 + la->origin, la->stride, la->wcu, la->hcu, invQscale, intraCost, im.data(), lc.data(), rowSatds, sums2);
    for (int i = 0; i < ncu; i++) { intraMode[i] = (uint8_t)im[(size_t)i]; lowresCosts[i] = (uint16_t)lc[(size_t)i]; }
    return X265HIP_OK;
}
int x265hip_la_estimate(x265hip_la* la, const x265hip_la_estimate_desc* d)
{
    if (!la || !d) return fail("la_estimate: bad arguments");
    { std::lock_guard<std::mutex> g(la->lock); la->launches++; if (getenv("X265MOCK_FAIL_AT") && la->launches == atol(getenv("X265MOCK_FAIL_AT"))) return fail("call %lld fails on request (X265MOCK_FAIL_AT)", (long long)la->launches); }
    return one_estimate(la, d);          /* (concurrent callers work on different (b, list, distance) arrays: the encoder's own workers do) */
}
int x265hip_la_estimate_batch(x265hip_la* la, const x265hip_la_estimate_desc* descs, int n)
{
    if (!la || !descs || n < 1) return fail("la_estimate_batch: bad arguments");
    /* an estimate of a call must not reuse a list search another estimate of the same call makes: the arrays a searching estimate writes are nobody's input */
    std::set<const void*> written;
    for (int i = 0; i < n; i++) for (int l = 0; l < 2; l++) if (descs[i].doSearch[l] && descs[i].mvCosts[l]) written.insert(descs[i].mvCosts[l]);
    for (int i = 0; i < n; i++) for (int l = 0; l < 2; l++)
        if (!descs[i].doSearch[l] && descs[i].mvCosts[l] && written.count(descs[i].mvCosts[l])) return fail("estimate_batch: estimate %d reuses the list-%d search another estimate of the same call makes", i, l);
    std::set<std::tuple<uint64_t, uint64_t, uint64_t>> seen;
    for (int i = 0; i < n; i++) if (!seen.insert(std::make_tuple(descs[i].key[0], descs[i].key[1], descs[i].key[2])).second) return fail("estimate_batch: the same (p0, b, p1) twice in one call");
    { std::lock_guard<std::mutex> g(la->lock); la->launches += (n + X265HIP_LA_MAX_BATCH - 1) / X265HIP_LA_MAX_BATCH; if (getenv("X265MOCK_FAIL_AT") && la->launches >= atol(getenv("X265MOCK_FAIL_AT"))) return fail("call %lld fails on request (X265MOCK_FAIL_AT)", (long long)la->launches); }
    for (int i = 0; i < n; i++) { const int rc = one_estimate(la, &descs[i]); if (rc) return rc; }
    return X265HIP_OK;
}
int x265hip_la_batch_stats(const x265hip_la* la, int64_t* launches, int64_t* estimates) { if (!la) return X265HIP_EARG; if (launches) *launches = la->launches; if (estimates) *estimates = la->estimates; return X265HIP_OK; }
int x265hip_la_cutree_propagate(x265hip_la* la, const x265hip_la_cutree_desc* d)
{
    if (!la || !d || !d->intraCost || !d->lowresCosts || !d->invQscale || !d->mvs0 || !d->propB || !d->prop0 || (d->distP1 > 0 && (!d->mvs1 || !d->prop1))) return fail("la_cutree_propagate: arrays missing");
    const int ncu = la->wcu * la->hcu;
    std::vector<int32_t> m0((size_t)ncu * 2), m1((size_t)ncu * 2, 0);
    for (int i = 0; i < 2 * ncu; i++) { m0[(size_t)i] = d->mvs0[i]; if (d->mvs1) m1[(size_t)i] = d->mvs1[i]; }
    std::vector<uint16_t> dummy((size_t)ncu, 0);
    g_o.propagate(la->wcu, la->hcu, d->distP0, d->distP1, d->weightedBiPred, d->fpsFactor, d->referenced, d->intraCost, d->lowresCosts, d->invQscale, m0.data(), m1.data(),
                  const_cast<uint16_t*>(d->propB), d->prop0, d->prop1 ? d->prop1 : dummy.data());
    return X265HIP_OK;
}
} // extern "C"
