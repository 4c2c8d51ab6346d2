// xh_la.cpp -- host side of the lookahead frame-cost path for a C / C++ caller (include/x265hip_ctx.h: x265hip_la_*): the half-resolution pictures of the frames in the
// lookahead stay on the device under the caller's key; LookaheadTLD::lowresIntraEstimate (encoder/slicetype.cpp:755-864) of one picture and
// CostEstimateGroup::estimateFrameCost (:4366-4463, with estimateCUCost :4467-4640) of one (p0, b, p1) choice are one call each, host arrays in / out, on top of
// x265hip_lookahead_intra_batch / x265hip_lookahead_cost_batch (kern_lookahead.hip).  What integration/lookahead_adapter.cpp binds inside the reference encoder.
#include "xh_common.h"
#include "../../include/x265hip_ctx.h"
#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>
using namespace xh;

namespace { struct Req; constexpr int kLaBatch = X265HIP_LA_MAX_BATCH; }      // estimates per launch at most

struct x265hip_la
{
    x265hip_ctx* ctx = nullptr;
    int wcu = 0, hcu = 0, ncu = 0, maxPics = 0, costHalf = 0;
    intptr_t stride = 0; int64_t planeElems = 0, origin = 0;
    pixel* low = nullptr;                        // (maxPics + kLaBatch) pictures x 4 planes; the last kLaBatch places hold the weighted copies of the estimates of one launch
    int32_t* intraCost = nullptr; int32_t* invq = nullptr; bool haveInvq = false;      // per picture slot
    uint8_t* intraMode = nullptr; uint16_t* intraLc = nullptr; int32_t* intraRows = nullptr; int64_t* intraSums = nullptr;      // one picture's worth (the one being estimated)
    uint16_t* costRow = nullptr;
    int16_t* mvs = nullptr; int32_t* mvCosts = nullptr; uint16_t* lc = nullptr; int32_t* rows = nullptr; int64_t* sums = nullptr; x265hip_la_task* task = nullptr;
    // --hme: the quarter-resolution pictures (Lowres::lowerResBuffer[0]: four planes) at the same places, the level-0 MV / cost slots of a launch
    bool hme = false; int wcu4 = 0, hcu4 = 0, ncu4 = 0; intptr_t stride4 = 0; int64_t planeElems4 = 0, origin4 = 0;
    pixel* low4 = nullptr; int16_t* mvs4 = nullptr; int32_t* mvCosts4 = nullptr;
    struct Slot { uint64_t key = 0; uint64_t used = 0; bool lower = false; bool haveIntra = false; };      // haveIntra: the slot's intra costs are THIS picture's (a picture that came up as a reference has none yet)
    std::vector<Slot> slots; uint64_t tick = 0;
    // cuTree (x265hip_la_cutree_propagate): staging of one propagation step, allocated at its first call
    int32_t *ctIntra = nullptr, *ctInvq = nullptr; uint16_t *ctLc = nullptr, *ctProp[3] = {}; int16_t *ctMv[2] = {}; uint64_t* ctWork = nullptr;
    std::mutex mu;                                // the device side: one call at a time on the context's stream
    // estimates that arrive while a launch is in flight are queued and go up together (x265hip_la_estimate)
    std::mutex qmu; std::condition_variable qcv; bool leader = false;
    std::vector<Req*> queue;
    int64_t batches = 0, batched = 0;
    std::vector<void*> owned;
    template<class T> int alloc(T*& p, size_t n, const char* file = __builtin_FILE(), int line = __builtin_LINE())
    {
        void* v = nullptr;
        XH_HIP(xh::dev_alloc(&v, n * sizeof(T), xh::alloc_tag(file, line)));
        owned.push_back(v); p = (T*)v;
        return X265HIP_OK;
    }
    int find(uint64_t key) const { for (int i = 0; i < (int)slots.size(); i++) if (slots[i].key == key) return i; return -1; }
};

namespace { constexpr int kLaHalfMin = 1 << 14; }

extern "C" int x265hip_la_create(x265hip_ctx* ctx, int widthInCU, int heightInCU, intptr_t stride, int64_t planeElems, int64_t origin, int maxPictures, x265hip_la** out)
{
    if (!ctx || !out || widthInCU < 1 || heightInCU < 1 || stride < widthInCU * 8 || planeElems < stride * heightInCU * 8 || origin < 0 || origin >= planeElems || maxPictures < 3 || maxPictures > 1024)
    { set_error("la_create: bad geometry"); return X265HIP_EARG; }
    // the MVD cost row: every |mv - mvp| of a picture this size, and the doubled vector the raster of a star level of --hme costs one placement in four at (motion.cpp:1392)
    const int costHalf = std::max(kLaHalfMin, 12 * 8 * (widthInCU > heightInCU ? widthInCU : heightInCU) + 192);
    XH_HIP(hipSetDevice(x265hip_ctx_device(ctx)));
    x265hip_la* a = new (std::nothrow) x265hip_la();
    if (!a) return X265HIP_EARG;
    a->ctx = ctx; a->wcu = widthInCU; a->hcu = heightInCU; a->ncu = widthInCU * heightInCU; a->stride = stride; a->planeElems = planeElems; a->origin = origin; a->maxPics = maxPictures; a->costHalf = costHalf;
    a->slots.resize((size_t)maxPictures);
    const size_t np = (size_t)maxPictures + kLaBatch, ncu = (size_t)a->ncu;      // + one weighted copy per estimate of a batch
    if ((int64_t)np * 4 * planeElems >= ((int64_t)1 << 31)) { set_error("la_create: %d pictures of this size do not fit the 2^31-element lowres buffer", maxPictures); delete a; return X265HIP_EARG; }
    int rc;
    if ((rc = a->alloc(a->low, np * 4 * (size_t)planeElems)) || (rc = a->alloc(a->intraCost, np * ncu)) || (rc = a->alloc(a->invq, np * ncu)) || (rc = a->alloc(a->intraMode, ncu)) ||
        (rc = a->alloc(a->intraLc, ncu)) || (rc = a->alloc(a->intraRows, (size_t)heightInCU)) || (rc = a->alloc(a->intraSums, 2)) || (rc = a->alloc(a->costRow, (size_t)2 * costHalf + 1)) ||
        (rc = a->alloc(a->mvs, (size_t)kLaBatch * 2 * ncu * 2)) || (rc = a->alloc(a->mvCosts, (size_t)kLaBatch * 2 * ncu)) || (rc = a->alloc(a->lc, (size_t)kLaBatch * ncu)) ||
        (rc = a->alloc(a->rows, (size_t)kLaBatch * heightInCU)) || (rc = a->alloc(a->sums, (size_t)kLaBatch * 3)) || (rc = a->alloc(a->task, (size_t)kLaBatch)))
    { x265hip_la_destroy(a); return rc; }
    std::vector<uint16_t> row((size_t)2 * costHalf + 1);
    if ((rc = x265hip_mvcost_row(x265hip_lookahead_qp(), costHalf, row.data()))) { x265hip_la_destroy(a); return rc; }
    if (hipMemcpy(a->costRow, row.data(), row.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess) { x265hip_la_destroy(a); return X265HIP_EDEVICE; }
    *out = a;
    return X265HIP_OK;
}
extern "C" int x265hip_la_enable_hme(x265hip_la* a, int widthInCU4, int heightInCU4, intptr_t stride4, int64_t planeElems4, int64_t origin4)
{
    if (!a || widthInCU4 < 1 || heightInCU4 < 1 || stride4 < widthInCU4 * 8 || planeElems4 < stride4 * heightInCU4 * 8 || origin4 < 0 || origin4 >= planeElems4)
    { set_error("la_enable_hme: bad geometry"); return X265HIP_EARG; }
    std::lock_guard<std::mutex> g(a->mu);
    if (a->hme) return X265HIP_OK;
    XH_HIP(hipSetDevice(x265hip_ctx_device(a->ctx)));
    const size_t np = (size_t)a->maxPics + kLaBatch;
    if ((int64_t)np * 4 * planeElems4 >= ((int64_t)1 << 31)) { set_error("la_enable_hme: the quarter-resolution buffer does not fit 2^31 elements"); return X265HIP_EARG; }
    a->wcu4 = widthInCU4; a->hcu4 = heightInCU4; a->ncu4 = widthInCU4 * heightInCU4; a->stride4 = stride4; a->planeElems4 = planeElems4; a->origin4 = origin4;
    int rc;
    if ((rc = a->alloc(a->low4, np * 4 * (size_t)planeElems4)) || (rc = a->alloc(a->mvs4, (size_t)kLaBatch * 2 * a->ncu4 * 2)) || (rc = a->alloc(a->mvCosts4, (size_t)kLaBatch * 2 * a->ncu4))) return rc;
    a->hme = true;
    return X265HIP_OK;
}
extern "C" void x265hip_la_destroy(x265hip_la* a)
{
    if (!a) return;
    (void)hipSetDevice(x265hip_ctx_device(a->ctx));
    (void)hipStreamSynchronize((hipStream_t)x265hip_ctx_stream(a->ctx));
    for (void* p : a->owned) (void)xh::dev_free(p);
    delete a;
}

namespace {
// the picture's slot, uploading its planes (and AQ factors / intra costs when given) if it is not on the device yet; `pinned`: slots this call must not evict
int ensure_picture(x265hip_la* a, hipStream_t st, uint64_t key, const void* planes4, const int32_t* invQscale, const int32_t* intraCost, const int* pinned, int nPinned, int* slotOut,
                   const void* lowerPlanes4 = nullptr)
{
    if (!key) { set_error("la: picture key 0"); return X265HIP_EARG; }
    int s = a->find(key);
    if (s < 0)
    {
        if (!planes4) { set_error("la: picture %llu is not on the device and no planes were given", (unsigned long long)key); return X265HIP_EARG; }
        for (int i = 0; i < (int)a->slots.size(); i++)
        {
            bool pin = false;
            for (int k = 0; k < nPinned; k++) pin |= pinned[k] == i;
            if (!pin && (s < 0 || a->slots[i].used < a->slots[s].used)) s = i;
        }
        if (s < 0) { set_error("la: every one of the %d picture places is in use by this launch (create the producer with maxPictures >= 3 x the estimates that arrive together)", (int)a->slots.size()); return X265HIP_EARG; }
        a->slots[s].key = 0; a->slots[s].lower = false; a->slots[s].haveIntra = false;
        XH_HIP(hipMemcpyAsync(a->low + (size_t)s * 4 * a->planeElems, planes4, (size_t)4 * a->planeElems * sizeof(pixel), hipMemcpyHostToDevice, st));
        if (invQscale) { XH_HIP(hipMemcpyAsync(a->invq + (size_t)s * a->ncu, invQscale, (size_t)a->ncu * sizeof(int32_t), hipMemcpyHostToDevice, st)); a->haveInvq = true; }
        if (intraCost) XH_HIP(hipMemcpyAsync(a->intraCost + (size_t)s * a->ncu, intraCost, (size_t)a->ncu * sizeof(int32_t), hipMemcpyHostToDevice, st));
        XH_HIP(hipStreamSynchronize(st));                    // the caller's buffers are free again (and pageable copies are staged anyway)
        a->slots[s].key = key;                               // named once its planes are there
        a->slots[s].haveIntra = intraCost != nullptr;
    }
    else if (invQscale || (intraCost && !a->slots[s].haveIntra))
    {   // the factors of a resident picture may have moved since (--aq-motion rewrites them during the slice-type analysis): they are small, send them again.
        // A picture that came up as somebody's REFERENCE (no intra costs given) and is estimated now still holds the costs of the picture its place held before
        if (invQscale) { XH_HIP(hipMemcpyAsync(a->invq + (size_t)s * a->ncu, invQscale, (size_t)a->ncu * sizeof(int32_t), hipMemcpyHostToDevice, st)); a->haveInvq = true; }
        if (intraCost && !a->slots[s].haveIntra)
        {
            XH_HIP(hipMemcpyAsync(a->intraCost + (size_t)s * a->ncu, intraCost, (size_t)a->ncu * sizeof(int32_t), hipMemcpyHostToDevice, st));
            a->slots[s].haveIntra = true;
        }
        XH_HIP(hipStreamSynchronize(st));
    }
    if (a->hme && lowerPlanes4 && !a->slots[s].lower)
    {   // (the intra estimate brought the picture up without its quarter-resolution planes)
        XH_HIP(hipMemcpyAsync(a->low4 + (size_t)s * 4 * a->planeElems4, lowerPlanes4, (size_t)4 * a->planeElems4 * sizeof(pixel), hipMemcpyHostToDevice, st));
        XH_HIP(hipStreamSynchronize(st));
        a->slots[s].lower = true;
    }
    a->slots[s].used = ++a->tick;
    *slotOut = s;
    return X265HIP_OK;
}
}

extern "C" int x265hip_la_picture(x265hip_la* a, uint64_t key, const void* planes4, const int32_t* invQscale, const int32_t* intraCost)
{
    if (!a) { set_error("la_picture: null"); return X265HIP_EARG; }
    std::lock_guard<std::mutex> g(a->mu);
    XH_HIP(hipSetDevice(x265hip_ctx_device(a->ctx)));
    int s;
    return ensure_picture(a, (hipStream_t)x265hip_ctx_stream(a->ctx), key, planes4, invQscale, intraCost, nullptr, 0, &s);
}
extern "C" int x265hip_la_forget(x265hip_la* a, uint64_t key)
{
    if (!a) return X265HIP_EARG;
    std::lock_guard<std::mutex> g(a->mu);
    const int s = a->find(key);
    if (s >= 0) a->slots[s].key = 0;
    return X265HIP_OK;
}

extern "C" int x265hip_la_intra(x265hip_la* a, uint64_t key, const void* planes4, const int32_t* invQscale, int32_t* intraCost, uint8_t* intraMode, uint16_t* lowresCosts,
                                int32_t* rowSatds, int64_t* sums2)
{
    if (!a || !intraCost || !intraMode || !lowresCosts || !rowSatds || !sums2) { set_error("la_intra: bad arguments"); return X265HIP_EARG; }
    std::lock_guard<std::mutex> g(a->mu);
    XH_HIP(hipSetDevice(x265hip_ctx_device(a->ctx)));
    hipStream_t st = (hipStream_t)x265hip_ctx_stream(a->ctx);
    int s, rc;
    if ((rc = ensure_picture(a, st, key, planes4, invQscale, nullptr, nullptr, 0, &s))) return rc;
    const size_t ncu = (size_t)a->ncu;
    if ((rc = x265hip_lookahead_intra_batch(st, a->low + (size_t)s * 4 * a->planeElems, a->planeElems, a->stride, a->origin, a->wcu, a->hcu, 1, a->haveInvq ? a->invq + (size_t)s * ncu : nullptr,
                                            a->intraCost + (size_t)s * ncu, a->intraMode, a->intraLc, a->intraRows, a->intraSums))) return rc;
    a->slots[s].haveIntra = true;
    XH_HIP(hipMemcpyAsync(intraCost, a->intraCost + (size_t)s * ncu, ncu * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    XH_HIP(hipMemcpyAsync(intraMode, a->intraMode, ncu, hipMemcpyDeviceToHost, st));
    XH_HIP(hipMemcpyAsync(lowresCosts, a->intraLc, ncu * sizeof(uint16_t), hipMemcpyDeviceToHost, st));
    XH_HIP(hipMemcpyAsync(rowSatds, a->intraRows, (size_t)a->hcu * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    XH_HIP(hipMemcpyAsync(sums2, a->intraSums, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    XH_HIP(hipStreamSynchronize(st));
    return X265HIP_OK;
}

namespace {
// One launch for every estimate that is waiting: the lookahead's workers ask concurrently (b-adapt 2 batches its frame costs, slicetype.cpp:1960-1990, 4236-4283), and a
// batch of estimates costs the device little more than one (the sweep over a picture's block wavefronts is a serial depth that a batch pays once).
struct Req { const x265hip_la_estimate_desc* d; int rc; bool done; };

int run_batch(x265hip_la* a, const x265hip_la_estimate_desc* const* descs, int n)
{
    std::lock_guard<std::mutex> g(a->mu);
    XH_HIP(hipSetDevice(x265hip_ctx_device(a->ctx)));
    hipStream_t st = (hipStream_t)x265hip_ctx_stream(a->ctx);
    const size_t ncu = (size_t)a->ncu;
    std::vector<int> pinned;
    x265hip_la_task tasks[kLaBatch];
    int rc;
    for (int i = 0; i < n; i++)
    {
        const x265hip_la_estimate_desc* d = descs[i];
        const bool isB = d->key[2] != d->key[1];
        int slot[3] = { -1, -1, -1 };
        const int order[3] = { 1, 0, 2 };      // b first (its intra costs and AQ factors are read), then the references; a picture this batch uses is not evicted for another
        for (int k : order)
        {
            if (k == 2 && !isB) { slot[2] = slot[1]; continue; }
            if ((rc = ensure_picture(a, st, d->key[k], d->planes[k], k == 1 ? d->invQscale : nullptr, k == 1 ? d->intraCost : nullptr, pinned.data(), (int)pinned.size(), &slot[k],
                                     d->hme ? d->lowerPlanes[k] : nullptr))) return rc;
            if (d->hme && (!a->hme || !a->slots[slot[k]].lower)) { set_error("la_estimate: --hme estimate without x265hip_la_enable_hme / quarter-resolution planes"); return X265HIP_EARG; }
            pinned.push_back(slot[k]);
        }
        if (!a->slots[slot[1]].haveIntra) { set_error("la_estimate: picture %llu has no intra costs on the device (x265hip_la_intra it first or hand over desc.intraCost)", (unsigned long long)d->key[1]); return X265HIP_EARG; }
        if (slot[0] == slot[1] || (isB && slot[2] == slot[1])) { set_error("la_estimate: the estimated picture is its own reference"); return X265HIP_EARG; }
        x265hip_la_task& t = tasks[i];
        t = x265hip_la_task{};
        t.b = slot[1]; t.p0 = slot[0]; t.p1 = slot[2];      // places in the lowres buffer; p1 == b marks the P estimate
        t.doSearch[0] = d->doSearch[0] != 0; t.doSearch[1] = isB && d->doSearch[1] != 0; t.mvSlot[0] = 2 * i; t.mvSlot[1] = 2 * i + 1; t.outSlot = i; t.weighted0 = 0;
        if (d->weightedPlanes)
        {   // list 0 is searched in the weighted copy LookaheadTLD::weightsAnalyse made (slicetype.cpp:4474); the bidirectional average uses p0 itself
            XH_HIP(hipMemcpyAsync(a->low + (size_t)(a->maxPics + i) * 4 * a->planeElems, d->weightedPlanes, (size_t)4 * a->planeElems * sizeof(pixel), hipMemcpyHostToDevice, st));
            t.weighted0 = 1 + a->maxPics + i;
        }
        for (int l = 0; l < (isB ? 2 : 1); l++)
            if (!t.doSearch[l])
            {   // the list's earlier result (the reference's bDoSearch caching, :4376-4377): MVs and costs are read
                XH_HIP(hipMemcpyAsync(a->mvs + (size_t)(2 * i + l) * ncu * 2, d->mvs[l], ncu * 2 * sizeof(int16_t), hipMemcpyHostToDevice, st));
                XH_HIP(hipMemcpyAsync(a->mvCosts + (size_t)(2 * i + l) * ncu, d->mvCosts[l], ncu * sizeof(int32_t), hipMemcpyHostToDevice, st));
            }
    }
    // all estimates of a batch sweep the same way (the caller's workers belong to one lookahead)
    const int rowsPerSlice = descs[0]->rowsPerSlice;
    const x265hip_la_estimate_desc* d0 = descs[0];
    for (int i = 1; i < n; i++)
        if (descs[i]->rowsPerSlice != rowsPerSlice || descs[i]->hme != d0->hme || (d0->hme && (memcmp(descs[i]->hmeMethod, d0->hmeMethod, sizeof(d0->hmeMethod)) || memcmp(descs[i]->hmeRange, d0->hmeRange, sizeof(d0->hmeRange)))))
        { set_error("la_estimate: concurrent estimates with different sweep parameters"); return X265HIP_EARG; }
    x265hip_la_hme H{};
    if (d0->hme)
    {
        H.lowerRes = a->low4; H.planeElems = a->planeElems4; H.stride = a->stride4; H.origin = a->origin4; H.widthInCU = a->wcu4; H.heightInCU = a->hcu4;
        H.method[0] = d0->hmeMethod[0]; H.method[1] = d0->hmeMethod[1]; H.range[0] = d0->hmeRange[0]; H.range[1] = d0->hmeRange[1];
        H.mvs = a->mvs4; H.mvCosts = a->mvCosts4;
    }
    XH_HIP(hipMemcpyAsync(a->task, tasks, (size_t)n * sizeof(x265hip_la_task), hipMemcpyHostToDevice, st));
    if ((rc = x265hip_lookahead_cost_batch_hme(st, a->low, a->planeElems, a->stride, a->origin, a->wcu, a->hcu, a->task, n, a->maxPics + kLaBatch, a->intraCost, a->haveInvq ? a->invq : nullptr,
                                               a->costRow, a->costHalf, rowsPerSlice, a->mvs, a->mvCosts, a->lc, a->rows, a->sums, d0->hme ? &H : nullptr))) return rc;
    for (int i = 0; i < n; i++)
    {
        const x265hip_la_estimate_desc* d = descs[i];
        const bool isB = d->key[2] != d->key[1];
        for (int l = 0; l < (isB ? 2 : 1); l++)
            if (tasks[i].doSearch[l])
            {
                if (d->hme && d->lowerMvs[l] && d->lowerMvCosts[l])
                {   // Lowres::lowerResMvs / lowerResMvCosts of the list
                    XH_HIP(hipMemcpyAsync(d->lowerMvs[l], a->mvs4 + (size_t)(2 * i + l) * a->ncu4 * 2, (size_t)a->ncu4 * 2 * sizeof(int16_t), hipMemcpyDeviceToHost, st));
                    XH_HIP(hipMemcpyAsync(d->lowerMvCosts[l], a->mvCosts4 + (size_t)(2 * i + l) * a->ncu4, (size_t)a->ncu4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
                }
                XH_HIP(hipMemcpyAsync(d->mvs[l], a->mvs + (size_t)(2 * i + l) * ncu * 2, ncu * 2 * sizeof(int16_t), hipMemcpyDeviceToHost, st));
                XH_HIP(hipMemcpyAsync(d->mvCosts[l], a->mvCosts + (size_t)(2 * i + l) * ncu, ncu * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            }
        XH_HIP(hipMemcpyAsync(d->lowresCosts, a->lc + (size_t)i * ncu, ncu * sizeof(uint16_t), hipMemcpyDeviceToHost, st));
        XH_HIP(hipMemcpyAsync(d->rowSatds, a->rows + (size_t)i * a->hcu, (size_t)a->hcu * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        XH_HIP(hipMemcpyAsync(d->sums, a->sums + (size_t)i * 3, 3 * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    }
    XH_HIP(hipStreamSynchronize(st));
    a->batches++; a->batched += n;
    return X265HIP_OK;
}
}

extern "C" int x265hip_la_estimate(x265hip_la* a, const x265hip_la_estimate_desc* d)
{
    if (!a || !d || !d->lowresCosts || !d->rowSatds || !d->sums || !d->mvs[0] || !d->mvCosts[0] || (d->key[2] != d->key[1] && (!d->mvs[1] || !d->mvCosts[1])))
    { set_error("la_estimate: bad arguments"); return X265HIP_EARG; }
    Req r{ d, X265HIP_OK, false };
    std::unique_lock<std::mutex> lk(a->qmu);
    a->queue.push_back(&r);
    while (!r.done)
    {
        if (a->leader) { a->qcv.wait(lk); continue; }
        // become the leader: everything that is waiting now (this request among it, unless more than a batch was ahead of it) goes up as one launch
        a->leader = true;
        Req* batch[kLaBatch]; const x265hip_la_estimate_desc* descs[kLaBatch];
        // a launch pins up to three pictures per estimate: never more estimates than the producer has places for (the rest goes up with the next launch)
        const int n = (int)std::min<size_t>(a->queue.size(), (size_t)std::min(kLaBatch, std::max(1, a->maxPics / 3)));
        for (int i = 0; i < n; i++) { batch[i] = a->queue[i]; descs[i] = batch[i]->d; }
        a->queue.erase(a->queue.begin(), a->queue.begin() + n);
        lk.unlock();
        const int rc = run_batch(a, descs, n);
        lk.lock();
        for (int i = 0; i < n; i++) { batch[i]->rc = rc; batch[i]->done = true; }      // (an error text belongs to the thread that ran the batch; the code reaches every caller)
        a->leader = false;
        a->qcv.notify_all();
    }
    return r.rc;
}
// A whole batch of the caller at once (CostEstimateGroup::finishBatch, slicetype.cpp:4271-4278: up to 512 queued estimates): launches of up to X265HIP_LA_MAX_BATCH estimates
// (and never more than the producer has picture places for), in the caller's order.  The caller keeps estimates that depend on each other's searches in different calls
extern "C" int x265hip_la_estimate_batch(x265hip_la* a, const x265hip_la_estimate_desc* descs, int n)
{
    if (!a || !descs || n < 0) { set_error("la_estimate_batch: bad arguments"); return X265HIP_EARG; }
    for (int i = 0; i < n; i++)
    {
        const x265hip_la_estimate_desc* d = descs + i;
        if (!d->lowresCosts || !d->rowSatds || !d->sums || !d->mvs[0] || !d->mvCosts[0] || (d->key[2] != d->key[1] && (!d->mvs[1] || !d->mvCosts[1])))
        { set_error("la_estimate_batch: estimate %d lacks an output array", i); return X265HIP_EARG; }
    }
    const int per = std::min(kLaBatch, std::max(1, a->maxPics / 3));
    for (int i0 = 0; i0 < n; i0 += per)
    {
        const x265hip_la_estimate_desc* p[kLaBatch];
        const int m = std::min(per, n - i0);
        for (int i = 0; i < m; i++) p[i] = descs + i0 + i;
        // (behind the leader of concurrent single estimates, if there is one: the device side is one call at a time -- run_batch takes a->mu)
        const int rc = run_batch(a, p, m);
        if (rc) return rc;
    }
    return X265HIP_OK;
}
extern "C" int x265hip_la_batch_stats(const x265hip_la* a, int64_t* launches, int64_t* estimates)
{
    if (!a) return X265HIP_EARG;
    if (launches) *launches = a->batches;
    if (estimates) *estimates = a->batched;
    return X265HIP_OK;
}


// Lookahead::estimateCUPropagate (slicetype.cpp:3850-3953) for a host caller: the step's arrays up, x265hip_cutree_propagate, the two references' accumulated costs back
extern "C" int x265hip_la_cutree_propagate(x265hip_la* a, const x265hip_la_cutree_desc* d)
{
    if (!a || !d) { set_error("la_cutree_propagate: null argument"); return X265HIP_EARG; }
    const bool bidir = d->distP1 > 0;
    if (d->distP0 < 1 || d->distP1 < 0 || !d->intraCost || !d->lowresCosts || !d->invQscale || !d->mvs0 || !d->propB || !d->prop0 || (bidir && (!d->mvs1 || !d->prop1)))
    { set_error("la_cutree_propagate: incomplete description"); return X265HIP_EARG; }
    std::lock_guard<std::mutex> g(a->mu);
    XH_HIP(hipSetDevice(x265hip_ctx_device(a->ctx)));
    hipStream_t st = (hipStream_t)x265hip_ctx_stream(a->ctx);
    const size_t n = (size_t)a->ncu;
    {   // staging buffers: each allocated once (a failure half way leaves the earlier ones in place for the next call)
        int rc;
        auto once = [&](auto*& p, size_t count) -> int { return p ? X265HIP_OK : a->alloc(p, count); };
        if ((rc = once(a->ctIntra, n)) || (rc = once(a->ctInvq, n)) || (rc = once(a->ctLc, n)) || (rc = once(a->ctProp[0], n)) || (rc = once(a->ctProp[1], n)) ||
            (rc = once(a->ctProp[2], n)) || (rc = once(a->ctMv[0], 2 * n)) || (rc = once(a->ctMv[1], 2 * n)) || (rc = once(a->ctWork, 2 * n))) return rc;
    }
    XH_HIP(hipMemcpyAsync(a->ctIntra, d->intraCost, n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    XH_HIP(hipMemcpyAsync(a->ctInvq, d->invQscale, n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    XH_HIP(hipMemcpyAsync(a->ctLc, d->lowresCosts, n * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    XH_HIP(hipMemcpyAsync(a->ctMv[0], d->mvs0, 2 * n * sizeof(int16_t), hipMemcpyHostToDevice, st));
    if (bidir) XH_HIP(hipMemcpyAsync(a->ctMv[1], d->mvs1, 2 * n * sizeof(int16_t), hipMemcpyHostToDevice, st));
    XH_HIP(hipMemcpyAsync(a->ctProp[0], d->propB, n * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    XH_HIP(hipMemcpyAsync(a->ctProp[1], d->prop0, n * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    if (bidir) XH_HIP(hipMemcpyAsync(a->ctProp[2], d->prop1, n * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    int rc = x265hip_cutree_propagate(st, a->wcu, a->hcu, d->distP0, d->distP1, d->weightedBiPred, d->fpsFactor, d->referenced, a->ctIntra, a->ctLc, a->ctInvq, a->ctMv[0],
                                      bidir ? a->ctMv[1] : nullptr, a->ctProp[0], a->ctProp[1], bidir ? a->ctProp[2] : nullptr, a->ctWork, 2 * n * sizeof(uint64_t));
    if (rc) return rc;
    XH_HIP(hipMemcpyAsync(d->prop0, a->ctProp[1], n * sizeof(uint16_t), hipMemcpyDeviceToHost, st));
    if (bidir) XH_HIP(hipMemcpyAsync(d->prop1, a->ctProp[2], n * sizeof(uint16_t), hipMemcpyDeviceToHost, st));
    XH_HIP(hipStreamSynchronize(st));
    return X265HIP_OK;
}
