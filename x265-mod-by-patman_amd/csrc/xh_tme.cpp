// xh_tme.cpp -- host side of the ThreadedME producer: the order in which Analysis::deriveMVsForCTU / computeMVForPUs (reference encoder/analysis.cpp:161-306) visit the
// PUs of a CTU, as a flat schedule.  PU k of every CTU of a picture has the same partition shape, position inside the CTU, slot in the CTU's MEData table
// (slice->m_ctuMV, MAX_NUM_PUS_PER_CTU = 593 slots, threadedme.h) and neighbour slots, so a frame batch can be stepped through the schedule with one launch sequence
// per entry over all CTUs.
#include "xh_common.h"
#include "../../include/x265hip_frame.h"

namespace {

struct PuShape { int w, h, part, amp; };
// g_puLookup (threadedme.h:67-92); part numbers = enum PartSize (common.h): 2Nx2N 0, 2NxN 1, Nx2N 2, NxN 3, 2NxnU 4, 2NxnD 5, nLx2N 6, nRx2N 7
const PuShape k_lookup[24] = {
    { 8, 4, 1, 0 }, { 4, 8, 2, 0 }, { 8, 8, 0, 0 }, { 16, 4, 4, 1 }, { 16, 12, 5, 1 }, { 4, 16, 6, 1 }, { 12, 16, 7, 1 }, { 16, 8, 1, 0 }, { 8, 16, 2, 0 }, { 16, 16, 0, 0 },
    { 32, 8, 4, 1 }, { 32, 24, 5, 1 }, { 8, 32, 6, 1 }, { 24, 32, 7, 1 }, { 32, 16, 1, 0 }, { 16, 32, 2, 0 }, { 32, 32, 0, 0 }, { 64, 16, 4, 1 }, { 64, 48, 5, 1 },
    { 16, 64, 6, 1 }, { 48, 64, 7, 1 }, { 64, 32, 1, 0 }, { 32, 64, 2, 0 }, { 64, 64, 0, 0 } };

struct Builder
{
    int ctu, minCu, rect, amp, n, maxSteps;
    int start[129][8];
    x265hip_tme_step* out;

    void startIndices()
    {   // ThreadedME::initPuStartIdx (threadedme.cpp:86-107)
        int s = 0;
        for (const PuShape& p : k_lookup)
        {
            if (p.w > ctu || p.h > ctu) continue;
            const int iw = p.amp ? (p.w > p.h ? p.w : p.h) : p.w, ih = p.amp ? iw : p.h;
            const int num = (ctu / iw) * (ctu / ih);
            start[p.w + p.h][p.part] = s;
            s += p.amp ? 2 * num : num;
        }
    }
    void visit(int cuX, int cuY, int size)
    {   // Analysis::computeMVForPUs (analysis.cpp:161-246): the four sub-CUs first (z order), then every PU shape of this CU in g_puLookup's order
        if (size > minCu)
            for (int s = 0; s < 4; s++) visit(cuX + (s & 1) * (size >> 1), cuY + (s >> 1) * (size >> 1), size >> 1);
        for (const PuShape& p : k_lookup)
        {
            if (p.w > size || p.h > size || (p.w != size && p.h != size)) continue;
            if (!amp && p.amp) continue;
            if (!rect && p.w != p.h && !p.amp) continue;
            const int bw = p.amp ? (p.w > p.h ? p.w : p.h) : p.w, bh = p.amp ? bw : p.h;
            const int cols = ctu / bw, rows = ctu / bh;
            int puOffset = 0;
            if (p.amp) puOffset = rows * cols;
            else if (p.part == 1) puOffset = cols;
            else if (p.part == 2) puOffset = 1;
            const int col = cuX / bw, row = cuY / bh, st = start[p.w + p.h][p.part];
            if (n < maxSteps)
            {
                x265hip_tme_step& o = out[n];
                o.part = (int16_t)p.part; o.cuSize = (int16_t)size; o.cuX = (int16_t)cuX; o.cuY = (int16_t)cuY; o.puOffset = (int16_t)puOffset;
                o.finalIdx = (int16_t)(st + row * cols + col);
                o.neighbor[0] = (int16_t)(col > 0 ? st + row * cols + col - 1 : -1);                          // MD_LEFT
                o.neighbor[1] = (int16_t)(row > 0 ? st + (row - 1) * cols + col : -1);                        // MD_ABOVE
                o.neighbor[2] = (int16_t)(row > 0 && col < cols - 1 ? st + (row - 1) * cols + col + 1 : -1);  // MD_ABOVE_RIGHT
                o.neighbor[3] = -1;                                                                           // MD_BELOW_LEFT: never
                o.neighbor[4] = (int16_t)(row > 0 && col > 0 ? st + (row - 1) * cols + col - 1 : -1);         // MD_ABOVE_LEFT
                // the partitions (PredictionUnit, CUData::getPartIndexAndSize): 2Nx2N one; 2NxN / Nx2N two halves; AMP a quarter and three quarters
                o.numPart = (int16_t)(p.part == 0 ? 1 : 2);
                for (int k = 0; k < 2; k++) { o.pu[k][0] = (int16_t)cuX; o.pu[k][1] = (int16_t)cuY; o.pu[k][2] = (int16_t)size; o.pu[k][3] = (int16_t)size; }
                const int q = size >> 2, hlf = size >> 1;
                switch (p.part)
                {
                case 1: o.pu[0][3] = o.pu[1][3] = (int16_t)hlf; o.pu[1][1] = (int16_t)(cuY + hlf); break;
                case 2: o.pu[0][2] = o.pu[1][2] = (int16_t)hlf; o.pu[1][0] = (int16_t)(cuX + hlf); break;
                case 4: o.pu[0][3] = (int16_t)q; o.pu[1][3] = (int16_t)(size - q); o.pu[1][1] = (int16_t)(cuY + q); break;
                case 5: o.pu[0][3] = (int16_t)(size - q); o.pu[1][3] = (int16_t)q; o.pu[1][1] = (int16_t)(cuY + size - q); break;
                case 6: o.pu[0][2] = (int16_t)q; o.pu[1][2] = (int16_t)(size - q); o.pu[1][0] = (int16_t)(cuX + q); break;
                case 7: o.pu[0][2] = (int16_t)(size - q); o.pu[1][2] = (int16_t)q; o.pu[1][0] = (int16_t)(cuX + size - q); break;
                default: break;
                }
            }
            n++;
        }
    }
};

} // namespace

extern "C" int x265hip_tme_schedule(int ctuSize, int minCuSize, int rect, int amp, x265hip_tme_step* steps, int maxSteps)
{
    if ((ctuSize != 64 && ctuSize != 32 && ctuSize != 16) || minCuSize < 8 || minCuSize > ctuSize || (minCuSize & (minCuSize - 1)) || (maxSteps > 0 && !steps)) return X265HIP_EARG;
    Builder b{};
    b.ctu = ctuSize; b.minCu = minCuSize; b.rect = rect; b.amp = amp; b.n = 0; b.maxSteps = maxSteps; b.out = steps;
    b.startIndices();
    b.visit(0, 0, ctuSize);
    return b.n;
}

// ---- host-level producer: one picture's MEData table from HOST inputs (include/x265hip_ctx.h: x265hip_tme_*) ----------------------------------------------
// Uploads the picture's planes, builds the phase planes, runs deriveMVsForCTU's first stage (x265hip_diamond_batch for the CTU and its four sub-CUs per reference,
// then the caller's collocated-median override), steps x265hip_tme_frame through the schedule and copies the table back.
#include "../../include/x265hip_ctx.h"
#include <algorithm>
#include <cstring>
#include <map>
#include <new>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstdlib>
using namespace xh;

int xh_diamond_rows(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                    const x265hip_me_task* tasks, int n, const uint16_t* costTable, int costHalfRange, x265hip_me_result* results);      // kern_diamond.hip
int xh_tme_area(void* stream, const x265hip_me_result* res, const int32_t* where, int nTasks, int nl, int numRef0, int numRef1, const int16_t* median, int16_t* areaBest);      // kern_tme.hip
int xh_tme_slots(void* stream, x265hip_inter_choice* table, x265hip_inter_choice* packed, const int32_t* slots, int nUsed, int nCtu, int toTable);      // kern_tme.hip

struct x265hip_tme
{
    x265hip_ctx* ctx = nullptr;
    int width = 0, height = 0, ctu = 64, nCtu = 0, nCtuX = 0, depthBytes = (int)sizeof(pixel);
    std::vector<x265hip_tme_step> steps;
    std::vector<void*> owned;
    std::map<int, uint16_t*> costRows;         // per qp, device (the diamond stage)
    std::map<int, std::vector<uint16_t>> hostRows;
    uint16_t* costTable = nullptr;             // [64][2 * kHalf + 1]: the rows of the picture's qps, in the order of desc->qps
    float* bitsRow = nullptr;
    pixel* cur = nullptr; pixel* plane[2][X265HIP_MAX_REF][2] = {}; pixel* phase[2][X265HIP_MAX_REF][2] = {};      // this picture's: kept buffers or the slot's own ones
    pixel* ownPlane[2][X265HIP_MAX_REF][2] = {}; pixel* ownPhase[2][X265HIP_MAX_REF][2] = {}; bool own[2][X265HIP_MAX_REF][2] = {};       // [list][ref][0 = searched plane, 1 = reconstructed picture]
    int64_t planeElems = 0;
    x265hip_inter_choice* table = nullptr; x265hip_inter_choice* refTable[2][X265HIP_MAX_REF] = {}; int16_t* lowres[2][X265HIP_MAX_REF] = {};
    int16_t* areaBest = nullptr; x265hip_tme_temporal* temporal = nullptr; uint8_t* qpIndex = nullptr; void* workspace = nullptr; size_t workspaceBytes = 0;
    x265hip_me_task* dTasks = nullptr; x265hip_me_result* dResults = nullptr; int32_t* dWhere = nullptr; int16_t* dMedian = nullptr;
    std::vector<x265hip_me_task> hTasks; std::vector<int32_t> hWhere;      // kept: the copies read them after the call that filled them returned
    // the slots of a CTU's table the schedule writes (and reads: neighbours are PUs of the same shape); sparse schedules move only these (pinned staging, packed [ctu][slot])
    std::vector<int32_t> slots; int32_t* dSlots = nullptr; x265hip_inter_choice* dPacked = nullptr; x265hip_inter_choice* hPacked = nullptr; bool sparse = false;
    // the staging pair holds one REGION (nCtu x slots records) per table a call moves -- the picture's and its references' own: every table of a call packs into its own region,
    // so no copy has to be waited for before the next table is packed (r05 drained the stream once per table: up to 1 + 2 x 16 times per call); regions grow on demand
    int packRegions = 0, packNext = 0; size_t regionRecs = 0;
    // reconstructed reference pictures stay on the device (plane + its 16 phase planes) under the caller's key; the least recently used of kKeep makes room
    struct Kept { uint64_t key = 0; pixel* plane = nullptr; pixel* phase = nullptr; uint64_t used = 0; int rowsSeen = 0; };      // rowsSeen: plane rows on the device so far (a reference that is still being reconstructed grows)
    std::vector<Kept> kept; uint64_t tick = 0; int evictions = 0;
    int rowQp[64];                                                // the qp whose MVD cost row sits in row q of costTable (rows are kept across pictures)
    bool prof = false, first = false; double sec[5] = {}; int pictures = 0;      // X265HIP_TME_PROF: upload, diamond stage, submit, drain, (total)
    template<class T> int alloc(T*& p, size_t n, const char* file = __builtin_FILE(), int line = __builtin_LINE())
    {
        void* v = nullptr;
        XH_HIP(xh::dev_alloc(&v, n * sizeof(T), xh::alloc_tag(file, line)));
        owned.push_back(v); p = (T*)v;
        return X265HIP_OK;
    }
    // at least `n` regions in the staging pair (called with the stream idle: at creation, and at the start of a call that moves more tables than any before it)
    int pack_regions(int n)
    {
        if (n <= packRegions) return X265HIP_OK;
        if (hPacked) { XH_HIP(hipHostFree(hPacked)); hPacked = nullptr; }
        if (dPacked) { owned.erase(std::remove(owned.begin(), owned.end(), (void*)dPacked), owned.end()); XH_HIP(xh::dev_free(dPacked)); dPacked = nullptr; }
        packRegions = 0;
        int rc = alloc(dPacked, regionRecs * n);
        if (rc) return rc;
        XH_HIP(hipHostMalloc((void**)&hPacked, regionRecs * n * sizeof(x265hip_inter_choice), hipHostMallocDefault));
        packRegions = n;
        return X265HIP_OK;
    }
};

namespace {
constexpr int kHalf = 1 << 15, kBitsHalf = 1 << 15;
constexpr int kKeepMax = 40, kKeepMin = 12;   // planes kept on the device under a key (16 references + the ones just replaced; with frame threads the references and weighted planes of every picture in flight); 17 planes each
constexpr size_t kKeepBytes = (size_t)32 << 30;   // ... within a byte budget: 40 slots of 4K 10 bit are 12.9 GB, of 8K they would be 50 GB -- 8K keeps 25 (X265HIP_TME_KEEP_BYTES overrides the budget)
int keep_slots(size_t elems)
{
    size_t budget = kKeepBytes;
    if (const char* e = getenv("X265HIP_TME_KEEP_BYTES")) { const long long v = atoll(e); if (v > 0) budget = (size_t)v; }
    const size_t perSlot = elems * 17 * sizeof(pixel);
    return (int)std::max<size_t>(kKeepMin, std::min<size_t>(kKeepMax, perSlot ? budget / perSlot : kKeepMax));
}
}

extern "C" int x265hip_tme_create(x265hip_ctx* ctx, int width, int height, int ctuSize, int minCuSize, int rect, int amp, x265hip_tme** out)
{
    if (!ctx || !out || width < 8 || height < 8) { set_error("tme_create: bad picture size"); return X265HIP_EARG; }
    // CUData::clipMv's quarter-pel limits, (size + 8 - pos - 1) << 2, travel as int16 in the task records
    if (width > X265HIP_MAX_PIC_DIM || height > X265HIP_MAX_PIC_DIM) { set_error("tme_create: %dx%d: pictures up to %d pixels a side", width, height, X265HIP_MAX_PIC_DIM); return X265HIP_EARG; }
    XH_HIP(hipSetDevice(x265hip_ctx_device(ctx)));
    const int n = x265hip_tme_schedule(ctuSize, minCuSize, rect, amp, nullptr, 0);
    if (n <= 0) { set_error("tme_create: bad CTU / CU sizes"); return X265HIP_EARG; }
    x265hip_tme* t = new (std::nothrow) x265hip_tme();
    if (!t) return X265HIP_EARG;
    for (int q = 0; q < 64; q++) t->rowQp[q] = -1;
    t->ctx = ctx; t->width = width; t->height = height; t->ctu = ctuSize; t->nCtuX = (width + ctuSize - 1) / ctuSize; t->nCtu = t->nCtuX * ((height + ctuSize - 1) / ctuSize);
    t->steps.resize(n);
    x265hip_tme_schedule(ctuSize, minCuSize, rect, amp, t->steps.data(), n);
    int rc;
    {
        std::vector<char> used(593, 0);
        for (const x265hip_tme_step& e : t->steps) for (int pi = 0; pi < e.numPart; pi++) { const int sl = e.finalIdx + pi * e.puOffset; if (sl >= 0 && sl < 593) used[sl] = 1; }
        for (int sl = 0; sl < 593; sl++) if (used[sl]) t->slots.push_back(sl);
        t->sparse = t->slots.size() * 2 < 593;
        if (t->sparse)
        {
            t->regionRecs = (size_t)t->nCtu * t->slots.size();
            if ((rc = t->alloc(t->dSlots, t->slots.size()))) { x265hip_tme_destroy(t); return rc; }
            if (hipMemcpy(t->dSlots, t->slots.data(), t->slots.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { x265hip_tme_destroy(t); return X265HIP_EDEVICE; }
            if ((rc = t->pack_regions(4))) { x265hip_tme_destroy(t); return rc; }          // the picture's table + three references' (preset medium); more when a picture brings more
        }
    }
    std::vector<float> bits(2 * kBitsHalf + 1);
    x265hip_mvbits_row(kBitsHalf, bits.data());
    if ((rc = t->alloc(t->bitsRow, bits.size()))) { x265hip_tme_destroy(t); return rc; }
    if (hipMemcpy(t->bitsRow, bits.data(), bits.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) { x265hip_tme_destroy(t); return X265HIP_EDEVICE; }
    t->workspaceBytes = x265hip_tme_workspace(t->nCtu);
    char* ws = nullptr;
    if ((rc = t->alloc(ws, t->workspaceBytes)) || (rc = t->alloc(t->table, (size_t)t->nCtu * 593)) || (rc = t->alloc(t->areaBest, (size_t)t->nCtu * 5 * 2 * X265HIP_MAX_REF * 2)) ||
        (rc = t->alloc(t->temporal, (size_t)t->nCtu * n * 2)) || (rc = t->alloc(t->qpIndex, (size_t)t->nCtu * n)) || (rc = t->alloc(t->dTasks, (size_t)t->nCtu * 5)) ||
        (rc = t->alloc(t->dResults, (size_t)t->nCtu * 5 * 2 * X265HIP_MAX_REF)) || (rc = t->alloc(t->dWhere, (size_t)t->nCtu * 5)) || (rc = t->alloc(t->dMedian, (size_t)t->nCtu * 2 * X265HIP_MAX_REF * 3)) || (rc = t->alloc(t->costTable, (size_t)64 * (2 * kHalf + 1))))
    { x265hip_tme_destroy(t); return rc; }
    t->workspace = ws;
    *out = t;
    return X265HIP_OK;
}
extern "C" void x265hip_tme_destroy(x265hip_tme* t)
{
    if (!t) return;
    if (t->prof && t->pictures)
        fprintf(stderr, "x265hip_tme: %d pictures, per picture: upload + phase planes %.2f ms, diamond stage %.2f ms, submit %.2f ms, drain + table down %.2f ms\n", t->pictures,
                1e3 * t->sec[0] / t->pictures, 1e3 * t->sec[1] / t->pictures, 1e3 * t->sec[2] / t->pictures, 1e3 * t->sec[3] / t->pictures);
    if (t->prof && t->evictions) fprintf(stderr, "x265hip_tme: %d kept planes were replaced while they held rows (%d slots)\n", t->evictions, (int)t->kept.size());
    if (t->hPacked) (void)hipHostFree(t->hPacked);
    for (void* p : t->owned) (void)xh::dev_free(p);
    for (auto& kv : t->costRows) (void)xh::dev_free(kv.second);
    delete t;
}
extern "C" int x265hip_tme_entries(const x265hip_tme* t, const x265hip_tme_step** steps) { if (!t) return 0; if (steps) *steps = t->steps.data(); return (int)t->steps.size(); }

extern "C" int x265hip_tme_picture(x265hip_tme* t, const x265hip_tme_picture_desc* d)
{
    if (!t || !d || !d->curPlane || !d->table || !d->temporal || d->nQp < 1 || d->nQp > 64 || !d->qpIndex || !d->areaQpIndex) { set_error("tme_picture: bad arguments"); return X265HIP_EARG; }
    const int nl = d->isP ? 1 : 2, nS = (int)t->steps.size();
    const int nCtuY = t->nCtu / t->nCtuX;
    if (d->ctuRowFirst < 0 || d->ctuRowCount < 0 || d->ctuRowFirst + d->ctuRowCount > nCtuY || (d->ctuRowFirst && !d->ctuRowCount))
    { set_error("tme_picture: CTU rows %d + %d of %d", d->ctuRowFirst, d->ctuRowCount, nCtuY); return X265HIP_EARG; }
    // the band: CTUs c0 .. c0 + nCtu - 1 of the picture (the whole picture without one).  Per-CTU arrays keep the picture's addressing: everything below moves and computes
    // the band's part of them only
    const int c0 = d->ctuRowFirst * t->nCtuX, nCtu = (d->ctuRowCount ? d->ctuRowCount : nCtuY) * t->nCtuX;
    for (int l = 0; l < nl; l++)      // before anything is indexed by it: refs[][] and every per-reference array here hold X265HIP_MAX_REF entries
        if (d->numRef[l] < 1 || d->numRef[l] > X265HIP_MAX_REF) { set_error("tme_picture: %d references in list %d (1..%d)", d->numRef[l], l, X265HIP_MAX_REF); return X265HIP_EARG; }
    if (d->width != t->width || d->height != t->height) { set_error("tme_picture: %dx%d picture on a %dx%d producer", d->width, d->height, t->width, t->height); return X265HIP_EARG; }
    if (d->pirStartCol < 0 || d->pirStartCol > t->nCtuX || (d->pirStartCol && (!d->isP || d->pirSafeX < -3 || d->pirSafeX > X265HIP_MAX_PIC_DIM)))
    { set_error("tme_picture: intra-refresh fields: start column %d of %d, safe x %d (%s picture)", d->pirStartCol, t->nCtuX, d->pirSafeX, d->isP ? "P" : "B"); return X265HIP_EARG; }
    // the band's qp indices name rows of the cost table and entries of `lambdas` on the device (nQp of them): an index beyond them would be an out-of-range device read
    for (size_t i = (size_t)c0 * 5; i < (size_t)(c0 + nCtu) * 5; i++)
        if (d->areaQpIndex[i] >= d->nQp) { set_error("tme_picture: areaQpIndex[%zu] = %d with %d qps", i, (int)d->areaQpIndex[i], d->nQp); return X265HIP_EARG; }
    for (size_t i = (size_t)c0 * nS; i < (size_t)(c0 + nCtu) * nS; i++)
        if (d->qpIndex[i] >= d->nQp) { set_error("tme_picture: qpIndex[%zu] = %d with %d qps", i, (int)d->qpIndex[i], d->nQp); return X265HIP_EARG; }
    XH_HIP(hipSetDevice(x265hip_ctx_device(t->ctx)));      // the caller may be any thread of the encoder's pool (a new thread starts on device 0)
    hipStream_t st = (hipStream_t)x265hip_ctx_stream(t->ctx);
    t->prof = (d->flags & X265HIP_TME_PROFILE) != 0;
    const int refLag = d->frameThreads > 1 ? d->searchRange : (d->sourceHeight > 0 ? d->sourceHeight : d->height);
    const int64_t elems = d->planeElems;
    int rc;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    auto lap = [&](int k, bool sync) { if (!t->prof) return; if (sync) (void)hipStreamSynchronize(st); const double t1 = now(); if (t->first) t->sec[k] += t1 - t0; t0 = t1; };      // the first picture (streams, code objects) is not counted
    if (t->sparse)
    {   // a staging region per table this call moves (the picture's + its references' own); the stream is idle here -- the call before ended with a synchronisation
        int nTables = 1;
        for (int l = 0; l < nl; l++) for (int r = 0; r < d->numRef[l]; r++) nTables += d->refs[l][r].refTable != nullptr;
        if ((rc = t->pack_regions(nTables))) return rc;
        t->packNext = 0;
    }
    // host table -> device table: whole, or the schedule's slots through the pinned staging pair (a region of its own per table: nothing is waited for in between)
    auto table_up = [&](x265hip_inter_choice* dev, const x265hip_inter_choice* host) -> int
    {
        if (!t->sparse) { XH_HIP(hipMemcpyAsync(dev + (size_t)c0 * 593, host + (size_t)c0 * 593, (size_t)nCtu * 593 * sizeof(x265hip_inter_choice), hipMemcpyHostToDevice, st)); return X265HIP_OK; }
        const int nU = (int)t->slots.size();
        if (t->packNext >= t->packRegions) { set_error("tme_picture: more tables than staging regions"); return X265HIP_EARG; }      // (sized at the start of the call)
        x265hip_inter_choice* h = t->hPacked + (size_t)t->packNext * t->regionRecs; x265hip_inter_choice* dv = t->dPacked + (size_t)t->packNext * t->regionRecs;
        t->packNext++;
        for (int c = 0; c < nCtu; c++) for (int k = 0; k < nU; k++) h[(size_t)c * nU + k] = host[(size_t)(c0 + c) * 593 + t->slots[k]];
        XH_HIP(hipMemcpyAsync(dv, h, (size_t)nCtu * nU * sizeof(x265hip_inter_choice), hipMemcpyHostToDevice, st));
        return xh_tme_slots(st, dev + (size_t)c0 * 593, dv, t->dSlots, nU, nCtu, 1);
    };
    if (t->planeElems != elems)
    {   // first picture (or another plane geometry): the device planes
        if (t->planeElems) { set_error("tme_picture: the plane geometry changed"); return X265HIP_EARG; }
        t->planeElems = elems;
        if ((rc = t->alloc(t->cur, (size_t)elems))) return rc;
    }
    const int rows = (int)(elems / d->stride);
    {   // the source picture: the rows the band's PUs lie in (CTUs cut by the picture edge reach into the bottom margin)
        const int top = (int)(d->origin / d->stride);
        const int y0 = d->ctuRowCount ? top + d->ctuRowFirst * t->ctu : 0, y1 = d->ctuRowCount ? std::min(rows, top + (d->ctuRowFirst + d->ctuRowCount) * t->ctu) : rows;
        XH_HIP(hipMemcpyAsync(t->cur + (size_t)y0 * d->stride, (const pixel*)d->curPlane + (size_t)y0 * d->stride, (size_t)(y1 - y0) * d->stride * sizeof(pixel), hipMemcpyHostToDevice, st));
    }
    for (int l = 0; l < nl; l++)
        for (int r = 0; r < d->numRef[l]; r++)
        {
            const x265hip_tme_host_ref& R = d->refs[l][r];
            if (!R.mePlane || !R.reconPlane) { set_error("tme_picture: planes of list %d reference %d missing", l, r); return X265HIP_EARG; }
            for (int k = 0; k < 2; k++)
            {
                if (k == 1 && R.reconPlane == R.mePlane) { t->plane[l][r][1] = t->plane[l][r][0]; t->phase[l][r][1] = t->phase[l][r][0]; continue; }
                const bool recon = k == 1 || R.reconPlane == R.mePlane;                      // else: the weighted plane, which belongs to the current picture
                const uint64_t key = recon ? R.reconKey : R.meKey;
                const int valid0 = recon ? R.reconRowsValid : R.meRowsValid;
                if (valid0 < 0 || valid0 > rows) { set_error("tme_picture: %d valid rows of a %d-row plane (list %d reference %d)", valid0, rows, l, r); return X265HIP_EARG; }
                const int valid = valid0 ? valid0 : rows;                                  // plane rows that are final now
                int seen = 0;                                                              // ... and how many of them the device holds already
                x265hip_tme::Kept* slot = nullptr;
                uint64_t* pendingKey = nullptr;
                if (key)
                {
                    for (auto& kp : t->kept) if (kp.key == key) slot = &kp;
                    if (!slot)
                    {
                        if ((int)t->kept.size() < keep_slots((size_t)elems))
                        {
                            t->kept.emplace_back(); slot = &t->kept.back();
                            if ((rc = t->alloc(slot->plane, (size_t)elems)) || (rc = t->alloc(slot->phase, (size_t)elems * 16))) { t->kept.pop_back(); return rc; }
                        }
                        else
                        {   // least recently used, not one this picture already took
                            for (auto& kp : t->kept) if (kp.used != t->tick + 1 && (!slot || kp.used < slot->used)) slot = &kp;
                            if (!slot) { set_error("tme_picture: more distinct reference pictures than kept planes"); return X265HIP_EARG; }
                            if (slot->rowsSeen) t->evictions++;      // a plane that was on the device goes: if its picture is asked for again it is uploaded whole (reported by x265hip_tme_destroy under X265HIP_TME_PROFILE)
                        }
                        slot->key = 0;                      // named only once its planes are on their way (below): a failed call must not leave a keyed slot with stale planes
                        slot->rowsSeen = 0;
                        pendingKey = &slot->key;
                    }
                    slot->used = t->tick + 1;
                    t->plane[l][r][k] = slot->plane; t->phase[l][r][k] = slot->phase;
                    seen = slot->rowsSeen;
                    if (valid <= seen) continue;
                }
                else
                {
                    if (!t->own[l][r][k])
                    {
                        if ((rc = t->alloc(t->ownPlane[l][r][k], (size_t)elems)) || (rc = t->alloc(t->ownPhase[l][r][k], (size_t)elems * 16))) return rc;
                        t->own[l][r][k] = true;
                    }
                    t->plane[l][r][k] = t->ownPlane[l][r][k]; t->phase[l][r][k] = t->ownPhase[l][r][k];
                }
                // rows seen .. valid - 1 go up; the phase rows whose vertical taps reached past `seen` last time (the last 4; 8 taken) are made again with the new ones
                const pixel* src = (const pixel*)(k ? R.reconPlane : R.mePlane);
                XH_HIP(hipMemcpyAsync(t->plane[l][r][k] + (size_t)seen * d->stride, src + (size_t)seen * d->stride, (size_t)(valid - seen) * d->stride * sizeof(pixel), hipMemcpyHostToDevice, st));
                if ((rc = x265hip_subpel_planes_rows(st, t->plane[l][r][k], d->stride, rows, std::max(0, seen - 8), valid, t->phase[l][r][k], elems))) return rc;
                if (slot) slot->rowsSeen = valid;
                if (pendingKey) *pendingKey = key;
            }
            if (R.refTable)
            {
                if (!t->refTable[l][r] && (rc = t->alloc(t->refTable[l][r], (size_t)t->nCtu * 593))) return rc;      // the whole picture's table, whatever band comes first
                if ((rc = table_up(t->refTable[l][r], R.refTable))) return rc;
            }
            if (R.lowresMv)
            {
                const size_t nb = (size_t)d->lowresBlocksX * ((d->height + 15) / 16) * 2;
                if (!t->lowres[l][r] && (rc = t->alloc(t->lowres[l][r], nb))) return rc;
                XH_HIP(hipMemcpyAsync(t->lowres[l][r], R.lowresMv, nb * sizeof(int16_t), hipMemcpyHostToDevice, st));
            }
        }
    for (int q = 0; q < d->nQp; q++)
        if (!t->costRows.count(d->qps[q]))
        {
            std::vector<uint16_t> row(2 * kHalf + 1);
            if ((rc = x265hip_mvcost_row(d->qps[q], kHalf, row.data()))) return rc;
            void* v = nullptr;
            XH_HIP(xh::dev_alloc(&v, row.size() * sizeof(uint16_t), XH_ALLOC_TAG));
            XH_HIP(hipMemcpy(v, row.data(), row.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            t->costRows[d->qps[q]] = (uint16_t*)v;
            t->hostRows[d->qps[q]] = row;
        }
    for (int q = 0; q < d->nQp; q++)
        if (t->rowQp[q] != d->qps[q])
        {   // device to device: the row of this qp exists on the device since the first picture that used it
            XH_HIP(hipMemcpyAsync(t->costTable + (size_t)q * (2 * kHalf + 1), t->costRows[d->qps[q]], (size_t)(2 * kHalf + 1) * sizeof(uint16_t), hipMemcpyDeviceToDevice, st));
            t->rowQp[q] = d->qps[q];
        }
    lap(0, true);
    // ---- deriveMVsForCTU's first stage (analysis.cpp:262-299): diamondSearch at range 32 around (0,0) for the CTU (area 0) and its four sub-CUs (areas 1..4), per reference;
    //      m_areaBestMV starts as zero for every area a search does not write; the collocated median, where there is one, replaces all five ----
    // The tasks do not depend on the reference: built once (grouped by CU size and qp: a launch has one PU size and one cost row), uploaded once; every
    // (reference, group) is a launch into its own part of the result array, and m_areaBestMV is assembled on the device -- no round trip inside the picture.
    std::vector<x265hip_me_task>& tasks = t->hTasks; std::vector<int32_t>& where = t->hWhere;
    tasks.clear(); where.clear();
    // (r05: a launch per CU size and reference -- the task names its qp's row of the cost table, xh_diamond_rows -- instead of one per (size, qp, reference): AQ / cuTree
    //  give a picture a dozen qps and more, and the launches were a millisecond of every call)
    struct Group { int size, first, n; };
    std::vector<Group> groups;
    for (int size = t->ctu; size >= t->ctu / 2; size >>= 1)
    {
        const int first = (int)tasks.size();
        for (int c = c0; c < c0 + nCtu; c++)
            for (int a = (size == t->ctu ? 0 : 1); a < (size == t->ctu ? 1 : 5); a++)
            {
                const int cx = (c % t->nCtuX) * t->ctu + (a ? ((a - 1) & 1) * size : 0), cy = (c / t->nCtuX) * t->ctu + (a ? ((a - 1) >> 1) * size : 0);
                x265hip_me_task k{};
                k.curOff = k.refOff = (int32_t)(d->origin + (int64_t)cy * d->stride + cx);
                // Search::setSearchRange(cu, MV(0,0), 32) >> 2 (search.cpp:4969-5021) with CUData::clipMv's limits of this CU
                const int xmin = -((t->ctu + 8 + cx - 1) << 2), ymin = -((t->ctu + 8 + cy - 1) << 2), ymax = (d->height + 8 - cy - 1) << 2;
                int xmax = (d->width + 8 - cx - 1) << 2;
                if (cx / t->ctu < d->pirStartCol) xmax = std::min(xmax, (d->pirSafeX - cx) * 4);       // --intra-refresh (search.cpp:4987-4996): the min is outermost below as it is there
                const int dd = 32 << 2;
                k.mvmin[0] = (int16_t)(std::min(xmax, std::max(xmin, -dd)) >> 2); k.mvmin[1] = (int16_t)(std::min(ymax, std::max(ymin, -dd)) >> 2);
                k.mvmin[1] = (int16_t)std::min((int)k.mvmin[1], refLag);                                  // m_refLagPixels on both ends (search.cpp:5017-5018)
                k.mvmax[0] = (int16_t)(std::min(xmax, std::max(xmin, dd)) >> 2); k.mvmax[1] = (int16_t)(std::max(std::min(std::min(ymax, std::max(ymin, dd)) >> 2, refLag), (int)k.mvmin[1]));
                k.mvpFrom = d->areaQpIndex[c * 5 + a];                                                    // the row of costTable (the order of desc->qps)
                tasks.push_back(k); where.push_back(c * 5 + a);
            }
        if ((int)tasks.size() > first) groups.push_back(Group{ size, first, (int)tasks.size() - first });
    }
    const int nTasks = (int)tasks.size();                                                 // nCtu * 5
    XH_HIP(hipMemcpyAsync(t->dTasks, tasks.data(), (size_t)nTasks * sizeof(x265hip_me_task), hipMemcpyHostToDevice, st));
    XH_HIP(hipMemcpyAsync(t->dWhere, where.data(), (size_t)nTasks * sizeof(int32_t), hipMemcpyHostToDevice, st));
    constexpr size_t kMedianPer = 2 * X265HIP_MAX_REF * 3, kAreaPer = 5 * 2 * X265HIP_MAX_REF * 2;       // int16 per CTU
    if (d->median) XH_HIP(hipMemcpyAsync(t->dMedian + c0 * kMedianPer, d->median + c0 * kMedianPer, (size_t)nCtu * kMedianPer * sizeof(int16_t), hipMemcpyHostToDevice, st));
    XH_HIP(hipMemsetAsync(t->areaBest + c0 * kAreaPer, 0, (size_t)nCtu * kAreaPer * sizeof(int16_t), st));
    for (int l = 0; l < nl; l++)
        for (int r = 0; r < d->numRef[l]; r++)
            for (const Group& g : groups)
                if ((rc = xh_diamond_rows(st, g.size, g.size, t->cur, d->stride, t->plane[l][r][0], d->stride, t->dTasks + g.first, g.n, t->costTable, kHalf,
                                          t->dResults + (size_t)(l * X265HIP_MAX_REF + r) * nTasks + g.first))) return rc;
    if ((rc = xh_tme_area(st, t->dResults, t->dWhere, nTasks, nl, d->numRef[0], d->isP ? 0 : d->numRef[1], d->median ? t->dMedian : nullptr, t->areaBest))) return rc;
    if ((rc = table_up(t->table, d->table))) return rc;
    XH_HIP(hipMemcpyAsync(t->temporal + (size_t)c0 * nS * 2, d->temporal + (size_t)c0 * nS * 2, (size_t)nCtu * nS * 2 * sizeof(x265hip_tme_temporal), hipMemcpyHostToDevice, st));
    XH_HIP(hipMemcpyAsync(t->qpIndex + (size_t)c0 * nS, d->qpIndex + (size_t)c0 * nS, (size_t)nCtu * nS, hipMemcpyHostToDevice, st));
    lap(1, true);
    x265hip_tme_args a{};
    a.isP = d->isP; a.numRef[0] = d->numRef[0]; a.numRef[1] = d->numRef[1]; a.curPOC = d->curPOC; a.temporalMvp = d->temporalMvp;
    std::memcpy(a.refPOC, d->refPOC, sizeof(a.refPOC));
    a.searchRange = d->searchRange; a.searchMethod = d->searchMethod; a.subpelRefine = d->subpelRefine;
    a.picWidth = d->width; a.picHeight = d->height; a.ctuSize = t->ctu; a.lowresBlocksX = d->lowresBlocksX;
    a.refLagPixels = refLag; a.frameParallel = d->frameThreads > 1; a.flags = d->flags;
    a.pirStartCol = d->pirStartCol; a.pirSafeX = d->pirSafeX;
    a.curPlane = t->cur; a.stride = d->stride; a.origin = d->origin; a.planeElems = elems;
    for (int l = 0; l < nl; l++)
        for (int r = 0; r < d->numRef[l]; r++)
        {
            a.refs[l][r].mePlane = t->plane[l][r][0]; a.refs[l][r].mePhase = t->phase[l][r][0]; a.refs[l][r].reconPhase = t->phase[l][r][1];
            a.refs[l][r].refTable = d->refs[l][r].refTable ? t->refTable[l][r] : nullptr; a.refs[l][r].lowresMv = d->refs[l][r].lowresMv ? t->lowres[l][r] : nullptr;
        }
    a.table = t->table; a.areaBest = t->areaBest; a.temporal = t->temporal;
    a.nQp = d->nQp; a.qpIndex = t->qpIndex; a.costHalfRange = kHalf;
    a.costRows = t->costTable;
    for (int q = 0; q < d->nQp; q++) a.lambdas[q] = x265hip_rd_lambda(d->qps[q]);
    a.bitsRow = t->bitsRow; a.bitsHalfRange = kBitsHalf; a.steps = t->steps.data(); a.nSteps = nS; a.workspace = t->workspace; a.workspaceBytes = t->workspaceBytes;
    a.ctuFirst = c0; a.ctuCount = d->ctuRowCount ? nCtu : 0;
    if ((rc = x265hip_tme_frame(st, &a))) return rc;
    lap(2, false);
    if (!t->sparse) XH_HIP(hipMemcpyAsync(d->table + (size_t)c0 * 593, t->table + (size_t)c0 * 593, (size_t)nCtu * 593 * sizeof(x265hip_inter_choice), hipMemcpyDeviceToHost, st));
    else
    {
        const int nU = (int)t->slots.size();
        if ((rc = xh_tme_slots(st, t->table + (size_t)c0 * 593, t->dPacked, t->dSlots, nU, nCtu, 0))) return rc;
        XH_HIP(hipMemcpyAsync(t->hPacked, t->dPacked, (size_t)nCtu * nU * sizeof(x265hip_inter_choice), hipMemcpyDeviceToHost, st));
    }
    if (d->areaBestOut) XH_HIP(hipMemcpyAsync(d->areaBestOut + c0 * kAreaPer, t->areaBest + c0 * kAreaPer, (size_t)nCtu * kAreaPer * sizeof(int16_t), hipMemcpyDeviceToHost, st));
    XH_HIP(hipStreamSynchronize(st));
    if (t->sparse)
    {
        const int nU = (int)t->slots.size();
        for (int c = 0; c < nCtu; c++) for (int k = 0; k < nU; k++) d->table[(size_t)(c0 + c) * 593 + t->slots[k]] = t->hPacked[(size_t)c * nU + k];
    }
    t->tick++;
    lap(3, false); if (t->first) t->pictures++; t->first = true;
    return X265HIP_OK;
}

// Long-lived host buffers of the caller (PicYuv planes, the FrameData tables) can be page-locked once: the producer's copies from / to them are then DMA transfers
// instead of the runtime's staged pageable copies (which bound x265hip_tme_picture at preset medium: ~9 MB per 1080p picture).
extern "C" int x265hip_host_register(void* p, size_t bytes)
{
    if (!p || !bytes) { set_error("host_register: bad arguments"); return X265HIP_EARG; }
    XH_HIP(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return X265HIP_OK;
}
extern "C" int x265hip_host_unregister(void* p)
{
    if (!p) return X265HIP_EARG;
    XH_HIP(hipHostUnregister(p));
    return X265HIP_OK;
}
