// xh_ctx.cpp -- host side of the frame-batched path for a C / C++ caller (include/x265hip_ctx.h): context, resident planes, pyramid task
// lists, one step = phase planes -> ME 64 / 32 / 16 / 8 -> TQ.  Plain HIP runtime calls around the batch entry points of x265hip_frame.h.
#include "xh_common.h"
#include "../../include/x265hip_ctx.h"
#include "xh_internal.h"
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace xh;

struct x265hip_batch;
struct x265hip_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<x265hip_batch*> batches;      // batches of this context (their sub-streams are joined into `stream` before anything waits on it)
    // The large device blocks (plane stacks, phase planes: hundreds of MB to tens of GB) of a destroyed batch stay with the context and are handed to its next batch instead of
    // going back to the driver: a long-lived host that creates and destroys 8K batches would otherwise unmap and map multi-GB ranges all the time -- slow, and the pattern the
    // runtime's virtual-memory fault of round 5 needs (profiles/r05_fence_flake.txt: hipMemset on a freshly mapped 4.6 GB block after other multi-GB blocks were unmapped).
    // Best fit within +25 %; at most kSpareMax bytes are kept; x265hip_ctx_trim / x265hip_ctx_destroy give them back.  (The fence build keeps every block its own reservation.)
    struct Spare { void* p; size_t bytes; };
    std::vector<Spare> spare; size_t spareBytes = 0;
    static constexpr size_t kSpareMin = (size_t)64 << 20, kSpareMax = (size_t)96 << 30;
    void* take(size_t bytes, size_t* actual)
    {
        if (xh::kFence || bytes < kSpareMin) return nullptr;
        int best = -1;
        for (int i = 0; i < (int)spare.size(); i++)
            if (spare[i].bytes >= bytes && spare[i].bytes <= bytes + bytes / 4 && (best < 0 || spare[i].bytes < spare[best].bytes)) best = i;
        if (best < 0) return nullptr;
        void* p = spare[best].p; spareBytes -= spare[best].bytes; *actual = spare[best].bytes;
        spare.erase(spare.begin() + best);
        return p;
    }
    bool keep(void* p, size_t bytes)
    {
        if (xh::kFence || bytes < kSpareMin || spareBytes + bytes > kSpareMax) return false;
        spare.push_back(Spare{ p, bytes }); spareBytes += bytes;
        return true;
    }
    void trim() { for (const Spare& s : spare) (void)xh::dev_free(s.p); spare.clear(); spareBytes = 0; }
};

namespace {
constexpr int CTU = 64;
const int kLevels[4] = { 64, 32, 16, 8 };
constexpr int kHalf = 1 << 15;                 // MVD cost row: d in [-32768, 32768] quarter-pels
constexpr int kTimingSets = 64;                // event sets kept between two x265hip_batch_read_timing calls
constexpr int kBitsHalf = 1 << 14;             // MVD bit-size row of the per-PU choice among references
constexpr int kRect0 = 4, kAmp0 = 12, kSlots = 24;

bool desc_ok(const x265hip_batch_desc* d)
{
    return d && d->width >= CTU && d->height >= CTU && d->width <= X265HIP_MAX_PIC_DIM && d->height <= X265HIP_MAX_PIC_DIM && d->width % CTU == 0 && d->height % CTU == 0 && d->frames >= 1 && d->margin >= CTU + 16 + 8 &&
           d->margin % 4 == 0 && d->qp >= 0 && d->qp <= 51 && d->merange >= 1 && d->subme >= 0 && d->subme <= 7 && d->tuLog2 >= 2 && d->tuLog2 <= 5 &&
           d->refs >= 0 && d->refs <= X265HIP_MAX_REF && d->refs1 >= 0 && d->refs1 <= X265HIP_MAX_REF && d->streams >= 0 && d->streams <= 8 && d->bandRows >= 0 &&
           ((d->refs <= 1 && d->refs1 == 0) || d->usePlanes)
           && d->bandRows == 0           // the band-major schedule was a measured loss (profiles/r03_band_major_ab.txt) and left the library in round 5
           ;
}
int level_index(int level) { for (int i = 0; i < 4; i++) if (kLevels[i] == level) return i; return -1; }
int64_t stride_of(const x265hip_batch_desc* d) { return d->width + 2 * d->margin; }
int64_t plane_of(const x265hip_batch_desc* d) { return stride_of(d) * (d->height + 2 * d->margin); }
}

struct x265hip_batch
{
    x265hip_ctx* ctx = nullptr;
    x265hip_batch_desc d{};
    int refs = 1, nsub = 1;                     // refs = nref[0]
    int64_t stride = 0, plane = 0;
    pixel *cur = nullptr, *recon = nullptr;
    int nref[2] = { 1, 0 };                                                                    // references searched per list (list 1: B pictures)
    pixel* ref[2][X265HIP_MAX_REF] = {}; pixel* planes[2][X265HIP_MAX_REF] = {};              // per (list, reference): the plane stack and its 16-slot phase planes
    // every searched PU shape has a slot: 0..3 the 2Nx2N PUs of the levels 64 / 32 / 16 / 8; 4..11 the 2NxN / Nx2N PUs (desc.rect: 4 + 2 * level + (0: 2NxN, 1: Nx2N));
    // 12..23 the AMP PUs of the levels 64 / 32 / 16 (desc.amp: 12 + 4 * level + (0: 2N x N/2, 1: 2N x 3N/2, 2: N/2 x 2N, 3: 3N/2 x 2N)).  res[list][ref][slot]: that
    // reference's own chain down the pyramid; choice[slot]: the per-PU choice among references / lists (several references or a B picture)
    x265hip_me_task* tasks[kSlots] = {}; int ntasks[kSlots] = {}; x265hip_me_result* res[2][X265HIP_MAX_REF][kSlots] = {}; x265hip_inter_choice* choice[kSlots] = {};
    bool needChoice = false;
    x265hip_tu_task* tu = nullptr; int ntu = 0, mvLevel = 0;
    int16_t* coeff = nullptr; uint32_t* numSig = nullptr; uint64_t* sse = nullptr;
    uint16_t* costRow = nullptr; float* bitsRow = nullptr; uint64_t lambda = 0;
    // sub-batches of whole pictures on their own streams (desc.streams): stream 0 is the context's
    hipStream_t sub[8] = {}; hipEvent_t evFork = nullptr, evJoin[8] = {};
    // two sub-batches of whole pictures (streams = 2): the 64x64 level (window in LDS, two workgroups fill a CU) of the two streams ALTERNATES -- a stream takes it when
    // the other has finished its own -- so that it always runs beside the other stream's 32x32 .. 8x8 levels, never beside itself; and the sub-batches are not joined
    // between steps: they are joined when something waits on the context's stream (join_subs, x265hip_batch_join).  (A step forks every sub-stream behind what the
    // context's stream holds at that moment -- its own previous pass included -- so sub-stream 1's pass k + 1 starts behind sub-stream 0's pass k; sub-stream 0 never
    // waits for sub-stream 1.)
    hipEvent_t evTok[8] = {}; bool tokSet[8] = {}; int pingpong = 0; bool unjoined = false;
    // The size-specialised kernels address a 16-slot plane buffer with 32-bit byte offsets.  A batch whose buffer is larger than that (8K 10 bit: 1.14 GB per picture) is
    // cut into GROUPS of groupFrames pictures: group g keeps its 16 slots back to back inside the one allocation (slots groupFrames * plane apart), and the pointer the
    // kernels get is biased by the group's first picture, so that the tasks' absolute offsets (picture * plane + ...) address it unchanged.  0 = one group
    int groupFrames = 0; bool groupsForced = false;
    bool ownStart64 = true;                  // STAR: the 64x64 level without its start-stage launch (kern_me_star.hip xh_me_star_own64); x265hip_batch_set_mode(X265HIP_BATCH_START64_LAUNCH) turns it off for A/B
    // per-stage events of sub-batch 0 (x265hip_batch_set_timing)
    bool timing = false; std::vector<std::string> stageNames; std::vector<hipEvent_t> evStage; int timedSteps = 0;      // evStage: kTimingSets sets of 2 events per stage
    std::vector<xh::KernelEvents> evStar; bool starValid[kTimingSets] = {}; int starSteps = 0; const xh::KernelEvents* starNow = nullptr;                // star64_kernel alone, first reference of sub-batch 0 (x265hip_batch_read_kernel_timing)
    std::vector<void*> owned; std::vector<size_t> ownedBytes;
    template<class T> int alloc(T*& p, size_t n, const char* file = __builtin_FILE(), int line = __builtin_LINE())
    {
        size_t got = n * sizeof(T);                              // (a kept block may be up to a quarter larger than asked for: it goes back with its own size)
        void* v = ctx->take(n * sizeof(T), &got);                // a large block of an earlier batch of this context, or a fresh one
        if (!v)
        {
            hipError_t e = xh::dev_alloc(&v, n * sizeof(T), xh::alloc_tag(file, line));
            if (e == hipErrorOutOfMemory && ctx->spareBytes) { (void)hipGetLastError(); ctx->trim(); e = xh::dev_alloc(&v, n * sizeof(T), xh::alloc_tag(file, line)); }      // the kept blocks did not fit this batch: give them back and ask again
            XH_HIP(e);
        }
        owned.push_back(v); ownedBytes.push_back(got); p = (T*)v;
        return X265HIP_OK;
    }
};

// the sub-batches' streams back into the context's stream (a step of the alternating schedule leaves them running)
static int join_subs(x265hip_batch* b)
{
    if (!b->unjoined) return X265HIP_OK;
    for (int s = 1; s < b->nsub; s++)
    {
        XH_HIP(hipEventRecord(b->evJoin[s], b->sub[s]));
        XH_HIP(hipStreamWaitEvent(b->sub[0], b->evJoin[s], 0));
    }
    b->unjoined = false;
    return X265HIP_OK;
}

extern "C" int x265hip_ctx_create(int device, x265hip_ctx** out)
{
    if (!out) { set_error("ctx_create: null output"); return X265HIP_EARG; }
    int n = 0;
    XH_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) { set_error("ctx_create: device %d out of range (%d devices)", device, n); return X265HIP_EDEVICE; }
    XH_HIP(hipSetDevice(device));
    x265hip_ctx* c = new (std::nothrow) x265hip_ctx();
    if (!c) { set_error("ctx_create: out of host memory"); return X265HIP_EARG; }
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreate"); }
    *out = c;
    return X265HIP_OK;
}
extern "C" void x265hip_ctx_destroy(x265hip_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream); x265hip_tme_release_stream(c->stream); (void)hipStreamDestroy(c->stream);
    c->trim();
    delete c;
}
// the device blocks the context keeps from destroyed batches go back to the driver now; returns the bytes released
extern "C" size_t x265hip_ctx_trim(x265hip_ctx* c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    const size_t n = c->spareBytes;
    c->trim();
    return n;
}
extern "C" void* x265hip_ctx_stream(x265hip_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int x265hip_ctx_device(const x265hip_ctx* c) { return c ? c->device : -1; }
extern "C" int x265hip_ctx_sync(x265hip_ctx* c)
{
    if (!c) { set_error("ctx_sync: null context"); return X265HIP_EARG; }
    XH_HIP(hipSetDevice(c->device));
    for (x265hip_batch* b : c->batches) { const int rc = join_subs(b); if (rc) return rc; }
    XH_HIP(hipStreamSynchronize(c->stream));
    return X265HIP_OK;
}

extern "C" int x265hip_batch_task_count(const x265hip_batch_desc* d, int level)
{
    if (!desc_ok(d) || level_index(level) < 0) return X265HIP_EARG;
    return d->frames * (d->width / level) * (d->height / level);
}
extern "C" int x265hip_batch_build_me_tasks(const x265hip_batch_desc* d, int level, x265hip_me_task* out)
{
    if (!desc_ok(d) || level_index(level) < 0 || !out) { set_error("batch_build_me_tasks: bad arguments"); return X265HIP_EARG; }
    const int W = d->width, H = d->height, nx = W / level, ny = H / level, pnx = W / (2 * level), pny = H / (2 * level);
    const int64_t stride = stride_of(d), plane = plane_of(d);
    if (plane * d->frames >= ((int64_t)1 << 31)) { set_error("batch: plane stack beyond 2^31 elements (use fewer frames per batch)"); return X265HIP_EARG; }
    x265hip_me_task* t = out;
    for (int f = 0; f < d->frames; f++)
        for (int by = 0; by < ny; by++)
            for (int bx = 0; bx < nx; bx++, t++)
            {
                const int x = bx * level, y = by * level;
                memset(t, 0, sizeof(*t));
                t->curOff = t->refOff = (int32_t)(f * plane + (int64_t)(d->margin + y) * stride + d->margin + x);
                // CUData::clipMv limits in quarter-pels (offset 8, maxCUSize 64), the window itself is derived on the device (setSearchRange)
                t->mvmin[0] = (int16_t)(-((CTU + 8 + x - 1) << 2)); t->mvmin[1] = (int16_t)(-((CTU + 8 + y - 1) << 2));
                t->mvmax[0] = (int16_t)((W + 8 - x - 1) << 2);      t->mvmax[1] = (int16_t)((H + 8 - y - 1) << 2);
                t->flags = X265HIP_ME_WINDOW;
                t->mvpFrom = level == CTU ? -1 : f * (pnx * pny) + (by / 2) * pnx + (bx / 2);
            }
    return X265HIP_OK;
}
extern "C" int x265hip_batch_tu_count(const x265hip_batch_desc* d)
{
    if (!desc_ok(d)) return X265HIP_EARG;
    return d->frames * (d->width >> d->tuLog2) * (d->height >> d->tuLog2);
}
extern "C" int x265hip_batch_build_tu_tasks(const x265hip_batch_desc* d, x265hip_tu_task* out)
{
    if (!desc_ok(d) || !out) { set_error("batch_build_tu_tasks: bad arguments"); return X265HIP_EARG; }
    const int n = 1 << d->tuLog2, W = d->width, H = d->height, nx = W / n, ny = H / n;
    const int mvLevel = n < 8 ? 8 : n, lnx = W / mvLevel, lny = H / mvLevel;         // pyramid level whose MVs drive the TUs
    const int64_t stride = stride_of(d), plane = plane_of(d);
    x265hip_tu_task* t = out;
    for (int f = 0; f < d->frames; f++)
        for (int by = 0; by < ny; by++)
            for (int bx = 0; bx < nx; bx++, t++)
            {
                const int x = bx * n, y = by * n;
                memset(t, 0, sizeof(*t));
                t->curOff = t->refOff = t->reconOff = (int32_t)(f * plane + (int64_t)(d->margin + y) * stride + d->margin + x);
                t->mvFrom = f * (lnx * lny) + (y / mvLevel) * lnx + (x / mvLevel);
            }
    return X265HIP_OK;
}

// ---- rectangular partitions: for every CU of a pyramid level its two 2NxN and its two Nx2N PUs (g_puLookup, encoder/threadedme.h:67-92), each seeded with the MV of
//      its own CU's 2Nx2N search (mvpFrom indexes that level's results); limits = CUData::clipMv on the CU's position (cudata.cpp:2094-2107) ----
namespace {
// slot -> PU shape; lv = the CU size the shape belongs to.  false: no such shape (AMP splits CUs of 16x16 and up -- maxAMPDepth, slice.h)
bool slot_shape(int slot, int& w, int& h, int& lv)
{
    if (slot < kRect0) { lv = kLevels[slot]; w = h = lv; return true; }
    if (slot < kAmp0) { const int k = slot - kRect0; lv = kLevels[k >> 1]; if (k & 1) { w = lv >> 1; h = lv; } else { w = lv; h = lv >> 1; } return true; }
    const int k = slot - kAmp0; lv = kLevels[k >> 2];
    if (lv < 16) return false;
    switch (k & 3) { case 0: w = lv; h = lv >> 2; break; case 1: w = lv; h = 3 * (lv >> 2); break; case 2: w = lv >> 2; h = lv; break; default: w = 3 * (lv >> 2); h = lv; }
    return true;
}
int amp_slot(int w, int h)
{
    const int lv = w > h ? w : h, li = level_index(lv), s = w > h ? h : w;
    if (li < 0 || lv < 16 || (4 * s != lv && 4 * s != 3 * lv)) return -1;
    return kAmp0 + 4 * li + (w > h ? 0 : 2) + (4 * s == lv ? 0 : 1);
}
}
extern "C" int x265hip_batch_rect_task_count(const x265hip_batch_desc* d, int w, int h)
{
    if (!desc_ok(d) || level_index(w > h ? w : h) < 0 || (w != 2 * h && h != 2 * w)) return X265HIP_EARG;
    return d->frames * (d->width / w) * (d->height / h);
}
extern "C" int x265hip_batch_build_rect_tasks(const x265hip_batch_desc* d, int w, int h, x265hip_me_task* out)
{
    if (x265hip_batch_rect_task_count(d, w, h) < 0 || !out) { set_error("batch_build_rect_tasks: bad arguments"); return X265HIP_EARG; }
    const int W = d->width, H = d->height, lv = w > h ? w : h, nx = W / w, ny = H / h;
    const int64_t stride = stride_of(d), plane = plane_of(d);
    x265hip_me_task* t = out;
    for (int f = 0; f < d->frames; f++)
        for (int by = 0; by < ny; by++)
            for (int bx = 0; bx < nx; bx++, t++)
            {
                const int x = bx * w, y = by * h, cx = (x / lv) * lv, cy = (y / lv) * lv;
                memset(t, 0, sizeof(*t));
                t->curOff = t->refOff = (int32_t)(f * plane + (int64_t)(d->margin + y) * stride + d->margin + x);
                t->mvmin[0] = (int16_t)(-((CTU + 8 + cx - 1) << 2)); t->mvmin[1] = (int16_t)(-((CTU + 8 + cy - 1) << 2));
                t->mvmax[0] = (int16_t)((W + 8 - cx - 1) << 2);      t->mvmax[1] = (int16_t)((H + 8 - cy - 1) << 2);
                t->flags = X265HIP_ME_WINDOW;
                t->mvpFrom = f * ((W / lv) * (H / lv)) + (y / lv) * (W / lv) + (x / lv);
            }
    return X265HIP_OK;
}

// ---- asymmetric partitions (param->bEnableAMP: presets slower and up, param.cpp:592-593; searched at analysis.cpp:2756-2860): every CU of 64, 32 and 16 pixels has the
//      modes 2NxnU (PUs 2N x N/2 over 2N x 3N/2), 2NxnD (2N x 3N/2 over 2N x N/2), nLx2N and nRx2N (the same, left / right) -- g_puLookup, encoder/threadedme.h:67-92.  A shape
//      w x h occurs twice per CU: once as the first PU of a mode (at the CU's origin) and once as the second PU of the sibling mode (behind the larger PU).  Task order of a
//      shape: picture, CU row, occurrence (0: at the CU's origin, 1: the other one), CU column -- a range of CTU rows is a contiguous range of the list.  Each PU is seeded by
//      its CU's 2Nx2N result (mvpFrom), limits as for the rectangles. ----
extern "C" int x265hip_batch_amp_task_count(const x265hip_batch_desc* d, int w, int h)
{
    if (!desc_ok(d) || amp_slot(w, h) < 0) return X265HIP_EARG;
    const int lv = w > h ? w : h;
    return d->frames * (d->width / lv) * (d->height / lv) * 2;
}
extern "C" int x265hip_batch_build_amp_tasks(const x265hip_batch_desc* d, int w, int h, x265hip_me_task* out)
{
    if (x265hip_batch_amp_task_count(d, w, h) < 0 || !out) { set_error("batch_build_amp_tasks: bad arguments"); return X265HIP_EARG; }
    const int W = d->width, H = d->height, lv = w > h ? w : h, nx = W / lv, ny = H / lv;
    const int64_t stride = stride_of(d), plane = plane_of(d);
    x265hip_me_task* t = out;
    for (int f = 0; f < d->frames; f++)
        for (int cy = 0; cy < ny; cy++)
            for (int k = 0; k < 2; k++)
                for (int cx = 0; cx < nx; cx++, t++)
                {   // occurrence 1 starts where the sibling mode's first PU (the complementary size) ends
                    const int x = cx * lv + (k && w < lv ? lv - w : 0), y = cy * lv + (k && h < lv ? lv - h : 0);
                    memset(t, 0, sizeof(*t));
                    t->curOff = t->refOff = (int32_t)(f * plane + (int64_t)(d->margin + y) * stride + d->margin + x);
                    t->mvmin[0] = (int16_t)(-((CTU + 8 + cx * lv - 1) << 2)); t->mvmin[1] = (int16_t)(-((CTU + 8 + cy * lv - 1) << 2));
                    t->mvmax[0] = (int16_t)((W + 8 - cx * lv - 1) << 2);      t->mvmax[1] = (int16_t)((H + 8 - cy * lv - 1) << 2);
                    t->flags = X265HIP_ME_WINDOW;
                    t->mvpFrom = f * (nx * ny) + cy * nx + cx;
                }
    return X265HIP_OK;
}

extern "C" void x265hip_batch_destroy(x265hip_batch* b)
{
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    { auto& v = b->ctx->batches; v.erase(std::remove(v.begin(), v.end(), b), v.end()); }
    (void)hipStreamSynchronize(b->ctx->stream);
    for (int i = 0; i < 8; i++) if (b->evTok[i]) (void)hipEventDestroy(b->evTok[i]);
    for (int i = 1; i < 8; i++) if (b->sub[i]) { (void)hipStreamSynchronize(b->sub[i]); x265hip_tme_release_stream(b->sub[i]); (void)hipStreamDestroy(b->sub[i]); }
    for (int i = 0; i < 8; i++) if (b->evJoin[i]) (void)hipEventDestroy(b->evJoin[i]);
    if (b->evFork) (void)hipEventDestroy(b->evFork);
    for (hipEvent_t e : b->evStage) (void)hipEventDestroy(e);
    for (auto& k : b->evStar) { (void)hipEventDestroy(k.before); (void)hipEventDestroy(k.after); }
    for (size_t i = 0; i < b->owned.size(); i++) if (!b->ctx->keep(b->owned[i], b->ownedBytes[i])) (void)xh::dev_free(b->owned[i]);      // (everything queued on the blocks has finished: the streams were drained above)
    delete b;
}

namespace { int choose_group_frames(const x265hip_batch* b, int forced); }
extern "C" int x265hip_batch_create(x265hip_ctx* ctx, const x265hip_batch_desc* d, x265hip_batch** out)
{
    if (!ctx || !out || !desc_ok(d)) { set_error("batch_create: bad arguments"); return X265HIP_EARG; }
    XH_HIP(hipSetDevice(ctx->device));
    x265hip_batch* b = new (std::nothrow) x265hip_batch();
    if (!b) { set_error("batch_create: out of host memory"); return X265HIP_EARG; }
    b->ctx = ctx; b->d = *d; b->stride = stride_of(d); b->plane = plane_of(d);
    b->refs = d->refs > 1 ? d->refs : 1; b->nref[0] = b->refs; b->nref[1] = d->refs1;
    b->needChoice = b->refs > 1 || b->nref[1] > 0;
    b->nsub = d->streams > 1 ? (d->streams < d->frames ? d->streams : d->frames) : 1;
    b->sub[0] = ctx->stream;
    const size_t elems = (size_t)b->plane * d->frames;
    int rc = X265HIP_OK;
    auto fail = [&](int code) { x265hip_batch_destroy(b); return code; };
#define XB(call) do { rc = (call); if (rc != X265HIP_OK) return fail(rc); } while (0)
#define XBH(call, what) do { if ((call) != hipSuccess) return fail(hip_fail(hipErrorUnknown, what)); } while (0)
    XBH(hipEventCreateWithFlags(&b->evFork, hipEventDisableTiming), "hipEventCreate");
    ctx->batches.push_back(b);
    b->groupFrames = choose_group_frames(b, 0);
    if ((b->nsub == 2 || (b->nsub > 2 && xh_experiment("X265HIP_RING"))) && !xh_experiment("X265HIP_NO_PINGPONG"))
    {
        b->pingpong = 1;                                         // (a batch in plane groups steps every group as a sub-batch of its own and does not pass the token: step_group)
        for (int i = 0; i < b->nsub; i++) XBH(hipEventCreateWithFlags(&b->evTok[i], hipEventDisableTiming), "hipEventCreate");
    }
    for (int i = 0; i < b->nsub; i++)
    {
        if (i) XBH(hipStreamCreateWithFlags(&b->sub[i], hipStreamNonBlocking), "hipStreamCreate");
        XBH(hipEventCreateWithFlags(&b->evJoin[i], hipEventDisableTiming), "hipEventCreate");
    }
    XB(b->alloc(b->cur, elems));
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < b->nref[l]; r++)
        {
            XB(b->alloc(b->ref[l][r], elems));
            if (d->usePlanes) XB(b->alloc(b->planes[l][r], 16 * elems));
        }
    if (d->recon) { XB(b->alloc(b->recon, elems)); XB(b->alloc(b->sse, (size_t)x265hip_batch_tu_count(d))); }
    std::vector<x265hip_me_task> host;
    for (int slot = 0; slot < kSlots; slot++)
    {
        int w, h, lv;
        if (!slot_shape(slot, w, h, lv) || (slot >= kRect0 && slot < kAmp0 && !d->rect) || (slot >= kAmp0 && !d->amp)) continue;
        b->ntasks[slot] = slot < kRect0 ? x265hip_batch_task_count(d, lv) : slot < kAmp0 ? x265hip_batch_rect_task_count(d, w, h) : x265hip_batch_amp_task_count(d, w, h);
        host.resize((size_t)b->ntasks[slot]);
        XB(slot < kRect0 ? x265hip_batch_build_me_tasks(d, lv, host.data()) : slot < kAmp0 ? x265hip_batch_build_rect_tasks(d, w, h, host.data()) : x265hip_batch_build_amp_tasks(d, w, h, host.data()));
        XB(b->alloc(b->tasks[slot], host.size()));
        XBH(hipMemcpy(b->tasks[slot], host.data(), host.size() * sizeof(x265hip_me_task), hipMemcpyHostToDevice), "hipMemcpy(tasks)");
        for (int l = 0; l < 2; l++)
            for (int r = 0; r < b->nref[l]; r++)
            {
                XB(b->alloc(b->res[l][r][slot], host.size()));
                XBH(hipMemset(b->res[l][r][slot], 0, host.size() * sizeof(x265hip_me_result)), "hipMemset(results)");
            }
        if (b->needChoice) XB(b->alloc(b->choice[slot], host.size()));
    }
    b->ntu = x265hip_batch_tu_count(d);
    b->mvLevel = (1 << d->tuLog2) < 8 ? 8 : (1 << d->tuLog2);
    std::vector<x265hip_tu_task> tu((size_t)b->ntu);
    XB(x265hip_batch_build_tu_tasks(d, tu.data()));
    XB(b->alloc(b->tu, tu.size()));
    XBH(hipMemcpy(b->tu, tu.data(), tu.size() * sizeof(x265hip_tu_task), hipMemcpyHostToDevice), "hipMemcpy(tu tasks)");
    XB(b->alloc(b->coeff, (size_t)b->ntu << (2 * d->tuLog2))); XB(b->alloc(b->numSig, (size_t)b->ntu));
    std::vector<uint16_t> row(2 * kHalf + 1);
    XB(x265hip_mvcost_row(d->qp, kHalf, row.data()));
    XB(b->alloc(b->costRow, row.size()));
    XBH(hipMemcpy(b->costRow, row.data(), row.size() * sizeof(uint16_t), hipMemcpyHostToDevice), "hipMemcpy(cost row)");
    if (b->needChoice)
    {   // the per-PU choice among references prices MVDs in bits (BitCost::bitcost) and weighs them with the RD lambda (RDCost::getCost)
        std::vector<float> bits(2 * kBitsHalf + 1);
        XB(x265hip_mvbits_row(kBitsHalf, bits.data()));
        XB(b->alloc(b->bitsRow, bits.size()));
        XBH(hipMemcpy(b->bitsRow, bits.data(), bits.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(bits row)");
        b->lambda = x265hip_rd_lambda(d->qp);
    }
    // the stages of one sub-batch, in launch order
    if (d->usePlanes) b->stageNames.push_back("planes");
    for (int i = 0; i < 4; i++)
    {
        b->stageNames.push_back("me" + std::to_string(kLevels[i]));
        if (d->rect) b->stageNames.push_back("rect" + std::to_string(kLevels[i]));
        if (d->amp && kLevels[i] >= 16) b->stageNames.push_back("amp" + std::to_string(kLevels[i]));
    }
    b->stageNames.push_back("tq");
#undef XB
#undef XBH
    *out = b;
    return X265HIP_OK;
}

extern "C" int x265hip_batch_upload_plane(x265hip_batch* b, int which, int frame, const void* pixels, intptr_t strideElems)
{
    if (!b || which < 0 || which > b->nref[0] + b->nref[1] || frame < 0 || frame >= b->d.frames || !pixels || strideElems < b->d.width)
    { set_error("batch_upload_plane: bad arguments"); return X265HIP_EARG; }
    XH_HIP(hipSetDevice(b->ctx->device));
    { const int jrc = join_subs(b); if (jrc) return jrc; }
    const x265hip_batch_desc& d = b->d;
    pixel* plane = (!which ? b->cur : which <= b->nref[0] ? b->ref[0][which - 1] : b->ref[1][which - 1 - b->nref[0]]) + (size_t)frame * b->plane;
    pixel* org = plane + (size_t)d.margin * b->stride + d.margin;
    hipStream_t st = b->ctx->stream;
    // host rows -> straight into the padded plane, then the borders on the device (extendPicBorder)
    XH_HIP(hipMemcpy2DAsync(org, (size_t)b->stride * sizeof(pixel), pixels, (size_t)strideElems * sizeof(pixel), (size_t)d.width * sizeof(pixel), d.height, hipMemcpyHostToDevice, st));
    return x265hip_extend_pic_border(st, org, b->stride, d.width, d.height, d.margin, d.margin, 1, 0);
}

namespace {
inline int group_frames(const x265hip_batch* b) { return b->groupFrames > 0 ? b->groupFrames : b->d.frames; }
// the end of the group that holds picture f (a range of pictures is cut at these)
inline int group_end(const x265hip_batch* b, int f) { const int G = group_frames(b); return std::min(b->d.frames, (f / G + 1) * G); }
// the 16-slot buffer of the group that holds picture f, as the kernels take it: planeElems = the distance of its slots; the pointer is biased by the group's first picture
// (groups before it fill 16 * f0 * plane elements of the allocation; the tasks' offsets count pictures from the batch's first: - f0 * plane)
inline pixel* group_planes(const x265hip_batch* b, int l, int r, int f, int64_t& planeElems)
{
    const int G = group_frames(b), f0 = f / G * G, n = std::min(G, b->d.frames - f0);
    planeElems = (int64_t)n * b->plane;
    return b->planes[l][r] ? b->planes[l][r] + (size_t)15 * f0 * b->plane : nullptr;
}
// largest group the 32-bit byte offsets reach: the last slot of a group ends at (16 * n + f0) * plane elements from the biased pointer
int choose_group_frames(const x265hip_batch* b, int forced)
{
    const int F = b->d.frames;
    // (and a group's offsets -- its first picture's bias included -- stay below 16 x its slot distance, the bound the LDS-window load of the 32x32 kernel clamps to: f0 <= 15 n)
    auto fits = [&](int G) { const int nLast = F % G ? F % G : G; return ((uint64_t)(16 * G + F) * (uint64_t)b->plane + 4096) * sizeof(pixel) < (1ull << 32) && F - nLast <= 15 * nLast; };
    if (forced > 0) return forced < F ? forced : 0;
    if (!b->d.usePlanes || fits(F)) return 0;
    int G = F;
    while (G > 1 && !fits(G)) G--;
    if (!fits(G)) return 0;                                      // not even one picture: the generic kernels take the batch, as before
    for (int g = G; g >= 1; g--)                                 // prefer a count of groups the sub-batches share evenly
        if (((F + g - 1) / g) % b->nsub == 0 && F % g == 0) return g;
    return G;
}

// the phase planes of pictures f0 .. f1 - 1 (every reference); the range lies inside one group
int planes_range(x265hip_batch* b, int f0, int f1, hipStream_t st)
{
    const x265hip_batch_desc& d = b->d;
    const int rowsPerPic = d.height + 2 * d.margin;
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < b->nref[l]; r++)
        {
            int64_t planeElems;                                          // the 16 phase-plane slots of the group are planeElems apart; a sub-batch addresses its pictures inside them
            pixel* planes = group_planes(b, l, r, f0, planeElems);
            const pixel* src = b->ref[l][r] + (size_t)f0 * b->plane; pixel* dst = planes + (size_t)f0 * b->plane;
            const int rc = x265hip_subpel_planes(st, src, b->stride, (f1 - f0) * rowsPerPic, dst, planeElems);
            if (rc != X265HIP_OK) return rc;
        }
    return X265HIP_OK;
}

// One sub-batch on stream st: the CTU rows g0 .. g1 - 1 of the batch, counted through the pictures (global row g = picture * ctuRows + row).  Tasks of every
// shape are laid out picture-major, then raster: a range of global CTU rows is a contiguous range of every task list.  withPlanes: the range is whole pictures and
// their phase planes are made first; ev != nullptr: events around every stage (2 per stage)
int step_group(x265hip_batch* b, int g0, int g1, bool withPlanes, hipStream_t st, hipEvent_t* ev, int sub);
// (whole pictures: cut at the plane groups; every piece is a sub-batch of its own on the same stream)
int step_range(x265hip_batch* b, int g0, int g1, bool withPlanes, hipStream_t st, hipEvent_t* ev, int sub = -1)
{
    const int ctuRows = b->d.height / CTU;
    if (group_frames(b) >= b->d.frames) return step_group(b, g0, g1, withPlanes, st, ev, sub);
    for (int f = g0 / ctuRows; f < g1 / ctuRows;)
    {
        const int e = std::min(g1 / ctuRows, group_end(b, f));
        const int rc = step_group(b, f * ctuRows, e * ctuRows, withPlanes, st, ev, sub);      // (per-stage events: the last piece's)
        if (rc != X265HIP_OK) return rc;
        f = e;
    }
    return X265HIP_OK;
}
int step_group(x265hip_batch* b, int g0, int g1, bool withPlanes, hipStream_t st, hipEvent_t* ev, int sub)
{
    const x265hip_batch_desc& d = b->d;
    const int ctuRows = d.height / CTU;
    int64_t planeElems = 0;
    pixel* grpPlanes[2][X265HIP_MAX_REF] = {};                          // this group's plane buffers (biased pointers)
    for (int l = 0; l < 2; l++) for (int r = 0; r < b->nref[l]; r++) grpPlanes[l][r] = group_planes(b, l, r, g0 / ctuRows, planeElems);
    const bool up = d.usePlanes != 0;
    int rc, stage = 0;
    auto mark = [&](int end) -> int { if (ev) XH_HIP(hipEventRecord(ev[2 * stage + end], st)); if (end) stage++; return X265HIP_OK; };
    if (up)
    {
        if (withPlanes)
        {
            if ((rc = mark(0))) return rc;
            if ((rc = planes_range(b, g0 / ctuRows, g1 / ctuRows, st)) != X265HIP_OK) return rc;
            if ((rc = mark(1))) return rc;
        }
        else stage++;                                    // (band-major: the planes of the whole batch were made, and timed, before the bands)
    }
    // one task list (slot) searched in every reference of every list (each with its own parent chain), then the per-PU choice (Search::puMotionEstimation's tail,
    // search.cpp:258-556).  parentSlot < 0: the top level
    auto search = [&](int slot, int parentSlot, int first, int n) -> int
    {
        int w, h, lv; slot_shape(slot, w, h, lv);
        const x265hip_me_task* tasks = b->tasks[slot] + first;
        for (int l = 0; l < 2; l++)
            for (int r = 0; r < b->nref[l]; r++)
            {
                x265hip_me_result* out = b->res[l][r][slot] + first;
                const x265hip_me_result* parent = parentSlot >= 0 ? b->res[l][r][parentSlot] : nullptr;
                // star64_kernel alone: its event pair is armed only around a call that can launch it, and never outlives the call (the kernel's launcher clears the pointer
                // when it records the events; a call that took another kernel -- merange beyond the band, an odd stride -- leaves it, and the step is then not counted)
                const xh::KernelEvents* armed = nullptr;
                if (w == CTU && h == CTU && !parent && up && d.method == X265HIP_ME_STAR && l == 0 && r == 0 && ev && b->starNow) { armed = b->starNow; b->starNow = nullptr; xh::tl_star64Events = armed; }
                if (w == CTU && h == CTU && !parent && up && d.method == X265HIP_ME_STAR && b->ownStart64)       // the top level's own tasks: zero predictor, no candidates
                    rc = xh_me_star_own64(st, b->cur, b->stride, b->ref[l][r], b->stride, tasks, n, b->costRow, kHalf, d.merange, d.subme, out, grpPlanes[l][r], planeElems);
                else
                    rc = x265hip_me_batch(st, w, h, b->cur, b->stride, b->ref[l][r], b->stride, tasks, n, b->costRow, kHalf, d.merange, d.method, d.subme, out, parent,
                                          up ? grpPlanes[l][r] : nullptr, up ? planeElems : 0);
                if (armed)
                {
                    if (!xh::tl_star64Events && rc == X265HIP_OK) { b->starValid[armed - b->evStar.data()] = true; b->starSteps++; }
                    xh::tl_star64Events = nullptr;
                }
                if (rc != X265HIP_OK) return rc;
            }
        if (b->needChoice)
        {
            x265hip_merge_params p{};
            p.numRef[0] = b->nref[0]; p.numRef[1] = b->nref[1];
            for (int l = 0; l < 2; l++)
                for (int r = 0; r < b->nref[l]; r++)
                { p.results[l][r] = b->res[l][r][slot] + first; p.mvpSource[l][r] = parentSlot >= 0 ? b->res[l][r][parentSlot] : nullptr; p.subpelPlanes[l][r] = grpPlanes[l][r]; }
            p.planeElems = planeElems; p.bitsRow = b->bitsRow; p.bitsHalfRange = kBitsHalf; p.lambda = b->lambda; p.sourceMaxDim = d.width > d.height ? d.width : d.height;
            // the bidirectional candidate of a B picture exists for the PUs of a split CU -- not for 2Nx2N ("handled elsewhere": checkBidir2Nx2N belongs to the mode decision)
            // and not inside an 8x8 CU (CUData::isBipredRestriction): search.cpp:421-422
            p.bidir = b->nref[1] > 0 && w != h && lv > 8;
            if ((rc = x265hip_inter_merge_batch(st, w, h, b->cur, b->stride, b->stride, tasks, n, &p, b->choice[slot] + first)) != X265HIP_OK) return rc;
        }
        return X265HIP_OK;
    };
    for (int i = 0; i < 4; i++)
    {
        const int lv = kLevels[i], per = (d.width / lv) * (CTU / lv);               // PUs of this level per CTU row
        if ((rc = mark(0))) return rc;
        const bool token = b->pingpong && sub >= 0 && i == 0 && group_frames(b) >= d.frames;
        const int prev = (sub + b->nsub - 1) % b->nsub;                             // the token goes round the streams
        if (token && b->tokSet[prev]) XH_HIP(hipStreamWaitEvent(st, b->evTok[prev], 0));
        if ((rc = search(i, i ? i - 1 : -1, g0 * per, (g1 - g0) * per))) return rc;
        if (token) { XH_HIP(hipEventRecord(b->evTok[sub], st)); b->tokSet[sub] = true; }
        if ((rc = mark(1))) return rc;
        // the PUs of the split forms of this level's CUs: seeded by the CU's own 2Nx2N result in the same reference
        if (d.rect)
        {
            if ((rc = mark(0))) return rc;
            for (int slot = kRect0 + 2 * i; slot < kRect0 + 2 * i + 2; slot++)
            {
                const int k = b->ntasks[slot] / (d.frames * ctuRows);                // PUs of the shape per CTU row
                if ((rc = search(slot, i, g0 * k, (g1 - g0) * k))) return rc;
            }
            if ((rc = mark(1))) return rc;
        }
        if (d.amp && lv >= 16)
        {
            if ((rc = mark(0))) return rc;
            for (int slot = kAmp0 + 4 * i; slot < kAmp0 + 4 * i + 4; slot++)
            {
                const int k = b->ntasks[slot] / (d.frames * ctuRows);
                if ((rc = search(slot, i, g0 * k, (g1 - g0) * k))) return rc;
            }
            if ((rc = mark(1))) return rc;
        }
    }
    const int n = 1 << d.tuLog2, tper = (d.width / n) * (CTU / n), t0 = g0 * tper, nt = (g1 - g0) * tper, mi = level_index(b->mvLevel);
    if ((rc = mark(0))) return rc;
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < b->nref[l]; r++)
        {   // one launch per reference plane: every TU is compensated from the reference its (2Nx2N) PU chose -- list 0 or list 1; a 2Nx2N PU has no bidirectional candidate here
            x265hip_tq_params p{};
            p.qp = d.qp; p.add = 85; p.subpelPlanes = up ? grpPlanes[l][r] : nullptr; p.planeElems = up ? planeElems : 0;
            if (b->needChoice) { p.choice = b->choice[mi]; p.choiceList = l; p.choiceRef = r; }
            rc = x265hip_tq_batch
                                 (st, d.tuLog2, b->cur, b->stride, b->ref[l][r], b->stride, b->tu + t0, nt, &p, b->coeff + ((size_t)t0 << (2 * d.tuLog2)), b->numSig + t0,
                                  d.recon ? b->recon : nullptr, b->stride, d.recon ? b->sse + t0 : nullptr, b->needChoice ? nullptr : b->res[0][0][mi]);
            if (rc != X265HIP_OK) return rc;
        }
    return mark(1);
}
}

// the event pair of this step's star64_kernel launch (first reference of the timed sub-batch)
static int arm_star_events(x265hip_batch* b)
{
    if (b->evStar.size() < (size_t)kTimingSets)
    {
        b->evStar.resize(kTimingSets);
        for (auto& k : b->evStar) { XH_HIP(hipEventCreate(&k.before)); XH_HIP(hipEventCreate(&k.after)); }
    }
    b->starNow = &b->evStar[b->starSteps % kTimingSets];
    return X265HIP_OK;
}

extern "C" int x265hip_batch_step(x265hip_batch* b)
{
    int rc0;
    if (!b) { set_error("batch_step: null batch"); return X265HIP_EARG; }
    XH_HIP(hipSetDevice(b->ctx->device));
    const int F = b->d.frames, S = b->nsub, ctuRows = b->d.height / CTU, G = F * ctuRows;
    hipEvent_t* ev = nullptr;
    if (b->timing)
    {   // the next event set (the sets of the steps since the last read_timing; beyond kTimingSets the oldest are overwritten)
        const size_t per = 2 * b->stageNames.size(), set = (size_t)(b->timedSteps % kTimingSets);
        while (b->evStage.size() < (set + 1) * per) { hipEvent_t e; XH_HIP(hipEventCreate(&e)); b->evStage.push_back(e); }
        ev = b->evStage.data() + set * per;
        b->timedSteps++;
        if ((rc0 = arm_star_events(b))) return rc0;
    }
    if (S == 1) return step_range(b, 0, G, true, b->sub[0], ev);
    int rc;
    {   // Independent pictures: the levels of one picture depend on each other (a level's predictor is its parent CU's MV), pictures do not.  Sub-batch s runs on its
        // own stream, so the LDS-bound 64x64 search of one runs beside the latency-bound 16x16 / 8x8 searches of another.  Everything is ordered after the work already
        // queued on the context's stream and joined back into it.
        XH_HIP(hipEventRecord(b->evFork, b->sub[0]));
        for (int s = 0; s < S; s++)
        {
            if (s) XH_HIP(hipStreamWaitEvent(b->sub[s], b->evFork, 0));
            if ((rc = step_range(b, (F * s / S) * ctuRows, (F * (s + 1) / S) * ctuRows, true, b->sub[s], s == 0 ? ev : nullptr, s)) != X265HIP_OK) return rc;
        }
        if (b->pingpong) { b->unjoined = true; return X265HIP_OK; }               // joined by x265hip_batch_sync
    }
    for (int s = 1; s < S; s++)
    {
        XH_HIP(hipEventRecord(b->evJoin[s], b->sub[s]));
        XH_HIP(hipStreamWaitEvent(b->sub[0], b->evJoin[s], 0));
    }
    return X265HIP_OK;
}

// the whole batch as ONE sub-batch on the context's stream, whatever desc.streams says: the stages then run one after the other, which is what per-stage times need
extern "C" int x265hip_batch_step_one_stream(x265hip_batch* b)
{
    if (!b) { set_error("batch_step_one_stream: null batch"); return X265HIP_EARG; }
    XH_HIP(hipSetDevice(b->ctx->device));
    int rc = join_subs(b);
    if (rc) return rc;
    for (int s = 1; s < b->nsub; s++)
    {   // (a batch without the alternating schedule joins at the end of every step already; this makes the call safe after either)
        XH_HIP(hipEventRecord(b->evJoin[s], b->sub[s]));
        XH_HIP(hipStreamWaitEvent(b->sub[0], b->evJoin[s], 0));
    }
    hipEvent_t* ev = nullptr;
    if (b->timing)
    {
        const size_t per = 2 * b->stageNames.size(), set = (size_t)(b->timedSteps % kTimingSets);
        while (b->evStage.size() < (set + 1) * per) { hipEvent_t e; XH_HIP(hipEventCreate(&e)); b->evStage.push_back(e); }
        ev = b->evStage.data() + set * per;
        b->timedSteps++;
        if ((rc = arm_star_events(b))) return rc;
    }
    rc = step_range(b, 0, b->d.frames * (b->d.height / CTU), true, b->sub[0], ev);
    if (rc) return rc;
    // the sub-streams' next pass starts behind this one
    XH_HIP(hipEventRecord(b->evFork, b->sub[0]));
    for (int s = 1; s < b->nsub; s++) XH_HIP(hipStreamWaitEvent(b->sub[s], b->evFork, 0));
    return X265HIP_OK;
}

// the sub-batches' streams joined into the context's stream: what a GPU-side consumer of x265hip_batch_device_ptr's arrays queues on x265hip_ctx_stream() behind this
// call sees every sub-batch's results (a step of the alternating schedule returns with sub-stream 1 still on its own)
extern "C" int x265hip_batch_join(x265hip_batch* b)
{
    if (!b) { set_error("batch_join: null batch"); return X265HIP_EARG; }
    XH_HIP(hipSetDevice(b->ctx->device));
    return join_subs(b);
}

extern "C" int x265hip_batch_set_mode(x265hip_batch* b, int on);
extern "C" int x265hip_batch_set_fused(x265hip_batch* b, int on) { return x265hip_batch_set_mode(b, on); }
extern "C" int x265hip_batch_set_mode(x265hip_batch* b, int on)
{
    if (!b) return X265HIP_EARG;
    if (on & X265HIP_BATCH_PLANE_GROUPS_OF_2) { b->groupFrames = choose_group_frames(b, 2); b->groupsForced = true; on &= ~X265HIP_BATCH_PLANE_GROUPS_OF_2; }
    else if (b->groupsForced) { b->groupFrames = choose_group_frames(b, 0); b->groupsForced = false; }
    if (on & ~X265HIP_BATCH_START64_LAUNCH) { set_error("batch_set_mode: flags %d: the fused lower levels and the tiled phase planes were measured losses (profiles/r03_fused_ab.txt, r03_tiled_ab.txt) and left the library in round 5", on & ~X265HIP_BATCH_START64_LAUNCH); return X265HIP_EARG; }
    b->ownStart64 = !(on & 4);
    return X265HIP_OK;
}
extern "C" int x265hip_batch_set_timing(x265hip_batch* b, int on) { if (!b) return X265HIP_EARG; b->timing = on != 0; return X265HIP_OK; }
// the longest single kernel of a pass on its own: star64_kernel of the first reference (STAR search with the phase planes, one stream), mean milliseconds over the timed steps
// since the last call; returns the number of steps averaged, 0 when no such launch was timed
extern "C" int x265hip_batch_read_kernel_timing(x265hip_batch* b, float* ms)
{
    if (!b || !ms) { set_error("batch_read_kernel_timing: null argument"); return X265HIP_EARG; }
    *ms = 0;
    if (b->starSteps < 1) return 0;
    XH_HIP(hipSetDevice(b->ctx->device));
    for (int s = 0; s < b->nsub; s++) XH_HIP(hipStreamSynchronize(b->sub[s]));
    const int sets = b->starSteps < kTimingSets ? b->starSteps : kTimingSets;
    double sum = 0;
    int used = 0;
    for (int k = 0; k < sets; k++)
    {
        if (!b->starValid[k]) continue;
        float t = 0; XH_HIP(hipEventElapsedTime(&t, b->evStar[k].before, b->evStar[k].after)); sum += t; used++;
    }
    b->starSteps = 0;
    for (bool& v : b->starValid) v = false;
    if (!used) return 0;
    *ms = (float)(sum / used);
    return used;
}
extern "C" int x265hip_batch_stage_count(const x265hip_batch* b) { return b ? (int)b->stageNames.size() : 0; }
extern "C" const char* x265hip_batch_stage_name(const x265hip_batch* b, int i) { return (b && i >= 0 && i < (int)b->stageNames.size()) ? b->stageNames[i].c_str() : nullptr; }
extern "C" int x265hip_batch_read_timing(x265hip_batch* b, float* ms)
{
    if (!b || !ms || b->timedSteps < 1) { set_error("batch_read_timing: no timed step"); return X265HIP_EARG; }
    XH_HIP(hipSetDevice(b->ctx->device));
    for (int s = 0; s < b->nsub; s++) XH_HIP(hipStreamSynchronize(b->sub[s]));
    const size_t per = 2 * b->stageNames.size();
    const int sets = b->timedSteps < kTimingSets ? b->timedSteps : kTimingSets;
    for (size_t i = 0; i < b->stageNames.size(); i++)
    {
        double sum = 0;
        for (int k = 0; k < sets; k++) { float t = 0; XH_HIP(hipEventElapsedTime(&t, b->evStage[k * per + 2 * i], b->evStage[k * per + 2 * i + 1])); sum += t; }
        ms[i] = (float)(sum / sets);
    }
    b->timedSteps = 0;
    return sets;
}

namespace {
// the slot of a searched shape of THIS batch, or -1
int shape_slot(const x265hip_batch* b, int w, int h)
{
    int slot = -1;
    if (w == h) slot = level_index(w);
    else if (level_index(w > h ? w : h) >= 0 && (w == 2 * h || h == 2 * w)) slot = kRect0 + 2 * level_index(w > h ? w : h) + (h > w ? 1 : 0);
    else slot = amp_slot(w, h);
    return (slot >= 0 && b->tasks[slot]) ? slot : -1;
}
}
extern "C" int x265hip_batch_read_results(x265hip_batch* b, int level, x265hip_me_result* out) { return x265hip_batch_read_results_list(b, level, level, 0, 0, out); }
extern "C" int x265hip_batch_read_results_ref(x265hip_batch* b, int w, int h, int ref, x265hip_me_result* out) { return x265hip_batch_read_results_list(b, w, h, 0, ref, out); }
extern "C" int x265hip_batch_read_results_list(x265hip_batch* b, int w, int h, int list, int ref, x265hip_me_result* out)
{
    const int i = b ? shape_slot(b, w, h) : -1;
    if (!b || i < 0 || list < 0 || list > 1 || ref < 0 || ref >= b->nref[list] || !out) { set_error("batch_read_results: bad arguments (shape %dx%d, list %d, reference %d)", w, h, list, ref); return X265HIP_EARG; }
    XH_HIP(hipSetDevice(b->ctx->device));
    { const int jrc = join_subs(b); if (jrc) return jrc; }
    XH_HIP(hipMemcpyAsync(out, b->res[list][ref][i], (size_t)b->ntasks[i] * sizeof(x265hip_me_result), hipMemcpyDeviceToHost, b->ctx->stream));
    XH_HIP(hipStreamSynchronize(b->ctx->stream));
    return X265HIP_OK;
}
extern "C" int x265hip_batch_read_choices(x265hip_batch* b, int w, int h, x265hip_inter_choice* out)
{
    const int i = b ? shape_slot(b, w, h) : -1;
    if (!b || i < 0 || !b->needChoice || !out) { set_error("batch_read_choices: bad arguments (choices exist with refs > 1 or refs1 > 0)"); return X265HIP_EARG; }
    XH_HIP(hipSetDevice(b->ctx->device));
    { const int jrc = join_subs(b); if (jrc) return jrc; }
    XH_HIP(hipMemcpyAsync(out, b->choice[i], (size_t)b->ntasks[i] * sizeof(x265hip_inter_choice), hipMemcpyDeviceToHost, b->ctx->stream));
    XH_HIP(hipStreamSynchronize(b->ctx->stream));
    return X265HIP_OK;
}
extern "C" int x265hip_batch_read_coeffs(x265hip_batch* b, int16_t* coeff, uint32_t* numSig)
{
    if (!b || (!coeff && !numSig)) { set_error("batch_read_coeffs: bad arguments"); return X265HIP_EARG; }
    XH_HIP(hipSetDevice(b->ctx->device));
    { const int jrc = join_subs(b); if (jrc) return jrc; }
    if (coeff) XH_HIP(hipMemcpyAsync(coeff, b->coeff, ((size_t)b->ntu << (2 * b->d.tuLog2)) * sizeof(int16_t), hipMemcpyDeviceToHost, b->ctx->stream));
    if (numSig) XH_HIP(hipMemcpyAsync(numSig, b->numSig, (size_t)b->ntu * sizeof(uint32_t), hipMemcpyDeviceToHost, b->ctx->stream));
    XH_HIP(hipStreamSynchronize(b->ctx->stream));
    return X265HIP_OK;
}
extern "C" int x265hip_batch_read_plane(x265hip_batch* b, int which, int frame, void* out)
{
    if (!b || which < 0 || which > b->nref[0] + b->nref[1] || frame < 0 || frame >= b->d.frames || !out) { set_error("batch_read_plane: bad arguments"); return X265HIP_EARG; }
    XH_HIP(hipSetDevice(b->ctx->device));
    { const int jrc = join_subs(b); if (jrc) return jrc; }
    const pixel* plane = (!which ? b->cur : which <= b->nref[0] ? b->ref[0][which - 1] : b->ref[1][which - 1 - b->nref[0]]) + (size_t)frame * b->plane;
    XH_HIP(hipMemcpyAsync(out, plane, (size_t)b->plane * sizeof(pixel), hipMemcpyDeviceToHost, b->ctx->stream));
    XH_HIP(hipStreamSynchronize(b->ctx->stream));
    return X265HIP_OK;
}
extern "C" void* x265hip_batch_device_ptr(x265hip_batch* b, int what)
{
    if (!b) return nullptr;
    if (what >= 100 && what < 100 + b->nref[0]) return b->ref[0][what - 100];
    if (what >= 200 && what < 200 + b->nref[0]) return b->planes[0][what - 200];
    if (what >= 300 && what < 300 + b->nref[1]) return b->ref[1][what - 300];
    if (what >= 400 && what < 400 + b->nref[1]) return b->planes[1][what - 400];
    switch (what)
    {
    case 0: return b->cur; case 1: return b->ref[0][0]; case 2: return b->planes[0][0]; case 3: return b->coeff; case 4: return b->numSig; case 5: return b->recon;
    case 10: return b->res[0][0][3]; case 11: return b->res[0][0][2]; case 12: return b->res[0][0][1]; case 13: return b->res[0][0][0];
    default: return nullptr;
    }
}
