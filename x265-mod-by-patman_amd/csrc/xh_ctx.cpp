// xh_ctx.cpp -- host side of the frame-batched path for a C / C++ caller (include/x265hip_ctx.h): context, resident planes, pyramid task
// lists, one step = phase planes -> ME 64 / 32 / 16 / 8 -> TQ.  Plain HIP runtime calls around the batch entry points of x265hip_frame.h.
#include "xh_common.h"
#include "../../include/x265hip_ctx.h"
#include <cstring>
#include <new>
#include <vector>

using namespace xh;

struct x265hip_ctx
{
    int device = 0;
    hipStream_t stream = nullptr;
};

namespace {
constexpr int CTU = 64;
const int kLevels[4] = { 64, 32, 16, 8 };
constexpr int kHalf = 1 << 15;                 // MVD cost row: d in [-32768, 32768] quarter-pels

bool desc_ok(const x265hip_batch_desc* d)
{
    return d && d->width >= CTU && d->height >= CTU && d->width <= X265HIP_MAX_PIC_DIM && d->height <= X265HIP_MAX_PIC_DIM && d->width % CTU == 0 && d->height % CTU == 0 && d->frames >= 1 && d->margin >= CTU + 16 + 8 &&
           d->margin % 4 == 0 && d->qp >= 0 && d->qp <= 51 && d->merange >= 1 && d->subme >= 0 && d->subme <= 7 && d->tuLog2 >= 2 && d->tuLog2 <= 5;
}
int level_index(int level) { for (int i = 0; i < 4; i++) if (kLevels[i] == level) return i; return -1; }
int64_t stride_of(const x265hip_batch_desc* d) { return d->width + 2 * d->margin; }
int64_t plane_of(const x265hip_batch_desc* d) { return stride_of(d) * (d->height + 2 * d->margin); }
}

struct x265hip_batch
{
    x265hip_ctx* ctx = nullptr;
    x265hip_batch_desc d{};
    int64_t stride = 0, plane = 0;
    pixel *cur = nullptr, *ref = nullptr, *planes = nullptr, *recon = nullptr;
    x265hip_me_task* tasks[4] = {}; x265hip_me_result* results[4] = {}; int ntasks[4] = {};
    x265hip_tu_task* tu = nullptr; int ntu = 0, mvLevel = 0;
    int16_t* coeff = nullptr; uint32_t* numSig = nullptr; uint64_t* sse = nullptr;
    uint16_t* costRow = nullptr;
    std::vector<void*> owned;
    template<class T> int alloc(T*& p, size_t n)
    {
        void* v = nullptr;
        XH_HIP(hipMalloc(&v, n * sizeof(T)));
        owned.push_back(v); p = (T*)v;
        return X265HIP_OK;
    }
};

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
    delete c;
}
extern "C" void* x265hip_ctx_stream(x265hip_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int x265hip_ctx_device(const x265hip_ctx* c) { return c ? c->device : -1; }
extern "C" int x265hip_ctx_sync(x265hip_ctx* c)
{
    if (!c) { set_error("ctx_sync: null context"); return X265HIP_EARG; }
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

extern "C" void x265hip_batch_destroy(x265hip_batch* b)
{
    if (!b) return;
    (void)hipStreamSynchronize(b->ctx->stream);
    for (void* p : b->owned) (void)hipFree(p);
    delete b;
}

extern "C" int x265hip_batch_create(x265hip_ctx* ctx, const x265hip_batch_desc* d, x265hip_batch** out)
{
    if (!ctx || !out || !desc_ok(d)) { set_error("batch_create: bad arguments"); return X265HIP_EARG; }
    XH_HIP(hipSetDevice(ctx->device));
    x265hip_batch* b = new (std::nothrow) x265hip_batch();
    if (!b) { set_error("batch_create: out of host memory"); return X265HIP_EARG; }
    b->ctx = ctx; b->d = *d; b->stride = stride_of(d); b->plane = plane_of(d);
    const size_t elems = (size_t)b->plane * d->frames;
    int rc = X265HIP_OK;
    auto fail = [&](int code) { x265hip_batch_destroy(b); return code; };
#define XB(call) do { rc = (call); if (rc != X265HIP_OK) return fail(rc); } while (0)
    XB(b->alloc(b->cur, elems)); XB(b->alloc(b->ref, elems));
    if (d->usePlanes) XB(b->alloc(b->planes, 16 * elems));
    if (d->recon) { XB(b->alloc(b->recon, elems)); XB(b->alloc(b->sse, (size_t)x265hip_batch_tu_count(d))); }
    std::vector<x265hip_me_task> host;
    for (int i = 0; i < 4; i++)
    {
        b->ntasks[i] = x265hip_batch_task_count(d, kLevels[i]);
        host.resize((size_t)b->ntasks[i]);
        XB(x265hip_batch_build_me_tasks(d, kLevels[i], host.data()));
        XB(b->alloc(b->tasks[i], host.size())); XB(b->alloc(b->results[i], host.size()));
        if (hipMemcpy(b->tasks[i], host.data(), host.size() * sizeof(x265hip_me_task), hipMemcpyHostToDevice) != hipSuccess) return fail(hip_fail(hipErrorUnknown, "hipMemcpy(tasks)"));
        if (hipMemset(b->results[i], 0, host.size() * sizeof(x265hip_me_result)) != hipSuccess) return fail(hip_fail(hipErrorUnknown, "hipMemset(results)"));
    }
    b->ntu = x265hip_batch_tu_count(d);
    b->mvLevel = (1 << d->tuLog2) < 8 ? 8 : (1 << d->tuLog2);
    std::vector<x265hip_tu_task> tu((size_t)b->ntu);
    XB(x265hip_batch_build_tu_tasks(d, tu.data()));
    XB(b->alloc(b->tu, tu.size()));
    if (hipMemcpy(b->tu, tu.data(), tu.size() * sizeof(x265hip_tu_task), hipMemcpyHostToDevice) != hipSuccess) return fail(hip_fail(hipErrorUnknown, "hipMemcpy(tu tasks)"));
    XB(b->alloc(b->coeff, (size_t)b->ntu << (2 * d->tuLog2))); XB(b->alloc(b->numSig, (size_t)b->ntu));
    std::vector<uint16_t> row(2 * kHalf + 1);
    XB(x265hip_mvcost_row(d->qp, kHalf, row.data()));
    XB(b->alloc(b->costRow, row.size()));
    if (hipMemcpy(b->costRow, row.data(), row.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess) return fail(hip_fail(hipErrorUnknown, "hipMemcpy(cost row)"));
#undef XB
    *out = b;
    return X265HIP_OK;
}

extern "C" int x265hip_batch_upload_plane(x265hip_batch* b, int which, int frame, const void* pixels, intptr_t strideElems)
{
    if (!b || (which != 0 && which != 1) || frame < 0 || frame >= b->d.frames || !pixels || strideElems < b->d.width)
    { set_error("batch_upload_plane: bad arguments"); return X265HIP_EARG; }
    const x265hip_batch_desc& d = b->d;
    pixel* plane = (which ? b->ref : b->cur) + (size_t)frame * b->plane;
    pixel* org = plane + (size_t)d.margin * b->stride + d.margin;
    hipStream_t st = b->ctx->stream;
    // host rows -> straight into the padded plane, then the borders on the device (extendPicBorder)
    XH_HIP(hipMemcpy2DAsync(org, (size_t)b->stride * sizeof(pixel), pixels, (size_t)strideElems * sizeof(pixel), (size_t)d.width * sizeof(pixel), d.height, hipMemcpyHostToDevice, st));
    return x265hip_extend_pic_border(st, org, b->stride, d.width, d.height, d.margin, d.margin, 1, 0);
}

extern "C" int x265hip_batch_step(x265hip_batch* b)
{
    if (!b) { set_error("batch_step: null batch"); return X265HIP_EARG; }
    const x265hip_batch_desc& d = b->d;
    hipStream_t st = b->ctx->stream;
    const int64_t planeElems = b->plane * d.frames;
    int rc;
    if (d.usePlanes && (rc = x265hip_subpel_planes(st, b->ref, b->stride, d.frames * (d.height + 2 * d.margin), b->planes, planeElems)) != X265HIP_OK) return rc;
    for (int i = 0; i < 4; i++)
    {
        const int lv = kLevels[i];
        rc = x265hip_me_batch(st, lv, lv, b->cur, b->stride, b->ref, b->stride, b->tasks[i], b->ntasks[i], b->costRow, kHalf, d.merange, d.method, d.subme,
                              b->results[i], i ? b->results[i - 1] : nullptr, d.usePlanes ? b->planes : nullptr, d.usePlanes ? planeElems : 0);
        if (rc != X265HIP_OK) return rc;
    }
    x265hip_tq_params p{};
    p.qp = d.qp; p.add = 85; p.quantCoeff = nullptr; p.deltaU = nullptr; p.subpelPlanes = d.usePlanes ? b->planes : nullptr; p.planeElems = d.usePlanes ? planeElems : 0;
    return x265hip_tq_batch(st, d.tuLog2, b->cur, b->stride, b->ref, b->stride, b->tu, b->ntu, &p, b->coeff, b->numSig, d.recon ? b->recon : nullptr, b->stride,
                            d.recon ? b->sse : nullptr, b->results[level_index(b->mvLevel)]);
}

extern "C" int x265hip_batch_read_results(x265hip_batch* b, int level, x265hip_me_result* out)
{
    const int i = level_index(level);
    if (!b || i < 0 || !out) { set_error("batch_read_results: bad arguments"); return X265HIP_EARG; }
    XH_HIP(hipMemcpyAsync(out, b->results[i], (size_t)b->ntasks[i] * sizeof(x265hip_me_result), hipMemcpyDeviceToHost, b->ctx->stream));
    XH_HIP(hipStreamSynchronize(b->ctx->stream));
    return X265HIP_OK;
}
extern "C" int x265hip_batch_read_coeffs(x265hip_batch* b, int16_t* coeff, uint32_t* numSig)
{
    if (!b || (!coeff && !numSig)) { set_error("batch_read_coeffs: bad arguments"); return X265HIP_EARG; }
    if (coeff) XH_HIP(hipMemcpyAsync(coeff, b->coeff, ((size_t)b->ntu << (2 * b->d.tuLog2)) * sizeof(int16_t), hipMemcpyDeviceToHost, b->ctx->stream));
    if (numSig) XH_HIP(hipMemcpyAsync(numSig, b->numSig, (size_t)b->ntu * sizeof(uint32_t), hipMemcpyDeviceToHost, b->ctx->stream));
    XH_HIP(hipStreamSynchronize(b->ctx->stream));
    return X265HIP_OK;
}
extern "C" void* x265hip_batch_device_ptr(x265hip_batch* b, int what)
{
    if (!b) return nullptr;
    switch (what)
    {
    case 0: return b->cur; case 1: return b->ref; case 2: return b->planes; case 3: return b->coeff; case 4: return b->numSig; case 5: return b->recon;
    case 10: return b->results[3]; case 11: return b->results[2]; case 12: return b->results[1]; case 13: return b->results[0];
    default: return nullptr;
    }
}
