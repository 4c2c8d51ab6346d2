// xh_ff.cpp -- host side of the in-loop filter path for a C / C++ caller (include/x265hip_ctx.h: x265hip_ff_*): one reconstructed picture goes up with CUData's
// per-partition arrays, is deblocked on the device (x265hip_deblock_frame = Deblock::deblockCTU for every CTU, common/deblock.cpp:37-497), the SAO statistics of every
// CTU are collected on the deblocked picture (x265hip_sao_stats_frame = SAO::calcSaoStatsCTU, encoder/sao.cpp:729-905), and the deblocked planes and the statistics come
// back.  What integration/filter_adapter.cpp binds inside the reference encoder (FrameFilter::processRow / ParallelFilter::processTasks, encoder/framefilter.cpp:451-664).
#include "xh_common.h"
#include "../../include/x265hip_ctx.h"
#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>
using namespace xh;

struct x265hip_ff
{
    x265hip_ctx* ctx = nullptr;
    int width = 0, height = 0, ctu = 0, nctu = 0, npart = 0;
    intptr_t strideY = 0, strideC = 0;
    pixel *recon[3] = {}, *fenc[3] = {};                    // device planes, the host's pitches
    uint8_t *log2CUSize = nullptr, *partSize = nullptr, *tuDepth = nullptr, *predMode = nullptr, *cbfLuma = nullptr, *tqBypass = nullptr;
    int8_t *qp = nullptr, *refIdx0 = nullptr, *refIdx1 = nullptr;
    int32_t *mv0 = nullptr, *mv1 = nullptr;
    int32_t* stats[3] = {};
    uint8_t* sliceFirstRow = nullptr; int nrows = 0;   // --slices: device copy of desc.pic.sliceFirstRow (nrows + 1 bytes)
    std::mutex mu;
    std::vector<void*> owned;
    template<class T> int alloc(T*& p, size_t n, const char* file = __builtin_FILE(), int line = __builtin_LINE())
    {
        void* v = nullptr;
        XH_HIP(xh::dev_alloc(&v, n * sizeof(T), xh::alloc_tag(file, line)));
        owned.push_back(v); p = (T*)v;
        return X265HIP_OK;
    }
};

extern "C" int x265hip_ff_create(x265hip_ctx* ctx, int width, int height, int ctuSize, intptr_t strideY, intptr_t strideC, x265hip_ff** out)
{
    if (!ctx || !out || width < 8 || height < 8 || (width & 7) || (height & 7) || width > X265HIP_MAX_PIC_DIM || height > X265HIP_MAX_PIC_DIM ||
        (ctuSize != 16 && ctuSize != 32 && ctuSize != 64) || strideY < width || strideC < width / 2)
    { set_error("ff_create: bad geometry (dimensions are multiples of 8, CTU 16/32/64)"); return X265HIP_EARG; }
    XH_HIP(hipSetDevice(x265hip_ctx_device(ctx)));
    x265hip_ff* f = new (std::nothrow) x265hip_ff();
    if (!f) return X265HIP_EARG;
    f->ctx = ctx; f->width = width; f->height = height; f->ctu = ctuSize; f->strideY = strideY; f->strideC = strideC;
    f->nctu = ((width + ctuSize - 1) / ctuSize) * ((height + ctuSize - 1) / ctuSize); f->npart = (ctuSize / 4) * (ctuSize / 4);
    const size_t n = (size_t)f->nctu * f->npart, ly = (size_t)strideY * height, lc = (size_t)strideC * height;      // chroma planes: up to the luma's height (4:2:2 / 4:4:4)
    int rc = 0;
    for (int p = 0; p < 3 && !rc; p++)
        if (!(rc = f->alloc(f->recon[p], p ? lc : ly)) && !(rc = f->alloc(f->fenc[p], p ? lc : ly))) rc = f->alloc(f->stats[p], (size_t)f->nctu * 320);
    if (!rc) rc = f->alloc(f->log2CUSize, n); if (!rc) rc = f->alloc(f->partSize, n); if (!rc) rc = f->alloc(f->tuDepth, n); if (!rc) rc = f->alloc(f->predMode, n);
    if (!rc) rc = f->alloc(f->cbfLuma, n); if (!rc) rc = f->alloc(f->tqBypass, n); if (!rc) rc = f->alloc(f->qp, n); if (!rc) rc = f->alloc(f->refIdx0, n);
    if (!rc) rc = f->alloc(f->refIdx1, n); if (!rc) rc = f->alloc(f->mv0, 2 * n); if (!rc) rc = f->alloc(f->mv1, 2 * n);
    f->nrows = (height + ctuSize - 1) / ctuSize;
    if (!rc) rc = f->alloc(f->sliceFirstRow, (size_t)f->nrows + 1);
    if (rc) { x265hip_ff_destroy(f); return rc; }
    *out = f;
    return X265HIP_OK;
}

extern "C" void x265hip_ff_destroy(x265hip_ff* f)
{
    if (!f) return;
    (void)hipSetDevice(x265hip_ctx_device(f->ctx));
    (void)hipStreamSynchronize((hipStream_t)x265hip_ctx_stream(f->ctx));
    for (void* p : f->owned) (void)xh::dev_free(p);
    delete f;
}

extern "C" int x265hip_ff_picture(x265hip_ff* f, const x265hip_ff_picture_desc* d)
{
    if (!f || !d) { set_error("ff_picture: null argument"); return X265HIP_EARG; }
    const x265hip_deblock_pic& P = d->pic;
    if (P.width != f->width || P.height != f->height || P.ctuSize != f->ctu || !d->reconY || !d->reconCb || !d->reconCr)
    { set_error("ff_picture: the picture is not the one the producer was created for"); return X265HIP_EARG; }
    // the chroma planes' subsampling (desc.pic.chromaFormat: X265_CSP_I420 = 1 (or 0), I422 = 2, I444 = 3)
    if (P.chromaFormat < 0 || P.chromaFormat > 3) { set_error("ff_picture: chroma format %d", P.chromaFormat); return X265HIP_EARG; }
    const int hs = P.chromaFormat == 3 ? 0 : 1, vs = (P.chromaFormat == 2 || P.chromaFormat == 3) ? 0 : 1;
    if (f->strideC < (f->width >> hs)) { set_error("ff_picture: chroma pitch %ld for %d samples", (long)f->strideC, f->width >> hs); return X265HIP_EARG; }
    if (d->deblock && (!P.log2CUSize || !P.partSize || !P.tuDepth || !P.predMode || !P.cbfLuma || !P.qp || !P.refIdx0 || !P.mv0 || (!P.sliceIsP && (!P.refIdx1 || !P.mv1)) ||
                       (P.tqBypassEnabled && !P.tqBypass)))
    { set_error("ff_picture: incomplete picture description"); return X265HIP_EARG; }
    if ((d->saoStats & 1) && (!d->fencY || !d->stats[0])) { set_error("ff_picture: luma statistics without the source plane / the output"); return X265HIP_EARG; }
    if ((d->saoStats & 2) && (!d->fencCb || !d->fencCr || !d->stats[1] || !d->stats[2])) { set_error("ff_picture: chroma statistics without the source planes / the outputs"); return X265HIP_EARG; }
    const int nRows = f->nrows, nx = (f->width + f->ctu - 1) / f->ctu;
    if (d->ctuRowFirst < 0 || d->ctuRowCount < 0 || d->ctuRowFirst + d->ctuRowCount > nRows || (d->ctuRowFirst && !d->ctuRowCount))
    { set_error("ff_picture: CTU rows %d + %d of %d", d->ctuRowFirst, d->ctuRowCount, nRows); return X265HIP_EARG; }
    // the band (the whole picture without one): CTU rows [r0, r1); moved and computed: the CU arrays from the row above the band, the planes from 8 luma lines above it
    const int r0 = d->ctuRowFirst, r1 = d->ctuRowCount ? r0 + d->ctuRowCount : nRows;
    // the band's top edge is filtered unless the band begins the picture or a slice (desc.pic.sliceFirstRow: the CTUs above are no neighbours).  Only then the row above takes
    // part: its CUs are the P side of the edge, 4 of its luma lines are read and 3 written (8 are moved up, 4 down).  Otherwise NOTHING above the band is touched -- with --slices
    // under frame threads the rows above belong to another slice, whose own bands (and the encoder's SAO behind them) may be working on them at this very moment
    const bool topEdge = r0 > 0 && !(P.sliceFirstRow && P.sliceFirstRow[r0]);
    const int rA = topEdge ? r0 - 1 : r0;
    const int ys = r0 * f->ctu, y0 = topEdge ? ys - 8 : ys, yBack = topEdge ? ys - 4 : ys, y1 = std::min(r1 * f->ctu, f->height);      // luma lines: first of the band, first moved up, first moved back down
    std::lock_guard<std::mutex> g(f->mu);
    XH_HIP(hipSetDevice(x265hip_ctx_device(f->ctx)));
    hipStream_t st = (hipStream_t)x265hip_ctx_stream(f->ctx);
    const size_t a0 = (size_t)rA * nx * f->npart, n = (size_t)(r1 - rA) * nx * f->npart;      // partitions of the CTU rows [rA, r1)
    void* const hostRecon[3] = { d->reconY, d->reconCb, d->reconCr };
    const void* const hostFenc[3] = { d->fencY, d->fencCb, d->fencCr };
    auto pitch = [&](int p) { return (size_t)(p ? f->strideC : f->strideY) * sizeof(pixel); };
    auto wbytes = [&](int p) { return (size_t)(p ? f->width >> hs : f->width) * sizeof(pixel); };
    auto line = [&](int p, int y) { return (size_t)(p ? y >> vs : y); };                      // a luma line's line of plane p (y0 / ys / y1 are even)
    auto at = [&](const void* base, int p, int y) { return (void*)((char*)base + line(p, y) * pitch(p)); };
    // --slices: the rows that begin a slice (a host array like the rest of the description); the entry behind the last row is always 0
    const uint8_t* sfr = nullptr;
    if (P.sliceFirstRow)
    {
        XH_HIP(hipMemcpyAsync(f->sliceFirstRow, P.sliceFirstRow, (size_t)f->nrows, hipMemcpyHostToDevice, st));
        XH_HIP(hipMemsetAsync(f->sliceFirstRow + f->nrows, 0, 1, st));
        sfr = f->sliceFirstRow;
    }
    for (int p = 0; p < 3; p++)
    {
        XH_HIP(hipMemcpy2DAsync(at(f->recon[p], p, y0), pitch(p), at(hostRecon[p], p, y0), pitch(p), wbytes(p), line(p, y1) - line(p, y0), hipMemcpyHostToDevice, st));
        if ((p == 0 && (d->saoStats & 1)) || (p > 0 && (d->saoStats & 2)))
            XH_HIP(hipMemcpy2DAsync(at(f->fenc[p], p, ys), pitch(p), at(hostFenc[p], p, ys), pitch(p), wbytes(p), line(p, y1) - line(p, ys), hipMemcpyHostToDevice, st));
    }
    if (d->deblock)
    {
        x265hip_deblock_pic D = P;
        D.sliceFirstRow = sfr;
#define XF_UP(field, per) XH_HIP(hipMemcpyAsync(f->field + a0 * (per), P.field + a0 * (per), n * (per) * sizeof(*P.field), hipMemcpyHostToDevice, st)); D.field = f->field
        XF_UP(log2CUSize, 1); XF_UP(partSize, 1); XF_UP(tuDepth, 1); XF_UP(predMode, 1); XF_UP(cbfLuma, 1); XF_UP(qp, 1); XF_UP(refIdx0, 1); XF_UP(mv0, 2);
        if (P.tqBypassEnabled) { XF_UP(tqBypass, 1); } else D.tqBypass = nullptr;
        if (!P.sliceIsP) { XF_UP(refIdx1, 1); XF_UP(mv1, 2); } else { D.refIdx1 = nullptr; D.mv1 = nullptr; }
#undef XF_UP
        int rc = x265hip_deblock_rows(st, &D, f->recon[0], f->strideY, f->recon[1], f->recon[2], f->strideC, nullptr, r0, r1);
        if (rc) return rc;
        for (int p = 0; p < 3; p++)
            XH_HIP(hipMemcpy2DAsync(at(hostRecon[p], p, yBack), pitch(p), at(f->recon[p], p, yBack), pitch(p), wbytes(p), line(p, y1) - line(p, yBack), hipMemcpyDeviceToHost, st));
    }
    for (int p = 0; p < 3; p++)
    {
        if (!((p == 0 && (d->saoStats & 1)) || (p > 0 && (d->saoStats & 2)))) continue;
        // a chroma plane: its own width / height / CTU width and height (the format's shifts) and planeOffset 2 (sao.cpp:748-756, :773)
        int rc = x265hip_sao_stats_rows(st, f->fenc[p], f->recon[p], p ? f->strideC : f->strideY, p ? f->width >> hs : f->width, p ? f->height >> vs : f->height,
                                        p ? f->ctu >> hs : f->ctu, p ? f->ctu >> vs : f->ctu, d->saoNonDeblocked ? 1 : 0, p ? 2 : 0, f->stats[p], sfr, r0, r1);
        if (rc) return rc;
        const size_t s0 = (size_t)r0 * nx * 320;
        XH_HIP(hipMemcpyAsync(d->stats[p] + s0, f->stats[p] + s0, (size_t)(r1 - r0) * nx * 320 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    XH_HIP(hipStreamSynchronize(st));
    return X265HIP_OK;
}
