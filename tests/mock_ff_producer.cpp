/*
 * mock_ff_producer.cpp -- TEST INFRASTRUCTURE ONLY (tests/test_ff_adapter_cpu.py builds it with g++): the in-loop filter producer's entry points (include/x265hip_ctx.h:
 * x265hip_ff_create / _picture) answered WITHOUT a GPU by the oracle's plain-C deblocking filter and SAO statistics (oracle/x265_oracle.c in oracle/libx265oracle_me_8.so, named
 * by X265MOCK_ORACLE_LIB; pinned to the reference's Deblock / SAO classes by tests/test_deblock_oracle_vs_ref.py, tests/test_sao_oracle_vs_ref.py), so that the host half of the seam --
 * integration/filter_adapter.cpp: the gather of CUData's arrays, the deferral of a picture's filters to its last row, the replay of the encoder's row loop behind the call -- can be
 * driven by the compiled reference encoder (oracle/_ref/x265e2e_8) on the CPU: the encode must write the plain encoder's bitstream.
 * It checks what x265hip_ff_picture checks (the description complete for what is asked) and X265MOCK_FAIL_AT=n makes the n-th call fail.
 */
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <vector>
#include "../include/x265hip_ctx.h"

#if defined(MOCK_DEPTH) && MOCK_DEPTH > 8
typedef uint16_t xo_pixel;     /* -DMOCK_DEPTH=10: the 10-bit encoder with the 10-bit oracle */
#else
typedef uint8_t xo_pixel;      /* the 8-bit encoder */
#endif
struct x265hip_ctx { int device; };
struct x265hip_ff { int width, height, ctu; intptr_t strideY, strideC; long calls = 0, bands = 0, pictures = 0; std::mutex mu; std::map<const void*, std::map<int, int>> rowsDone; std::map<const void*, int> rowsTotal; /* per picture in flight (keyed by its luma plane): per slice (its first row) the rows filtered so far; rows in all */ };

namespace {
char g_err[512] = "";
int fail(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    fprintf(stderr, "mock_ff_producer: PROTOCOL VIOLATION: %s\n", g_err);
    return X265HIP_EARG;
}
/* (x265hip_deblock_pic and the oracle's xo_deblock_pic are the same record: the library's description was modelled on it) */
void (*g_deblock)(const x265hip_deblock_pic*, xo_pixel*, intptr_t, xo_pixel*, xo_pixel*, intptr_t, uint8_t*, int, int);
void (*g_stats)(const xo_pixel*, const xo_pixel*, intptr_t, int, int, int, int, int, int, int32_t*, const uint8_t*, int, int);      /* xo_sao_stats_rows_wh */
} // namespace

extern "C" {
#ifndef MOCK_NO_COMMON      /* (the three mocks in one library -- tests/test_all_adapters_cpu.py -- keep one copy of the context functions: the ThreadedME mock's) */
const char* x265hip_last_error(void) { return g_err; }
int x265hip_ctx_create(int device, x265hip_ctx** out) { *out = new x265hip_ctx{ device }; return X265HIP_OK; }
void x265hip_ctx_destroy(x265hip_ctx* c) { delete c; }
#endif

int x265hip_ff_create(x265hip_ctx* ctx, int width, int height, int ctuSize, intptr_t strideY, intptr_t strideC, x265hip_ff** out)
{
    if (!ctx || !out || width < 8 || height < 8) return fail("ff_create: bad arguments");
    const char* path = getenv("X265MOCK_ORACLE_LIB");
    void* lib = path ? dlopen(path, RTLD_NOW | RTLD_LOCAL) : nullptr;
    if (!lib) return fail("X265MOCK_ORACLE_LIB (%s) does not load: %s", path ? path : "unset", dlerror());
    *(void**)&g_deblock = dlsym(lib, "xo_deblock_rows"); *(void**)&g_stats = dlsym(lib, "xo_sao_stats_rows_wh");      /* (the whole picture is the band of all its rows) */
    if (!g_deblock || !g_stats) return fail("%s lacks xo_deblock_rows / xo_sao_stats_rows_wh", path);
    x265hip_ff* f = new x265hip_ff();
    f->width = width; f->height = height; f->ctu = ctuSize; f->strideY = strideY; f->strideC = strideC;
    *out = f;
    return X265HIP_OK;
}
void x265hip_ff_destroy(x265hip_ff* f)
{
    if (!f) return;
    fprintf(stderr, "mock_ff_producer: %ld pictures in %ld calls, %ld of them bands of a picture\n", f->pictures, f->calls, f->bands);
    if (!f->rowsDone.empty()) fprintf(stderr, "mock_ff_producer: PROTOCOL VIOLATION: %zu pictures were left unfinished\n", f->rowsDone.size());
    delete f;
}

int x265hip_ff_picture(x265hip_ff* f, const x265hip_ff_picture_desc* d)
{
    if (!f || !d) return fail("ff_picture: null argument");
    const x265hip_deblock_pic& P = d->pic;
    if (P.width != f->width || P.height != f->height || P.ctuSize != f->ctu || !d->reconY || !d->reconCb || !d->reconCr) return fail("ff_picture: the picture is not the one the producer was created for");
    if (d->deblock && (!P.log2CUSize || !P.partSize || !P.tuDepth || !P.predMode || !P.cbfLuma || !P.qp || !P.refIdx0 || !P.mv0 || (!P.sliceIsP && (!P.refIdx1 || !P.mv1)) || (P.tqBypassEnabled && !P.tqBypass)))
        return fail("ff_picture: incomplete picture description");
    if ((d->saoStats & 1) && (!d->fencY || !d->stats[0])) return fail("ff_picture: luma statistics without the source plane / the output");
    if ((d->saoStats & 2) && (!d->fencCb || !d->fencCr || !d->stats[1] || !d->stats[2])) return fail("ff_picture: chroma statistics without the source planes / the outputs");
    std::lock_guard<std::mutex> g(f->mu);
    f->calls++;
    if (getenv("X265MOCK_FAIL_AT") && f->calls == atol(getenv("X265MOCK_FAIL_AT"))) return fail("call %ld fails on request (X265MOCK_FAIL_AT)", f->calls);
    const int nrows = (f->height + f->ctu - 1) / f->ctu;
    /* bands of a picture (include/x265hip_ctx.h: desc.ctuRowFirst / ctuRowCount): in increasing order, contiguous, each row once; bands of different pictures may interleave */
    if (d->ctuRowFirst < 0 || d->ctuRowCount < 0 || d->ctuRowFirst + d->ctuRowCount > nrows || (d->ctuRowFirst && !d->ctuRowCount)) return fail("ff_picture: CTU rows %d + %d of %d", d->ctuRowFirst, d->ctuRowCount, nrows);
    const int r0 = d->ctuRowFirst, r1 = d->ctuRowCount ? r0 + d->ctuRowCount : nrows;
    {   /* a picture's bands: inside one slice each, a slice's bands in increasing order and contiguous, every row once; the slices of a picture (and pictures) may interleave */
        int s0 = r0;
        while (s0 > 0 && !(P.sliceFirstRow && P.sliceFirstRow[s0])) s0--;
        for (int r = r0 + 1; r < r1; r++) if (P.sliceFirstRow && P.sliceFirstRow[r] && (r0 > 0 || r1 < nrows)) return fail("ff_picture: the band of CTU rows %d..%d crosses the slice that begins at row %d", r0, r1 - 1, r);
        std::map<int, int>& slices = f->rowsDone[d->reconY];
        const bool whole = r0 == 0 && r1 == nrows;
        if (!whole)
        {
            if (!slices.count(s0)) slices[s0] = s0;
            if (r0 != slices[s0]) return fail("ff_picture: band starts at CTU row %d, the rows done of its slice (first row %d) are %d", r0, s0, slices[s0]);
            slices[s0] = r1;
        }
        int& total = f->rowsTotal[d->reconY];
        total += r1 - r0;
        if (total > nrows) return fail("ff_picture: %d rows of a picture of %d", total, nrows);
        if (total == nrows) { f->rowsDone.erase(d->reconY); f->rowsTotal.erase(d->reconY); f->pictures++; }
    }
    if (d->ctuRowCount) f->bands++;
    if (P.chromaFormat < 0 || P.chromaFormat > 3) return fail("ff_picture: chroma format %d", P.chromaFormat);
    const int hs = P.chromaFormat == 3 ? 0 : 1, vs = (P.chromaFormat == 2 || P.chromaFormat == 3) ? 0 : 1;      /* the chroma planes' subsampling */
    std::vector<uint8_t> sfr;
    x265hip_deblock_pic D = P;
    if (P.sliceFirstRow) { sfr.assign(P.sliceFirstRow, P.sliceFirstRow + nrows); sfr.push_back(0); D.sliceFirstRow = sfr.data(); }
    if (d->deblock) g_deblock(&D, (xo_pixel*)d->reconY, f->strideY, (xo_pixel*)d->reconCb, (xo_pixel*)d->reconCr, f->strideC, nullptr, r0, r1);
    const void* fenc[3] = { d->fencY, d->fencCb, d->fencCr }; void* rec[3] = { d->reconY, d->reconCb, d->reconCr };
    for (int p = 0; p < 3; p++)
        if ((p == 0 && (d->saoStats & 1)) || (p > 0 && (d->saoStats & 2)))
            g_stats((const xo_pixel*)fenc[p], (const xo_pixel*)rec[p], p ? f->strideC : f->strideY, p ? f->width >> hs : f->width, p ? f->height >> vs : f->height, p ? f->ctu >> hs : f->ctu,
                    p ? f->ctu >> vs : f->ctu, d->saoNonDeblocked ? 1 : 0, p ? 2 : 0, d->stats[p], P.sliceFirstRow ? sfr.data() : nullptr, r0, r1);
    return X265HIP_OK;
}
} // extern "C"
