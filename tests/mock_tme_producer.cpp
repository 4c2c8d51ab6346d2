/*
 * mock_tme_producer.cpp -- TEST INFRASTRUCTURE ONLY (tests/test_tme_adapter_cpu.py builds it with g++): a stand-in for the ThreadedME producer's six entry points that runs
 * WITHOUT a GPU, so that the host half of the seam -- integration/tme_adapter.cpp: the band protocol under frame threads, the harvest, the write-back, the workers' queues --
 * can be driven by the compiled reference encoder (oracle/_ref/x265tmegpu_8) on the CPU.  It searches nothing.  What it does:
 *
 *   - CHECKS the protocol of include/x265hip_ctx.h on every call: one call at a time; the bands of a picture arrive in order, contiguous, never twice; the valid-row counts
 *     cover what the band's searches may read (the encoder's row-lag rule; a band declares what ITS searches need, a later band of another picture may declare fewer rows of
 *     the same plane); array pointers there.  A violation is printed and the call fails.
 *   - Writes, for every CTU of the band and every slot of the schedule, a record that is a pure function of WHAT THE ADAPTER HANDED OVER FOR THAT CTU: the qps of its entries,
 *     its own pixels, its collocated neighbours and medians, the reference tables' records, the lookahead's MVs of its blocks, and the reference planes' rows the CTU's searches could read (rows the caller declared final).
 *     Two encodes of the same clip therefore write the same bitstream whatever the bands were and however the threads interleaved -- unless the adapter handed over state that
 *     was not final yet (a row still being reconstructed, a table still being written): then the hash, the record's MV and the bitstream move.  The records are legal for the
 *     encoder (list 0, reference 0, a quarter-pel MV within +-3).
 *   - X265MOCK_FAIL_AT=n: the n-th call fails (the adapter must end the encode at once, loudly).
 *   - Sleeps X265MOCK_CALL_US + X265MOCK_ROW_US x (CTU rows of the band) microseconds per call: the producer as the host sees it -- a latency -- for timing the adapter's
 *     scheduling on CPUs alone.
 *
 * The schedule is the real one: x265hip_tme_schedule of the library named by X265MOCK_REAL_LIB (host code; the library loads without a GPU).
 */
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include "../include/x265hip_ctx.h"
#ifndef MOCK_PIXEL_BYTES
#define MOCK_PIXEL_BYTES 1      /* the 8-bit encoder; -DMOCK_PIXEL_BYTES=2 for a 10-bit one */
#endif

struct x265hip_ctx { int device; };
struct x265hip_tme
{
    int width, height, ctu, nCtuX, nCtuY;
    std::vector<x265hip_tme_step> steps;
    std::vector<int> slots;
    std::map<int, int> rowsDone;                 /* per POC: CTU rows that have their records */
    long calls = 0, bands = 0, pirPictures = 0;
};

namespace {
char g_err[512] = "";
std::atomic<int> g_inCall{0};
int g_callUs = 0, g_rowUs = 0;
int fail(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    fprintf(stderr, "mock_tme_producer: PROTOCOL VIOLATION: %s\n", g_err);
    return X265HIP_EARG;
}
inline uint64_t mix(uint64_t h, uint64_t v) { h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2); return h * 0xFF51AFD7ED558CCDull; }
uint64_t hash_bytes(uint64_t h, const void* p, size_t n)
{
    const uint8_t* b = (const uint8_t*)p;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t v; memcpy(&v, b + i, 8); h = mix(h, v); }
    uint64_t v = 0; if (i < n) { memcpy(&v, b + i, n - i); h = mix(h, v); }
    return h;
}
} // namespace

extern "C" {
const char* x265hip_last_error(void) { return g_err; }
int x265hip_ctx_create(int device, x265hip_ctx** out) { *out = new x265hip_ctx{ device }; return X265HIP_OK; }
void x265hip_ctx_destroy(x265hip_ctx* c) { delete c; }

int x265hip_tme_create(x265hip_ctx* ctx, int width, int height, int ctuSize, int minCuSize, int rect, int amp, x265hip_tme** out)
{
    if (!ctx || !out) return fail("tme_create: null arguments");
    const char* real = getenv("X265MOCK_REAL_LIB");
    void* lib = real ? dlopen(real, RTLD_NOW | RTLD_LOCAL) : nullptr;
    if (!lib) return fail("tme_create: X265MOCK_REAL_LIB (%s) does not load: %s", real ? real : "unset", dlerror());
    typedef int (*sched_t)(int, int, int, int, x265hip_tme_step*, int);
    sched_t sched = (sched_t)dlsym(lib, "x265hip_tme_schedule");
    if (!sched) return fail("tme_create: no x265hip_tme_schedule in %s", real);
    const int n = sched(ctuSize, minCuSize, rect, amp, nullptr, 0);
    if (n <= 0) return fail("tme_create: bad CTU / CU sizes");
    x265hip_tme* t = new x265hip_tme();
    t->width = width; t->height = height; t->ctu = ctuSize; t->nCtuX = (width + ctuSize - 1) / ctuSize; t->nCtuY = (height + ctuSize - 1) / ctuSize;
    t->steps.resize(n);
    sched(ctuSize, minCuSize, rect, amp, t->steps.data(), n);
    std::vector<char> used(593, 0);
    for (const x265hip_tme_step& e : t->steps) for (int pi = 0; pi < e.numPart; pi++) { const int sl = e.finalIdx + pi * e.puOffset; if (sl >= 0 && sl < 593) used[sl] = 1; }
    for (int sl = 0; sl < 593; sl++) if (used[sl]) t->slots.push_back(sl);
    if (getenv("X265MOCK_CALL_US")) g_callUs = atoi(getenv("X265MOCK_CALL_US"));
    if (getenv("X265MOCK_ROW_US")) g_rowUs = atoi(getenv("X265MOCK_ROW_US"));
    *out = t;
    return X265HIP_OK;
}
void x265hip_tme_destroy(x265hip_tme* t)
{
    if (!t) return;
    fprintf(stderr, "mock_tme_producer: %ld calls, %ld of them bands of a picture\n", t->calls, t->bands);
    if (t->pirPictures) fprintf(stderr, "mock_tme_producer: %ld pictures with an intra-refresh window limit\n", t->pirPictures);
    delete t;
}
int x265hip_tme_entries(const x265hip_tme* t, const x265hip_tme_step** steps) { if (!t) return 0; if (steps) *steps = t->steps.data(); return (int)t->steps.size(); }

int x265hip_tme_picture(x265hip_tme* t, const x265hip_tme_picture_desc* d)
{
    if (g_inCall.fetch_add(1) != 0) { g_inCall.fetch_sub(1); return fail("two x265hip_tme_picture calls at once on one producer"); }
    struct Leave { ~Leave() { g_inCall.fetch_sub(1); } } leave;
    if (!t || !d || !d->curPlane || !d->table || !d->temporal || !d->qpIndex || !d->areaQpIndex || d->nQp < 1 || d->nQp > 64) return fail("tme_picture: missing arrays");
    if (d->width != t->width || d->height != t->height) return fail("tme_picture: %dx%d picture on a %dx%d producer", d->width, d->height, t->width, t->height);
    const int nl = d->isP ? 1 : 2, nS = (int)t->steps.size();
    for (int l = 0; l < nl; l++) if (d->numRef[l] < 1 || d->numRef[l] > X265HIP_MAX_REF) return fail("tme_picture: %d references in list %d", d->numRef[l], l);
    if (d->ctuRowFirst < 0 || d->ctuRowCount < 0 || d->ctuRowFirst + d->ctuRowCount > t->nCtuY || (d->ctuRowFirst && !d->ctuRowCount)) return fail("tme_picture: CTU rows %d + %d of %d", d->ctuRowFirst, d->ctuRowCount, t->nCtuY);
    const int row0 = d->ctuRowFirst, row1 = d->ctuRowCount ? row0 + d->ctuRowCount : t->nCtuY;
    /* --intra-refresh (search.cpp:4987-4996): only P pictures carry the limit; the safe column lies at or right of the picture's own refresh column (encoder.cpp:1095-1106) */
    if (d->pirStartCol < 0 || d->pirStartCol > t->nCtuX || (d->pirStartCol && (!d->isP || d->pirSafeX < d->pirStartCol * t->ctu - 3 || d->pirSafeX >= (t->nCtuX + 1) * t->ctu)))
        return fail("POC %d: intra-refresh fields: start column %d, safe x %d (%s picture, %d CTU columns)", d->curPOC, d->pirStartCol, d->pirSafeX, d->isP ? "P" : "B", t->nCtuX);
    if (d->pirStartCol) t->pirPictures += row0 == 0;
    t->calls++; if (d->ctuRowCount) t->bands++;
    if (getenv("X265MOCK_FAIL_AT") && t->calls == atol(getenv("X265MOCK_FAIL_AT"))) return fail("call %ld fails on request (X265MOCK_FAIL_AT)", t->calls);
    /* bands of a picture: in order, contiguous, once */
    int& done = t->rowsDone[d->curPOC];
    if (row0 != done) return fail("POC %d: band starts at CTU row %d, rows done %d", d->curPOC, row0, done);
    done = row1;
    if (done == t->nCtuY) t->rowsDone.erase(d->curPOC);            /* (the POC comes round again in the next GOP of a long clip) */
    const int rows = (int)(d->planeElems / d->stride), top = (int)(d->origin / d->stride);
    /* what the band's searches may read of a reference: rows down to the last band row's window (search range + sub-pel taps); the caller must have declared them final */
    const bool parallel = d->frameThreads > 1;
    const int reachRow = top + std::min(row1 * t->ctu + d->searchRange + 4, d->height);
    for (int l = 0; l < nl; l++)
        for (int r = 0; r < d->numRef[l]; r++)
        {
            const x265hip_tme_host_ref& R = d->refs[l][r];
            if (!R.mePlane || !R.reconPlane) return fail("POC %d: planes of list %d reference %d missing", d->curPOC, l, r);
            for (int k = 0; k < 2; k++)
            {
                const int v0 = k ? R.reconRowsValid : R.meRowsValid;
                if (v0 < 0 || v0 > rows) return fail("POC %d: %d valid rows of a %d-row plane", d->curPOC, v0, rows);
                if (!parallel && v0) return fail("POC %d: %d valid rows declared with one frame thread (references are complete there)", d->curPOC, v0);
                const int valid = v0 ? v0 : rows;
                if (valid < reachRow) return fail("POC %d rows %d..%d: list %d reference %d (%s plane) declares %d final rows, the band's searches reach row %d", d->curPOC, row0, row1 - 1, l, r, k ? "reconstructed" : "search", valid, reachRow);
            }
        }
    /* the records: a pure function of the CTU's inputs */
    const size_t px = MOCK_PIXEL_BYTES;
    for (int c = row0 * t->nCtuX; c < row1 * t->nCtuX; c++)
    {
        const int cy = c / t->nCtuX, cx = c % t->nCtuX;
        uint64_t h = mix(0x1234, (uint64_t)d->curPOC * 4099 + (uint64_t)c);
        if (cx < d->pirStartCol) h = mix(h, (uint64_t)d->pirSafeX + 77);       /* the window limit of this CTU's CUs is an input of its records */
        for (int k = 0; k < nS; k++) h = mix(h, (uint64_t)d->qps[d->qpIndex[(size_t)c * nS + k]]);
        for (int a = 0; a < 5; a++) h = mix(h, (uint64_t)d->qps[d->areaQpIndex[(size_t)c * 5 + a]]);
        h = hash_bytes(h, d->temporal + (size_t)c * nS * 2, (size_t)nS * 2 * sizeof(x265hip_tme_temporal));
        if (d->median) h = hash_bytes(h, d->median + (size_t)c * 2 * X265HIP_MAX_REF * 3, (size_t)2 * X265HIP_MAX_REF * 3 * sizeof(int16_t));
        for (int sl : t->slots) h = hash_bytes(h, &d->table[(size_t)c * 593 + sl], sizeof(x265hip_inter_choice));
        {   /* the CTU's own pixels */
            const uint8_t* cur = (const uint8_t*)d->curPlane;
            const int ox = (int)(d->origin % d->stride);
            for (int y = top + cy * t->ctu; y < top + std::min((cy + 1) * t->ctu, d->height); y++)
                h = hash_bytes(h, cur + ((size_t)y * d->stride + ox + cx * t->ctu) * px, (size_t)std::min(t->ctu, d->width - cx * t->ctu) * px);
        }
        for (int l = 0; l < nl; l++)
            for (int r = 0; r < d->numRef[l]; r++)
            {
                const x265hip_tme_host_ref& R = d->refs[l][r];
                if (R.refTable) for (int sl : t->slots) h = hash_bytes(h, &R.refTable[(size_t)c * 593 + sl], sizeof(x265hip_inter_choice));
                if (R.lowresMv)      /* the lookahead's MVs of the CTU's 16x16 blocks */
                    for (int by = cy * t->ctu / 16; by < std::min((cy + 1) * t->ctu, d->height + 15) / 16; by++)
                        for (int bx = cx * t->ctu / 16; bx < std::min((cx + 1) * t->ctu / 16, d->lowresBlocksX); bx++)
                            h = hash_bytes(h, R.lowresMv + 2 * ((size_t)by * d->lowresBlocksX + bx), 2 * sizeof(int16_t));
                h = mix(h, (uint64_t)(R.lowresMv != nullptr) * 2 + (R.refTable != nullptr));
                for (int k = 0; k < 2; k++)
                {
                    if (k && R.reconPlane == R.mePlane) break;
                    const int v0 = k ? R.reconRowsValid : R.meRowsValid, valid = v0 ? v0 : rows;
                    /* the CTU's window in the reference, cut to the rows the caller declared final: its own rows -/+ the search range and the sub-pel taps, 8 columns either side */
                    const int y0 = std::max(0, top + cy * t->ctu - d->searchRange - 4), y1 = std::min(valid, top + std::min((cy + 1) * t->ctu, d->height) + d->searchRange + 4);
                    const int ox = (int)(d->origin % d->stride);
                    const int x0 = std::max(0, ox + cx * t->ctu - 8), x1 = std::min((int)d->stride, ox + (cx + 1) * t->ctu + 8);
                    const uint8_t* plane = (const uint8_t*)(k ? R.reconPlane : R.mePlane);
                    for (int y = y0; y < y1; y++) h = hash_bytes(h, plane + ((size_t)y * d->stride + x0) * px, (size_t)(x1 - x0) * px);
                }
            }
        for (int sl : t->slots)
        {
            x265hip_inter_choice& o = d->table[(size_t)c * 593 + sl];
            const uint64_t g = mix(h, (uint64_t)sl);
            memset(&o, 0, sizeof(o));
            o.mv[0][0] = (int16_t)((int)(g & 7) - 3); o.mv[0][1] = (int16_t)((int)((g >> 3) & 3) - 2);
            o.mvp[0][0] = o.mv[0][0]; o.mvp[0][1] = o.mv[0][1];
            o.ref[0] = 0; o.ref[1] = -1;
            o.bits = 12; o.cost = 4000 + (uint32_t)((g >> 8) & 1023);
        }
    }
    const int us = g_callUs + g_rowUs * (row1 - row0);
    if (us > 0) std::this_thread::sleep_for(std::chrono::microseconds(us));
    return X265HIP_OK;
}
} // extern "C"
