// xh_slots.cpp -- the drop-in table: per-slot functions with the reference's exact C signatures
// (primitives.h:133-236) that stage caller-owned host blocks to the GPU, run the batched kernel
// for ONE item and copy the result back.  Reentrant: all state is per calling thread.
// x265hip_setup_primitives() is the "one more overwrite pass" of primitives.cpp:336-376.
#include "xh_runtime.h"
#include "xh_internal.h"
#include <cstring>

using namespace xh;

#define XH_FOR_EACH_PU(X) \
    X(0, 4, 4) X(1, 8, 8) X(2, 16, 16) X(3, 32, 32) X(4, 64, 64) X(5, 8, 4) X(6, 4, 8) X(7, 16, 8) X(8, 8, 16) \
    X(9, 32, 16) X(10, 16, 32) X(11, 64, 32) X(12, 32, 64) X(13, 16, 12) X(14, 12, 16) X(15, 16, 4) X(16, 4, 16) \
    X(17, 32, 24) X(18, 24, 32) X(19, 32, 8) X(20, 8, 32) X(21, 64, 48) X(22, 48, 64) X(23, 64, 16) X(24, 16, 64)
#define XH_FOR_EACH_CU(X) X(0, 4) X(1, 8) X(2, 16) X(3, 32) X(4, 64)
#define XH_FOR_EACH_TU(X) X(0, 4) X(1, 8) X(2, 16) X(3, 32)

namespace {

const int32_t* upload_i32(ThreadCtx& c, const int32_t* v, int n)
{
    int32_t* d = (int32_t*)c.dalloc(n * 4);
    // pinned staging is reused within a call; offsets are tiny so copy synchronously-safe via a private slice
    static thread_local int slice = 0;
    int32_t* p = (int32_t*)(c.pinned + 1024 + (slice++ & 7) * 256);
    memcpy(p, v, n * 4);
    if (hipMemcpyAsync(d, p, n * 4, hipMemcpyHostToDevice, c.stream) != hipSuccess) fatal("upload_i32");
    return d;
}
template<class R> R fetch_scalar(ThreadCtx& c, const void* dev)
{
    if (hipMemcpyAsync(c.pinned, dev, sizeof(R), hipMemcpyDeviceToHost, c.stream) != hipSuccess) fatal("fetch_scalar");
    c.sync();
    R r; memcpy(&r, c.pinned, sizeof(R)); return r;
}
#define XH_OK(call) do { if ((call) != X265HIP_OK) fatal(#call); } while (0)

// ---------------- pixel compare ----------------
int slot_cmp(int op, int w, int h, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    ThreadCtx& c = ThreadCtx::get(); c.reset();
    DevBlock A = stage_in(c, a, sa, w, h, sizeof(pixel)), B = stage_in(c, b, sb, w, h, sizeof(pixel));
    void* out = c.dalloc(8);
    const int32_t* z = dev_zero_offsets(c);
    XH_OK(x265hip_pixelcmp_batch(c.stream, op, w, h, A.ptr, A.stride, z, B.ptr, B.stride, z, 1, out));
    return fetch_scalar<int32_t>(c, out);
}
sse_t slot_sse_pp(int n, const pixel* a, intptr_t sa, const pixel* b, intptr_t sb)
{
    ThreadCtx& c = ThreadCtx::get(); c.reset();
    DevBlock A = stage_in(c, a, sa, n, n, sizeof(pixel)), B = stage_in(c, b, sb, n, n, sizeof(pixel));
    void* out = c.dalloc(8);
    const int32_t* z = dev_zero_offsets(c);
    XH_OK(x265hip_pixelcmp_batch(c.stream, X265HIP_CMP_SSE_PP, n, n, A.ptr, A.stride, z, B.ptr, B.stride, z, 1, out));
    return (sse_t)fetch_scalar<uint64_t>(c, out);
}
sse_t slot_sse_ss(int n, const int16_t* a, intptr_t sa, const int16_t* b, intptr_t sb)
{
    ThreadCtx& c = ThreadCtx::get(); c.reset();
    DevBlock A = stage_in(c, a, sa, n, n, 2), B = stage_in(c, b, sb, n, n, 2);
    void* out = c.dalloc(8);
    const int32_t* z = dev_zero_offsets(c);
    XH_OK(x265hip_pixelcmp_batch(c.stream, X265HIP_CMP_SSE_SS, n, n, A.ptr, A.stride, z, B.ptr, B.stride, z, 1, out));
    return (sse_t)fetch_scalar<uint64_t>(c, out);
}
sse_t slot_ssd_s(int n, const int16_t* a, intptr_t sa)
{
    ThreadCtx& c = ThreadCtx::get(); c.reset();
    DevBlock A = stage_in(c, a, sa, n, n, 2);
    void* out = c.dalloc(8);
    const int32_t* z = dev_zero_offsets(c);
    XH_OK(x265hip_pixelcmp_batch(c.stream, X265HIP_CMP_SSD_S, n, n, A.ptr, A.stride, z, A.ptr, A.stride, z, 1, out));
    return (sse_t)fetch_scalar<uint64_t>(c, out);
}
// sad_x3 / sad_x4: fenc stride is the fixed FENC_STRIDE (pixel.cpp:89,113)
void slot_sad_xn(int nref, int w, int h, const pixel* fenc, const pixel* const* refs, intptr_t rs, int32_t* res)
{
    ThreadCtx& c = ThreadCtx::get(); c.reset();
    DevBlock F = stage_in(c, fenc, XH_FENC_STRIDE, w, h, sizeof(pixel));
    pixel* R = (pixel*)c.dalloc((size_t)nref * w * h * sizeof(pixel));
    int32_t offs[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < nref; i++)
    {
        offs[i] = i * w * h;
        if (rs >= w)
        {
            if (hipMemcpy2DAsync(R + offs[i], w * sizeof(pixel), refs[i], rs * sizeof(pixel), w * sizeof(pixel), h, hipMemcpyHostToDevice, c.stream) != hipSuccess)
                fatal("sad_xn H2D");
        }
        else
            fatal("sad_x3/x4 called with frefstride < width");
    }
    const int32_t* dOff = upload_i32(c, offs, 4);
    int32_t* out = (int32_t*)c.dalloc(16);
    XH_OK(x265hip_pixelcmp_batch(c.stream, X265HIP_CMP_SAD, w, h, F.ptr, F.stride, dev_zero_offsets(c), R, w, dOff, nref, out));
    if (hipMemcpyAsync(c.pinned, out, 16, hipMemcpyDeviceToHost, c.stream) != hipSuccess) fatal("sad_xn D2H");
    c.sync();
    memcpy(res, c.pinned, nref * 4);
}

template<int W, int H> int s_sad(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return slot_cmp(X265HIP_CMP_SAD, W, H, a, sa, b, sb); }
template<int W, int H> int s_satd(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return slot_cmp(X265HIP_CMP_SATD, W, H, a, sa, b, sb); }
template<int N> int s_sa8d(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return slot_cmp(X265HIP_CMP_SA8D, N, N, a, sa, b, sb); }
template<int N> int s_psy(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return slot_cmp(X265HIP_CMP_PSY_COST, N, N, a, sa, b, sb); }
template<int N> sse_t s_sse_pp(const pixel* a, intptr_t sa, const pixel* b, intptr_t sb) { return slot_sse_pp(N, a, sa, b, sb); }
template<int N> sse_t s_sse_ss(const int16_t* a, intptr_t sa, const int16_t* b, intptr_t sb) { return slot_sse_ss(N, a, sa, b, sb); }
template<int N> sse_t s_ssd_s(const int16_t* a, intptr_t sa) { return slot_ssd_s(N, a, sa); }
template<int W, int H> void s_sad_x3(const pixel* f, const pixel* r0, const pixel* r1, const pixel* r2, intptr_t rs, int32_t* res)
{ const pixel* r[3] = { r0, r1, r2 }; slot_sad_xn(3, W, H, f, r, rs, res); }
template<int W, int H> void s_sad_x4(const pixel* f, const pixel* r0, const pixel* r1, const pixel* r2, const pixel* r3, intptr_t rs, int32_t* res)
{ const pixel* r[4] = { r0, r1, r2, r3 }; slot_sad_xn(4, W, H, f, r, rs, res); }

// pu[].ads (pixel.cpp:121-165); which variant a PU gets is the reference's slot map (pixel.cpp:1122-1146)
constexpr int ads_parts(int w, int h)
{
    return ((w == 4 && h == 4) || (w == 8 && h == 8) || (w == 16 && h == 12) || (w == 12 && h == 16) || (w == 16 && h == 4) || (w == 4 && h == 16)) ? 1
         : ((w == 8 && h == 4) || (w == 4 && h == 8) || (w == 16 && h == 8) || (w == 8 && h == 16) || (w == 32 && h == 16) || (w == 16 && h == 32) ||
            (w == 64 && h == 32) || (w == 32 && h == 64)) ? 2 : 4;
}
template<int W, int H> int s_ads(int* encDC, uint32_t* sums, int delta, uint16_t* costMvX, int16_t* mvs, int width, int thresh)
{
    constexpr int P = ads_parts(W, H);
    ThreadCtx& c = ThreadCtx::get(); c.reset();
    if (width <= 0) return 0;
    const int span = width + (P > 1 ? delta : 0) + (P == 4 ? (W >> 1) : 0);
    DevBlock E = stage_in(c, encDC, 4, P, 1, 4), S = stage_in(c, sums, span, span, 1, 4), M = stage_in(c, costMvX, width, width, 1, 2);
    int16_t* dm = (int16_t*)c.dalloc((size_t)width * 2); int32_t* dn = (int32_t*)c.dalloc(8);
    XH_OK(x265hip_ads(c.stream, P, W, (const int32_t*)E.ptr, (const uint32_t*)S.ptr, delta, (const uint16_t*)M.ptr, dm, width, thresh, dn));
    const int nmv = fetch_scalar<int32_t>(c, dn);
    if (nmv > 0)
    {
        if (hipMemcpyAsync(mvs, dm, (size_t)nmv * 2, hipMemcpyDeviceToHost, c.stream) != hipSuccess) fatal("ads D2H");
        c.sync();
    }
    return nmv;
}

#include "xh_slots_blk.inc"
#include "xh_slots_tr.inc"
#include "xh_slots_ip.inc"
#include "xh_slots_intra.inc"

} // namespace

extern "C" int x265hip_setup_primitives(void* encoder_primitives, int bit_depth, uint32_t flags)
{
    if (!encoder_primitives) { set_error("NULL table"); return X265HIP_EARG; }
    if (bit_depth != X265_DEPTH) { set_error("bit depth %d requested, library built for %d", bit_depth, X265_DEPTH); return X265HIP_EABI; }
    int ndev = 0;
    XH_HIP(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { set_error("no HIP device"); return X265HIP_EDEVICE; }
    void** t = (void**)encoder_primitives;
#define PU(i, s) t[X265HIP_OFF_PU / 8 + (i) * X265HIP_PU_PTRS + (s)]
#define CU(i, s) t[X265HIP_OFF_CU / 8 + (i) * X265HIP_CU_PTRS + (s)]
#define SC(off) t[(off) / 8]
#define CH420_PU(i, s) t[(X265HIP_OFF_CHROMA + 1 * X265HIP_CHROMA_BYTES) / 8 + (i) * X265HIP_CHROMA_PU_PTRS + (s)]
#define CH420_CU(i, s) t[(X265HIP_OFF_CHROMA + 1 * X265HIP_CHROMA_BYTES) / 8 + X265HIP_NUM_PU * X265HIP_CHROMA_PU_PTRS + (i) * X265HIP_CHROMA_CU_PTRS + (s)]

#define FILL_PU(i, W, H) \
    PU(i, X265HIP_PU_SAD) = (void*)s_sad<W, H>; PU(i, X265HIP_PU_SAD_X3) = (void*)s_sad_x3<W, H>; \
    PU(i, X265HIP_PU_SAD_X4) = (void*)s_sad_x4<W, H>; PU(i, X265HIP_PU_SATD) = (void*)s_satd<W, H>; PU(i, X265HIP_PU_ADS) = (void*)s_ads<W, H>;
    XH_FOR_EACH_PU(FILL_PU)
#define FILL_CU(i, N) \
    CU(i, X265HIP_CU_SA8D) = (void*)s_sa8d<N>; CU(i, X265HIP_CU_PSY_COST_PP) = (void*)s_psy<N>; \
    CU(i, X265HIP_CU_SSE_PP) = (void*)s_sse_pp<N>; CU(i, X265HIP_CU_SSE_SS) = (void*)s_sse_ss<N>; \
    CU(i, X265HIP_CU_SSD_S) = (void*)s_ssd_s<N>; CU(i, X265HIP_CU_SSD_S_ALIGNED) = (void*)s_ssd_s<N>;
    XH_FOR_EACH_CU(FILL_CU)
    fill_blk(t);
    fill_transform(t);
    fill_interp(t);
    fill_intra(t, flags);
    return X265HIP_OK;
}
