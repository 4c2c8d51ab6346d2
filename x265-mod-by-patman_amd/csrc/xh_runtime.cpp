// xh_runtime.cpp -- error plumbing, per-thread stream/arena, host<->device staging.
#include "xh_runtime.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <atomic>
#include "../../include/x265hip_frame.h"

namespace xh {
thread_local const KernelEvents* tl_star64Events = nullptr;

static thread_local char g_err[512];

void set_error(const char* fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
int hip_fail(hipError_t e, const char* what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return X265HIP_EDEVICE;
}
void fatal(const char* what)
{
    fprintf(stderr, "x265hip: FATAL: %s -- %s\n(no CPU fallback exists in this library)\n", what, g_err);
    abort();
}

static const size_t kArena = 8u << 20;     // first arena; plane-sized slots (frameInitLowres, weight_pp on 4K / 8K planes) grow it on demand
static std::atomic<int> g_device{-1};      // device chosen by x265hip_device_init; the reference's pool threads inherit it here

// The context of a thread dies with the thread (the reference's pool workers come and go with the encoder).
struct CtxHolder
{
    ThreadCtx* c = nullptr;
    ~CtxHolder()
    {
        if (!c) return;
        (void)hipStreamSynchronize(c->stream);
        for (char* p : c->retired) { if (xh::kFence) xh::dev_free_pooled(p); else (void)xh::dev_free(p); }
        (void)xh::dev_free(c->arena); (void)xh::dev_free(c->zeros); (void)hipHostFree(c->pinned); (void)hipStreamDestroy(c->stream);
        delete c;
    }
};
static thread_local CtxHolder t_ctx;

ThreadCtx& ThreadCtx::get()
{
    if (t_ctx.c) return *t_ctx.c;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { hip_fail(e, "hipGetDeviceCount"); fatal("no HIP device available"); }
    const int dev = g_device.load();
    if (dev >= 0 && hipSetDevice(dev) != hipSuccess) fatal("hipSetDevice failed");     // a new host thread starts on device 0 otherwise
    ThreadCtx* c = new ThreadCtx();
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) fatal("hipStreamCreate failed");
    if (xh::dev_alloc((void**)&c->arena, kArena, "slot arena") != hipSuccess) fatal("hipMalloc(arena) failed");
    c->arenaSize = kArena;
    if (xh::dev_alloc((void**)&c->zeros, 256, "zero offsets") != hipSuccess) fatal("hipMalloc(zeros) failed");
    if (hipHostMalloc((void**)&c->pinned, 4096, hipHostMallocDefault) != hipSuccess) fatal("hipHostMalloc failed");
    t_ctx.c = c;
    if (hipMemsetAsync(c->zeros, 0, 256, c->stream) != hipSuccess) fatal("hipMemset failed");
    return *c;
}

// A slot call that needs more than the arena holds gets a new, larger chunk; blocks handed out earlier in the same call stay valid
// (their chunk is retired, not freed) until the next call's reset(), which runs after that call's sync().
void* ThreadCtx::dalloc(size_t bytes)
{
    if (xh::kFence)
    {   // fence build: every block of a slot call is its own fenced allocation (freed by the next reset())
        void* p = nullptr;
        if (xh::dev_alloc_pooled(&p, bytes, "slot block") != hipSuccess) { set_error("device memory exhausted (fence build, %zu bytes)", bytes); fatal("arena"); }
        retired.push_back((char*)p);
        return p;
    }
    size_t off = (arenaUsed + 255) & ~(size_t)255;
    if (off + bytes > arenaSize)
    {
        size_t want = arenaSize * 2;
        while (want < bytes + 256) want *= 2;
        char* bigger = nullptr;
        if (xh::dev_alloc((void**)&bigger, want, "slot arena") != hipSuccess) { set_error("device memory exhausted growing the slot arena to %zu bytes", want); fatal("arena"); }
        retired.push_back(arena);
        arena = bigger; arenaSize = want; off = 0;
    }
    arenaUsed = off + bytes;
    return arena + off;
}
void ThreadCtx::reset()
{
    arenaUsed = 0;
    if (!retired.empty())
    {   // every slot call ends with sync(), so nothing queued on the stream still reads the retired chunks
        (void)hipStreamSynchronize(stream);
        for (char* p : retired) { if (xh::kFence) xh::dev_free_pooled(p); else (void)xh::dev_free(p); }
        retired.clear();
    }
}
void ThreadCtx::sync()
{
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { hip_fail(e, "hipStreamSynchronize"); fatal("kernel execution failed"); }
}
const int32_t* dev_zero_offsets(ThreadCtx& c) { return (const int32_t*)c.zeros; }

DevBlock stage_in(ThreadCtx& c, const void* host, intptr_t stride, int w, int h, int es)
{
    DevBlock b;
    if (stride >= w)
    {
        b.ptr = c.dalloc((size_t)w * h * es); b.stride = w;
        if (hipMemcpy2DAsync(b.ptr, (size_t)w * es, host, (size_t)stride * es, (size_t)w * es, h, hipMemcpyHostToDevice, c.stream) != hipSuccess)
            fatal("hipMemcpy2DAsync H2D failed");
    }
    else
    {   // overlapping rows: copy the span [first element, last element]
        size_t span = ((size_t)(h - 1) * (stride < 0 ? 0 : stride) + w) * es;
        b.ptr = c.dalloc(span); b.stride = stride;
        if (hipMemcpyAsync(b.ptr, host, span, hipMemcpyHostToDevice, c.stream) != hipSuccess) fatal("hipMemcpyAsync H2D failed");
    }
    return b;
}
void* stage_out_alloc(ThreadCtx& c, int w, int h, int es) { return c.dalloc((size_t)w * h * es); }
void stage_out_copy(ThreadCtx& c, void* host, intptr_t stride, const void* dev, int w, int h, int es)
{
    if (hipMemcpy2DAsync(host, (size_t)stride * es, dev, (size_t)w * es, (size_t)w * es, h, hipMemcpyDeviceToHost, c.stream) != hipSuccess)
        fatal("hipMemcpy2DAsync D2H failed");
}

} // namespace xh

extern "C" const char* x265hip_last_error(void) { return xh::g_err; }
extern "C" int x265hip_bit_depth(void) { return X265_DEPTH; }
extern "C" int x265hip_device_init(int device)
{
    int n = 0;
    XH_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) { xh::set_error("device %d out of range (%d devices)", device, n); return X265HIP_EDEVICE; }
    XH_HIP(hipSetDevice(device));
    xh::g_device.store(device);          // the table slots run on the caller's pool threads: ThreadCtx::get() selects this device there
    return X265HIP_OK;
}
extern "C" int x265hip_abi_check(size_t sizeof_table, int bit_depth)
{
    if (sizeof_table != X265HIP_SIZEOF_TABLE) { xh::set_error("sizeof(EncoderPrimitives) %zu != %d", sizeof_table, X265HIP_SIZEOF_TABLE); return X265HIP_EABI; }
    if (bit_depth != X265_DEPTH) { xh::set_error("bit depth %d requested, library built for %d", bit_depth, X265_DEPTH); return X265HIP_EABI; }
    return X265HIP_OK;
}

// BitCost::CalculateLogs + setQP (bitcost.cpp:30-105): float table of bit sizes from a DOUBLE log, scaled by the
// 4-decimal lambda table (constants.cpp:28-116: pow(2, qp/6 - 2) * (1 << (depth - 8))), + 0.5f, capped at 2^15 - 1.
extern "C" int x265hip_mvcost_row(int qp, int halfRange, uint16_t* out)
{
    if (qp < 0 || qp > 69 || halfRange < 0 || !out) { xh::set_error("mvcost_row: bad arguments"); return X265HIP_EARG; }
    const double lambda = std::floor(std::pow(2.0, (double)qp / 6.0 - 2.0) * (double)(1 << (X265_DEPTH - 8)) * 10000.0 + 0.5) / 10000.0;
    const float log2_2 = (float)(2.0f / std::log((double)2.0f));
    for (int i = 0; i <= halfRange; i++)
    {
        const float bits = i ? (float)(std::log((double)(float)(i + 1)) * log2_2 + 1.718f) : 0.718f;
        double c = bits * lambda + 0.5f;
        if (c > 32767.0) c = 32767.0;
        out[halfRange + i] = out[halfRange - i] = (uint16_t)c;
    }
    return X265HIP_OK;
}
