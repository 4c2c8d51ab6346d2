/*
 * tests/emu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.  A stand-in for <hip/hip_runtime.h> that lets the library's kernel SOURCES (csrc/*.hip, unchanged) be compiled
 * for the HOST and run on the CPU, so that their logic -- indexing, band / row arithmetic, reductions, LDS protocols -- can be checked against the oracle where no GPU is
 * available (tests/test_emu_*.py build tests/emu/libx265hip_emu_<depth>.so from the same sources the GPU library is built from).  It is NOT a CPU path of the product: nothing
 * under x265-mod-by-patman_amd/ includes it, the emulated library is only ever loaded by tests, and what it cannot show is everything that makes the GPU the GPU -- timing,
 * races between wavefronts, the compiler's code for gfx950, LDS bank behaviour, out-of-range LDS accesses.
 *
 * Execution model: a launch runs its workgroups one after the other; the work-items of a workgroup are FIBERS (ucontext) on the calling thread, scheduled round-robin.  A
 * fiber runs until it reaches __syncthreads() (waits for every live work-item of the group), a cross-lane operation of its wavefront (waits until every live lane of the
 * wavefront -- 64 consecutive work-items -- has arrived at a cross-lane operation, then the operation is evaluated on the values the lanes published: DPP controls, readlane,
 * ballot, shuffles, wave barrier), or its end.  Lanes that have returned count as inactive (EXEC = 0): a DPP read from one gives bound_ctrl's zero / the old value.
 * __shared__ variables are function-level statics (workgroups do not overlap in time); __constant__ are statics; device memory is the host heap.
 */
#pragma once
#define XH_EMU 1
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

/* ---- qualifiers ---- */
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __constant__ static

/* ---- runtime types ---- */
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorUnknown = 999 };
typedef struct emu_stream_* hipStream_t;
typedef struct emu_event_* hipEvent_t;
typedef int hipDevice_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipEventDefault = 0, hipHostMallocDefault = 0, hipHostRegisterDefault = 0 };
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct uint3 { unsigned x, y, z; };
struct int2 { int x, y; }; struct uint2 { unsigned x, y; }; struct int4 { int x, y, z, w; }; struct uint4 { unsigned x, y, z, w; };
struct float2 { float x, y; }; struct float4 { float x, y, z, w; }; struct short2 { short x, y; }; struct ushort2 { unsigned short x, y; }; struct uchar4 { unsigned char x, y, z, w; };
struct short4 { short x, y, z, w; }; struct ushort4 { unsigned short x, y, z, w; }; struct longlong2 { long long x, y; }; struct ulonglong2 { unsigned long long x, y; };
inline int2 make_int2(int x, int y) { return int2{ x, y }; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{ x, y }; }
inline int4 make_int4(int x, int y, int z, int w) { return int4{ x, y, z, w }; }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{ x, y, z, w }; }
inline float2 make_float2(float x, float y) { return float2{ x, y }; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{ x, y, z, w }; }
inline short2 make_short2(short x, short y) { return short2{ x, y }; }
inline ushort2 make_ushort2(unsigned short x, unsigned short y) { return ushort2{ x, y }; }
inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { return uchar4{ x, y, z, w }; }

inline uint3 threadIdx, blockIdx;
inline dim3 blockDim, gridDim;

/* ---- the fiber scheduler ---- */
namespace emu {
enum State { READY, AT_BARRIER, AT_WAVE, DONE };
/* what a cross-lane operation's released lanes see: the values published by the lanes that took part (mask) */
struct Snapshot { uint64_t val[64]; uint64_t mask; int users; };
/* a context switch of our own (ucontext's swapcontext makes a system call per switch: the signal mask): save the callee-saved registers on the stack we leave, change stacks, restore */
__attribute__((naked, noinline)) inline void switch_stack(void** /*saveSp: rdi*/, void* /*toSp: rsi*/)
{
    asm volatile("pushq %rbp\n pushq %rbx\n pushq %r12\n pushq %r13\n pushq %r14\n pushq %r15\n"
                 "movq %rsp, (%rdi)\n movq %rsi, %rsp\n"
                 "popq %r15\n popq %r14\n popq %r13\n popq %r12\n popq %rbx\n popq %rbp\n ret\n");
}
/* AddressSanitizer (make ASAN=1) has to be told about stack switches it did not make */
#ifdef XH_EMU_ASAN
extern "C" void __sanitizer_start_switch_fiber(void** fakeStackSave, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fakeStackSave, const void** bottomOld, size_t* sizeOld);
#define EMU_ASAN_START(save, bottom, size) __sanitizer_start_switch_fiber(save, bottom, size)
#define EMU_ASAN_FINISH(save, bottomOld, sizeOld) __sanitizer_finish_switch_fiber(save, bottomOld, sizeOld)
#else
#define EMU_ASAN_START(save, bottom, size) ((void)0)
#define EMU_ASAN_FINISH(save, bottomOld, sizeOld) ((void)0)
#endif
struct Fiber
{
    void* sp = nullptr; char* stack = nullptr; void* fake = nullptr; State st = READY; uint3 tid; int lin = 0;
    int site = -1; uint64_t pub = 0, scope = 0, stamp = 0; Snapshot* snap = nullptr;      /* the cross-lane operation the lane waits at */
};
struct Group
{
    std::vector<Fiber> f; void* mainSp = nullptr; void* mainFake = nullptr; const void* mainBottom = nullptr; size_t mainSize = 0; int cur = -1, live = 0, atBarrier = 0, n = 0; const std::function<void()>* body = nullptr;
    std::vector<char> dyn;                       /* dynamic LDS of the launch (HIP_DYNAMIC_SHARED) */
};
inline Group* g = nullptr;
inline uint64_t g_stamp = 0;
inline size_t g_dynBytes = 0;
constexpr size_t kStack = 192 << 10;
inline std::vector<char*>& stack_pool() { static std::vector<char*> p; return p; }
inline void die(const char* what) { fprintf(stderr, "hip emulation: %s\n", what); abort(); }

inline void yield()
{
    Fiber& me = g->f[g->cur];
    EMU_ASAN_START(&me.fake, g->mainBottom, g->mainSize);
    switch_stack(&me.sp, g->mainSp);
    EMU_ASAN_FINISH(g->f[g->cur].fake, nullptr, nullptr);
}
inline void trampoline()
{
    EMU_ASAN_FINISH(nullptr, &g->mainBottom, &g->mainSize);
    (*g->body)();
    Fiber& me = g->f[g->cur];
    me.st = DONE; g->live--;
    EMU_ASAN_START(nullptr, g->mainBottom, g->mainSize);
    switch_stack(&me.sp, g->mainSp);
    die("a finished work-item was resumed");
}
inline void* dyn_lds() { return g->dyn.data(); }

/* Release the cross-lane operations of one wavefront that can go: a lane's operation is COMPLETE when every live lane of its scope (the lanes its result can depend on: its
   quad / row of 16 / 32 lanes / the wavefront) waits at the same operation -- lane groups of one wavefront that run different control flow (several PUs per wavefront, each with
   its own search) then proceed independently, as they do under EXEC masks.  `force`: nothing else in the workgroup can run -- the lanes of ONE pending operation (chosen below)
   go with the lanes that are there (the others count as inactive: they returned, or sit in a branch that never comes here). */
inline bool release_wave(Group& G, int w0, bool force)
{
    const int w1 = std::min(w0 + 64, G.n);
    uint64_t live = 0;
    for (int i = w0; i < w1; i++) if (G.f[i].st != DONE) live |= 1ull << (i - w0);
    bool any = false;
    int sites[64], ns = 0;
    for (int i = w0; i < w1; i++)
        if (G.f[i].st == AT_WAVE) { bool seen = false; for (int k = 0; k < ns; k++) seen |= sites[k] == G.f[i].site; if (!seen) sites[ns++] = G.f[i].site; }
    int forcedSite = -1;
    if (force)
    {   /* the most LOCAL pending operation first (a quad / row operation inside divergent code comes before the wavefront-wide one at the point where the lanes meet again);
           among equals the one reached last (lanes still inside a loop come back to their operation, the lanes that left it wait behind it) */
        int bestScope = 65; uint64_t bestStamp = 0;
        for (int i = w0; i < w1; i++)
            if (G.f[i].st == AT_WAVE)
            {
                const int sc = __builtin_popcountll(G.f[i].scope);
                if (sc < bestScope || (sc == bestScope && G.f[i].stamp >= bestStamp)) { bestScope = sc; bestStamp = G.f[i].stamp; forcedSite = G.f[i].site; }
            }
    }
    for (int k = 0; k < ns; k++)
    {
        uint64_t at = 0;
        for (int i = w0; i < w1; i++) if (G.f[i].st == AT_WAVE && G.f[i].site == sites[k]) at |= 1ull << (i - w0);
        uint64_t go = 0;
        for (int i = w0; i < w1; i++)
            if ((at >> (i - w0)) & 1) { const uint64_t need = G.f[i].scope & live; if ((need & ~at) == 0 || sites[k] == forcedSite) go |= 1ull << (i - w0); }
        if (!go) continue;
        Snapshot* sn = new Snapshot(); sn->mask = at; sn->users = __builtin_popcountll(go);
        for (int i = w0; i < w1; i++) sn->val[i - w0] = ((at >> (i - w0)) & 1) ? G.f[i].pub : 0;
        for (int i = w0; i < w1; i++) if ((go >> (i - w0)) & 1) { G.f[i].snap = sn; G.f[i].st = READY; }
        any = true;
    }
    return any;
}

inline void run_group(dim3 block, const std::function<void()>& body)
{
    Group G; g = &G;
    const int n = (int)(block.x * block.y * block.z);
    G.f.resize(n); G.live = n; G.n = n; G.body = &body; G.dyn.assign(g_dynBytes + 64, 0);
    for (int i = 0; i < n; i++)
    {
        Fiber& f = G.f[i];
        f.lin = i; f.tid = uint3{ (unsigned)(i % block.x), (unsigned)((i / block.x) % block.y), (unsigned)(i / (block.x * block.y)) };
        if (stack_pool().empty()) f.stack = (char*)malloc(kStack); else { f.stack = stack_pool().back(); stack_pool().pop_back(); }
        /* the first switch to the fiber pops six registers and returns into trampoline(); the ABI wants rsp = 8 (mod 16) at a function's first instruction */
        void** top = (void**)(((uintptr_t)f.stack + kStack) & ~(uintptr_t)15) - 2;
        top[0] = (void*)trampoline; top[1] = nullptr;
        for (int k = 1; k <= 6; k++) top[-k] = nullptr;
        f.sp = (void*)(top - 6);
    }
    /* The order in which ready work-items get their turn is the emulation's choice, not the program's: a kernel whose result depends on it has a race (an LDS exchange without
       its __syncthreads() / wave_sync()).  X265HIP_EMU_ORDER = reverse | random[:seed] picks another order (tests/test_emu_kernels.py runs a selection under all three) */
    static const int orderMode = [] { const char* e = getenv("X265HIP_EMU_ORDER"); return !e ? 0 : !strncmp(e, "reverse", 7) ? 1 : !strncmp(e, "random", 6) ? 2 : 0; }();
    static uint32_t rnd = [] { const char* e = getenv("X265HIP_EMU_ORDER"); const char* c = e ? strchr(e, ':') : nullptr; return c ? (uint32_t)atoi(c + 1) * 2654435761u + 1u : 12345u; }();
    std::vector<int> order(n);
    for (int i = 0; i < n; i++) order[i] = orderMode == 1 ? n - 1 - i : i;
    while (G.live > 0)
    {
        bool progressed = false;
        if (orderMode == 2)
            for (int i = n - 1; i > 0; i--) { rnd = rnd * 1664525u + 1013904223u; const int j = (int)((rnd >> 8) % (uint32_t)(i + 1)); std::swap(order[i], order[j]); }
        for (int oi = 0; oi < n; oi++)
        {
            const int i = order[oi];
            Fiber& f = G.f[i];
            if (f.st != READY) continue;
            G.cur = i; threadIdx = f.tid; progressed = true;
            EMU_ASAN_START(&G.mainFake, f.stack, kStack);
            switch_stack(&G.mainSp, f.sp);
            EMU_ASAN_FINISH(G.mainFake, nullptr, nullptr);
        }
        if (G.live == 0) break;
        if (G.atBarrier == G.live) { for (auto& f : G.f) if (f.st == AT_BARRIER) f.st = READY; G.atBarrier = 0; progressed = true; }
        for (int w0 = 0; w0 < n; w0 += 64) progressed |= release_wave(G, w0, false);
        if (!progressed)
        {
            bool forced = false;
            for (int w0 = 0; w0 < n; w0 += 64) forced |= release_wave(G, w0, true);
            if (!forced) die("deadlock: work-items wait at a __syncthreads() the others never reach");
        }
    }
    for (auto& f : G.f) stack_pool().push_back(f.stack);
    g = nullptr;
}

inline void block_barrier() { Fiber& me = g->f[g->cur]; me.st = AT_BARRIER; g->atBarrier++; yield(); }
inline int lane_id() { return g->f[g->cur].lin & 63; }

/* scopes (as lane masks of the wavefront, for lane l) */
inline uint64_t scope_quad(int l) { return 0xFull << (l & ~3); }
inline uint64_t scope_row(int l) { return 0xFFFFull << (l & ~15); }
inline uint64_t scope_half(int l) { return 0xFFFFFFFFull << (l & 32); }
inline uint64_t scope_width(int l, int w) { return w >= 64 ? ~0ull : (((1ull << w) - 1) << (l & ~(w - 1))); }
constexpr uint64_t kScopeWave = ~0ull;

/* publish `v` at operation `site`, wait until the operation is released for this lane (see release_wave), return the released snapshot (free it with done()) */
inline Snapshot* wave_exchange(int site, uint64_t v, uint64_t scope)
{
    Fiber& me = g->f[g->cur];
    me.site = site; me.pub = v; me.scope = scope; me.stamp = ++g_stamp; me.st = AT_WAVE; me.snap = nullptr;
    yield();
    return g->f[g->cur].snap;
}
inline void done(Snapshot* s) { if (--s->users == 0) delete s; }

inline std::recursive_mutex& launch_mutex() { static std::recursive_mutex m; return m; }
template<class F> void launch(dim3 grid, dim3 block, size_t dynBytes, F&& body)
{
    std::lock_guard<std::recursive_mutex> guard(launch_mutex());      /* the scheduler's state and the kernels' __shared__ statics are the process's: one launch at a time, whichever host thread asks */
    std::function<void()> fn = body;
    gridDim = grid; blockDim = block; g_dynBytes = dynBytes;
    for (unsigned z = 0; z < grid.z; z++) for (unsigned y = 0; y < grid.y; y++) for (unsigned x = 0; x < grid.x; x++) { blockIdx = uint3{ x, y, z }; run_group(block, fn); }
}
} // namespace emu

template<class K, class... A> inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t dynLds, hipStream_t, A... args)
{
    emu::launch(grid, block, dynLds, [&] { kernel(args...); });
}
inline void __syncthreads() { emu::block_barrier(); }
#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)emu::dyn_lds();
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
template<class F> inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }

/* ---- cross-lane operations ---- */
namespace emu {
/* v_mov_b32 with a DPP control (ISA, "DPP"): the lane that lane `l` reads, -1 = none (outside its row / the wavefront) */
inline int dpp_source(int l, int ctrl)
{
    const int row = l & ~15, r = l & 15;
    if (ctrl <= 0xFF) return (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);                         /* quad_perm */
    if (ctrl >= 0x101 && ctrl <= 0x10F) { const int s = r + (ctrl & 15); return s < 16 ? row | s : -1; }       /* row_shl:n -- lane l takes lane l + n */
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const int s = r - (ctrl & 15); return s >= 0 ? row | s : -1; }       /* row_shr:n -- lane l takes lane l - n */
    if (ctrl >= 0x121 && ctrl <= 0x12F) return row | ((r - (ctrl & 15)) & 15);               /* row_ror:n */
    if (ctrl == 0x130) return l + 1 < 64 ? l + 1 : -1;                                         /* wave_shl:1 */
    if (ctrl == 0x134) return (l + 1) & 63;                                                    /* wave_rol:1 */
    if (ctrl == 0x138) return l - 1 >= 0 ? l - 1 : -1;                                         /* wave_shr:1 */
    if (ctrl == 0x13C) return (l - 1) & 63;                                                    /* wave_ror:1 */
    if (ctrl == 0x140) return row | (15 - r);                                                  /* row_mirror */
    if (ctrl == 0x141) return row | (r < 8 ? 7 - r : 23 - r);                                  /* row_half_mirror */
    if (ctrl == 0x142) return (l >> 4) > 0 ? row - 1 : -1;                                     /* row_bcast:15 -- lane 15 of a row to the next row */
    if (ctrl == 0x143) return l >= 32 ? 31 : -1;                                               /* row_bcast:31 -- lane 31 to rows 2 and 3 */
    die("DPP control not modelled");
    return -1;
}
inline uint64_t dpp_scope(int l, int ctrl) { return ctrl <= 0xFF ? scope_quad(l) : ((ctrl >= 0x101 && ctrl <= 0x12F) || ctrl == 0x140 || ctrl == 0x141) ? scope_row(l) : kScopeWave; }
template<class T> inline T update_dpp(int site, T old, T src, int ctrl, int rowMask, int bankMask, bool boundCtrl)
{
    static_assert(sizeof(T) == 4, "32-bit DPP");
    uint32_t bits; memcpy(&bits, &src, 4);
    const int l = lane_id();
    Snapshot* sn = wave_exchange(site, bits, dpp_scope(l, ctrl));
    T out = old;
    if (((rowMask >> (l >> 4)) & 1) && ((bankMask >> ((l >> 2) & 3)) & 1))                    /* else: this lane's row / bank is masked, no write */
    {
        const int s = dpp_source(l, ctrl);
        if (s < 0 || !((sn->mask >> s) & 1)) { if (boundCtrl) memset(&out, 0, 4); }           /* no source lane, or it is inactive: bound_ctrl's zero / the old value */
        else { const uint32_t r = (uint32_t)sn->val[s]; memcpy(&out, &r, 4); }
    }
    done(sn);
    return out;
}
template<class T> inline T readlane(int site, T v, int lane)
{
    uint32_t bits = 0; memcpy(&bits, &v, sizeof(T) < 4 ? sizeof(T) : 4);
    Snapshot* sn = wave_exchange(site, bits, kScopeWave);
    const uint32_t r = (uint32_t)sn->val[lane & 63];
    done(sn);
    T out; memcpy(&out, &r, sizeof(T) < 4 ? sizeof(T) : 4);
    return out;
}
template<class T> inline T readfirstlane(int site, T v)
{
    uint32_t bits = 0; memcpy(&bits, &v, sizeof(T) < 4 ? sizeof(T) : 4);
    Snapshot* sn = wave_exchange(site, bits, kScopeWave);
    const uint32_t r = (uint32_t)sn->val[__builtin_ctzll(sn->mask)];
    done(sn);
    T out; memcpy(&out, &r, sizeof(T) < 4 ? sizeof(T) : 4);
    return out;
}
inline uint64_t ballot(int site, bool p)
{
    Snapshot* sn = wave_exchange(site, p ? 1 : 0, kScopeWave);
    uint64_t m = 0;
    for (int l = 0; l < 64; l++) if (((sn->mask >> l) & 1) && sn->val[l]) m |= 1ull << l;
    done(sn);
    return m;
}
template<class T> inline T shfl(int site, T v, int srcLane, int width = 64)
{
    uint64_t bits = 0; memcpy(&bits, &v, sizeof(T));
    const int l = lane_id(), s = (l & ~(width - 1)) | (srcLane & (width - 1));
    Snapshot* sn = wave_exchange(site, bits, scope_width(l, width));
    const uint64_t r = ((sn->mask >> s) & 1) ? sn->val[s] : bits;
    done(sn);
    T out; memcpy(&out, &r, sizeof(T));
    return out;
}
/* a wave barrier (wave_sync: data handed between lanes through LDS).  Lanes of a wavefront run in lockstep on the hardware; here the lanes that share the data must have
   stored before any of them loads.  Which lanes those are the call does not say: the whole wavefront is the scope, and lane groups that run apart are released when nothing else
   can run (release_wave's `force`) */
inline void wave_barrier(int site) { done(wave_exchange(site, 0, kScopeWave)); }
/* ds_swizzle_b32: quad-permute mode (bit 15) or the and / or / xor masks inside groups of 32 lanes */
inline int ds_swizzle(int site, int v, int pattern)
{
    const int l = lane_id();
    Snapshot* sn = wave_exchange(site, (uint32_t)v, (pattern & 0x8000) ? scope_quad(l) : scope_half(l));
    int s;
    if (pattern & 0x8000) s = (l & ~3) | ((pattern >> (2 * (l & 3))) & 3);
    else { const int andm = pattern & 31, orm = (pattern >> 5) & 31, xorm = (pattern >> 10) & 31; s = (l & 32) | ((((l & 31) & andm) | orm) ^ xorm); }
    const int r = ((sn->mask >> s) & 1) ? (int)(uint32_t)sn->val[s] : 0;
    done(sn);
    return r;
}
/* v_permlane16_swap / v_permlane32_swap (gfx950): the odd rows of vdst trade with the even rows of src0 (16) / the upper half of vdst with the lower half of src0 (32); the
   builtin returns { vdst', src0' } */
typedef unsigned v2u __attribute__((ext_vector_type(2)));
inline v2u permlane_swap(int site, unsigned a, unsigned b, int n)
{
    const int l = lane_id();
    Snapshot* sn = wave_exchange(site, ((uint64_t)b << 32) | a, n == 16 ? scope_half(l) : kScopeWave);
    v2u r = { a, b };
    const bool upper = n == 16 ? ((l >> 4) & 1) : (l >= 32);
    if (upper) { if ((sn->mask >> (l - n)) & 1) r[0] = (unsigned)(sn->val[l - n] >> 32); }      /* vdst[upper] <- src0[lower] */
    else { if ((sn->mask >> (l + n)) & 1) r[1] = (unsigned)sn->val[l + n]; }                   /* src0[lower] <- vdst[upper] */
    done(sn);
    return r;
}
/* v_mfma_i32_32x32x32_i8 (gfx950), operands as csrc/xh_dct32.h lays them out: lane (r = lane & 31, g = lane >> 5) holds A[r][16 g + s] / B[16 g + s][r] in byte s of its 16
   operand bytes, and D[(i & 3) + 8 (i >> 2) + 4 g][r] in element i of its 16 results */
typedef int v4i_ __attribute__((ext_vector_type(4)));
typedef int v16i_ __attribute__((ext_vector_type(16)));
struct MfmaPub { v4i_ a, b; v16i_ c; };
inline v16i_ mfma_i32_32x32x32_i8(int site, v4i_ a, v4i_ b, v16i_ c)
{
    const MfmaPub mine{ a, b, c };                             /* every lane publishes where its operands are, reads the others', and leaves only when all have read */
    const int l = lane_id();
    Snapshot* sn = wave_exchange(site, (uint64_t)(uintptr_t)&mine, kScopeWave);
    const int r = l & 31, gq = l >> 5;
    v16i_ d;
    for (int i = 0; i < 16; i++)
    {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * gq;
        int acc = c[i];
        for (int k = 0; k < 32; k++)
        {
            const int ga = k >> 4, sa = k & 15;
            const MfmaPub* pa = (const MfmaPub*)(uintptr_t)sn->val[row + 32 * ga];
            const MfmaPub* pb = (const MfmaPub*)(uintptr_t)sn->val[r + 32 * ga];
            if (!pa || !pb) die("MFMA with inactive lanes");
            const int av = (int)(int8_t)((unsigned)pa->a[sa >> 2] >> (8 * (sa & 3)));
            const int bv = (int)(int8_t)((unsigned)pb->b[sa >> 2] >> (8 * (sa & 3)));
            acc += av * bv;
        }
        d[i] = acc;
    }
    done(sn);
    done(wave_exchange(site + (1 << 24), 0, kScopeWave));
    return d;
}
} // namespace emu

#define EMU_SITE __COUNTER__
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu::update_dpp(EMU_SITE, old, src, ctrl, rm, bm, bc)
#define __builtin_amdgcn_mov_dpp(src, ctrl, rm, bm, bc) emu::update_dpp(EMU_SITE, src, src, ctrl, rm, bm, bc)
#define __builtin_amdgcn_readlane(v, lane) emu::readlane(EMU_SITE, v, lane)
#define __builtin_amdgcn_readfirstlane(v) emu::readfirstlane(EMU_SITE, v)
#define __builtin_amdgcn_ballot_w64(p) emu::ballot(EMU_SITE, p)
#define __ballot(p) emu::ballot(EMU_SITE, (p) != 0)
#define __shfl(v, lane, ...) emu::shfl(EMU_SITE, v, lane, ##__VA_ARGS__)
#define __shfl_xor(v, m, ...) emu::shfl(EMU_SITE, v, emu::lane_id() ^ (m), ##__VA_ARGS__)
#define __builtin_amdgcn_wave_barrier() emu::wave_barrier(EMU_SITE)
#define __builtin_amdgcn_ds_swizzle(v, pat) emu::ds_swizzle(EMU_SITE, v, pat)
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) emu::permlane_swap(EMU_SITE, a, b, 16)
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) emu::permlane_swap(EMU_SITE, a, b, 32)
#define __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, x, y, z) emu::mfma_i32_32x32x32_i8(EMU_SITE, a, b, c)
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_s_barrier() emu::block_barrier()

/* ---- per-lane arithmetic builtins (pure functions) ---- */
inline unsigned __builtin_amdgcn_alignbyte(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)((((uint64_t)hi << 32) | lo) >> (8 * (sh & 3))); }
inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)((((uint64_t)hi << 32) | lo) >> (sh & 31)); }
inline unsigned __builtin_amdgcn_perm(unsigned s0, unsigned s1, unsigned sel)
{   /* v_perm_b32: byte k of the result = byte sel[k] of { s0 (bytes 7..4), s1 (bytes 3..0) }; selectors 8-11: sign of byte 1 / 3 / 5 / 7, 12: 0x00, >= 13: 0xFF */
    const uint64_t both = ((uint64_t)s0 << 32) | s1;
    unsigned r = 0;
    for (int k = 0; k < 4; k++)
    {
        const unsigned s = (sel >> (8 * k)) & 0xFF;
        unsigned b;
        if (s <= 7) b = (unsigned)(both >> (8 * s)) & 0xFF;
        else if (s <= 11) b = ((both >> (8 * (2 * (s - 8) + 1) + 7)) & 1) ? 0xFF : 0x00;
        else if (s == 12) b = 0x00;
        else b = 0xFF;
        r |= b << (8 * k);
    }
    return r;
}
inline unsigned __builtin_amdgcn_sad_u8(unsigned a, unsigned b, unsigned c)
{
    for (int k = 0; k < 4; k++) { const int x = (a >> (8 * k)) & 0xFF, y = (b >> (8 * k)) & 0xFF; c += (unsigned)(x > y ? x - y : y - x); }
    return c;
}
inline unsigned __builtin_amdgcn_sad_u16(unsigned a, unsigned b, unsigned c)
{
    for (int k = 0; k < 2; k++) { const int x = (a >> (16 * k)) & 0xFFFF, y = (b >> (16 * k)) & 0xFFFF; c += (unsigned)(x > y ? x - y : y - x); }
    return c;
}
inline int __mul24(int a, int b) { return (int)(((int64_t)((a << 8) >> 8) * ((b << 8) >> 8))); }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFF) * (b & 0xFFFFFF); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
typedef short emu_s16x2 __attribute__((ext_vector_type(2)));
inline int __builtin_amdgcn_sdot2(emu_s16x2 a, emu_s16x2 b, int c, bool) { return c + (int)a[0] * (int)b[0] + (int)a[1] * (int)b[1]; }
/* v_ashr_pk_u8_i32 (gfx950): { sat_u8(a >> sh), sat_u8(b >> sh) } in the low two bytes */
inline unsigned short __builtin_amdgcn_ashr_pk_u8_i32(int a, int b, unsigned sh)
{
    auto sat = [](int v) { return (unsigned)(v < 0 ? 0 : v > 255 ? 255 : v); };
    return (unsigned short)(sat(a >> (sh & 31)) | (sat(b >> (sh & 31)) << 8));
}
inline int __builtin_amdgcn_sdot4(int a, int b, int c, bool) { for (int k = 0; k < 4; k++) c += (int)(int8_t)(a >> (8 * k)) * (int)(int8_t)(b >> (8 * k)); return c; }

/* ---- atomics (one thread runs all fibers) ---- */
/* (T may carry an address space -- the sources' LDS pointers, __attribute__((address_space(3))) -- which `auto` drops) */
template<class T, class U> inline auto atomicAdd(T* p, U v) { auto o = *p; *p = (decltype(o))(o + (decltype(o))v); return o; }
template<class T, class U> inline auto atomicMin(T* p, U v) { auto o = *p; if ((decltype(o))v < o) *p = (decltype(o))v; return o; }
template<class T, class U> inline auto atomicMax(T* p, U v) { auto o = *p; if ((decltype(o))v > o) *p = (decltype(o))v; return o; }
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd(p, v)
#define __hip_atomic_fetch_min(p, v, order, scope) atomicMin(p, v)
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (void)(*(p) = (v))
#define __HIP_MEMORY_SCOPE_WAVEFRONT 1
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __HIP_MEMORY_SCOPE_SYSTEM 4
inline void __threadfence() {}
inline void __threadfence_block() {}
#ifndef __clang__
template<class T> inline void __builtin_nontemporal_store(T v, T* p) { *p = v; }
template<class T> inline T __builtin_nontemporal_load(const T* p) { return *p; }
#endif

/* ---- integer min / max / abs as HIP offers them for mixed operands ---- */
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline long min(long a, long b) { return a < b ? a : b; }
inline long max(long a, long b) { return a > b ? a : b; }
inline long long min(long long a, long long b) { return a < b ? a : b; }
inline long long max(long long a, long long b) { return a > b ? a : b; }
inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
inline float min(float a, float b) { return a < b ? a : b; }
inline float max(float a, float b) { return a > b ? a : b; }
inline int min(int a, unsigned b) { return a < (int)b ? a : (int)b; }
inline int min(unsigned a, int b) { return (int)a < b ? (int)a : b; }
inline int max(int a, unsigned b) { return a > (int)b ? a : (int)b; }
inline int max(unsigned a, int b) { return (int)a > b ? (int)a : b; }
inline long min(long a, int b) { return a < b ? a : b; }
inline long min(int a, long b) { return a < b ? a : b; }
inline long max(long a, int b) { return a > b ? a : b; }
inline long max(int a, long b) { return a > b ? a : b; }
using std::abs;

/* ---- the runtime API the host side calls: device memory is the heap, streams and events do nothing ---- */
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template<class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = nullptr)
{
    for (size_t y = 0; y < h; y++) memmove((char*)d + y * dp, (const char*)s + y * sp, w);
    return hipSuccess;
}
inline hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind k) { return hipMemcpy2DAsync(d, dp, s, sp, w, h, k); }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (hipStream_t)malloc(1); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamGetDevice(hipStream_t, hipDevice_t* d) { *d = 0; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(1); return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.f; return hipSuccess; }      /* (no clock here: a millisecond, so that callers' rates are finite) */
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : e == hipErrorOutOfMemory ? "hipErrorOutOfMemory" : "hip error (emulation)"; }
inline const char* hipGetErrorName(hipError_t e) { return hipGetErrorString(e); }
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; int major, minor; };
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof(*p)); strcpy(p->name, "host emulation"); strcpy(p->gcnArchName, "gfx950-emu"); p->multiProcessorCount = 1; return hipSuccess; }
inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = *t = (size_t)8 << 30; return hipSuccess; }
