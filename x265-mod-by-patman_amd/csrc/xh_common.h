// xh_common.h -- shared definitions for the gfx950 back end (device + host side).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/x265hip.h"
#include "xh_fence.h"

#ifndef X265_DEPTH
#error "build with -DX265_DEPTH=8, 10 or 12"
#endif
#if X265_DEPTH > 8
typedef uint16_t pixel;
typedef uint64_t sse_t;
#else
typedef uint8_t pixel;
typedef uint32_t sse_t;
#endif

#define XH_PIXEL_MAX ((1 << X265_DEPTH) - 1)
#define XH_IF_INTERNAL_PREC 14     // reference common.h:304-310
#define XH_IF_FILTER_PREC 6
#define XH_IF_INTERNAL_OFFS (1 << (XH_IF_INTERNAL_PREC - 1))
#define XH_FENC_STRIDE 64          // reference common.h:71
#define XH_WAVE 64

// Experiment switches (environment variables of the A/B runs under profiles/) exist only in builds with -DX265HIP_EXPERIMENTS; a release library never reads
// the environment, so a stray variable cannot change which kernel runs.
#include <cstdlib>
#ifdef X265HIP_EXPERIMENTS
inline const char* xh_experiment(const char* name) { return getenv(name); }
#else
inline const char* xh_experiment(const char*) { return nullptr; }
#endif

namespace xh {

// A pair of events the NEXT star64_kernel launch of this thread is bracketed with (x265hip_batch_set_timing: the batch host's figure for the longest single kernel of a pass,
// next to its per-stage times); NULL = none.  Cleared by the launch that used it.
struct KernelEvents { hipEvent_t before, after; };
extern thread_local const KernelEvents* tl_star64Events;

// ---- error plumbing ----
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);          // records + returns X265HIP_EDEVICE
[[noreturn]] void fatal(const char* what);             // slot functions cannot return errors

#define XH_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return xh::hip_fail(e_, #call); } while (0)
// every kernel launch of the library: a fence build (xh_fence.h) notes it and waits for it, so that a page fault names its kernel
#ifdef X265HIP_FENCE
#define XH_KLAUNCH(k, ...) do { hipLaunchKernelGGL(k, __VA_ARGS__); xh::launch_note(__FILE__ " " #k, __LINE__); } while (0)
#else
#define XH_KLAUNCH(...) hipLaunchKernelGGL(__VA_ARGS__)
#endif
#define XH_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return xh::hip_fail(e_, "kernel launch"); } while (0)

// An empty volatile asm with "+v" operands pins values to VGPRs at a program point: an ordering barrier for the compiler's scheduler (xh_mc.h store4, star64_body.inc pin).
// XH_EMU is defined by tests/emu/hip/hip_runtime.h, the host emulation the CPU tests compile these sources with: it has no VGPRs (and no scheduler to hold back).
#ifdef XH_EMU
#define XH_PIN_VGPRS(...) ((void)0)
#else
#define XH_PIN_VGPRS(...) asm volatile("" : __VA_ARGS__)
#endif

// ---- device helpers ----
__device__ __forceinline__ int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ pixel clip_pixel(int v) { return (pixel)clip3(0, XH_PIXEL_MAX, v); }
__device__ __forceinline__ int16_t clip16(int v) { return (int16_t)clip3(-32768, 32767, v); }

// Wave-level synchronisation point for data exchanged between lanes through LDS.  Lanes are independent
// threads to the compiler: without a CONVERGENT barrier plus release/acquire fences it may order one lane's
// LDS store after another lane's load of the same word (e.g. by threading the load into both arms of a
// divergent `if (lane == 0)` store).  Costs no instruction beyond the waitcnt the hardware needs anyway.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Full-wave (64 lane) integer sum, result valid in every lane.  Four DPP steps (quad_perm x2, row_half_mirror,
// row_mirror: VALU-rate cross-lane moves inside each 16-lane row) leave the row sum in every lane of the row; the
// four row sums are then combined through v_readlane (SGPRs).  The ds_bpermute butterfly (__shfl_xor) it replaces
// costs a dependent LDS-crossbar round trip per step and dominated the first ME kernel.
__device__ __forceinline__ int row_sum16(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);    // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);    // row_mirror
    return v;
}
__device__ __forceinline__ int wave_sum(int v)
{
    v = row_sum16(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v)
{   // callers' partial sums are < 2^26 * 64 lanes: reduce as 16-bit-safe halves
    unsigned lo = (unsigned)(v & 0xFFFFFFu), mid = (unsigned)((v >> 24) & 0xFFFFFFu), hi = (unsigned)(v >> 48);
    return (unsigned long long)(unsigned)wave_sum((int)lo) + ((unsigned long long)(unsigned)wave_sum((int)mid) << 24) +
           ((unsigned long long)(unsigned)wave_sum((int)hi) << 48);
}

// Workgroups are dealt to the 8 XCDs round-robin (workgroup b runs on XCD b % 8) and every XCD has its own L2.
// Renumbering gives XCD k the k-th CONTIGUOUS eighth of the work list, so neighbouring blocks of a picture -- which
// share reference rows and search aprons -- meet in one L2 instead of being replicated in all eight.
__device__ __forceinline__ int xcd_contiguous_block(int b, int nblocks)
{
    const int xcd = b & 7, idx = b >> 3, q = nblocks >> 3, r = nblocks & 7;
    return xcd * q + min(xcd, r) + idx;
}

// The same in chunks of C consecutive blocks per XCD (keeps neighbours together without handing whole pictures --
// whose search effort differs -- to single XCDs).  Blocks past the last full round of 8*C keep their number.
__device__ __forceinline__ int xcd_chunked_block(int b, int nblocks, int C)
{
    const int round = 8 * C, full = (nblocks / round) * round;
    if (b >= full) return b;
    const int base = (b / round) * round, o = b - base;
    return base + (o & 7) * C + (o >> 3);
}

// HEVC core-transform coefficient by angle index (the numbers of constants.cpp:270-344)
static __device__ const int8_t k_cos33[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                              61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
__device__ __forceinline__ int dct_coef(int kfull, int j)   // kfull = k * (32 / N)
{
    int th = (kfull * (2 * j + 1)) & 127;
    int a = th <= 32 ? th : th <= 64 ? 64 - th : th <= 96 ? th - 64 : 128 - th;
    int v = k_cos33[a];
    return (th > 32 && th <= 96) ? -v : v;
}

} // namespace xh
