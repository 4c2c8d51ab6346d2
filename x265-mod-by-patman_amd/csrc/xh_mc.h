// xh_mc.h -- wave-level building blocks shared by the fused kernels (kern_me.hip, kern_tq.hip):
// typed LDS pointers, 4-pixel ("quad") load/store/SAD helpers and motion compensation of one block into LDS
// (the copy_pp | luma_hpp | luma_vpp | luma_hvpp dispatch of predict.cpp:279-300 / motion.cpp:1797-1801).
#pragma once
#include "xh_common.h"

namespace xh {

__device__ const int8_t k_lumaTaps[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
                                             { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };   // constants.cpp:250-256

// LDS pointers carry their address space explicitly so that every access is a ds_* instruction.  (With generic
// pointers the compiler mixes ds_* at inlined sites with flat_* inside non-inlined helpers; a ds_write followed by
// a flat_load of the same LDS word is not ordered by the hardware and returned stale candidate lists.)
#define XH_LDS __attribute__((address_space(3)))
typedef XH_LDS pixel lpixel;
typedef XH_LDS int16_t lshort;
typedef XH_LDS int lint;
typedef XH_LDS uint32_t lu32;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef XH_LDS u32x2 lu2;

// Geometry + LDS views of one block handled by one wavefront.  w is a multiple of 4; a "quad" is 4 horizontally
// adjacent pixels (one packed dword at 8 bit, two at 16 bit).
struct McCtx
{
    const pixel* fref; intptr_t rs;      // co-located block origin in the reference plane
    lpixel* pred; lshort* immed;         // per-wave LDS: candidate / predicted block (stride w), 14-bit hv intermediate
    int w, h, lane, gsize, qpr, nquads, qdivm;   // lane = index inside the lane GROUP that owns this block; gsize = lanes per group
    __device__ __forceinline__ void setGeometry(int w_, int h_, int lane_, int gsize_ = 64)
    { w = w_; h = h_; lane = lane_; gsize = gsize_; qpr = w_ >> 2; nquads = qpr * h_; qdivm = ((1 << 20) / qpr) + 1; }
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int wsum_u(int v) { return uni(wave_sum(v)); }

// Sum over an aligned group of G lanes (G = 8, 16, 32 or 64); every lane of the group gets the group's sum.
// Small PUs are packed 64/G per wavefront: each group runs its own search (its lanes hold identical control
// state and therefore take identical branches), only these reductions and the LDS exchanges are cooperative.
template<int G> __device__ __forceinline__ int group_sum(int v)
{
    if (G == 64) return wsum_u(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);    // row_half_mirror: 8 lanes
    if (G >= 16) v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror: 16 lanes
    if (G == 32) v += __shfl_xor(v, 16, 64);
    return v;
}

// ---- 4-pixel helpers -------------------------------------------------------------------------
__device__ __forceinline__ void load4(const lpixel* p, int* v)      // p aligned to 4 pixels (LDS)
{
#if X265_DEPTH == 8
    uint32_t a = *(const lu32*)p;
    v[0] = a & 0xFF; v[1] = (a >> 8) & 0xFF; v[2] = (a >> 16) & 0xFF; v[3] = a >> 24;
#else
    u32x2 a = *(const lu2*)p;
    v[0] = a.x & 0xFFFF; v[1] = a.x >> 16; v[2] = a.y & 0xFFFF; v[3] = a.y >> 16;
#endif
}
__device__ __forceinline__ void load4u(const pixel* p, int* v)     // unaligned (global)
{
#if X265_DEPTH == 8
    uint32_t a; __builtin_memcpy(&a, p, 4);
    v[0] = a & 0xFF; v[1] = (a >> 8) & 0xFF; v[2] = (a >> 16) & 0xFF; v[3] = a >> 24;
#else
    u32x2 a; __builtin_memcpy(&a, p, 8);
    v[0] = a.x & 0xFFFF; v[1] = a.x >> 16; v[2] = a.y & 0xFFFF; v[3] = a.y >> 16;
#endif
}
__device__ __forceinline__ void store4(lpixel* p, const int* v)     // aligned (LDS)
{
#if X265_DEPTH == 8
    // ROCm 7.2 / gfx950: clamp(x >> s, 0, 255) pairs feeding a byte pack are selected as v_ashr_pk_u8_i32, whose
    // result the compiler then ORs with bytes 2-3 as if its upper 16 bits were zero -- they keep the old register
    // contents (observed: bytes 2/3 of every packed quad corrupted).  Materialise the four values first.
    int v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
    XH_PIN_VGPRS("+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    *(lu32*)p = (uint32_t)v0 | ((uint32_t)v1 << 8) | ((uint32_t)v2 << 16) | ((uint32_t)v3 << 24);
#else
    u32x2 a; a.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16); a.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
    *(lu2*)p = a;
#endif
}
__device__ __forceinline__ unsigned sad4(const lpixel* f /*LDS aligned*/, const pixel* r /*global*/, unsigned acc)
{
#if X265_DEPTH == 8
    uint32_t a = *(const lu32*)f, b; __builtin_memcpy(&b, r, 4);
    return __builtin_amdgcn_sad_u8(a, b, acc);
#else
    u32x2 a = *(const lu2*)f, b; __builtin_memcpy(&b, r, 8);
    acc = __builtin_amdgcn_sad_u16(a.x, b.x, acc);
    return __builtin_amdgcn_sad_u16(a.y, b.y, acc);
#endif
}
// The lane unit of the size-specialised kernels: 8 bytes of one row (8 pixels at 8 bit, 4 pixels at 16 bit) as packed
// register data.  8-byte units keep row chunks >= 16 bytes for every PU width >= 16, which is what the L1/TA front end
// needs to stay at 4 lanes per clock (profiles/micro/l1bench.hip: 4-byte lanes on 8-byte rows cost 1 lane per clock).
// Tiled phase planes (16-bit builds; me_body.inc XH_TILED, kern_planes.hip, kern_tq.hip): slots 1..15 of the plane buffer in tiles of 16 x 4 pixels = one 128-byte line
// (row pitch rs a multiple of 16, rows a multiple of 4); element offset of pixel (X, Y) inside its slot
__device__ __forceinline__ uint32_t tile_off(uint32_t X, uint32_t Y, uint32_t rs) { return (Y >> 2) * (4u * rs) + ((X >> 4) << 6) + ((Y & 3u) << 4) + (X & 15u); }
typedef u32x2 fquad;
#define XH_UNITPX (8 / (int)sizeof(pixel))
__device__ __forceinline__ fquad ldq(const pixel* p) { u32x2 a; __builtin_memcpy(&a, p, 8); return a; }              // unaligned, global
__device__ __forceinline__ fquad ldf(const lpixel* p) { return *(const lu2*)p; }                                     // 8-byte aligned, LDS
// The same unit at a byte-misaligned address: one DWORD-ALIGNED 12-byte load + two funnel shifts.  The L1/TA front end
// splits a sub-dword-misaligned wide load into dwords at twice the cost, while dword-aligned x2/x3/x4 loads run at the
// full 4 lanes per clock whatever their 8/16-byte alignment (profiles/micro/l1bench.hip).  a = address rounded down to 4, m = address & 3.
__device__ __forceinline__ fquad ldq_a(const char* a, unsigned m)
{
#if defined(XH_LDQ_UNALIGNED) && X265_DEPTH != 8
    // experiment (profiles/r03_ldq_unaligned_ab.txt): the 8 bytes straight from their pixel-aligned address -- no funnel shifts, a misaligned load for half of the lanes
    fquad u; __builtin_memcpy(&u, __builtin_assume_aligned(a + m, 2), 8); return u;
#endif
    struct W3 { uint32_t x, y, z; } w;
#ifdef XH_LDQ_NT      // experiment (profiles/r04_nt_ab.txt): candidate loads as streaming loads
    const uint32_t* pa = (const uint32_t*)__builtin_assume_aligned(a, 4);
    w.x = __builtin_nontemporal_load(pa); w.y = __builtin_nontemporal_load(pa + 1); w.z = __builtin_nontemporal_load(pa + 2);
#else
    __builtin_memcpy(&w, __builtin_assume_aligned(a, 4), 12);
#endif
    fquad r; r.x = __builtin_amdgcn_alignbyte(w.y, w.x, m); r.y = __builtin_amdgcn_alignbyte(w.z, w.y, m);
    return r;
}
#if X265_DEPTH == 8
__device__ __forceinline__ unsigned sadq(fquad f, fquad r, unsigned acc) { return __builtin_amdgcn_sad_u8(f.y, r.y, __builtin_amdgcn_sad_u8(f.x, r.x, acc)); }
__device__ __forceinline__ void unpackq(fquad a, int* v)
{
    v[0] = a.x & 0xFF; v[1] = (a.x >> 8) & 0xFF; v[2] = (a.x >> 16) & 0xFF; v[3] = a.x >> 24;
    v[4] = a.y & 0xFF; v[5] = (a.y >> 8) & 0xFF; v[6] = (a.y >> 16) & 0xFF; v[7] = a.y >> 24;
}
#else
__device__ __forceinline__ unsigned sadq(fquad f, fquad r, unsigned acc) { return __builtin_amdgcn_sad_u16(f.y, r.y, __builtin_amdgcn_sad_u16(f.x, r.x, acc)); }
__device__ __forceinline__ void unpackq(fquad a, int* v) { v[0] = a.x & 0xFFFF; v[1] = a.x >> 16; v[2] = a.y & 0xFFFF; v[3] = a.y >> 16; }
#endif
// 4 pixels at byte offset `bo` from a (wave-uniform) base: dword-aligned load(s) + funnel shift, like ldq_a
__device__ __forceinline__ void load4a(const char* base, uint32_t bo, int* v)
{
    const uint32_t m = bo & 3u;
    const char* a = base + (size_t)(bo - m);
#if X265_DEPTH == 8
    u32x2 w; __builtin_memcpy(&w, __builtin_assume_aligned(a, 4), 8);
    const uint32_t x = __builtin_amdgcn_alignbyte(w.y, w.x, m);
    v[0] = x & 0xFF; v[1] = (x >> 8) & 0xFF; v[2] = (x >> 16) & 0xFF; v[3] = x >> 24;
#else
    const fquad q = ldq_a(a, m);
    v[0] = q.x & 0xFFFF; v[1] = q.x >> 16; v[2] = q.y & 0xFFFF; v[3] = q.y >> 16;
#endif
}
// 11 consecutive pixels starting at p (unaligned, global): what a 4-wide 8-tap horizontal filter needs
__device__ __forceinline__ void load11u(const pixel* p, int* v)
{
#if X265_DEPTH == 8
    uint32_t a[3]; __builtin_memcpy(a, p, 12);
#pragma unroll
    for (int i = 0; i < 11; i++) v[i] = (a[i >> 2] >> (8 * (i & 3))) & 0xFF;
#else
    uint32_t a[6]; __builtin_memcpy(a, p, 24);
#pragma unroll
    for (int i = 0; i < 11; i++) v[i] = (a[i >> 1] >> (16 * (i & 1))) & 0xFFFF;
#endif
}


// ---- unaligned reads from an LDS-staged reference window ----------------------------------------
// 32-bit DS reads at byte-granular addresses are native on gfx950 (unaligned DS access mode); wider misaligned DS
// reads are replayed slowly, so every wide read is issued as separate (volatile => never merged) dwords.
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
typedef const XH_LDS u32_unaligned lu32u;
__device__ __forceinline__ uint32_t lds_u32(const lpixel* p, int byteOff) { return *(lu32u*)((const XH_LDS char*)p + byteOff); }
__device__ __forceinline__ void load4u(const lpixel* p, int* v)
{
#if X265_DEPTH == 8
    uint32_t a = lds_u32(p, 0);
    v[0] = a & 0xFF; v[1] = (a >> 8) & 0xFF; v[2] = (a >> 16) & 0xFF; v[3] = a >> 24;
#else
    uint32_t a = lds_u32(p, 0), b = lds_u32(p, 4);
    v[0] = a & 0xFFFF; v[1] = a >> 16; v[2] = b & 0xFFFF; v[3] = b >> 16;
#endif
}
// ---- 4x4 Hadamard of a difference block in packed 16-bit lanes (satd_4x4 / satd_8x4's transform, pixel.cpp:210-260) --------
// Four pixels of a row travel as packed data (px4); their differences are two registers of two int16 (U, V).  Every butterfly
// between rows and the first butterfly along the row are whole-register v_pk_add/sub_i16; the last butterfly along the row pairs the
// halves of one register, and only sum |coef| is wanted, so it is |a + b| + |a - b| = 2 max(|a|, |b|).  10 bit: |difference| <= 1023,
// three butterfly stages <= 8184 < 2^15; the eight maxima of a 4x4 add up to <= 65472 < 2^16.  12 bit: |difference| <= 4095, three stages <= 32760 still fit int16,
// the eight maxima (<= 262080) do not fit 16 bits: they are added up as 32-bit integers there.
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 pk16(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
#if X265_DEPTH == 8
typedef uint32_t px4;
__device__ __forceinline__ px4 ld4p(const lpixel* p) { return *(const lu32*)p; }                                       // aligned to 4 pixels (LDS)
__device__ __forceinline__ px4 ld4pu(const pixel* p) { uint32_t a; __builtin_memcpy(&a, p, 4); return a; }             // unaligned (global)
__device__ __forceinline__ px4 ld4pu(const lpixel* p) { return lds_u32(p, 0); }                                         // unaligned (LDS)
__device__ __forceinline__ px4 ld4pa(const char* base, uint32_t bo)                                                    // byte offset off a uniform base: aligned loads + funnel shift
{
    const uint32_t m = bo & 3u;
    u32x2 w; __builtin_memcpy(&w, __builtin_assume_aligned(base + (size_t)(bo - m), 4), 8);
    return __builtin_amdgcn_alignbyte(w.y, w.x, m);
}
// pixels (0, 2) and (1, 3) of the four, zero-extended to 16 bits
__device__ __forceinline__ void px4_split(px4 a, s16x2& e, s16x2& o) { e = pk16(a & 0x00FF00FFu); o = pk16(__builtin_amdgcn_perm(0u, a, 0x0C030C01u)); }
#else
typedef u32x2 px4;
__device__ __forceinline__ px4 ld4p(const lpixel* p) { return *(const lu2*)p; }
__device__ __forceinline__ px4 ld4pu(const pixel* p) { u32x2 a; __builtin_memcpy(&a, p, 8); return a; }
__device__ __forceinline__ px4 ld4pu(const lpixel* p) { u32x2 a; a.x = lds_u32(p, 0); a.y = lds_u32(p, 4); return a; }
__device__ __forceinline__ px4 ld4pa(const char* base, uint32_t bo) { const uint32_t m = bo & 3u; return ldq_a(base + (size_t)(bo - m), m); }
__device__ __forceinline__ void px4_split(px4 a, s16x2& e, s16x2& o) { e = pk16(a.x); o = pk16(a.y); }                  // pixels (0, 1) and (2, 3)
#endif
__device__ __forceinline__ void px4_diff(px4 a, px4 b, s16x2& u, s16x2& v) { s16x2 ae, ao, be, bo; px4_split(a, ae, ao); px4_split(b, be, bo); u = ae - be; v = ao - bo; }
// max(|lo|, |hi|) of a register, in its low half
__device__ __forceinline__ u16x2 pk_absmax(s16x2 v)
{
    const s16x2 a = __builtin_elementwise_abs(v), sw = a.yx;
    return __builtin_bit_cast(u16x2, __builtin_elementwise_max(a, sw));
}
__device__ __forceinline__ void had4p(s16x2& a, s16x2& b, s16x2& c, s16x2& d)
{
    const s16x2 t0 = a + b, t1 = a - b, t2 = c + d, t3 = c - d;
    a = t0 + t2; c = t0 - t2; b = t1 + t3; d = t1 - t3;
}
// sum |coef| of the 4x4 Hadamard transform of a - b, rows given as packed pixels
__device__ __forceinline__ int had4x4_pk(const px4 (&a)[4], const px4 (&b)[4])
{
    s16x2 u[4], v[4];
#pragma unroll
    for (int y = 0; y < 4; y++) px4_diff(a[y], b[y], u[y], v[y]);
    had4p(u[0], u[1], u[2], u[3]); had4p(v[0], v[1], v[2], v[3]);
#if X265_DEPTH <= 10
    u16x2 acc = 0;
#pragma unroll
    for (int y = 0; y < 4; y++) { acc += pk_absmax(u[y] + v[y]); acc += pk_absmax(u[y] - v[y]); }
    return 2 * (int)acc.x;
#else
    int acc = 0;
#pragma unroll
    for (int y = 0; y < 4; y++) { acc += (int)pk_absmax(u[y] + v[y]).x; acc += (int)pk_absmax(u[y] - v[y]).x; }
    return 2 * acc;
#endif
}
__device__ __forceinline__ unsigned sad4(const lpixel* f /*LDS aligned*/, const lpixel* r /*LDS window, unaligned*/, unsigned acc)
{
#if X265_DEPTH == 8
    return __builtin_amdgcn_sad_u8(*(const lu32*)f, lds_u32(r, 0), acc);
#else
    u32x2 a = *(const lu2*)f;
    acc = __builtin_amdgcn_sad_u16(a.x, lds_u32(r, 0), acc);
    return __builtin_amdgcn_sad_u16(a.y, lds_u32(r, 4), acc);
#endif
}
__device__ __forceinline__ void load11u(const lpixel* p, int* v)
{
#if X265_DEPTH == 8
    uint32_t a[3] = { lds_u32(p, 0), lds_u32(p, 4), lds_u32(p, 8) };
#pragma unroll
    for (int i = 0; i < 11; i++) v[i] = (a[i >> 2] >> (8 * (i & 3))) & 0xFF;
#else
    uint32_t a[6] = { lds_u32(p, 0), lds_u32(p, 4), lds_u32(p, 8), lds_u32(p, 12), lds_u32(p, 16), lds_u32(p, 20) };
#pragma unroll
    for (int i = 0; i < 11; i++) v[i] = (a[i >> 1] >> (16 * (i & 1))) & 0xFFFF;
#endif
}

// A reference "view": pointer to the pixel co-located with the block's top-left corner + row stride, either in
// global memory (the padded plane) or in LDS (a staged window).  at(x, y) is the pixel x right / y below.
struct GView { const pixel* p; intptr_t s; __device__ __forceinline__ const pixel* at(int x, int y) const { return p + (intptr_t)y * s + x; } };
struct LView { const lpixel* p; int s; __device__ __forceinline__ const lpixel* at(int x, int y) const { return p + y * s + x; } };

#define QUAD_LOOP(c, q, y, x4) for (int q = (c).lane; q < (c).nquads; q += (c).gsize) { const int y = (q * (c).qdivm) >> 20; const int x4 = (q - y * (c).qpr) * 4;
#define QUAD_END }

// ---- sub-pel candidate: build the interpolated block in LDS (motion.cpp:1797-1801 dispatch) ----
template<class C, class V> __device__ __forceinline__ void build_pred(const C& c, const V& ref, int qx, int qy)
{
    const int ix = qx >> 2, iy = qy >> 2;
    const int xf = qx & 3, yf = qy & 3;
    const int headRoom = XH_IF_INTERNAL_PREC - X265_DEPTH;
    if (!(xf | yf))
    {
        QUAD_LOOP(c, q, y, x4)
            int v[4]; load4u(ref.at(ix + x4, iy + y), v); store4(c.pred + y * c.w + x4, v);
        QUAD_END
    }
    else if (!yf)
    {   // luma_hpp, ipfilter.cpp:79-118
        int t[8];
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = k_lumaTaps[xf][i];
        QUAD_LOOP(c, q, y, x4)
            int px[11], o[4];
            load11u(ref.at(ix + x4 - 3, iy + y), px);
#pragma unroll
            for (int e = 0; e < 4; e++)
            {
                int s = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) s += px[e + i] * t[i];
                o[e] = clip3(0, XH_PIXEL_MAX, (int)(int16_t)((s + 32) >> 6));
            }
            store4(c.pred + y * c.w + x4, o);
        QUAD_END
    }
    else if (!xf)
    {   // luma_vpp, ipfilter.cpp:164-203
        int t[8];
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = k_lumaTaps[yf][i];
        QUAD_LOOP(c, q, y, x4)
            int s[4] = { 0, 0, 0, 0 }, o[4];
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                int v[4]; load4u(ref.at(ix + x4, iy + y - 3 + i), v);
#pragma unroll
                for (int e = 0; e < 4; e++) s[e] += v[e] * t[i];
            }
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = clip3(0, XH_PIXEL_MAX, (int)(int16_t)((s[e] + 32) >> 6));
            store4(c.pred + y * c.w + x4, o);
        QUAD_END
    }
    else
    {   // luma_hvpp = hps(row-extended) into the 14-bit intermediate, then vertical sp; ipfilter.cpp:120-162,319-369
        int t[8];
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = k_lumaTaps[xf][i];
        const int shift1 = XH_IF_FILTER_PREC - headRoom, offset1 = (int)((unsigned)-XH_IF_INTERNAL_OFFS << shift1);
        const int rows = c.h + 7, nq = c.qpr * rows;
        for (int q = c.lane; q < nq; q += c.gsize)
        {
            const int y = (q * c.qdivm) >> 20, x4 = (q - y * c.qpr) * 4;
            int px[11];
            load11u(ref.at(ix + x4 - 3, iy + y - 3), px);
            int16_t o[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
            {
                int s = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) s += px[e + i] * t[i];
                o[e] = (int16_t)((s + offset1) >> shift1);
            }
            u32x2 pk; pk.x = (uint16_t)o[0] | ((uint32_t)(uint16_t)o[1] << 16); pk.y = (uint16_t)o[2] | ((uint32_t)(uint16_t)o[3] << 16);
            *(lu2*)(c.immed + y * c.w + x4) = pk;
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = k_lumaTaps[yf][i];
        const int shift2 = XH_IF_FILTER_PREC + headRoom, offset2 = (1 << (shift2 - 1)) + (XH_IF_INTERNAL_OFFS << XH_IF_FILTER_PREC);
        QUAD_LOOP(c, q, y, x4)
            int s[4] = { 0, 0, 0, 0 }, o[4];
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                u32x2 a = *(const lu2*)(c.immed + (y + i) * c.w + x4);
                s[0] += (int)(int16_t)(a.x & 0xFFFF) * t[i]; s[1] += (int)(int16_t)(a.x >> 16) * t[i];
                s[2] += (int)(int16_t)(a.y & 0xFFFF) * t[i]; s[3] += (int)(int16_t)(a.y >> 16) * t[i];
            }
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = clip3(0, XH_PIXEL_MAX, (int)(int16_t)((s[e] + offset2) >> shift2));
            store4(c.pred + y * c.w + x4, o);
        QUAD_END
    }
    wave_sync();
}

} // namespace xh

namespace xh {
// ---- chroma motion compensation of one 4:2:0 block into LDS (Predict::predInterChromaPixel, predict.cpp:340-380: copy | filter_hpp | filter_vpp |
// filter_hps + filter_vsp by the eighth-pel MV's fractions), shared by the chroma SATD terms of the search (me_body.inc) and the chroma TUs of the TQ chain ----
// ---- 14-bit prediction for bi-directional averaging: Predict::predInterLumaShort (predict.cpp:302-338) = convert_p2s | luma_hps | luma_vps |
// luma_hps (row-extended) + luma_vss (ipfilter.cpp:40-57, 120-162, 205-239, 284-317) into `dst` (stride c.w), then addAvg (pixel.cpp:834-854) ----
template<class C, class V> __device__ __forceinline__ void build_pred_short(const C& c, const V& ref, int qx, int qy, lshort* dst)
{
    const int ix = qx >> 2, iy = qy >> 2, xf = qx & 3, yf = qy & 3;
    const int headRoom = XH_IF_INTERNAL_PREC - X265_DEPTH;
    const int shift1 = XH_IF_FILTER_PREC - headRoom, offset1 = (int)((unsigned)-XH_IF_INTERNAL_OFFS << shift1);
    auto put4 = [&](lshort* p, const int (&o)[4]) {
        u32x2 pk; pk.x = (uint16_t)o[0] | ((uint32_t)(uint16_t)o[1] << 16); pk.y = (uint16_t)o[2] | ((uint32_t)(uint16_t)o[3] << 16);
        *(lu2*)p = pk;
    };
    if (!(xf | yf))
    {
        QUAD_LOOP(c, q, y, x4)
            int v[4], o[4]; load4u(ref.at(ix + x4, iy + y), v);
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = (int16_t)((v[e] << headRoom) - XH_IF_INTERNAL_OFFS);
            put4(dst + y * c.w + x4, o);
        QUAD_END
    }
    else if (!yf || !xf)
    {
        int t[8];
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = k_lumaTaps[yf ? yf : xf][i];
        QUAD_LOOP(c, q, y, x4)
            int s[4] = { 0, 0, 0, 0 }, o[4];
            if (!yf)
            {
                int px[11]; load11u(ref.at(ix + x4 - 3, iy + y), px);
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int i = 0; i < 8; i++) s[e] += px[e + i] * t[i];
            }
            else
            {
#pragma unroll
                for (int i = 0; i < 8; i++)
                {
                    int v[4]; load4u(ref.at(ix + x4, iy + y - 3 + i), v);
#pragma unroll
                    for (int e = 0; e < 4; e++) s[e] += v[e] * t[i];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = (int16_t)((s[e] + offset1) >> shift1);
            put4(dst + y * c.w + x4, o);
        QUAD_END
    }
    else
    {
        int t[8];
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = k_lumaTaps[xf][i];
        const int rows = c.h + 7, nq = c.qpr * rows;
        for (int q = c.lane; q < nq; q += c.gsize)
        {
            const int y = (q * c.qdivm) >> 20, x4 = (q - y * c.qpr) * 4;
            int px[11], o[4];
            load11u(ref.at(ix + x4 - 3, iy + y - 3), px);
#pragma unroll
            for (int e = 0; e < 4; e++)
            {
                int s = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) s += px[e + i] * t[i];
                o[e] = (int16_t)((s + offset1) >> shift1);
            }
            put4(c.immed + y * c.w + x4, o);
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = k_lumaTaps[yf][i];
        QUAD_LOOP(c, q, y, x4)
            int s[4] = { 0, 0, 0, 0 }, o[4];
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                u32x2 a = *(const lu2*)(c.immed + (y + i) * c.w + x4);
                s[0] += (int)(int16_t)(a.x & 0xFFFF) * t[i]; s[1] += (int)(int16_t)(a.x >> 16) * t[i];
                s[2] += (int)(int16_t)(a.y & 0xFFFF) * t[i]; s[3] += (int)(int16_t)(a.y >> 16) * t[i];
            }
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = (int16_t)(s[e] >> XH_IF_FILTER_PREC);        // filterVertical_ss: no offset (ipfilter.cpp:284-317)
            put4(dst + y * c.w + x4, o);
        QUAD_END
    }
    wave_sync();
}
// addAvg (pixel.cpp:834-854): c.pred = clip((a + b + offset) >> shift), shift = 15 - depth, offset = (1 << (shift - 1)) + 2 * 8192
template<class C> __device__ __forceinline__ void add_avg(const C& c, const lshort* a, const lshort* b)
{
    const int shiftNum = XH_IF_INTERNAL_PREC + 1 - X265_DEPTH, offset = (1 << (shiftNum - 1)) + 2 * XH_IF_INTERNAL_OFFS;
    QUAD_LOOP(c, q, y, x4)
        const u32x2 u = *(const lu2*)(a + y * c.w + x4), v = *(const lu2*)(b + y * c.w + x4);
        int o[4];
        o[0] = clip3(0, XH_PIXEL_MAX, ((int)(int16_t)(u.x & 0xFFFF) + (int)(int16_t)(v.x & 0xFFFF) + offset) >> shiftNum);
        o[1] = clip3(0, XH_PIXEL_MAX, ((int)(int16_t)(u.x >> 16) + (int)(int16_t)(v.x >> 16) + offset) >> shiftNum);
        o[2] = clip3(0, XH_PIXEL_MAX, ((int)(int16_t)(u.y & 0xFFFF) + (int)(int16_t)(v.y & 0xFFFF) + offset) >> shiftNum);
        o[3] = clip3(0, XH_PIXEL_MAX, ((int)(int16_t)(u.y >> 16) + (int)(int16_t)(v.y >> 16) + offset) >> shiftNum);
        store4(c.pred + y * c.w + x4, o);
    QUAD_END
    wave_sync();
}

struct CCtx { const pixel* ref[2]; intptr_t rs; lpixel* fenc[2]; lpixel* pred; lshort* immed; int w, h, qpr, nquads, lane; bool on; };
__device__ const int8_t k_chromaTaps[8][4] = { { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
                                               { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };   // constants.cpp:258-268
template<int G> __device__ __forceinline__ void gsync() { if (G > 64) __syncthreads(); else wave_sync(); }

template<int G> __device__ __forceinline__ void chroma_pred(const CCtx& cc, int plane, int mvx, int mvy)
{
    const pixel* r = cc.ref[plane] + (mvx >> 3) + (intptr_t)(mvy >> 3) * cc.rs;
    const int xf = mvx & 7, yf = mvy & 7, w = cc.w;
    const int headRoom = XH_IF_INTERNAL_PREC - X265_DEPTH;
    int t[4];
    if (!(xf | yf))
    {
        for (int q = cc.lane; q < cc.nquads; q += G)
        {
            const int y = q / cc.qpr, x4 = (q - y * cc.qpr) * 4;
            int v[4]; load4u(r + (intptr_t)y * cc.rs + x4, v); store4(cc.pred + y * w + x4, v);
        }
    }
    else if (!yf || !xf)
    {   // filter_hpp / filter_vpp (ipfilter.cpp:79-118, 164-203 with N = 4)
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = k_chromaTaps[yf ? yf : xf][i];
        const intptr_t step = yf ? cc.rs : 1;
        for (int q = cc.lane; q < cc.nquads; q += G)
        {
            const int y = q / cc.qpr, x4 = (q - y * cc.qpr) * 4;
            const pixel* p = r + (intptr_t)y * cc.rs + x4 - step;
            int o[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
            {
                int sum = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) sum += (int)p[e + i * step] * t[i];
                o[e] = clip3(0, XH_PIXEL_MAX, (int)(int16_t)((sum + 32) >> 6));
            }
            store4(cc.pred + y * w + x4, o);
        }
    }
    else
    {   // filter_hps (row-extended) into the 14-bit intermediate, then filter_vsp (ipfilter.cpp:120-162, 319-360 with N = 4)
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = k_chromaTaps[xf][i];
        const int shift1 = XH_IF_FILTER_PREC - headRoom, offset1 = (int)((unsigned)-XH_IF_INTERNAL_OFFS << shift1);
        const int nq = cc.qpr * (cc.h + 3);
        for (int q = cc.lane; q < nq; q += G)
        {
            const int y = q / cc.qpr, x4 = (q - y * cc.qpr) * 4;
            const pixel* p = r + (intptr_t)(y - 1) * cc.rs + x4 - 1;
            int16_t o[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
            {
                int sum = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) sum += (int)p[e + i] * t[i];
                o[e] = (int16_t)((sum + offset1) >> shift1);
            }
            u32x2 pk; pk.x = (uint16_t)o[0] | ((uint32_t)(uint16_t)o[1] << 16); pk.y = (uint16_t)o[2] | ((uint32_t)(uint16_t)o[3] << 16);
            *(lu2*)(cc.immed + y * w + x4) = pk;
        }
        gsync<G>();
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = k_chromaTaps[yf][i];
        const int shift2 = XH_IF_FILTER_PREC + headRoom, offset2 = (1 << (shift2 - 1)) + (XH_IF_INTERNAL_OFFS << XH_IF_FILTER_PREC);
        for (int q = cc.lane; q < cc.nquads; q += G)
        {
            const int y = q / cc.qpr, x4 = (q - y * cc.qpr) * 4;
            int sum[4] = { 0, 0, 0, 0 }, o[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
            {
                const u32x2 a = *(const lu2*)(cc.immed + (y + i) * w + x4);
                sum[0] += (int)(int16_t)(a.x & 0xFFFF) * t[i]; sum[1] += (int)(int16_t)(a.x >> 16) * t[i];
                sum[2] += (int)(int16_t)(a.y & 0xFFFF) * t[i]; sum[3] += (int)(int16_t)(a.y >> 16) * t[i];
            }
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = clip3(0, XH_PIXEL_MAX, (int)(int16_t)((sum[e] + offset2) >> shift2));
            store4(cc.pred + y * w + x4, o);
        }
    }
    gsync<G>();
}
} // namespace xh

namespace xh {
// convenience: the whole-plane (global memory) view of a context
template<class C> __device__ __forceinline__ GView plane_view(const C& c) { GView v; v.p = c.fref; v.s = c.rs; return v; }
template<class C> __device__ __forceinline__ void build_pred(const C& c, int qx, int qy) { build_pred(c, plane_view(c), qx, qy); }
}
