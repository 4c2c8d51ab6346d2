// kern_planes.hip -- quarter-pel phase planes of a reference picture.
//
// The reference interpolates a fresh block for every sub-pel candidate of every PU (motion.cpp:1797-1801:
// luma_hpp | luma_vpp | luma_hvpp into a 64x64 stack buffer).  The interpolated value at a position is a pure
// function of the reference pixels around it, so on a device with 288 GB of HBM we compute each of the 15 phases
// ONCE per reference picture -- the idea x265 itself uses for its lookahead's half-pel planes (lowres.h:104-124)
// extended to quarter-pel -- and every sub-pel candidate of the motion search becomes a plain SATD against
// plane[yFrac*4 + xFrac] at an integer offset.  Arithmetic per phase is exactly the primitive's:
//   yFrac == 0: luma_hpp (ipfilter.cpp:79-118)     xFrac == 0: luma_vpp (:164-203)
//   else:       luma_hvpp = hps (row extended, :120-162) -> vsp (:319-369)
// Slot 0 of the output receives a copy of the reference plane, so that the motion search addresses every candidate --
// integer or sub-pel -- as one buffer base plus a 32-bit offset.
// Streaming kernel: 1 plane read, 16 written.  Two forms: the 8-bit build filters with packed dot products on 128x8 tiles
// (below), the 16-bit builds with scalar MACs on 64x16 tiles (at the end of the file; already at the store ceiling).
#include "xh_mc.h"
#include <cstdlib>
#include "../../include/x265hip_frame.h"
using namespace xh;

#ifndef XP_NT_STORES
#define XP_NT_STORES 1      // the phase planes are written once and read milliseconds later: streaming stores (profiles/r04_nt_ab.txt: planes 0.586 -> 0.563 ms at 4K 10 bit)
#endif
namespace {

constexpr int TW = 64, TH = 16, SROWS = TH + 7, SCOLS = 76;          // source tile: rows y0-3 .. y0+TH+3, cols x0-4 .. x0+71
constexpr int SSTRIDE = sizeof(pixel) == 1 ? 76 : 78;               // odd number of dwords per row

__device__ __forceinline__ void store4g(pixel* p, const int* v)      // 4 pixels, p 4-pixel aligned
{
#if X265_DEPTH == 8
    int v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
    XH_PIN_VGPRS("+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));      // see xh_mc.h store4(): v_ashr_pk_u8_i32 miscompile
#if XP_NT_STORES
    __builtin_nontemporal_store((uint32_t)v0 | ((uint32_t)v1 << 8) | ((uint32_t)v2 << 16) | ((uint32_t)v3 << 24), (uint32_t*)p);
#else
    *(uint32_t*)p = (uint32_t)v0 | ((uint32_t)v1 << 8) | ((uint32_t)v2 << 16) | ((uint32_t)v3 << 24);
#endif
#else
    uint2 a; a.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16); a.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
#if XP_NT_STORES
    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
    u2v b = { a.x, a.y };
    __builtin_nontemporal_store(b, (u2v*)p);
#else
    *(uint2*)p = a;
#endif
#endif
}

#if X265_DEPTH == 8
// ---- 8-bit build: the filters as packed dot products -------------------------------------------------------------
// Pixels are kept as signed bytes q = p - 128, so an 8-tap sum is two v_dot4_i32_i8 (taps are int8):
//   sum(t*p) = dot(t, q) + 128*64, hence the 14-bit intermediate of luma_hps (sum - 8192, shift 0 at 8 bit) IS the dot
// and the vertical pass over intermediates is four v_dot2_i32_i16 on vertically packed pairs.  Horizontal windows come
// from the row-major tile, vertical windows from a byte-transposed copy (v_perm 4x4 transposes), both realigned with
// v_alignbyte/v_alignbit.  ~2x fewer vector instructions per pixel than the scalar-MAC form below.
// Tile of the 8-bit kernel: 128 x 8 outputs (128-byte row segments: full-line stores, profiles/micro/RESULTS.md), 16 source rows,
// 256 work items in each of its two compute phases.
constexpr int TW8 = 128, TH8 = 8, TROWS8 = 16, DW8 = TW8 / 4 + 3;
constexpr int SROW = 48;                    // dwords per row of the row-major tile (35 used; 2 rows = 32 banks apart)
constexpr int NT = 4;                       // tiles per workgroup (vertical walk)
__device__ __forceinline__ constexpr uint32_t pk4(int a, int b, int c, int d) { return (uint32_t)(uint8_t)a | ((uint32_t)(uint8_t)b << 8) | ((uint32_t)(uint8_t)c << 16) | ((uint32_t)(uint8_t)d << 24); }
__device__ __forceinline__ constexpr uint32_t pk2(int a, int b) { return (uint32_t)(uint16_t)a | ((uint32_t)(uint16_t)b << 16); }
// (Builtins, not inline asm: dot results have read-after-write hazards against other VALU opcodes that the compiler
// only pads with wait states when it can see both instructions.)
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot8(uint32_t lo, uint32_t hi, uint32_t tlo, uint32_t thi, int acc)
{
    return __builtin_amdgcn_sdot4((int)lo, (int)tlo, __builtin_amdgcn_sdot4((int)hi, (int)thi, acc, false), false);
}
__device__ __forceinline__ int dot2(uint32_t a, uint32_t t, int acc)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, t), acc, false);
}
// four pixels clip(v[i] >> SH, 0, 255) packed into a dword: v_ashr_pk_u8_i32 shifts, saturates and packs two at a time
template<int SH> __device__ __forceinline__ uint32_t sat_pack4(int a, int b, int c, int d)
{
    u16x2 v = { __builtin_amdgcn_ashr_pk_u8_i32(a, b, SH), __builtin_amdgcn_ashr_pk_u8_i32(c, d, SH) };
    return __builtin_bit_cast(uint32_t, v);
}
// Plane stores: the plane / row part of the address is wave-uniform (an SGPR base), the thread part a 32-bit offset.
__device__ __forceinline__ void store_px4(pixel* uniformBase, uint32_t threadOff, uint32_t v)
{
#if XP_NT_STORES
    __builtin_nontemporal_store(v, (uint32_t*)((char*)uniformBase + (size_t)threadOff));
#else
    *(uint32_t*)((char*)uniformBase + (size_t)threadOff) = v;
#endif
}
__device__ __forceinline__ uint32_t pack_u8(int a, int b, int c, int d)
{
    XH_PIN_VGPRS("+v"(a), "+v"(b), "+v"(c), "+v"(d));          // see xh_mc.h store4(): v_ashr_pk_u8_i32 miscompile
    return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
}
// the 8 bytes starting `sh` bytes into the 12-byte window (w0, w1, w2), sh = 0..3 (compile-time)
template<int SH> __device__ __forceinline__ void window8(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t& lo, uint32_t& hi)
{
    if (SH == 0) { lo = w0; hi = w1; }
    else { lo = __builtin_amdgcn_alignbyte(w1, w0, SH); hi = __builtin_amdgcn_alignbyte(w2, w1, SH); }
}

__global__ __launch_bounds__(256) void subpel_planes_kernel(const pixel* __restrict__ ref, intptr_t stride, int rows,
                                                            pixel* __restrict__ out, int64_t planeElems, int by0, int rowsEnd)
{
    __shared__ uint32_t s_src[TROWS8 * SROW];        // rows y0-3 .. y0+12, bytes x0-4 .. x0+135, as q = p - 128
    // vertically packed copies, [row group][column]: a thread's four adjacent columns are one 16-byte LDS access
    __shared__ __attribute__((aligned(16))) uint32_t s_srcT[(TROWS8 / 4) * TW8];     // dword = rows 4g .. 4g+3 of one column (bytes)
    __shared__ __attribute__((aligned(16))) uint32_t s_imT[3][(TROWS8 / 2) * TW8];   // [xFrac-1]: dword = rows 2p, 2p+1 of one column (14-bit intermediates, int16)
    const int x0 = blockIdx.x * TW8, t = threadIdx.x;
    constexpr uint32_t TLO[4] = { 0, pk4(-1, 4, -10, 58), pk4(-1, 4, -11, 40), pk4(0, 1, -5, 17) };
    constexpr uint32_t THI[4] = { 0, pk4(17, -5, 1, 0), pk4(40, -11, 4, -1), pk4(58, -10, 4, -1) };
    constexpr uint32_t TP[4][4] = { { 0, 0, 0, 0 }, { pk2(-1, 4), pk2(-10, 58), pk2(17, -5), pk2(1, 0) },
                                    { pk2(-1, 4), pk2(-11, 40), pk2(40, -11), pk2(4, -1) }, { pk2(0, 1), pk2(-5, 17), pk2(58, -10), pk2(4, -1) } };

    // A workgroup walks NT vertically adjacent tiles; the source rows of tile i+1 are fetched into registers while tile i is
    // being filtered, so the HBM latency of the only read of this kernel hides behind the arithmetic.
    const int rbA = t / DW8, dcA = t - rbA * DW8, gxA = x0 - 4 + 4 * dcA;         // loader role (t < 4*DW8): 4 rows x 1 dword
    auto fetch = [&](int y0_, uint32_t (&v)[4]) {
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const int gy = min(max(y0_ - 3 + 4 * rbA + r, 0), rows - 1);
            const pixel* row = ref + (intptr_t)gy * stride;
            if (gxA >= 0 && gxA + 3 < (int)stride) v[r] = *(const uint32_t*)(row + gxA);
            else
            {
                v[r] = 0;
#pragma unroll
                for (int e = 0; e < 4; e++) v[r] |= (uint32_t)row[min(max(gxA + e, 0), (int)stride - 1)] << (8 * e);
            }
        }
    };
    uint32_t nxt[4] = { 0, 0, 0, 0 };
    if (t < 4 * DW8 && (by0 + (int)blockIdx.y) * NT * TH8 < rowsEnd) fetch((by0 + (int)blockIdx.y) * NT * TH8, nxt);
    for (int it = 0; it < NT; it++)
    {
    const int y0 = ((by0 + (int)blockIdx.y) * NT + it) * TH8;
    if (y0 >= rowsEnd) break;                                                          // uniform
    // ---- A: slot-0 copy, signed bytes, transposed copy (coordinates were clamped into the allocation by fetch) ----
    if (t < 4 * DW8)
    {
        const int rb = rbA, dc = dcA, gx = gxA;
        uint32_t v[4];
#pragma unroll
        for (int r = 0; r < 4; r++)
        {
            const int gyu = y0 - 3 + 4 * rb + r;
            v[r] = nxt[r];
            if (dc >= 1 && dc <= TW8 / 4 && 4 * rb + r >= 3 && 4 * rb + r < 3 + TH8 && gyu < rowsEnd && gx < (int)stride)
                *(uint32_t*)(out + (intptr_t)gyu * stride + gx) = v[r];            // slot 0: the reference plane itself
            v[r] ^= 0x80808080u;
            s_src[(4 * rb + r) * SROW + dc] = v[r];
        }
        if (dc >= 1 && dc <= TW8 / 4)
        {   // 4x4 byte transpose: c_i = (v0.b_i, v1.b_i, v2.b_i, v3.b_i)
            const uint32_t t0 = __builtin_amdgcn_perm(v[1], v[0], 0x05010400), t1 = __builtin_amdgcn_perm(v[1], v[0], 0x07030602);
            const uint32_t t2 = __builtin_amdgcn_perm(v[3], v[2], 0x05010400), t3 = __builtin_amdgcn_perm(v[3], v[2], 0x07030602);
            const int col = 4 * (dc - 1);
            uint4 cols;
            cols.x = __builtin_amdgcn_perm(t2, t0, 0x05040100); cols.y = __builtin_amdgcn_perm(t2, t0, 0x07060302);
            cols.z = __builtin_amdgcn_perm(t3, t1, 0x05040100); cols.w = __builtin_amdgcn_perm(t3, t1, 0x07060302);
            *(uint4*)(s_srcT + rb * TW8 + col) = cols;
        }
        if (it + 1 < NT && y0 + TH8 < rowsEnd) fetch(y0 + TH8, nxt);                     // in flight during H / V / HV
    }
    __syncthreads();

    {   // ---- H (all 256 threads): 2 rows x 4 pixels per thread: intermediates for the three xFracs + the yFrac == 0 planes ----
        const int rb = t >> 6, pb = (t >> 5) & 1, q = t & 31;
        const int gx = x0 + 4 * q;
        const uint32_t hoff = (uint32_t)(4 * rb) * (uint32_t)stride + 4 * q;         // thread part of the store address (bytes = pixels)
        {
            uint32_t lo[2][4], hi[2][4];
#pragma unroll
            for (int r = 0; r < 2; r++)
            {
                const uint32_t* w = s_src + (4 * rb + 2 * pb + r) * SROW + q;
                const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
                window8<1>(w0, w1, w2, lo[r][0], hi[r][0]); window8<2>(w0, w1, w2, lo[r][1], hi[r][1]);
                window8<3>(w0, w1, w2, lo[r][2], hi[r][2]); lo[r][3] = w1; hi[r][3] = w2;
            }
#pragma unroll
            for (int xf = 1; xf < 4; xf++)
            {
                // d = sum(t*p) + 32 = intermediate + 8224: the bias rides along into the LDS copy and is absorbed by the
                // rounding constant of the second pass (64 * 8224 == (1 << 11) + (8192 << 6))
                int d[2][4];
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int e = 0; e < 4; e++) d[r][e] = dot8(lo[r][e], hi[r][e], TLO[xf], THI[xf], 8192 + 32);
                {
                    uint4 pr;
                    pr.x = __builtin_amdgcn_perm((uint32_t)d[1][0], (uint32_t)d[0][0], 0x05040100); pr.y = __builtin_amdgcn_perm((uint32_t)d[1][1], (uint32_t)d[0][1], 0x05040100);
                    pr.z = __builtin_amdgcn_perm((uint32_t)d[1][2], (uint32_t)d[0][2], 0x05040100); pr.w = __builtin_amdgcn_perm((uint32_t)d[1][3], (uint32_t)d[0][3], 0x05040100);
                    *(uint4*)(s_imT[xf - 1] + (2 * rb + pb) * TW8 + 4 * q) = pr;
                }
#pragma unroll
                for (int r = 0; r < 2; r++)
                {
                    const int tr = 4 * rb + 2 * pb + r, gy = y0 + tr - 3;
                    if (tr >= 3 && tr < 3 + TH8 && gy < rowsEnd && gx < (int)stride)
                        store_px4(out + (int64_t)xf * planeElems, (uint32_t)((y0 - 3 + 2 * pb + r) * (int)stride + x0) + hoff, sat_pack4<6>(d[r][0], d[r][1], d[r][2], d[r][3]));
                }
            }
        }
    }
    __syncthreads();

    if (t < 192)
    {   // ---- HV: 4 columns x 4 rows per thread and xFrac: vertical taps over the intermediates ----
        const int xf = __builtin_amdgcn_readfirstlane(t >> 6), i = t & 63, rq = i >> 5, cq = i & 31;   // one xFrac per wavefront
        const int gx = x0 + 4 * cq;
        const uint32_t voff = (uint32_t)(4 * rq) * (uint32_t)stride + 4 * cq;
        pixel* xbase = out + (int64_t)(xf + 1) * planeElems;
        uint32_t P[4][5], Q[4][5];
        {
            uint4 w[6];
#pragma unroll
            for (int k = 0; k < 6; k++) w[k] = *(const uint4*)(s_imT[xf] + (2 * rq + k) * TW8 + 4 * cq);
#pragma unroll
            for (int k = 0; k < 5; k++) { P[0][k] = w[k].x; P[1][k] = w[k].y; P[2][k] = w[k].z; P[3][k] = w[k].w; }
            const uint32_t p5[4] = { w[5].x, w[5].y, w[5].z, w[5].w };
#pragma unroll
            for (int c = 0; c < 4; c++)
            {
#pragma unroll
                for (int k = 0; k < 4; k++) Q[c][k] = __builtin_amdgcn_alignbit(P[c][k + 1], P[c][k], 16);
                Q[c][4] = __builtin_amdgcn_alignbit(p5[c], P[c][4], 16);
            }
        }
        const int offset2 = (1 << 11) + (XH_IF_INTERNAL_OFFS << XH_IF_FILTER_PREC);
#pragma unroll
        for (int yf = 1; yf < 4; yf++)
#pragma unroll
            for (int e = 0; e < 4; e++)
            {
                const int gy = y0 + 4 * rq + e;
                int o[4];
#pragma unroll
                for (int c = 0; c < 4; c++)
                {
                    const uint32_t* a = (e & 1) ? Q[c] + (e >> 1) : P[c] + (e >> 1);
                    int sum = offset2 - 64 * (8192 + 32);                              // == 0: the bias of the intermediates is the rounding term
#pragma unroll
                    for (int k = 0; k < 4; k++) sum = dot2(a[k], TP[yf][k], sum);
                    o[c] = sum;
                }
                if (gy < rowsEnd && gx < (int)stride)
                    store_px4(xbase + (int64_t)(yf * 4) * planeElems, (uint32_t)((y0 + e) * (int)stride + x0) + voff, sat_pack4<12>(o[0], o[1], o[2], o[3]));
            }
    }
    else
    {   // ---- V (the fourth wavefront): 4 columns x 4 rows per thread straight from the pixels: the xFrac == 0 planes ----
        const int i = t - 192, rq = i >> 5, cq = i & 31;
        const int gx = x0 + 4 * cq;
        const uint32_t voff = (uint32_t)(4 * rq) * (uint32_t)stride + 4 * cq;
        uint32_t lo[4][4], hi[4][4];
        {
            uint4 g[3];
#pragma unroll
            for (int k = 0; k < 3; k++) g[k] = *(const uint4*)(s_srcT + (rq + k) * TW8 + 4 * cq);
            const uint32_t W[4][3] = { { g[0].x, g[1].x, g[2].x }, { g[0].y, g[1].y, g[2].y }, { g[0].z, g[1].z, g[2].z }, { g[0].w, g[1].w, g[2].w } };
#pragma unroll
            for (int c = 0; c < 4; c++)
            {
                window8<0>(W[c][0], W[c][1], W[c][2], lo[c][0], hi[c][0]); window8<1>(W[c][0], W[c][1], W[c][2], lo[c][1], hi[c][1]);
                window8<2>(W[c][0], W[c][1], W[c][2], lo[c][2], hi[c][2]); window8<3>(W[c][0], W[c][1], W[c][2], lo[c][3], hi[c][3]);
            }
        }
#pragma unroll
        for (int yf = 1; yf < 4; yf++)
#pragma unroll
            for (int e = 0; e < 4; e++)
            {
                const int gy = y0 + 4 * rq + e;
                int o[4];
#pragma unroll
                for (int c = 0; c < 4; c++) o[c] = dot8(lo[c][e], hi[c][e], TLO[yf], THI[yf], 8192 + 32);
                if (gy < rowsEnd && gx < (int)stride)
                    store_px4(out + (int64_t)(yf * 4) * planeElems, (uint32_t)((y0 + e) * (int)stride + x0) + voff, sat_pack4<6>(o[0], o[1], o[2], o[3]));
            }
    }
    __syncthreads();                         // the next tile's A rewrites s_src / s_srcT, which V has just read
    }
}
#else
// TILED: slots 1..15 are written as tiles of 16 x 4 pixels (xh_mc.h tile_off; a wavefront's store then fills four whole 128-byte lines); slot 0 by rows
template<bool TILED>
__global__ __launch_bounds__(256) void subpel_planes_kernel(const pixel* __restrict__ ref, intptr_t stride, int rows,
                                                            pixel* __restrict__ out, int64_t planeElems, int by0, int rowsEnd)
{
    __shared__ __attribute__((aligned(16))) pixel s_src[SROWS * SSTRIDE];
    __shared__ __attribute__((aligned(16))) int16_t s_im[3][SROWS * TW];
    const int x0 = blockIdx.x * TW, y0 = (by0 + (int)blockIdx.y) * TH, t = threadIdx.x;
    const lpixel* src = (const lpixel*)s_src;
    const int headRoom = XH_IF_INTERNAL_PREC - X265_DEPTH;

    // ---- source tile (coordinates clamped into the allocation; border tiles only feed margin pixels) ----
    for (int i = t; i < SROWS * (SCOLS / 4); i += 256)
    {
        const int rr = i / (SCOLS / 4), cq = i - rr * (SCOLS / 4);
        const int gy = min(max(y0 - 3 + rr, 0), rows - 1);
        int v[4];
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
            const int gx = min(max(x0 - 4 + cq * 4 + e, 0), (int)stride - 1);
            v[e] = ref[(intptr_t)gy * stride + gx];
        }
#if X265_DEPTH == 8
        *(lu32*)((lpixel*)s_src + rr * SSTRIDE + cq * 4) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
#else
        *(lu32*)((lpixel*)s_src + rr * SSTRIDE + cq * 4) = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
        *(lu32*)((lpixel*)s_src + rr * SSTRIDE + cq * 4 + 2) = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
#endif
    }
    __syncthreads();

    // ---- stage 1: horizontal 8-tap for the three fractions; 14-bit intermediates + the yFrac == 0 planes ----
    {
        const int shift1 = XH_IF_FILTER_PREC - headRoom, offset1 = (int)((unsigned)-XH_IF_INTERNAL_OFFS << shift1);
        for (int i = t; i < SROWS * (TW / 4); i += 256)
        {
            const int rr = i >> 4, x4 = (i & 15) * 4;
            int a[4], b[4], c4[4], px[11];
            load4u(src + rr * SSTRIDE + x4, a); load4u(src + rr * SSTRIDE + x4 + 4, b); load4u(src + rr * SSTRIDE + x4 + 8, c4);
            // pixels x-3 .. x+7 for x = x0 + x4: tile columns x4+1 .. x4+11
            px[0] = a[1]; px[1] = a[2]; px[2] = a[3]; px[3] = b[0]; px[4] = b[1]; px[5] = b[2]; px[6] = b[3];
            px[7] = c4[0]; px[8] = c4[1]; px[9] = c4[2]; px[10] = c4[3];
            const int gy = y0 - 3 + rr;
            const bool inTile = rr >= 3 && rr < 3 + TH && gy < rowsEnd && x0 + x4 < stride;
#pragma unroll
            for (int xf = 1; xf < 4; xf++)
            {
                int o[4], im[4];
#pragma unroll
                for (int e = 0; e < 4; e++)
                {
                    int s = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) s += px[e + k] * k_lumaTaps[xf][k];
                    o[e] = clip3(0, XH_PIXEL_MAX, (int)(int16_t)((s + 32) >> 6));
                    im[e] = (s + offset1) >> shift1;
                }
                u32x2 pk;
                pk.x = (uint32_t)(uint16_t)im[0] | ((uint32_t)(uint16_t)im[1] << 16);
                pk.y = (uint32_t)(uint16_t)im[2] | ((uint32_t)(uint16_t)im[3] << 16);
                *(lu2*)((lshort*)s_im[xf - 1] + rr * TW + x4) = pk;
                if (inTile) store4g(out + (int64_t)xf * planeElems + (TILED ? (intptr_t)tile_off((uint32_t)(x0 + x4), (uint32_t)gy, (uint32_t)stride) : (intptr_t)gy * stride + x0 + x4), o);
            }
        }
    }
    __syncthreads();

    // ---- stage 2: vertical 8-tap: xFrac == 0 from the pixels, xFrac != 0 from the intermediates ----
    const int y = t >> 4, x4 = (t & 15) * 4;
    const int gy = y0 + y;
    if (gy >= rowsEnd || x0 + x4 >= stride) return;
    pixel* o0 = out + (intptr_t)gy * stride + x0 + x4;
    pixel* ot = TILED ? out + (intptr_t)tile_off((uint32_t)(x0 + x4), (uint32_t)gy, (uint32_t)stride) : o0;      // slots 1..15
    {
        int col[8][4];
#pragma unroll
        for (int k = 0; k < 8; k++) load4u(src + (y + k) * SSTRIDE + x4 + 4, col[k]);
        store4g(o0, col[3]);                                          // slot 0: the reference plane itself (one base for every candidate)
#pragma unroll
        for (int yf = 1; yf < 4; yf++)
        {
            int o[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
            {
                int s = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) s += col[k][e] * k_lumaTaps[yf][k];
                o[e] = clip3(0, XH_PIXEL_MAX, (int)(int16_t)((s + 32) >> 6));
            }
            store4g(ot + (int64_t)(yf * 4) * planeElems, o);
        }
    }
    const int shift2 = XH_IF_FILTER_PREC + headRoom, offset2 = (1 << (shift2 - 1)) + (XH_IF_INTERNAL_OFFS << XH_IF_FILTER_PREC);
#pragma unroll
    for (int xf = 1; xf < 4; xf++)
    {
        int col[8][4];
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            u32x2 a = *(const lu2*)((const lshort*)s_im[xf - 1] + (y + k) * TW + x4);
            col[k][0] = (int)(int16_t)(a.x & 0xFFFF); col[k][1] = (int)(int16_t)(a.x >> 16);
            col[k][2] = (int)(int16_t)(a.y & 0xFFFF); col[k][3] = (int)(int16_t)(a.y >> 16);
        }
#pragma unroll
        for (int yf = 1; yf < 4; yf++)
        {
            int o[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
            {
                int s = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) s += col[k][e] * k_lumaTaps[yf][k];
                o[e] = clip3(0, XH_PIXEL_MAX, (int)(int16_t)((s + offset2) >> shift2));
            }
            store4g(ot + (int64_t)(yf * 4 + xf) * planeElems, o);
        }
    }
}

#endif

} // namespace

extern "C" int x265hip_subpel_planes(void* stream, const void* refPlane, intptr_t stride, int rows, void* outPlanes, int64_t planeElems)
{
    return x265hip_subpel_planes_rows(stream, refPlane, stride, rows, 0, rows, outPlanes, planeElems);
}

// the rows rowFirst .. rowEnd - 1 of the 16 planes (a reference picture that is still being reconstructed grows by CTU rows: frame threads, encoder/frameencoder.cpp:1029-1036).
// The vertical taps of a row read the source rows y - 3 .. y + 4 of the WHOLE plane (clamped at 0 and rows - 1 as the whole-plane call clamps them): a row whose taps reach
// source rows the caller has not filled yet holds no meaning until a later call covers it again -- the caller re-submits the last rows of the previous range
extern "C" int x265hip_subpel_planes_rows(void* stream, const void* refPlane, intptr_t stride, int rows, int rowFirst, int rowEnd, void* outPlanes, int64_t planeElems)
{
    if (!refPlane || !outPlanes || stride < 16 || rows < 8 || (stride & 3) || planeElems < (int64_t)stride * rows || (planeElems & 3) || rowFirst < 0 || rowEnd > rows || rowFirst >= rowEnd)
    { set_error("subpel_planes: bad arguments (stride and planeElems must be multiples of 4; 0 <= rowFirst < rowEnd <= rows)"); return X265HIP_EARG; }
    if (((uintptr_t)refPlane | (uintptr_t)outPlanes) & 7) { set_error("subpel_planes: planes must be 8-byte aligned"); return X265HIP_EARG; }
#if X265_DEPTH == 8
    constexpr int TILE_ROWS = TH8 * NT;
    constexpr int TILE_COLS = TW8;
#else
    constexpr int TILE_ROWS = TH;
    constexpr int TILE_COLS = TW;
#endif
    const int by0 = rowFirst / TILE_ROWS;                 // whole tiles: the rows of the first tile above rowFirst are simply written again (same values)
    dim3 grid((unsigned)((stride + TILE_COLS - 1) / TILE_COLS), (unsigned)((rowEnd - by0 * TILE_ROWS + TILE_ROWS - 1) / TILE_ROWS));
#if X265_DEPTH == 8
    XH_KLAUNCH(subpel_planes_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                       (const pixel*)refPlane, stride, rows, (pixel*)outPlanes, planeElems, by0, rowEnd);
#else
    XH_KLAUNCH(subpel_planes_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const pixel*)refPlane, stride, rows, (pixel*)outPlanes, planeElems, by0, rowEnd);
#endif
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

