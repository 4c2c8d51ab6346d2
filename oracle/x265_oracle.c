/*
 * x265_oracle.c -- TEST INFRASTRUCTURE ONLY (see x265_oracle.h).
 *
 * Plain-C restatement of the arithmetic of the reference's C primitives.  It is
 * written from the algorithm (what each slot computes, with the reference's
 * rounding/clipping/cast points), not from the reference's SWAR/butterfly code.
 * Every function cites the reference location whose results it must reproduce.
 * Pinned by tests/test_oracle_vs_ref.py (against oracle/_ref built from the real
 * sources) and tests/golden/ (vectors captured from that build).
 */
#include "x265_oracle.h"
#include <stdlib.h>
#include <string.h>

#define PIXEL_MAX ((1 << X265_DEPTH) - 1)
#define IF_INTERNAL_PREC 14   /* common.h:304-310 */
#define IF_FILTER_PREC 6
#define IF_INTERNAL_OFFS (1 << (IF_INTERNAL_PREC - 1))

static inline int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline xo_pixel clip_pixel(int v) { return (xo_pixel)clip3i(0, PIXEL_MAX, v); }
static inline int16_t clip16(int v) { return (int16_t)clip3i(-32768, 32767, v); }

int xo_bit_depth(void) { return X265_DEPTH; }

/* ------------------------------------------------------------------ */
/* pixel compare family                                                */
/* ------------------------------------------------------------------ */

/* pixel.cpp:40-55 sad<lx,ly> */
int xo_sad(int w, int h, const xo_pixel* a, intptr_t sa, const xo_pixel* b, intptr_t sb)
{
    int sum = 0;
    for (int y = 0; y < h; y++, a += sa, b += sb)
        for (int x = 0; x < w; x++)
            sum += abs((int)a[x] - (int)b[x]);
    return sum;
}

/* pixel.cpp:74-95 sad_x3: fenc stride is the fixed FENC_STRIDE = 64 (common.h:71) */
void xo_sad_x3(int w, int h, const xo_pixel* fenc, const xo_pixel* r0, const xo_pixel* r1, const xo_pixel* r2, intptr_t rs, int32_t* res)
{
    res[0] = xo_sad(w, h, fenc, 64, r0, rs);
    res[1] = xo_sad(w, h, fenc, 64, r1, rs);
    res[2] = xo_sad(w, h, fenc, 64, r2, rs);
}

/* pixel.cpp:97-119 sad_x4 */
void xo_sad_x4(int w, int h, const xo_pixel* fenc, const xo_pixel* r0, const xo_pixel* r1, const xo_pixel* r2, const xo_pixel* r3, intptr_t rs, int32_t* res)
{
    res[0] = xo_sad(w, h, fenc, 64, r0, rs);
    res[1] = xo_sad(w, h, fenc, 64, r1, rs);
    res[2] = xo_sad(w, h, fenc, 64, r2, rs);
    res[3] = xo_sad(w, h, fenc, 64, r3, rs);
}

/* 4-point Hadamard butterfly, in place on d[0..3] with stride st */
static inline void had4(int* d, int st)
{
    int t0 = d[0] + d[st], t1 = d[0] - d[st], t2 = d[2 * st] + d[3 * st], t3 = d[2 * st] - d[3 * st];
    d[0] = t0 + t2; d[2 * st] = t0 - t2; d[st] = t1 + t3; d[3 * st] = t1 - t3;
}

/* sum of |4x4 Hadamard coefficients| of (a-b), NOT yet halved (pixel.cpp:210-238) */
static int had4x4_abs(const xo_pixel* a, intptr_t sa, const xo_pixel* b, intptr_t sb)
{
    int d[16];
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++)
            d[y * 4 + x] = (int)a[y * sa + x] - (int)b[y * sb + x];
    for (int y = 0; y < 4; y++) had4(d + 4 * y, 1);
    for (int x = 0; x < 4; x++) had4(d + x, 4);
    int s = 0;
    for (int i = 0; i < 16; i++) s += abs(d[i]);
    return s;
}

/* pixel.cpp:1148-1172: which decomposition each PU's satd slot uses.
 *   4x4 -> satd_4x4 (>>1 per 4x4); 8x4 -> satd_8x4 (>>1 per 8x4 pair);
 *   4x8, 4x16, 12x16 -> satd4<w,h> (sum of per-4x4 halved values);
 *   everything else -> satd8<w,h> (sum of per-8x4 halved values). */
int xo_satd(int w, int h, const xo_pixel* a, intptr_t sa, const xo_pixel* b, intptr_t sb)
{
    int total = 0;
    int use4 = (w == 4) || (w == 12);
    if (use4)
    {
        for (int y = 0; y < h; y += 4)
            for (int x = 0; x < w; x += 4)
                total += had4x4_abs(a + y * sa + x, sa, b + y * sb + x, sb) >> 1;
    }
    else
    {
        for (int y = 0; y < h; y += 4)
            for (int x = 0; x < w; x += 8)
                total += (had4x4_abs(a + y * sa + x, sa, b + y * sb + x, sb) +
                          had4x4_abs(a + y * sa + x + 4, sa, b + y * sb + x + 4, sb)) >> 1;
    }
    return total;
}

/* raw 8x8 Hadamard |coef| sum (pixel.cpp:291-326 _sa8d_8x8) */
static int had8x8_abs(const xo_pixel* a, intptr_t sa, const xo_pixel* b, intptr_t sb)
{
    int d[64];
    for (int y = 0; y < 8; y++)
        for (int x = 0; x < 8; x++)
            d[y * 8 + x] = (int)a[y * sa + x] - (int)b[y * sb + x];
    /* 8-point Hadamard along rows then columns (order of +/- outputs is irrelevant for sum of abs) */
    for (int pass = 0; pass < 2; pass++)
    {
        int st = pass ? 8 : 1, ln = pass ? 1 : 8;
        for (int l = 0; l < 8; l++)
        {
            int* p = d + l * ln;
            for (int half = 4; half >= 1; half >>= 1)
                for (int base = 0; base < 8; base += 2 * half)
                    for (int k = 0; k < half; k++)
                    {
                        int u = p[(base + k) * st], v = p[(base + k + half) * st];
                        p[(base + k) * st] = u + v;
                        p[(base + k + half) * st] = u - v;
                    }
        }
    }
    int s = 0;
    for (int i = 0; i < 64; i++) s += abs(d[i]);
    return s;
}

/* pixel.cpp:328-369,1180-1184: 4 -> satd_4x4; 8 -> (raw+2)>>2; 16 -> (sum of four raw +2)>>2;
 * 32/64 -> sum of sa8d_16x16 values */
int xo_sa8d(int size, const xo_pixel* a, intptr_t sa, const xo_pixel* b, intptr_t sb)
{
    if (size == 4) return had4x4_abs(a, sa, b, sb) >> 1;
    if (size == 8) return (had8x8_abs(a, sa, b, sb) + 2) >> 2;
    int cost = 0;
    for (int y = 0; y < size; y += 16)
        for (int x = 0; x < size; x += 16)
        {
            const xo_pixel* p = a + y * sa + x; const xo_pixel* q = b + y * sb + x;
            int s = had8x8_abs(p, sa, q, sb) + had8x8_abs(p + 8, sa, q + 8, sb) +
                    had8x8_abs(p + 8 * sa, sa, q + 8 * sb, sb) + had8x8_abs(p + 8 * sa + 8, sa, q + 8 * sb + 8, sb);
            cost += (s + 2) >> 2;
        }
    return cost;
}

/* pixel.cpp:167-186 sse<>; sse_t is u32 below 10-bit, u64 from 10-bit (common.h:145-149) */
static inline uint64_t sse_wrap(uint64_t v)
{
#if X265_DEPTH < 10
    return (uint32_t)v;
#else
    return v;
#endif
}
uint64_t xo_sse_pp(int w, int h, const xo_pixel* a, intptr_t sa, const xo_pixel* b, intptr_t sb)
{
    uint64_t sum = 0;
    for (int y = 0; y < h; y++, a += sa, b += sb)
        for (int x = 0; x < w; x++) { int t = (int)a[x] - (int)b[x]; sum += (uint64_t)(int64_t)(int32_t)((uint32_t)t * (uint32_t)t); }
    return sse_wrap(sum);
}
uint64_t xo_sse_ss(int w, int h, const int16_t* a, intptr_t sa, const int16_t* b, intptr_t sb)
{
    uint64_t sum = 0;
    for (int y = 0; y < h; y++, a += sa, b += sb)
        for (int x = 0; x < w; x++) { int t = (int)a[x] - (int)b[x]; sum += (uint64_t)(int64_t)(int32_t)((uint32_t)t * (uint32_t)t); }
    return sse_wrap(sum);
}
/* pixel.cpp:371-383 pixel_ssd_s_c */
uint64_t xo_ssd_s(int size, const int16_t* a, intptr_t sa)
{
    uint64_t sum = 0;
    for (int y = 0; y < size; y++, a += sa)
        for (int x = 0; x < size; x++) sum += (uint64_t)(int64_t)((int)a[x] * (int)a[x]);
    return sse_wrap(sum);
}

/* pixel.cpp:718-749 psyCost_pp: per 8x8, |(sa8d(src,0) - sum(src)>>2) - (sa8d(rec,0) - sum(rec)>>2)|; 4x4 uses satd */
int xo_psy_cost_pp(int size, const xo_pixel* src, intptr_t ss, const xo_pixel* rec, intptr_t rs)
{
    static const xo_pixel zero[8] = { 0 };
    if (size == 4)
    {
        int se = (had4x4_abs(src, ss, zero, 0) >> 1) - (xo_sad(4, 4, src, ss, zero, 0) >> 2);
        int re = (had4x4_abs(rec, rs, zero, 0) >> 1) - (xo_sad(4, 4, rec, rs, zero, 0) >> 2);
        return abs(se - re);
    }
    uint32_t tot = 0;
    for (int i = 0; i < size; i += 8)
        for (int j = 0; j < size; j += 8)
        {
            int se = ((had8x8_abs(src + i * ss + j, ss, zero, 0) + 2) >> 2) - (xo_sad(8, 8, src + i * ss + j, ss, zero, 0) >> 2);
            int re = ((had8x8_abs(rec + i * rs + j, rs, zero, 0) + 2) >> 2) - (xo_sad(8, 8, rec + i * rs + j, rs, zero, 0) >> 2);
            tot += (uint32_t)abs(se - re);
        }
    return (int)tot;
}

/* ------------------------------------------------------------------ */
/* block ops                                                           */
/* ------------------------------------------------------------------ */

/* pixel.cpp:463-475 getResidual: ONE stride shared by fenc, pred and residual */
void xo_calcresidual(int size, const xo_pixel* fenc, const xo_pixel* pred, int16_t* resi, intptr_t stride)
{
    for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++)
            resi[y * stride + x] = (int16_t)((int)fenc[y * stride + x] - (int)pred[y * stride + x]);
}
/* pixel.cpp:806-818 */
void xo_sub_ps(int w, int h, int16_t* dst, intptr_t ds, const xo_pixel* s0, const xo_pixel* s1, intptr_t ss0, intptr_t ss1)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            dst[y * ds + x] = (int16_t)((int)s0[y * ss0 + x] - (int)s1[y * ss1 + x]);
}
/* pixel.cpp:820-832 */
void xo_add_ps(int w, int h, xo_pixel* dst, intptr_t ds, const xo_pixel* s0, const int16_t* s1, intptr_t ss0, intptr_t ss1)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            dst[y * ds + x] = clip_pixel((int)s0[y * ss0 + x] + (int)s1[y * ss1 + x]);
}
/* pixel.cpp:751-804 blockcopy_{pp,ss,sp,ps}_c */
void xo_copy_pp(int w, int h, xo_pixel* dst, intptr_t ds, const xo_pixel* src, intptr_t ss)
{ for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) dst[y * ds + x] = src[y * ss + x]; }
void xo_copy_ss(int w, int h, int16_t* dst, intptr_t ds, const int16_t* src, intptr_t ss)
{ for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) dst[y * ds + x] = src[y * ss + x]; }
void xo_copy_sp(int w, int h, xo_pixel* dst, intptr_t ds, const int16_t* src, intptr_t ss)
{ for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) dst[y * ds + x] = (xo_pixel)src[y * ss + x]; }
void xo_copy_ps(int w, int h, int16_t* dst, intptr_t ds, const xo_pixel* src, intptr_t ss)
{ for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) dst[y * ds + x] = (int16_t)src[y * ss + x]; }
/* pixel.cpp:385-391 */
void xo_blockfill_s(int size, int16_t* dst, intptr_t ds, int16_t val)
{ for (int y = 0; y < size; y++) for (int x = 0; x < size; x++) dst[y * ds + x] = val; }
/* pixel.cpp:393-461 */
void xo_cpy2Dto1D_shl(int size, int16_t* dst, const int16_t* src, intptr_t ss, int shift)
{ for (int i = 0; i < size; i++) for (int j = 0; j < size; j++) dst[i * size + j] = (int16_t)((uint32_t)(int32_t)src[i * ss + j] << shift); }
void xo_cpy2Dto1D_shr(int size, int16_t* dst, const int16_t* src, intptr_t ss, int shift)
{ int16_t round = (int16_t)(1 << (shift - 1)); for (int i = 0; i < size; i++) for (int j = 0; j < size; j++) dst[i * size + j] = (int16_t)(((int)src[i * ss + j] + round) >> shift); }
void xo_cpy1Dto2D_shl(int size, int16_t* dst, const int16_t* src, intptr_t ds, int shift)
{ for (int i = 0; i < size; i++) for (int j = 0; j < size; j++) dst[i * ds + j] = (int16_t)((uint32_t)(int32_t)src[i * size + j] << shift); }
void xo_cpy1Dto2D_shr(int size, int16_t* dst, const int16_t* src, intptr_t ds, int shift)
{ int16_t round = (int16_t)(1 << (shift - 1)); for (int i = 0; i < size; i++) for (int j = 0; j < size; j++) dst[i * ds + j] = (int16_t)(((int)src[i * size + j] + round) >> shift); }
/* pixel.cpp:477-483: dst dense size x size */
void xo_transpose(int size, xo_pixel* dst, const xo_pixel* src, intptr_t ss)
{ for (int k = 0; k < size; k++) for (int l = 0; l < size; l++) dst[k * size + l] = src[l * ss + k]; }
/* pixel.cpp:834-854 addAvg */
void xo_addAvg(int w, int h, const int16_t* s0, const int16_t* s1, xo_pixel* dst, intptr_t ss0, intptr_t ss1, intptr_t ds)
{
    int shift = IF_INTERNAL_PREC + 1 - X265_DEPTH;
    int offset = (1 << (shift - 1)) + 2 * IF_INTERNAL_OFFS;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            dst[y * ds + x] = clip_pixel(((int)s0[y * ss0 + x] + (int)s1[y * ss1 + x] + offset) >> shift);
}
/* pixel.cpp:537-549 */
void xo_pixelavg_pp(int w, int h, xo_pixel* dst, intptr_t ds, const xo_pixel* s0, intptr_t ss0, const xo_pixel* s1, intptr_t ss1)
{ for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) dst[y * ds + x] = (xo_pixel)(((int)s0[y * ss0 + x] + (int)s1[y * ss1 + x] + 1) >> 1); }
/* pixel.cpp:485-508 */
void xo_weight_sp(const int16_t* src, xo_pixel* dst, intptr_t ss, intptr_t ds, int w, int h, int w0, int round, int shift, int offset)
{ for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) dst[y * ds + x] = clip_pixel(((w0 * ((int)src[y * ss + x] + IF_INTERNAL_OFFS) + round) >> shift) + offset); }
/* pixel.cpp:510-535 */
void xo_weight_pp(const xo_pixel* src, xo_pixel* dst, intptr_t stride, int w, int h, int w0, int round, int shift, int offset)
{
    int correction = IF_INTERNAL_PREC - X265_DEPTH;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
        {
            int16_t val = (int16_t)((int)src[y * stride + x] << correction);
            dst[y * stride + x] = clip_pixel(((w0 * (int)val + round) >> shift) + offset);
        }
}
/* pixel.cpp:551-577: src = two rows of 128 (at src and src+128), dst = two rows of 64 */
void xo_scale1D_128to64(xo_pixel* dst, const xo_pixel* src)
{
    for (int x = 0; x < 128; x += 2)
    {
        dst[x >> 1] = (xo_pixel)(((int)src[x] + (int)src[x + 1] + 1) >> 1);
        dst[64 + (x >> 1)] = (xo_pixel)(((int)src[128 + x] + (int)src[128 + x + 1] + 1) >> 1);
    }
}
/* pixel.cpp:579-594 */
void xo_scale2D_64to32(xo_pixel* dst, const xo_pixel* src, intptr_t stride)
{
    for (int y = 0; y < 64; y += 2)
        for (int x = 0; x < 64; x += 2)
            dst[(y / 2) * 32 + x / 2] = (xo_pixel)(((int)src[y * stride + x] + (int)src[y * stride + x + 1] +
                                                    (int)src[(y + 1) * stride + x] + (int)src[(y + 1) * stride + x + 1] + 2) >> 2);
}

/* ------------------------------------------------------------------ */
/* transforms                                                          */
/* ------------------------------------------------------------------ */

/* HEVC core transform coefficients by angle index a: round-ish 64*sqrt2*cos(pi*a/64) as
 * fixed by the standard (the numbers in constants.cpp:270-344 g_t4..g_t32). */
static const int8_t k_cos[33] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64,
                                  61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0 };
static int16_t g_mat[4][32 * 32];
static int g_mat_ready;
static void build_mats(void)
{
    if (g_mat_ready) return;
    for (int li = 0; li < 4; li++)
    {
        int n = 4 << li, step = 32 / n;
        for (int k = 0; k < n; k++)
            for (int j = 0; j < n; j++)
            {
                int th = ((k * step) * (2 * j + 1)) & 127, v;
                if (th <= 32) v = k_cos[th];
                else if (th <= 64) v = -k_cos[64 - th];
                else if (th <= 96) v = -k_cos[th - 64];
                else v = k_cos[128 - th];
                g_mat[li][k * n + j] = (int16_t)v;
            }
    }
    g_mat_ready = 1;
}
static int log2i(int n) { int l = 0; while ((1 << l) < n) l++; return l; }
const int16_t* xo_dct_matrix(int n) { build_mats(); return g_mat[log2i(n) - 2]; }

/* DST-VII 4x4 basis (the matrix fastForwardDst/inversedst factor, dct.cpp:43-81) */
static const int16_t k_dst4[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };

/* forward stage: dst[k*n + j] = (sum_m M[k][m]*src[j*n+m] + add) >> shift, int16 cast, no clip
 * (dct.cpp:83-240,418-440 partialButterfly*, exact as a matrix product) */
static void fwd_stage(const int16_t* M, int n, const int16_t* src, int16_t* dst, int shift)
{
    int add = 1 << (shift - 1);
    for (int j = 0; j < n; j++)
        for (int k = 0; k < n; k++)
        {
            int s = 0;
            for (int m = 0; m < n; m++) s += (int)M[k * n + m] * (int)src[j * n + m];
            dst[k * n + j] = (int16_t)((s + add) >> shift);
        }
}
/* inverse stage: dst[j*n + k] = clip16((sum_m M[m][k]*src[m*n+j] + add) >> shift)
 * (dct.cpp:242-416 partialButterflyInverse*) */
static void inv_stage(const int16_t* M, int n, const int16_t* src, int16_t* dst, int shift)
{
    int add = 1 << (shift - 1);
    for (int j = 0; j < n; j++)
        for (int k = 0; k < n; k++)
        {
            int s = 0;
            for (int m = 0; m < n; m++) s += (int)M[m * n + k] * (int)src[m * n + j];
            dst[j * n + k] = clip16((s + add) >> shift);
        }
}

/* dct.cpp:443-526: shift1 = log2N - 1 + (depth-8), shift2 = log2N + 6 */
void xo_dct(int n, const int16_t* src, int16_t* dst, intptr_t srcStride)
{
    int16_t blk[1024], tmp[1024];
    int lg = log2i(n);
    for (int i = 0; i < n; i++) memcpy(blk + i * n, src + i * srcStride, n * sizeof(int16_t));
    const int16_t* M = xo_dct_matrix(n);
    fwd_stage(M, n, blk, tmp, lg - 1 + X265_DEPTH - 8);
    fwd_stage(M, n, tmp, dst, lg + 6);
}
/* The data-parallel half of Search::estIntraPredQT's mode scan (encoder/search.cpp:1655-1745): sa8d of the source block
 * against the DC, planar and 33 angular predictions.  size 4..64; 64x64 is scaled to 32x32 like the reference
 * (scale2D_64to32 / scale1D_128to64, no filtered neighbours, costs << 2).  nbRef / nbFilt = intraNeighbourBuf[0] / [1] as
 * Predict::initAdiPattern leaves them (4*size+1 pixels).  costs[mode], mode 0 = planar, 1 = DC, 2..34 angular. */
void xo_intra_costs(int size, const xo_pixel* fenc, intptr_t stride, const xo_pixel* nbRef, const xo_pixel* nbFilt, int32_t* costs)
{
    static const unsigned char flags[35] = { 0x38, 0x00, 0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
                                             0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x38 };
    xo_pixel scaled[32 * 32], nb0[129], nb1[129], fencT[32 * 32], pred[32 * 32];
    static xo_pixel angs[33 * 32 * 32];
    const xo_pixel* ref = nbRef; const xo_pixel* flt = nbFilt;
    int n = size, shift = 0; (void)flags;
    if (size > 32)
    {   /* search.cpp:1670-1688 */
        xo_scale2D_64to32(scaled, fenc, stride);
        fenc = scaled; stride = 32;
        nb0[0] = nbRef[0];
        xo_scale1D_128to64(nb0 + 1, nbRef + 1);
        memcpy(nb1, nb0, sizeof(nb0));
        ref = nb0; flt = nb1; n = 32; shift = 2;
    }
    const int bLuma = n <= 16;
    xo_intra_pred(n, pred, n, ref, 1, bLuma);                                              /* DC (:1702) */
    costs[1] = xo_sa8d(n, fenc, stride, pred, n) << shift;
    xo_intra_pred(n, pred, n, (size & (8 | 16 | 32)) ? flt : ref, 0, 0);                   /* planar (:1708-1713) */
    costs[0] = xo_sa8d(n, fenc, stride, pred, n) << shift;
    xo_transpose(n, fencT, fenc, stride);                                                  /* :1721-1723 */
    xo_intra_allangs(n, angs, ref, flt, bLuma);
    for (int mode = 2; mode < 35; mode++)                                                  /* TRY_ANGLE, :1728-1735 */
        costs[mode] = (mode < 18 ? xo_sa8d(n, fencT, n, angs + (mode - 2) * n * n, n) : xo_sa8d(n, fenc, stride, angs + (mode - 2) * n * n, n)) << shift;
}
/* pixel.cpp:596-622 frame_init_lowres_core (p.frameInitLowres / frameInitLowerRes): half-resolution plane + its three
 * half-pel companions, each a rounded average of rounded vertical averages ("slower than naive bilinear, but matches asm") */
void xo_frame_init_lowres(const xo_pixel* src0, xo_pixel* dst0, xo_pixel* dsth, xo_pixel* dstv, xo_pixel* dstc,
                          intptr_t srcStride, intptr_t dstStride, int width, int height)
{
#define XO_F(a, b, c, d) ((((a + b + 1) >> 1) + ((c + d + 1) >> 1) + 1) >> 1)
    for (int y = 0; y < height; y++)
    {
        const xo_pixel* s0 = src0 + (intptr_t)(2 * y) * srcStride; const xo_pixel* s1 = s0 + srcStride; const xo_pixel* s2 = s1 + srcStride;
        for (int x = 0; x < width; x++)
        {
            dst0[y * dstStride + x] = (xo_pixel)XO_F(s0[2 * x], s1[2 * x], s0[2 * x + 1], s1[2 * x + 1]);
            dsth[y * dstStride + x] = (xo_pixel)XO_F(s0[2 * x + 1], s1[2 * x + 1], s0[2 * x + 2], s1[2 * x + 2]);
            dstv[y * dstStride + x] = (xo_pixel)XO_F(s1[2 * x], s2[2 * x], s1[2 * x + 1], s2[2 * x + 1]);
            dstc[y * dstStride + x] = (xo_pixel)XO_F(s1[2 * x + 1], s2[2 * x + 1], s1[2 * x + 2], s2[2 * x + 2]);
        }
    }
#undef XO_F
}
/* ipfilter.cpp:59-77 extendCURowColBorder (the p.extendRowBorder slot): replicate the first / last pixel of each row into the margins */
void xo_extend_row_border(xo_pixel* txt, intptr_t stride, int width, int height, int marginX)
{
    for (int y = 0; y < height; y++, txt += stride)
        for (int x = 0; x < marginX; x++) { txt[-marginX + x] = txt[0]; txt[width + x] = txt[width - 1]; }
}
/* pixel.cpp:1044-1058 extendPicBorder: rows first, then the (already widened) top / bottom rows into the vertical margins */
void xo_extend_pic_border(xo_pixel* pic, intptr_t stride, int width, int height, int marginX, int marginY)
{
    xo_extend_row_border(pic, stride, width, height, marginX);
    xo_pixel* top = pic - marginX;
    xo_pixel* bot = pic - marginX + (intptr_t)(height - 1) * stride;
    for (int y = 0; y < marginY; y++)
    {
        memcpy(top - (intptr_t)(y + 1) * stride, top, (size_t)stride * sizeof(xo_pixel));
        memcpy(bot + (intptr_t)(y + 1) * stride, bot, (size_t)stride * sizeof(xo_pixel));
    }
}
/* lowpassdct.cpp:34-116: 2x2 means (int16 arithmetic) -> half-size dct -> top-left embed, DC = scaled block sum.
 * n = size of the full block (8, 16, 32).  The 8x8 variant accumulates the block sum in int16 (:37), the others in int32. */
void xo_lowpass_dct(int n, const int16_t* src, int16_t* dst, intptr_t srcStride)
{
    const int h = n / 2;
    int16_t avg[256], coef[256];
    int32_t total32 = 0; int16_t total16 = 0;
    for (int i = 0; i < h; i++)
        for (int j = 0; j < h; j++)
        {
            int16_t sum = (int16_t)(src[2 * i * srcStride + 2 * j] + src[2 * i * srcStride + 2 * j + 1]
                                    + src[(2 * i + 1) * srcStride + 2 * j] + src[(2 * i + 1) * srcStride + 2 * j + 1]);
            avg[i * h + j] = (int16_t)(sum >> 2);
            total32 += sum; total16 = (int16_t)(total16 + sum);
        }
    xo_dct(h, avg, coef, h);
    memset(dst, 0, (size_t)n * n * sizeof(int16_t));
    for (int i = 0; i < h; i++) memcpy(dst + i * n, coef + i * h, h * sizeof(int16_t));
    if (n == 8)
#if X265_DEPTH == 8
        dst[0] = (int16_t)(total16 << 1);
#else
        dst[0] = (int16_t)(total16 >> (X265_DEPTH - 9));
#endif
    else
        dst[0] = (int16_t)(total32 >> ((n == 16 ? 1 : 3) + (X265_DEPTH - 8)));
}
/* pixel.cpp:121-165 ads_x1 / x2 / x4 (SEA pre-filter): which of them a PU uses is the slot map pixel.cpp:1122-1146.
 * parts = 1, 2 or 4; lx = PU width (the x4 variant reads sums[lx >> 1]). */
int xo_ads(int parts, int lx, const int* encDC, const uint32_t* sums, int delta, const uint16_t* costMvX, int16_t* mvs, int width, int thresh)
{
    int nmv = 0;
    for (int i = 0; i < width; i++, sums++)
    {
        long a = labs((long)encDC[0] - (long)sums[0]);
        if (parts == 2) a += labs((long)encDC[1] - (long)sums[delta]);
        if (parts == 4)
            a += labs((long)encDC[1] - (long)sums[lx >> 1]) + labs((long)encDC[2] - (long)sums[delta]) + labs((long)encDC[3] - (long)sums[delta + (lx >> 1)]);
        int ads = (int)(a + costMvX[i]);
        if (ads < thresh) mvs[nmv++] = (int16_t)i;
    }
    return nmv;
}
/* dct.cpp:528-611: shift1 = 7, shift2 = 12 - (depth-8), clip to int16 after each stage */
void xo_idct(int n, const int16_t* src, int16_t* dst, intptr_t dstStride)
{
    int16_t tmp[1024], blk[1024];
    const int16_t* M = xo_dct_matrix(n);
    inv_stage(M, n, src, tmp, 7);
    inv_stage(M, n, tmp, blk, 12 - (X265_DEPTH - 8));
    for (int i = 0; i < n; i++) memcpy(dst + i * dstStride, blk + i * n, n * sizeof(int16_t));
}
/* dct.cpp:443-459 dst4_c */
void xo_dst4(const int16_t* src, int16_t* dst, intptr_t srcStride)
{
    int16_t blk[16], tmp[16];
    for (int i = 0; i < 4; i++) memcpy(blk + i * 4, src + i * srcStride, 4 * sizeof(int16_t));
    fwd_stage(k_dst4, 4, blk, tmp, 1 + X265_DEPTH - 8);
    fwd_stage(k_dst4, 4, tmp, dst, 8);
}
/* dct.cpp:528-544 idst4_c */
void xo_idst4(const int16_t* src, int16_t* dst, intptr_t dstStride)
{
    int16_t tmp[16], blk[16];
    inv_stage(k_dst4, 4, src, tmp, 7);
    inv_stage(k_dst4, 4, tmp, blk, 12 - (X265_DEPTH - 8));
    for (int i = 0; i < 4; i++) memcpy(dst + i * dstStride, blk + i * 4, 4 * sizeof(int16_t));
}

/* dct.cpp:666-688 quant_c.  int arithmetic wraps like the reference's int32 (done in uint32 to stay defined). */
uint32_t xo_quant(const int16_t* coef, const int32_t* quantCoeff, int32_t* deltaU, int16_t* qCoef, int qBits, int add, int numCoeff)
{
    int qBits8 = qBits - 8;
    uint32_t numSig = 0;
    for (int i = 0; i < numCoeff; i++)
    {
        int level = coef[i];
        int sign = level < 0 ? -1 : 1;
        int32_t tmplevel = (int32_t)((uint32_t)abs(level) * (uint32_t)quantCoeff[i]);
        level = (int32_t)((uint32_t)tmplevel + (uint32_t)add) >> qBits;
        deltaU[i] = (int32_t)((uint32_t)tmplevel - ((uint32_t)level << qBits)) >> qBits8;
        if (level) ++numSig;
        level *= sign;
        qCoef[i] = clip16(level);
    }
    return numSig;
}
/* dct.cpp:690-715 nquant_c: stores ABSOLUTE levels */
uint32_t xo_nquant(const int16_t* coef, const int32_t* quantCoeff, int16_t* qCoef, int qBits, int add, int numCoeff)
{
    uint32_t numSig = 0;
    for (int i = 0; i < numCoeff; i++)
    {
        int level = coef[i];
        int sign = level < 0 ? -1 : 1;
        int32_t tmplevel = (int32_t)((uint32_t)abs(level) * (uint32_t)quantCoeff[i]);
        level = (int32_t)((uint32_t)tmplevel + (uint32_t)add) >> qBits;
        if (level) ++numSig;
        level *= sign;
        qCoef[i] = (int16_t)abs(clip3i(-32768, 32767, level));
    }
    return numSig;
}
/* dct.cpp:614-636 */
void xo_dequant_normal(const int16_t* q, int16_t* coef, int num, int scale, int shift)
{
    int add = 1 << (shift - 1);
    for (int n = 0; n < num; n++)
        coef[n] = clip16((int32_t)((uint32_t)((int)q[n] * scale) + (uint32_t)add) >> shift);
}
/* dct.cpp:638-664 */
void xo_dequant_scaling(const int16_t* q, const int32_t* deq, int16_t* coef, int num, int per, int shift)
{
    shift += 4;
    if (shift > per)
    {
        int add = 1 << (shift - per - 1);
        for (int n = 0; n < num; n++)
            coef[n] = clip16((int32_t)((uint32_t)((int)q[n] * deq[n]) + (uint32_t)add) >> (shift - per));
    }
    else
    {
        for (int n = 0; n < num; n++)
        {
            int c = clip3i(-32768, 32767, (int)q[n] * deq[n]);
            coef[n] = clip16((int32_t)((uint32_t)c * (1u << (per - shift))));
        }
    }
}
/* dct.cpp:716-728 */
int xo_count_nonzero(int n, const int16_t* q)
{ int c = 0; for (int i = 0; i < n * n; i++) c += q[i] != 0; return c; }
/* dct.cpp:730-744 */
uint32_t xo_copy_count(int n, int16_t* coef, const int16_t* resi, intptr_t rs)
{
    uint32_t c = 0;
    for (int k = 0; k < n; k++) for (int j = 0; j < n; j++) { coef[k * n + j] = resi[k * rs + j]; c += resi[k * rs + j] != 0; }
    return c;
}
/* dct.cpp:746-757 */
void xo_denoise_dct(int16_t* coef, uint32_t* resSum, const uint16_t* offset, int num)
{
    for (int i = 0; i < num; i++)
    {
        int level = coef[i];
        int sign = level >> 31;
        level = (level + sign) ^ sign;
        resSum[i] += (uint32_t)level;
        level -= offset[i];
        coef[i] = (int16_t)(level < 0 ? 0 : (level ^ sign) - sign);
    }
}

/* ------------------------------------------------------------------ */
/* interpolation                                                       */
/* ------------------------------------------------------------------ */
/* constants.cpp:250-268 (the HEVC interpolation taps) */
static const int16_t k_luma[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
                                      { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
static const int16_t k_chroma[8][4] = { { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
                                        { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };
static const int16_t* taps_of(int taps, int idx) { return taps == 8 ? k_luma[idx] : k_chroma[idx]; }

/* generic FIR; `step` is the element distance between taps (1 = horizontal, stride = vertical) */
#define FIR(src, pos, step, c, taps, sum) do { sum = 0; for (int t_ = 0; t_ < (taps); t_++) sum += (int)(src)[(pos) + t_ * (step)] * (int)(c)[t_]; } while (0)

/* ipfilter.cpp:79-118 */
void xo_interp_hpp(int taps, int w, int h, const xo_pixel* src, intptr_t ss, xo_pixel* dst, intptr_t ds, int coeffIdx)
{
    const int16_t* c = taps_of(taps, coeffIdx);
    src -= taps / 2 - 1;
    for (int y = 0; y < h; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
        {
            int sum; FIR(src, x, 1, c, taps, sum);
            int16_t v = (int16_t)((sum + 32) >> 6);
            dst[x] = clip_pixel(v);
        }
}
/* ipfilter.cpp:120-162 */
void xo_interp_hps(int taps, int w, int h, const xo_pixel* src, intptr_t ss, int16_t* dst, intptr_t ds, int coeffIdx, int isRowExt)
{
    const int16_t* c = taps_of(taps, coeffIdx);
    int headRoom = IF_INTERNAL_PREC - X265_DEPTH, shift = IF_FILTER_PREC - headRoom;
    int offset = (int)((unsigned)-IF_INTERNAL_OFFS << shift);
    int rows = h;
    src -= taps / 2 - 1;
    if (isRowExt) { src -= (taps / 2 - 1) * ss; rows += taps - 1; }
    for (int y = 0; y < rows; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
        {
            int sum; FIR(src, x, 1, c, taps, sum);
            dst[x] = (int16_t)((sum + offset) >> shift);
        }
}
/* ipfilter.cpp:164-203 */
void xo_interp_vpp(int taps, int w, int h, const xo_pixel* src, intptr_t ss, xo_pixel* dst, intptr_t ds, int coeffIdx)
{
    const int16_t* c = taps_of(taps, coeffIdx);
    src -= (taps / 2 - 1) * ss;
    for (int y = 0; y < h; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
        {
            int sum; FIR(src, x, ss, c, taps, sum);
            int16_t v = (int16_t)((sum + 32) >> 6);
            dst[x] = clip_pixel(v);
        }
}
/* ipfilter.cpp:205-239 */
void xo_interp_vps(int taps, int w, int h, const xo_pixel* src, intptr_t ss, int16_t* dst, intptr_t ds, int coeffIdx)
{
    const int16_t* c = taps_of(taps, coeffIdx);
    int headRoom = IF_INTERNAL_PREC - X265_DEPTH, shift = IF_FILTER_PREC - headRoom;
    int offset = (int)((unsigned)-IF_INTERNAL_OFFS << shift);
    src -= (taps / 2 - 1) * ss;
    for (int y = 0; y < h; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
        {
            int sum; FIR(src, x, ss, c, taps, sum);
            dst[x] = (int16_t)((sum + offset) >> shift);
        }
}
/* ipfilter.cpp:241-282 (and filterVertical_sp_c :319-360) */
void xo_interp_vsp(int taps, int w, int h, const int16_t* src, intptr_t ss, xo_pixel* dst, intptr_t ds, int coeffIdx)
{
    const int16_t* c = taps_of(taps, coeffIdx);
    int headRoom = IF_INTERNAL_PREC - X265_DEPTH, shift = IF_FILTER_PREC + headRoom;
    int offset = (1 << (shift - 1)) + (IF_INTERNAL_OFFS << IF_FILTER_PREC);
    src -= (taps / 2 - 1) * ss;
    for (int y = 0; y < h; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
        {
            int sum; FIR(src, x, ss, c, taps, sum);
            int16_t v = (int16_t)((sum + offset) >> shift);
            dst[x] = clip_pixel(v);
        }
}
/* ipfilter.cpp:284-317 */
void xo_interp_vss(int taps, int w, int h, const int16_t* src, intptr_t ss, int16_t* dst, intptr_t ds, int coeffIdx)
{
    const int16_t* c = taps_of(taps, coeffIdx);
    src -= (taps / 2 - 1) * ss;
    for (int y = 0; y < h; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
        {
            int sum; FIR(src, x, ss, c, taps, sum);
            dst[x] = (int16_t)(sum >> IF_FILTER_PREC);
        }
}
/* ipfilter.cpp:362-369: hps with row extension into a w-strided int16 buffer, then vertical sp */
void xo_interp_hvpp(int taps, int w, int h, const xo_pixel* src, intptr_t ss, xo_pixel* dst, intptr_t ds, int idxX, int idxY)
{
    int16_t immed[64 * (64 + 7)];
    xo_interp_hps(taps, w, h, src, ss, immed, w, idxX, 1);
    xo_interp_vsp(taps, w, h, immed + (taps / 2 - 1) * w, w, dst, ds, idxY);
}
/* ipfilter.cpp:40-57 filterPixelToShort_c */
void xo_p2s(int w, int h, const xo_pixel* src, intptr_t ss, int16_t* dst, intptr_t ds)
{
    int shift = IF_INTERNAL_PREC - X265_DEPTH;
    for (int y = 0; y < h; y++, src += ss, dst += ds)
        for (int x = 0; x < w; x++)
        {
            int16_t v = (int16_t)((int)src[x] << shift);
            dst[x] = (int16_t)(v - (int16_t)IF_INTERNAL_OFFS);
        }
}

/* ------------------------------------------------------------------ */
/* intra                                                               */
/* ------------------------------------------------------------------ */
/* constants.cpp:561 g_intraFilterFlags */
static const uint8_t k_intraFilterFlags[35] = {
    0x38, 0x00, 0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x38 };

/* intrapred.cpp:31-51. Layout: [0]=top-left, [1..2N]=above+above-right, [2N+1..4N]=left+below-left */
void xo_intra_filter(int size, const xo_pixel* s, xo_pixel* f)
{
    int n2 = size * 2;
    int tl = s[0];
    for (int i = 1; i < n2; i++) f[i] = (xo_pixel)((2 * s[i] + s[i - 1] + s[i + 1] + 2) >> 2);
    f[n2] = s[n2];
    f[0] = (xo_pixel)((2 * tl + s[1] + s[n2 + 1] + 2) >> 2);
    f[n2 + 1] = (xo_pixel)((2 * s[n2 + 1] + tl + s[n2 + 2] + 2) >> 2);
    for (int i = n2 + 2; i < 2 * n2; i++) f[i] = (xo_pixel)((2 * s[i] + s[i - 1] + s[i + 1] + 2) >> 2);
    f[2 * n2] = s[2 * n2];
}

/* angular prediction into a DENSE size x size buffer `out`, WITHOUT the final transpose of
 * horizontal modes (intrapred.cpp:102-189) */
static void ang_core(int size, xo_pixel* out, const xo_pixel* srcPix0, int dirMode, int bFilter)
{
    static const int8_t angleTable[17] = { -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
    static const int16_t invAngleTable[8] = { 4096, 1638, 910, 630, 482, 390, 315, 256 };
    int n2 = size * 2, horMode = dirMode < 18;
    xo_pixel nb[129];
    const xo_pixel* sp = srcPix0;
    if (horMode)
    {
        nb[0] = srcPix0[0];
        for (int i = 0; i < n2; i++) { nb[1 + i] = srcPix0[n2 + 1 + i]; nb[n2 + 1 + i] = srcPix0[1 + i]; }
        sp = nb;
    }
    int angleOffset = horMode ? 10 - dirMode : dirMode - 26;
    int angle = angleTable[8 + angleOffset];
    if (!angle)
    {
        for (int y = 0; y < size; y++) for (int x = 0; x < size; x++) out[y * size + x] = sp[1 + x];
        if (bFilter)
        {
            int tl = sp[0], top = sp[1];
            for (int y = 0; y < size; y++)
                out[y * size] = clip_pixel((int16_t)(top + (((int)sp[n2 + 1 + y] - tl) >> 1)));
        }
        return;
    }
    xo_pixel refBuf[64 + 2];
    const xo_pixel* ref;
    if (angle < 0)
    {
        int nbProjected = -((size * angle) >> 5) - 1;
        xo_pixel* rp = refBuf + nbProjected + 1;
        int invAngle = invAngleTable[-angleOffset - 1], invAngleSum = 128;
        for (int i = 0; i < nbProjected; i++) { invAngleSum += invAngle; rp[-2 - i] = sp[n2 + (invAngleSum >> 8)]; }
        for (int i = 0; i < size + 1; i++) rp[-1 + i] = sp[i];
        ref = rp;
    }
    else
        ref = sp + 1;
    int angleSum = 0;
    for (int y = 0; y < size; y++)
    {
        angleSum += angle;
        int off = angleSum >> 5, frac = angleSum & 31;
        for (int x = 0; x < size; x++)
            out[y * size + x] = frac ? (xo_pixel)(((32 - frac) * ref[off + x] + frac * ref[off + x + 1] + 16) >> 5) : ref[off + x];
    }
}

/* intrapred.cpp:53-100 (DC=1, planar=0) and :102-204 (angular 2..34) */
void xo_intra_pred(int size, xo_pixel* dst, intptr_t ds, const xo_pixel* srcPix, int dirMode, int bFilter)
{
    const xo_pixel* above = srcPix + 1;
    const xo_pixel* left = srcPix + 2 * size + 1;
    if (dirMode == 0)
    {
        int lg = log2i(size), tr = above[size], bl = left[size];
        for (int y = 0; y < size; y++)
            for (int x = 0; x < size; x++)
                dst[y * ds + x] = (xo_pixel)(((size - 1 - x) * left[y] + (size - 1 - y) * above[x] + (x + 1) * tr + (y + 1) * bl + size) >> (lg + 1));
        return;
    }
    if (dirMode == 1)
    {
        int dc = size;
        for (int i = 0; i < size; i++) dc += above[i] + left[i];
        dc /= 2 * size;
        for (int y = 0; y < size; y++) for (int x = 0; x < size; x++) dst[y * ds + x] = (xo_pixel)dc;
        if (bFilter)
        {
            dst[0] = (xo_pixel)((above[0] + left[0] + 2 * dc + 2) >> 2);
            for (int x = 1; x < size; x++) dst[x] = (xo_pixel)((above[x] + 3 * dc + 2) >> 2);
            for (int y = 1; y < size; y++) dst[y * ds] = (xo_pixel)((left[y] + 3 * dc + 2) >> 2);
        }
        return;
    }
    xo_pixel tmp[32 * 32];
    ang_core(size, tmp, srcPix, dirMode, bFilter);
    int hor = dirMode < 18;
    for (int y = 0; y < size; y++)
        for (int x = 0; x < size; x++)
            dst[y * ds + x] = hor ? tmp[x * size + y] : tmp[y * size + x];
}

/* intrapred.cpp:206-234: 33 dense blocks, modes < 18 left un-transposed (i.e. predicted on flipped
 * neighbours); source = filtPix when g_intraFilterFlags[mode] & size */
void xo_intra_allangs(int size, xo_pixel* dst, const xo_pixel* refPix, const xo_pixel* filtPix, int bLuma)
{
    for (int mode = 2; mode <= 34; mode++)
    {
        const xo_pixel* sp = (k_intraFilterFlags[mode] & size) ? filtPix : refPix;
        ang_core(size, dst + (mode - 2) * size * size, sp, mode, bLuma);
    }
}

/* ---- SAO statistics (encoder/sao.cpp:1774-1937): per offset class, sum of (source - reconstruction) and pixel count ----
 * diff has the fixed stride MAX_CU_SIZE = 64; stats / count accumulate (+=).  type 0..3 = edge classes EO_0..EO_3 (neighbour
 * pairs left/right, up/down, up-left/down-right, up-right/down-left), 4 = band offset.  The sign buffers carry the "upper
 * neighbour" signs of the first row in and those of the row after the last one out, exactly like the reference's in-place updates. */
static inline int sgn(int v) { return (v > 0) - (v < 0); }
static const int k_eoTable[5] = { 1, 2, 0, 3, 4 };                                /* SAO::s_eoTable, sao.cpp:59-66 */
void xo_sao_stats(int type, const int16_t* diff, const xo_pixel* rec, intptr_t stride, int8_t* upBuff1, int8_t* upBufft,
                  int endX, int endY, int32_t* stats, int32_t* count)
{
    if (type == 4)
    {   /* saoCuStatsBO_c */
        const int boShift = X265_DEPTH - 5;
        for (int y = 0; y < endY; y++, diff += 64, rec += stride)
            for (int x = 0; x < endX; x++) { const int c = rec[x] >> boShift; stats[c] += diff[x]; count[c]++; }
        return;
    }
    int32_t ts[5] = { 0, 0, 0, 0, 0 }, tc[5] = { 0, 0, 0, 0, 0 };
    for (int y = 0; y < endY; y++)
    {
        if (type == 0)
        {   /* saoCuStatsE0_c */
            int signLeft = sgn(rec[0] - rec[-1]);
            for (int x = 0; x < endX; x++)
            {
                const int signRight = sgn(rec[x] - rec[x + 1]);
                const int e = signRight + signLeft + 2;
                signLeft = -signRight;
                ts[e] += diff[x]; tc[e]++;
            }
        }
        else if (type == 1)
        {   /* saoCuStatsE1_c */
            for (int x = 0; x < endX; x++)
            {
                const int signDown = sgn(rec[x] - rec[x + stride]);
                const int e = signDown + upBuff1[x] + 2;
                upBuff1[x] = (int8_t)(-signDown);
                ts[e] += diff[x]; tc[e]++;
            }
        }
        else if (type == 2)
        {   /* saoCuStatsE2_c: the two sign buffers swap roles every row */
            upBufft[0] = (int8_t)sgn(rec[stride] - rec[-1]);
            for (int x = 0; x < endX; x++)
            {
                const int signDown = sgn(rec[x] - rec[x + stride + 1]);
                const int e = signDown + upBuff1[x] + 2;
                upBufft[x + 1] = (int8_t)(-signDown);
                ts[e] += diff[x]; tc[e]++;
            }
            int8_t* t = upBuff1; upBuff1 = upBufft; upBufft = t;
        }
        else
        {   /* saoCuStatsE3_c */
            for (int x = 0; x < endX; x++)
            {
                const int signDown = sgn(rec[x] - rec[x + stride - 1]);
                const int e = signDown + upBuff1[x] + 2;
                upBuff1[x - 1] = (int8_t)(-signDown);
                ts[e] += diff[x]; tc[e]++;
            }
            upBuff1[endX - 1] = (int8_t)sgn(rec[endX - 1 + stride] - rec[endX]);
        }
        rec += stride; diff += 64;
    }
    for (int x = 0; x < 5; x++) { stats[k_eoTable[x]] += ts[x]; count[k_eoTable[x]] += tc[x]; }
}

/* SAO::calcSaoStatsCTU for the luma plane of every CTU of a picture (encoder/sao.cpp:729-905; one slice, bLimitSAO off): which pixels
 * of a CTU each offset class counts (the rows / columns the deblocking of the neighbours has not finalised are skipped: skipB / skipR),
 * composed from the primitives above.  out: per CTU [2][5][32] = offsetOrg then count, types in the order SAO_EO_0..3, SAO_BO. */
void xo_sao_stats_frame(const xo_pixel* fenc, const xo_pixel* recon, intptr_t stride, int picWidth, int picHeight, int ctuSize, int nonDeblocked, int planeOffset, int32_t* out)
{
    xo_sao_stats_frame_slices(fenc, recon, stride, picWidth, picHeight, ctuSize, nonDeblocked, planeOffset, out, NULL);
}
/* --slices: sliceFirstRow[r] != 0 where CTU row r begins a slice (CUData::m_bFirstRowInSlice of its CTUs; m_bLastRowInSlice of the row before it): sao.cpp:744-746, 763-766 */
void xo_sao_stats_frame_slices(const xo_pixel* fenc, const xo_pixel* recon, intptr_t stride, int picWidth, int picHeight, int ctuSize, int nonDeblocked, int planeOffset, int32_t* out,
                               const uint8_t* sliceFirstRow)
{
    xo_sao_stats_rows(fenc, recon, stride, picWidth, picHeight, ctuSize, nonDeblocked, planeOffset, out, sliceFirstRow, 0, (picHeight + ctuSize - 1) / ctuSize);
}
void xo_sao_stats_rows(const xo_pixel* fenc, const xo_pixel* recon, intptr_t stride, int picWidth, int picHeight, int ctuSize, int nonDeblocked, int planeOffset, int32_t* out,
                       const uint8_t* sliceFirstRow, int ctuRow0, int ctuRow1)
{
    xo_sao_stats_rows_wh(fenc, recon, stride, picWidth, picHeight, ctuSize, ctuSize, nonDeblocked, planeOffset, out, sliceFirstRow, ctuRow0, ctuRow1);
}
/* The CTUs of the rows [ctuRow0, ctuRow1) only (their entries of `out`; the others are not touched).  The rows below need not be deblocked yet: a CTU's statistics leave out the
 * lines the deblocking of the row below still changes (skipB) and read one line beyond them, which it does not change -- the order the reference works in (rdoSaoUnitCu of row r
 * runs before row r + 1 is deblocked, framefilter.cpp:490-500). */
/* ... with the CTU's width and height in the plane given apart: the chroma planes of 4:2:2 have CTUs half as wide as high (ctuWidth >>= m_hChromaShift, ctuHeight >>=
 * m_vChromaShift, sao.cpp:748-756) */
void xo_sao_stats_rows_wh(const xo_pixel* fenc, const xo_pixel* recon, intptr_t stride, int picWidth, int picHeight, int ctuW, int ctuH, int nonDeblocked, int planeOffset, int32_t* out,
                          const uint8_t* sliceFirstRow, int ctuRow0, int ctuRow1)
{   /* chroma planes: pass the PLANE's width / height / CTU size (already shifted, :748-756) and planeOffset = 2 (:773) */
    const int po = planeOffset;
    const int nx = (picWidth + ctuW - 1) / ctuW, ny = (picHeight + ctuH - 1) / ctuH;
    if (ctuRow1 > ny) ctuRow1 = ny;
    for (int addr = ctuRow0 * nx; addr < ctuRow1 * nx; addr++)
    {
        const int lpelx = (addr % nx) * ctuW, tpely = (addr / nx) * ctuH;
        const int row = addr / nx;
        const int firstRow = row == 0 || (sliceFirstRow && sliceFirstRow[row]), lastRow = row == ny - 1 || (sliceFirstRow && sliceFirstRow[row + 1]);
        const int bAboveUnavail = (!tpely) | firstRow;
        const int rpelx = lpelx + ctuW < picWidth ? lpelx + ctuW : picWidth, bpely = tpely + ctuH < picHeight ? tpely + ctuH : picHeight;
        const int ctuWidth = rpelx - lpelx, ctuHeight = bpely - tpely;
        const int picH = lastRow ? bpely : picHeight;
        const xo_pixel* fenc0 = fenc + tpely * stride + lpelx; const xo_pixel* rec0 = recon + tpely * stride + lpelx;
        int32_t* stats = out + (size_t)addr * 320; int32_t* count = stats + 160;
        memset(stats, 0, 320 * sizeof(int32_t));
        int16_t diff[64 * 64];
        for (int y = 0; y < ctuHeight; y++) for (int x = 0; x < ctuWidth; x++) diff[y * 64 + x] = (int16_t)(fenc0[y * stride + x] - rec0[y * stride + x]);
        int8_t buf[2 * (64 + 32)], *upBuff1 = buf + 16, *upBufft = upBuff1 + (64 + 32);
        int skipB = 4, skipR = 5, startX, startY, endX, endY;
        const xo_pixel* rec;
        /* SAO_BO (:800-812) */
        if (nonDeblocked) { skipB = 3; skipR = 4; }
        endX = rpelx == picWidth ? ctuWidth : ctuWidth - skipR + po;
        endY = bpely == picH ? ctuHeight : ctuHeight - skipB + po;
        xo_sao_stats(4, diff, rec0, stride, NULL, NULL, endX, endY, stats + 4 * 32, count + 4 * 32);
        /* SAO_EO_0 (:816-828) */
        if (nonDeblocked) { skipB = 3; skipR = 5; }
        startX = !lpelx;
        endX = rpelx == picWidth ? ctuWidth - 1 : ctuWidth - skipR + po;
        xo_sao_stats(0, diff + startX, rec0 + startX, stride, NULL, NULL, endX - startX, ctuHeight - skipB + po, stats, count);
        /* SAO_EO_1 (:831-851) */
        if (nonDeblocked) { skipB = 4; skipR = 4; }
        rec = rec0; startY = bAboveUnavail;
        endX = rpelx == picWidth ? ctuWidth : ctuWidth - skipR + po;
        endY = bpely == picH ? ctuHeight - 1 : ctuHeight - skipB + po;
        if (startY) rec += stride;
        for (int i = 0; i < ctuWidth; i++) upBuff1[i] = (int8_t)sgn(rec[i] - rec[i - stride]);
        xo_sao_stats(1, diff + startY * 64, rec0 + startY * stride, stride, upBuff1, NULL, endX, endY - startY, stats + 32, count + 32);
        /* SAO_EO_2 (:856-878) */
        if (nonDeblocked) { skipB = 4; skipR = 5; }
        rec = rec0; startX = !lpelx;
        endX = rpelx == picWidth ? ctuWidth - 1 : ctuWidth - skipR + po;
        startY = bAboveUnavail;
        endY = bpely == picH ? ctuHeight - 1 : ctuHeight - skipB + po;
        if (startY) rec += stride;
        for (int i = 0; i < endX - startX; i++) upBuff1[i] = (int8_t)sgn(rec[startX + i] - rec[startX + i - stride - 1]);
        xo_sao_stats(2, diff + startX + startY * 64, rec0 + startX + startY * stride, stride, upBuff1, upBufft, endX - startX, endY - startY, stats + 64, count + 64);
        /* SAO_EO_3 (:880-901) */
        rec = rec0; if (startY) rec += stride;
        for (int i = 0; i < endX - startX + 1; i++) upBuff1[i] = (int8_t)sgn(rec[startX - 1 + i] - rec[startX + i - stride]);
        xo_sao_stats(3, diff + startX + startY * 64, rec0 + startX + startY * stride, stride, upBuff1 + 1, NULL, endX - startX, endY - startY, stats + 96, count + 96);
    }
}

/* SAO::calcSaoStatsCu_BeforeDblk (encoder/sao.cpp:908-1207) for every CTU of one plane: the statistics of the bottom / right border the deblocked statistics
 * leave out (--sao-non-deblock), taken on the picture BEFORE deblocking.  Restated per pixel: every pixel is classified against its real neighbours (what the
 * reference's sign buffers hold, including the entries its comments say it recomputes), and a class counts it when it lies in the class's window
 * [firstX, endX) x [firstY, endY) and right of startX or below startY.  Same output layout and plane conventions as xo_sao_stats_frame. */
void xo_sao_stats_frame_predeblock(const xo_pixel* fenc, const xo_pixel* recon, intptr_t stride, int picWidth, int picHeight, int ctuSize, int planeOffset, int32_t* out)
{
    static const int eoTable[5] = { 1, 2, 0, 3, 4 };
    const int po = planeOffset, boShift = X265_DEPTH - 5;
    const int nx = (picWidth + ctuSize - 1) / ctuSize, ny = (picHeight + ctuSize - 1) / ctuSize;
    for (int addr = 0; addr < nx * ny; addr++)
    {
        const int lpelx = (addr % nx) * ctuSize, tpely = (addr / nx) * ctuSize;
        const int firstRow = addr < nx, lastRow = addr >= nx * ny - nx;
        const int above = (!tpely) | firstRow;
        const int rpelx = lpelx + ctuSize < picWidth ? lpelx + ctuSize : picWidth, bpely = tpely + ctuSize < picHeight ? tpely + ctuSize : picHeight;
        const int cw = rpelx - lpelx, ch = bpely - tpely;
        const int picH = lastRow ? bpely : picHeight;
        const int atRight = rpelx == picWidth, atBottom = bpely == picH, firstX = !lpelx;
        int32_t* stats = out + (size_t)addr * 320; int32_t* count = stats + 160;
        memset(stats, 0, 320 * sizeof(int32_t));
        /* per type: startX, startY (:985-986, 1008-1009, 1040-1041, 1084-1085, 1138-1139) */
        const int boX = atRight ? cw : cw - (4 - po), boY = atBottom ? ch : ch - (3 - po);
        const int e0X = atRight ? cw - 1 : cw - (5 - po), e0Y = atBottom ? ch : ch - (3 - po);
        const int e1X = atRight ? cw : cw - (4 - po), e1Y = atBottom ? ch - 1 : ch - (4 - po);
        const int e2X = atRight ? cw - 1 : cw - (5 - po), e2Y = atBottom ? ch - 1 : ch - (4 - po);
        for (int y = 0; y < ch; y++)
            for (int x = 0; x < cw; x++)
            {
                const xo_pixel* r = recon + (tpely + y) * stride + lpelx + x;
                const int c = r[0], d = (int)fenc[(tpely + y) * stride + lpelx + x] - c;
                if (x >= boX || y >= boY) { stats[4 * 32 + (c >> boShift)] += d; count[4 * 32 + (c >> boShift)]++; }
                if (x < cw - 1 && x >= (y < e0Y ? e0X : firstX))
                { const int e = eoTable[sgn(c - r[-1]) + sgn(c - r[1]) + 2]; stats[e] += d; count[e]++; }
                if (y >= above && y < ch - 1 && (x >= e1X || y >= e1Y))
                { const int e = eoTable[sgn(c - r[-stride]) + sgn(c - r[stride]) + 2]; stats[32 + e] += d; count[32 + e]++; }
                if (y >= above && y < ch - 1 && x >= firstX && x < cw - 1 && (x >= e2X || y >= e2Y))
                {
                    const int e2 = eoTable[sgn(c - r[-stride - 1]) + sgn(c - r[stride + 1]) + 2]; stats[64 + e2] += d; count[64 + e2]++;
                    const int e3 = eoTable[sgn(c - r[-stride + 1]) + sgn(c - r[stride - 1]) + 2]; stats[96 + e3] += d; count[96 + e3]++;
                }
            }
    }
}

/* Encoder::computeSSD (encoder/encoder.cpp:1203-1270): the sum of squared differences of two planes -- what PSNR is computed from.  Its
 * block-wise fast path adds the same integers as the "slow path" loop restated here. */
uint64_t xo_plane_ssd(const xo_pixel* fenc, const xo_pixel* rec, intptr_t stride, int width, int height)
{
    uint64_t ssd = 0;
    for (int y = 0; y < height; y++, fenc += stride, rec += stride)
        for (int x = 0; x < width; x++) { const int d = (int)fenc[x] - (int)rec[x]; ssd += (uint64_t)(d * d); }
    return ssd;
}

/* SAO of a whole luma plane (SAO::generateLumaOffsets + applyPixelOffsets, encoder/sao.cpp:268-623) restated OUT OF PLACE: the reference works in
 * place CTU by CTU and keeps unmodified copies of the neighbouring row / column (m_tmpU, m_tmpL), i.e. every pixel is classified on the
 * picture as it was before SAO.  params: per CTU { typeIdx (-1 off, 0..3 EO, 4 BO), bandPos, offset[4] } with merges already resolved. */
void xo_sao_apply_frame(const xo_pixel* in, xo_pixel* out, intptr_t stride, int picWidth, int picHeight, int ctuSize, const int32_t* params)
{
    const int nx = (picWidth + ctuSize - 1) / ctuSize, ny = (picHeight + ctuSize - 1) / ctuSize, pm = (1 << X265_DEPTH) - 1;
    for (int y = 0; y < picHeight; y++) memcpy(out + y * stride, in + y * stride, picWidth * sizeof(xo_pixel));
    static const int ax[4] = { -1, 0, -1, 1 }, ay[4] = { 0, -1, -1, -1 };
    for (int addr = 0; addr < nx * ny; addr++)
    {
        const int32_t* p = params + 6 * addr;
        const int type = p[0];
        if (type < 0) continue;
        const int lpelx = (addr % nx) * ctuSize, tpely = (addr / nx) * ctuSize;
        const int above = (!tpely) | (addr < nx), lastRow = addr >= nx * ny - nx;
        const int picH = lastRow ? (picHeight < tpely + ctuSize ? picHeight : tpely + ctuSize) : picHeight;
        const int rpelx = lpelx + ctuSize < picWidth ? lpelx + ctuSize : picWidth, bpely = tpely + ctuSize < picH ? tpely + ctuSize : picH;
        const int cw = rpelx - lpelx, ch = bpely - tpely;
        int offEo[5];                                                   /* m_offsetEo: by edge type through s_eoTable, class 0 = no offset (:609-618) */
        { const int off[5] = { 0, p[2], p[3], p[4], p[5] }; for (int e = 0; e < 5; e++) offEo[e] = (int8_t)off[k_eoTable[e]]; }
        int8_t offBo[32]; memset(offBo, 0, sizeof(offBo));
        for (int i = 0; i < 4; i++) offBo[(p[1] + i) & 31] = (int8_t)p[2 + i];
        const int startX = (type == 1 || type == 4) ? 0 : !lpelx, endX = (type == 1 || type == 4) ? cw : (rpelx == picWidth ? cw - 1 : cw);
        const int startY = (type == 0 || type == 4) ? 0 : above, endY = (type == 0 || type == 4) ? ch : (bpely == picH ? ch - 1 : ch);
        for (int y = startY; y < endY; y++)
            for (int x = startX; x < endX; x++)
            {
                const xo_pixel* r = in + (tpely + y) * stride + lpelx + x;
                int v;
                if (type == 4) v = r[0] + offBo[r[0] >> (X265_DEPTH - 5)];
                else v = r[0] + offEo[sgn(r[0] - r[ay[type] * stride + ax[type]]) + sgn(r[0] - r[-ay[type] * stride - ax[type]]) + 2];
                out[(tpely + y) * stride + lpelx + x] = (xo_pixel)(v < 0 ? 0 : v > pm ? pm : v);
            }
    }
}

/* SSIM of a picture as the frame filter accumulates it (FrameFilter::processPostRow, encoder/framefilter.cpp:704-722; calculateSSIM :839-865;
 * ssim_4x4x2_core / ssim_end_1 / ssim_end_4, common/pixel.cpp:623-693).  The 4x4 blocks sit on a grid shifted by (2,2); a window is 2x2 blocks;
 * every CTU row r accumulates its window rows in FLOAT, four windows at a time (the ssim_end_4 partial) added to the row's running sum, and the
 * frame total is the DOUBLE sum of the row results (m_ssim += ...).  Restated on the global block grid instead of the reference's two rolling
 * sum rows: CTU row r starts at block row B0 = 0 (r = 0) or r*ctu/4 - 3 and holds hb block rows.  Float expressions keep the reference's
 * order of operations (this file must be compiled without FMA contraction: x86-64 baseline has none). */
static void ssim_block(const xo_pixel* a, intptr_t sa, const xo_pixel* b, intptr_t sb, uint32_t s[4])
{
    s[0] = s[1] = s[2] = s[3] = 0;
    for (int y = 0; y < 4; y++)
        for (int x = 0; x < 4; x++)
        {
            const uint32_t p = a[y * sa + x], q = b[y * sb + x];
            s[0] += p; s[1] += q; s[2] += p * p + q * q; s[3] += p * q;
        }
}

static float ssim_window(int s1, int s2, int ss, int s12)
{
#if X265_DEPTH > 8
    const float c1 = (float)(.01 * .01 * ((1 << X265_DEPTH) - 1) * ((1 << X265_DEPTH) - 1) * 64), c2 = (float)(.03 * .03 * ((1 << X265_DEPTH) - 1) * ((1 << X265_DEPTH) - 1) * 64 * 63);
    const float f1 = (float)s1, f2 = (float)s2, fss = (float)ss, f12 = (float)s12;
    const float vars = fss * 64 - f1 * f1 - f2 * f2, covar = f12 * 64 - f1 * f2;
#else
    const int c1 = (int)(.01 * .01 * ((1 << X265_DEPTH) - 1) * ((1 << X265_DEPTH) - 1) * 64 + .5), c2 = (int)(.03 * .03 * ((1 << X265_DEPTH) - 1) * ((1 << X265_DEPTH) - 1) * 64 * 63 + .5);
    const int f1 = s1, f2 = s2;
    const int vars = ss * 64 - f1 * f1 - f2 * f2, covar = s12 * 64 - f1 * f2;
#endif
    return (float)(2 * f1 * f2 + c1) * (float)(2 * covar + c2) / ((float)(f1 * f1 + f2 * f2 + c1) * (float)(vars + c2));
}

/* rowSsim / rowCnt: one entry per CTU row; *total = sum of rowSsim in row order (double), *cnt = sum of rowCnt: the frame's SSIM is total / cnt
 * (Encoder::finishFrameStats, encoder.cpp:3193-3198) */
void xo_ssim_frame(const xo_pixel* rec, intptr_t stride1, const xo_pixel* fenc, intptr_t stride2, int width, int height, int ctuSize,
                   float* rowSsim, uint32_t* rowCnt, double* total, uint32_t* cnt)
{
    const int numRows = (height + ctuSize - 1) / ctuSize;
    const uint32_t wb = (uint32_t)(width - 2) >> 2;
    *total = 0; *cnt = 0;
    for (int r = 0; r < numRows; r++)
    {
        const int start = r == 0, end = r == numRows - 1;
        uint32_t minY = r * ctuSize - 4 * !start, maxY = (r + 1) * ctuSize - 4 * !end;
        if (maxY > (uint32_t)height) maxY = height;
        minY += start ? 2 : -6;
        const uint32_t hb = (maxY - minY) >> 2;
        float ssim = 0.0f;
        for (uint32_t y = 1; y < hb; y++)
            for (uint32_t x = 0; x + 1 < wb; x += 4)
            {
                float part = 0.0f;
                for (uint32_t i = x; i < x + 4 && i + 1 < wb; i++)
                {
                    uint32_t s[4] = { 0, 0, 0, 0 }, t[4];
                    for (int k = 0; k < 4; k++)
                    {
                        const intptr_t py = minY + 4 * (y - 1 + (k >> 1)), px = 2 + 4 * (i + (k & 1));
                        ssim_block(rec + py * stride1 + px, stride1, fenc + py * stride2 + px, stride2, t);
                        for (int c = 0; c < 4; c++) s[c] += t[c];
                    }
                    part += ssim_window((int)s[0], (int)s[1], (int)s[2], (int)s[3]);
                }
                ssim += part;
            }
        rowSsim[r] = ssim; rowCnt[r] = (hb - 1) * (wb - 1);
        *total += ssim; *cnt += rowCnt[r];
    }
}

/* ---------------------------------------------------------------------------------------------------------------------------------------------
 * Deblocking of a whole 4:2:0 picture (Deblock::deblockCTU / deblockCU / setEdgefilter* / getBoundaryStrength / edgeFilterLuma / edgeFilterChroma,
 * common/deblock.cpp:37-497; pelFilterLumaStrong_c / pelFilterChroma_*_c, common/loopfilter.cpp:136-232), restated per 4x4 unit instead of as the
 * reference's recursive walk over the CU quadtree: a unit knows its CU size, so "is this unit's left / top side a transform edge, a prediction edge,
 * a CU edge" is arithmetic on its position inside the CU.  The per-partition arrays are the reference's own (CUData, CTU after CTU, z-scan order).
 * One slice; all vertical edges first, then all horizontal edges.
 * --------------------------------------------------------------------------------------------------------------------------------------------- */
static const uint8_t k_dbkTc[54] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9,
                                     10, 11, 13, 14, 16, 18, 20, 22, 24 };                       /* HEVC table 8-12 (deblock.cpp:499-503) */
static const uint8_t k_dbkBeta[52] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36,
                                       38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };  /* deblock.cpp:505-509 */

static inline int dbk_clip3(int lo, int hi, int v) { return v < lo ? lo : v > hi ? hi : v; }
static inline int dbk_pix(int v) { return dbk_clip3(0, (1 << X265_DEPTH) - 1, v); }

static uint32_t dbk_part(const xo_deblock_pic* d, int ux, int uy)          /* index of unit (ux, uy) in the CTU-major z-scan arrays */
{
    const int upc = d->ctuSize >> 2, nx = (d->width + d->ctuSize - 1) / d->ctuSize;
    const int lx = ux % upc, ly = uy % upc;
    uint32_t z = 0;
    for (int b = 0; b < 4; b++) z |= ((lx >> b) & 1u) << (2 * b) | ((ly >> b) & 1u) << (2 * b + 1);
    return (uint32_t)((uy / upc) * nx + ux / upc) * (uint32_t)(upc * upc) + z;
}

static int dbk_mv_far(const int32_t* a, const int32_t* b) { return abs(a[0] - b[0]) >= 4 || abs(a[1] - b[1]) >= 4; }

/* boundary strength of the edge on the left (dir 0) / top (dir 1) side of unit (ux, uy); deblock.cpp:46-70, 72-101, 125-247 */
int xo_deblock_bs(const xo_deblock_pic* d, int ux, int uy, int dir)
{
    const uint32_t q = dbk_part(d, ux, uy);
    if (!d->predMode[q]) return 0;
    const int pos = (dir ? uy : ux) * 4, cuSize = 1 << d->log2CUSize[q], rel = pos & (cuSize - 1);
    int bs;
    if (!rel) bs = pos > 0 ? 2 : 0;                                                   /* CU edge: 2 when the neighbour exists (bsCuEdge) */
    else if (!(rel & ((cuSize >> d->tuDepth[q]) - 1))) bs = 2;                         /* transform edge (setEdgefilterTU) */
    else
    {                                                                                 /* prediction edge inside the CU (setEdgefilterPU) */
        const int ps = d->partSize[q];
        const int at = dir ? (ps == 1 || ps == 3 ? cuSize >> 1 : ps == 4 ? cuSize >> 2 : ps == 5 ? cuSize - (cuSize >> 2) : -1)
                           : (ps == 2 || ps == 3 ? cuSize >> 1 : ps == 6 ? cuSize >> 2 : ps == 7 ? cuSize - (cuSize >> 2) : -1);
        bs = rel == at ? 1 : 0;
    }
    if (!bs || ((dir ? uy : ux) & 1)) return bs;                                       /* only the 8x8 grid is examined further (bsCheck) */
    const uint32_t p = dir ? dbk_part(d, ux, uy - 1) : dbk_part(d, ux - 1, uy);
    if (d->predMode[p] == 2 || d->predMode[q] == 2) return 2;
    if (bs > 1 && (((d->cbfLuma[q] >> d->tuDepth[q]) & 1) || ((d->cbfLuma[p] >> d->tuDepth[p]) & 1))) return 1;
    static const int32_t zero[2] = { 0, 0 };
    const int rp0 = d->refIdx0[p] >= 0 ? d->refPic[0][d->refIdx0[p]] : -1, rq0 = d->refIdx0[q] >= 0 ? d->refPic[0][d->refIdx0[q]] : -1;
    const int32_t* mp0 = rp0 >= 0 ? d->mv0 + 2 * p : zero; const int32_t* mq0 = rq0 >= 0 ? d->mv0 + 2 * q : zero;
    if (d->sliceIsP) return rp0 != rq0 || dbk_mv_far(mq0, mp0);
    const int rp1 = d->refIdx1[p] >= 0 ? d->refPic[1][d->refIdx1[p]] : -1, rq1 = d->refIdx1[q] >= 0 ? d->refPic[1][d->refIdx1[q]] : -1;
    const int32_t* mp1 = rp1 >= 0 ? d->mv1 + 2 * p : zero; const int32_t* mq1 = rq1 >= 0 ? d->mv1 + 2 * q : zero;
    if ((rp0 == rq0 && rp1 == rq1) || (rp0 == rq1 && rp1 == rq0))
    {
        if (rp0 != rp1)
            return rp0 == rq0 ? (dbk_mv_far(mq0, mp0) || dbk_mv_far(mq1, mp1)) : (dbk_mv_far(mq1, mp0) || dbk_mv_far(mq0, mp1));
        return (dbk_mv_far(mq0, mp0) || dbk_mv_far(mq1, mp1)) && (dbk_mv_far(mq1, mp0) || dbk_mv_far(mq0, mp1));
    }
    return 1;
}

/* one 4-sample luma edge segment; src = first sample on the Q side, offset = across the edge, step = along it (deblock.cpp:249-415) */
static void dbk_luma_segment(xo_pixel* src, intptr_t step, intptr_t offset, int bs, int qp, int betaOffset, int tcOffset, int maskP, int maskQ)
{
    const int sh = X265_DEPTH - 8;
    const int beta = k_dbkBeta[dbk_clip3(0, 51, qp + betaOffset)] << sh;
#define S(line, k) ((int)src[(line) * step + (k) * offset])
    const int dp0 = abs(S(0, -3) - 2 * S(0, -2) + S(0, -1)), dq0 = abs(S(0, 0) - 2 * S(0, 1) + S(0, 2));
    const int dp3 = abs(S(3, -3) - 2 * S(3, -2) + S(3, -1)), dq3 = abs(S(3, 0) - 2 * S(3, 1) + S(3, 2));
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 >= beta) return;
    const int tc = k_dbkTc[dbk_clip3(0, 53, qp + 2 * (bs - 1) + tcOffset)] << sh;
    int strong = 2 * d0 < (beta >> 2) && 2 * d3 < (beta >> 2);
    for (int line = 0; line < 4 && strong; line += 3)
        strong = abs(S(line, -4) - S(line, -1)) + abs(S(line, 3) - S(line, 0)) < (beta >> 3) && abs(S(line, -1) - S(line, 0)) < ((tc * 5 + 1) >> 1);
    const int side = (beta + (beta >> 1)) >> 3;
    const int maskP1 = (dp0 + dp3 < side ? -1 : 0) & maskP, maskQ1 = (dq0 + dq3 < side ? -1 : 0) & maskQ;
    for (int i = 0; i < 4; i++)
    {
        const int m0 = S(i, -4), m1 = S(i, -3), m2 = S(i, -2), m3 = S(i, -1), m4 = S(i, 0), m5 = S(i, 1), m6 = S(i, 2), m7 = S(i, 3);
        xo_pixel* s = src + i * step;
        if (strong)
        {   /* pelFilterLumaStrong_c: no clip to the pixel range, the result is cast */
            const int tcP = (2 * tc) & maskP, tcQ = (2 * tc) & maskQ;
            s[-3 * offset] = (xo_pixel)(dbk_clip3(-tcP, tcP, ((2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3) - m1) + m1);
            s[-2 * offset] = (xo_pixel)(dbk_clip3(-tcP, tcP, ((m1 + m2 + m3 + m4 + 2) >> 2) - m2) + m2);
            s[-offset] = (xo_pixel)(dbk_clip3(-tcP, tcP, ((m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3) - m3) + m3);
            s[0] = (xo_pixel)(dbk_clip3(-tcQ, tcQ, ((m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3) - m4) + m4);
            s[offset] = (xo_pixel)(dbk_clip3(-tcQ, tcQ, ((m3 + m4 + m5 + m6 + 2) >> 2) - m5) + m5);
            s[2 * offset] = (xo_pixel)(dbk_clip3(-tcQ, tcQ, ((m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3) - m6) + m6);
            continue;
        }
        int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
        if (abs(delta) >= tc * 10) continue;
        delta = dbk_clip3(-tc, tc, delta);
        s[-offset] = (xo_pixel)dbk_pix(m3 + (delta & maskP));
        s[0] = (xo_pixel)dbk_pix(m4 - (delta & maskQ));
        if (maskP1) s[-2 * offset] = (xo_pixel)dbk_pix(m2 + dbk_clip3(-(tc >> 1), tc >> 1, (((m1 + m3 + 1) >> 1) - m2 + delta) >> 1));
        if (maskQ1) s[offset] = (xo_pixel)dbk_pix(m5 + dbk_clip3(-(tc >> 1), tc >> 1, (((m6 + m4 + 1) >> 1) - m5 - delta) >> 1));
    }
#undef S
}

static const uint8_t k_dbkChromaScale[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 33, 33,
                                              34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };   /* constants.cpp:346-350 (g_chromaScale) */

void xo_deblock_frame(const xo_deblock_pic* d, xo_pixel* Y, intptr_t strideY, xo_pixel* Cb, xo_pixel* Cr, intptr_t strideC, uint8_t* bsOut)
{
    xo_deblock_rows(d, Y, strideY, Cb, Cr, strideC, bsOut, 0, (d->height + d->ctuSize - 1) / d->ctuSize);
}
/* A band of CTU rows [ctuRow0, ctuRow1): the edges of those rows' CTUs, the band's top edge included -- it changes the last lines of the row above, which must hold that row's
 * own deblocking already.  What FrameFilter::processRow does for each row in turn (framefilter.cpp:576-676: ParallelFilter::processTasks deblocks row r when the row encoders are
 * `m_filterRowDelay` rows ahead).  Bands in increasing order over a picture give xo_deblock_frame's picture (tests/test_filters_bands.py): the vertical edges of a row read and
 * write that row's samples only; a horizontal edge reads 4 and writes 3 lines either side, and edges lie 8 lines apart. */
void xo_deblock_rows(const xo_deblock_pic* d, xo_pixel* Y, intptr_t strideY, xo_pixel* Cb, xo_pixel* Cr, intptr_t strideC, uint8_t* bsOut, int ctuRow0, int ctuRow1)
{
    const int uw = d->width >> 2, uh = d->height >> 2, upc = d->ctuSize / 4;
    const int uy0 = ctuRow0 * upc, uy1 = ctuRow1 * upc < uh ? ctuRow1 * upc : uh;
    for (int dir = 0; dir < 2; dir++)
        for (int uy = uy0; uy < uy1; uy++)
            for (int ux = 0; ux < uw; ux++)
            {
                int bs = xo_deblock_bs(d, ux, uy, dir);
                /* --slices: the CTU above the first row of a slice is no neighbour (m_cuAbove = NULL, cudata.cpp:323; CUData::getPUAbove returns NULL and Deblock::setLoopfilterParam leaves the top edge out, deblock.cpp:59-66) */
                if (dir && d->sliceFirstRow && !(uy % (d->ctuSize / 4)) && d->sliceFirstRow[uy / (d->ctuSize / 4)]) bs = 0;
                if (bsOut) bsOut[((size_t)dir * uh + uy) * uw + ux] = (uint8_t)bs;
                if (!bs || ((dir ? uy : ux) & 1)) continue;                             /* edges on the 8x8 grid only (DEBLOCK_SMALLEST_BLOCK) */
                const uint32_t q = dbk_part(d, ux, uy), p = dir ? dbk_part(d, ux, uy - 1) : dbk_part(d, ux - 1, uy);
                int maskP = -1, maskQ = -1;
                if (d->tqBypassEnabled)
                {
                    maskP = d->tqBypass[p] ? 0 : -1; maskQ = d->tqBypass[q] ? 0 : -1;
                    if (!(maskP | maskQ)) continue;
                }
                const int qp = (d->qp[p] + d->qp[q] + 1) >> 1;
                dbk_luma_segment(Y + (intptr_t)uy * 4 * strideY + ux * 4, dir ? 1 : strideY, dir ? strideY : 1, bs, qp, 2 * d->betaOffsetDiv2, 2 * d->tcOffsetDiv2, maskP, maskQ);
                /* chroma: edges on the 8-sample CHROMA grid across the edge (deblock.cpp:104-113: luma position a multiple of 8 << the chroma shift in that direction), intra
                   strength only, one 4-sample chroma segment per (1 << the chroma shift ALONG the edge) luma units, its strength and QPs those of the first of them (:457-459).
                   4:2:0: the 16-sample luma grid, a segment per two luma units; 4:2:2: 16 across vertical edges / 8 across horizontal ones, a segment per unit along vertical
                   edges / per two along horizontal ones; 4:4:4: the luma grid */
                const int hs = d->chromaFormat == 3 ? 0 : 1, vs = (d->chromaFormat == 2 || d->chromaFormat == 3) ? 0 : 1;
                const int across = dir ? vs : hs, along = dir ? hs : vs;
                if (bs < 2 || ((dir ? uy : ux) & ((2 << across) - 1)) || ((dir ? ux : uy) & ((1 << along) - 1))) continue;
                for (int c = 0; c < 2; c++)
                {
                    int cqp = qp + (c ? d->crQpOffset : d->cbQpOffset);
                    if (cqp >= 30) cqp = d->chromaFormat <= 1 ? k_dbkChromaScale[cqp > 57 ? 57 : cqp] : (cqp < 51 ? cqp : 51);      /* :483-484: the table for 4:2:0 only */
                    const int tc = k_dbkTc[dbk_clip3(0, 53, cqp + 2 + 2 * d->tcOffsetDiv2)] << (X265_DEPTH - 8);
                    xo_pixel* s = (c ? Cr : Cb) + (intptr_t)((uy * 4) >> vs) * strideC + ((ux * 4) >> hs);
                    const intptr_t step = dir ? 1 : strideC, off = dir ? strideC : 1;
                    for (int i = 0; i < 4; i++, s += step)
                    {
                        const int m2 = s[-2 * off], m3 = s[-off], m4 = s[0], m5 = s[off];
                        const int delta = dbk_clip3(-tc, tc, ((m4 - m3) * 4 + m2 - m5 + 4) >> 3);
                        s[-off] = (xo_pixel)dbk_pix(m3 + (delta & maskP)); s[0] = (xo_pixel)dbk_pix(m4 - (delta & maskQ));
                    }
                }
            }
}
