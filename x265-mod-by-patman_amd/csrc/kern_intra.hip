// kern_intra.hip -- batched HEVC intra prediction (reference intrapred.cpp:31-234):
// [1 2 1] reference smoothing, DC (+edge filter), planar, the 33 angular modes, and the
// "all angles" variant whose horizontal modes (< 18) stay transposed (intrapred.cpp:217-231).
// Neighbour layout (predict.h:75): [0] top-left, [1..2N] above + above-right, [2N+1..4N] left + below-left.
#include "xh_common.h"
using namespace xh;

namespace {

__device__ const int8_t k_angleTable[17] = { -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
__device__ const int16_t k_invAngleTable[8] = { 4096, 1638, 910, 630, 482, 390, 315, 256 };
__device__ const uint8_t k_intraFilterFlags[35] = {   // constants.cpp:561
    0x38, 0x00, 0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x38 };

__global__ __launch_bounds__(256) void intra_filter_kernel(int N, const pixel* __restrict__ nb, const int32_t* __restrict__ nOff,
                                                           pixel* __restrict__ flt, const int32_t* __restrict__ fOff, int n)
{   // intrapred.cpp:31-51; one wave per neighbour array
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (item >= n) return;
    const pixel* s = nb + (nOff ? nOff[item] : item * (4 * N + 1));
    pixel* f = flt + (fOff ? fOff[item] : item * (4 * N + 1));
    const int n2 = 2 * N, tl = s[0];
    for (int i = lane; i <= 4 * N; i += 64)
    {
        int v;
        if (i == 0) v = (2 * tl + s[1] + s[n2 + 1] + 2) >> 2;
        else if (i == n2 || i == 2 * n2) v = s[i];
        else if (i == n2 + 1) v = (2 * s[i] + tl + s[i + 1] + 2) >> 2;
        else v = (2 * s[i] + s[i - 1] + s[i + 1] + 2) >> 2;
        f[i] = (pixel)v;
    }
}

// Predict one N x N block for `mode` from neighbours `src` (global memory).  Output element (y, x) of the
// FINAL (un-flipped) prediction goes to out[y * os + x]; when keepFlipped is set, horizontal modes are
// written as computed on the flipped neighbours (the all-angs layout).
__device__ void predict_block(int N, const pixel* __restrict__ src, int mode, int bFilter, pixel* __restrict__ out, intptr_t os,
                              bool keepFlipped, pixel* sp /* LDS, 4N+1 */, pixel* refBuf /* LDS, 66 */)
{
    const int n2 = 2 * N, tid = threadIdx.x;
    const bool angular = mode >= 2;
    const bool hor = angular && mode < 18;
    // stage neighbours in LDS (flipped for horizontal modes, intrapred.cpp:111-120)
    for (int i = tid; i <= 4 * N; i += 256)
    {
        int j = i;
        if (hor && i >= 1) j = (i <= n2) ? i + n2 : i - n2;
        sp[i] = src[j];
    }
    __syncthreads();
    const int lg = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
    if (mode == 0)
    {   // planar, intrapred.cpp:87-100
        const pixel* above = sp + 1; const pixel* left = sp + n2 + 1;
        const int tr = above[N], bl = left[N];
        for (int i = tid; i < N * N; i += 256)
        {
            int y = i >> lg, x = i & (N - 1);
            out[y * os + x] = (pixel)(((N - 1 - x) * left[y] + (N - 1 - y) * above[x] + (x + 1) * tr + (y + 1) * bl + N) >> (lg + 1));
        }
        return;
    }
    if (mode == 1)
    {   // DC, intrapred.cpp:53-85
        const pixel* above = sp + 1; const pixel* left = sp + n2 + 1;
        int dc = N;
        for (int i = 0; i < N; i++) dc += above[i] + left[i];
        dc /= 2 * N;
        for (int i = tid; i < N * N; i += 256)
        {
            int y = i >> lg, x = i & (N - 1);
            int v = dc;
            if (bFilter)
            {
                if (x == 0 && y == 0) v = (above[0] + left[0] + 2 * dc + 2) >> 2;
                else if (y == 0) v = (above[x] + 3 * dc + 2) >> 2;
                else if (x == 0) v = (left[y] + 3 * dc + 2) >> 2;
            }
            out[y * os + x] = (pixel)v;
        }
        return;
    }
    // angular, intrapred.cpp:102-204
    const int angleOffset = hor ? 10 - mode : mode - 26;
    const int angle = k_angleTable[8 + angleOffset];
    const pixel* ref = sp + 1;
    if (angle < 0)
    {
        const int nbProjected = -((N * angle) >> 5) - 1;
        pixel* rp = refBuf + nbProjected + 1;
        const int invAngle = k_invAngleTable[-angleOffset - 1];
        for (int i = tid; i < nbProjected; i += 256) rp[-2 - i] = sp[n2 + ((128 + (i + 1) * invAngle) >> 8)];
        for (int i = tid; i < N + 1; i += 256) rp[-1 + i] = sp[i];
        __syncthreads();
        ref = rp;
    }
    const bool transposeOut = hor && !keepFlipped;
    for (int i = tid; i < N * N; i += 256)
    {
        int y = i >> lg, x = i & (N - 1);
        int v;
        if (!angle)
        {
            v = sp[1 + x];
            if (bFilter && x == 0) v = clip_pixel((int16_t)((int)sp[1] + (((int)sp[n2 + 1 + y] - (int)sp[0]) >> 1)));
        }
        else
        {
            int angleSum = (y + 1) * angle;
            int off = angleSum >> 5, frac = angleSum & 31;
            v = frac ? (((32 - frac) * ref[off + x] + frac * ref[off + x + 1] + 16) >> 5) : ref[off + x];
        }
        if (transposeOut) out[x * os + y] = (pixel)v;
        else out[y * os + x] = (pixel)v;
    }
}

__global__ __launch_bounds__(256) void intra_pred_kernel(int N, const pixel* __restrict__ nb, const int32_t* __restrict__ nOff,
                                                         pixel* __restrict__ dst, intptr_t ds, const int32_t* __restrict__ dOff,
                                                         const int32_t* __restrict__ modeFilter, int n)
{
    __shared__ pixel sp[132];
    __shared__ pixel refBuf[68];
    const int item = blockIdx.x;
    const int mf = modeFilter[item];
    predict_block(N, nb + (nOff ? nOff[item] : item * (4 * N + 1)), mf & 0xFF, (mf >> 8) & 1,
                  dst + (dOff ? dOff[item] : (intptr_t)item * N * N), ds, false, sp, refBuf);
}

__global__ __launch_bounds__(256) void intra_allangs_kernel(int N, const pixel* __restrict__ ref, const int32_t* __restrict__ rOff,
                                                            const pixel* __restrict__ flt, const int32_t* __restrict__ fOff,
                                                            pixel* __restrict__ dst, int bLuma, int n)
{   // intrapred.cpp:206-234: grid = (33 modes, n items)
    __shared__ pixel sp[132];
    __shared__ pixel refBuf[68];
    const int mode = 2 + blockIdx.x, item = blockIdx.y;
    const bool useFilt = (k_intraFilterFlags[mode] & N) != 0;
    const pixel* src = useFilt ? flt + (fOff ? fOff[item] : item * (4 * N + 1)) : ref + (rOff ? rOff[item] : item * (4 * N + 1));
    predict_block(N, src, mode, bLuma, dst + ((intptr_t)item * 33 + (mode - 2)) * N * N, N, true, sp, refBuf);
}


// ---- intra mode scan: the data-parallel half of Search::estIntraPredQT (encoder/search.cpp:1655-1745) -------------------
// One workgroup per (mode, CU): the prediction is built in LDS (never stored), the source block is staged beside it
// (transposed for the horizontal modes, which the all-angs layout keeps flipped -- search.cpp:1721-1731 compares those
// with the transposed source), and the workgroup returns sa8d(source, prediction) << costShift.
// sa8d per size as the table slots are composed (pixel.cpp:291-369, 1180-1184): 4x4 -> satd_4x4 = sum >> 1;
// 8x8 -> (sum + 2) >> 2; 16x16 -> (four raw 8x8 sums + 2) >> 2; 32x32 -> sum of its four 16x16 values.
__device__ int had8x8_lds(const pixel* a, const pixel* b, int N)
{   // raw 8x8 Hadamard |sum| of (a - b), rows N apart
    int d[8][8];
#pragma unroll
    for (int y = 0; y < 8; y++)
    {
#pragma unroll
        for (int x = 0; x < 8; x++) d[y][x] = (int)a[y * N + x] - (int)b[y * N + x];
#pragma unroll
        for (int st = 1; st < 8; st <<= 1)
#pragma unroll
            for (int x = 0; x < 8; x++)
                if (!(x & st)) { int u = d[y][x], v = d[y][x + st]; d[y][x] = u + v; d[y][x + st] = u - v; }
    }
    int s = 0;
#pragma unroll
    for (int x = 0; x < 8; x++)
    {
#pragma unroll
        for (int st = 1; st < 8; st <<= 1)
#pragma unroll
            for (int y = 0; y < 8; y++)
                if (!(y & st)) { int u = d[y][x], v = d[y + st][x]; d[y][x] = u + v; d[y + st][x] = u - v; }
#pragma unroll
        for (int y = 0; y < 8; y++) s += abs(d[y][x]);
    }
    return s;
}
__global__ __launch_bounds__(256) void intra_cost_kernel(int N, int origSize, const pixel* __restrict__ src, intptr_t ss, const int32_t* __restrict__ srcOff,
                                                         intptr_t srcItemStride, const pixel* __restrict__ nbRef, const pixel* __restrict__ nbFilt, int nbPitch,
                                                         int costShift, int32_t* __restrict__ costs, int n)
{
    __shared__ pixel sp[132];
    __shared__ pixel refBuf[68];
    __shared__ pixel s_pred[32 * 32];
    __shared__ pixel s_fenc[32 * 32];
    __shared__ int s_part[16];
    const int mode = blockIdx.x, item = blockIdx.y, tid = threadIdx.x;
    const bool hor = mode >= 2 && mode < 18;
    // which neighbour array, which edge filter (search.cpp:1702-1735)
    bool useFilt;
    if (mode == 1) useFilt = false;
    else if (mode == 0) useFilt = (origSize & (8 | 16 | 32)) != 0;
    else useFilt = (k_intraFilterFlags[mode] & N) != 0;
    const int bFilter = mode == 0 ? 0 : (N <= 16);
    const pixel* nb = (useFilt ? nbFilt : nbRef) + (intptr_t)item * nbPitch;
    predict_block(N, nb, mode, bFilter, s_pred, N, true, sp, refBuf);
    const pixel* f = src + (srcOff ? (intptr_t)srcOff[item] : (intptr_t)item * srcItemStride);
    const int lg = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
    for (int i = tid; i < N * N; i += 256)
    {
        const int y = i >> lg, x = i & (N - 1);
        s_fenc[i] = hor ? f[(intptr_t)x * ss + y] : f[(intptr_t)y * ss + x];
    }
    __syncthreads();
    int cost = 0;
    if (N == 4)
    {
        if (tid == 0)
        {   // satd_4x4 (pixel.cpp:210-231)
            int d[4][4], s = 0;
            for (int y = 0; y < 4; y++)
            {
                for (int x = 0; x < 4; x++) d[y][x] = (int)s_fenc[y * 4 + x] - (int)s_pred[y * 4 + x];
                int a0 = d[y][0] + d[y][1], a1 = d[y][0] - d[y][1], a2 = d[y][2] + d[y][3], a3 = d[y][2] - d[y][3];
                d[y][0] = a0 + a2; d[y][2] = a0 - a2; d[y][1] = a1 + a3; d[y][3] = a1 - a3;
            }
            for (int x = 0; x < 4; x++)
            {
                int a0 = d[0][x] + d[1][x], a1 = d[0][x] - d[1][x], a2 = d[2][x] + d[3][x], a3 = d[2][x] - d[3][x];
                s += abs(a0 + a2) + abs(a0 - a2) + abs(a1 + a3) + abs(a1 - a3);
            }
            cost = s >> 1;
        }
    }
    else
    {
        const int per = N >> 3, nsub = per * per;                 // 8x8 sub-blocks
        if (tid < nsub)
        {
            const int by = tid / per, bx = tid - by * per;
            s_part[tid] = had8x8_lds(s_fenc + (by * 8) * N + bx * 8, s_pred + (by * 8) * N + bx * 8, N);
        }
        __syncthreads();
        if (tid == 0)
        {
            if (N == 8) cost = (s_part[0] + 2) >> 2;
            else
            {   // per 16x16: four raw sums, one rounding (pixel.cpp:330-345); 32x32 adds its four 16x16 values
                for (int qy = 0; qy < per; qy += 2)
                    for (int qx = 0; qx < per; qx += 2)
                        cost += (s_part[qy * per + qx] + s_part[qy * per + qx + 1] + s_part[(qy + 1) * per + qx] + s_part[(qy + 1) * per + qx + 1] + 2) >> 2;
            }
        }
    }
    if (tid == 0) costs[(intptr_t)item * 35 + mode] = cost << costShift;
}
// 64x64 CUs: the reference scales source and neighbours to 32x32 first (search.cpp:1670-1688; pixel.cpp:551-594)
__global__ __launch_bounds__(256) void intra_scale64_kernel(const pixel* __restrict__ src, intptr_t ss, const int32_t* __restrict__ srcOff,
                                                            const pixel* __restrict__ nbRef, int nbPitch, pixel* __restrict__ fencS /* n x 1024 */,
                                                            pixel* __restrict__ nbS /* n x 129 */, int n)
{
    const int item = blockIdx.x, tid = threadIdx.x;
    const pixel* f = src + srcOff[item];
    for (int i = tid; i < 1024; i += 256)
    {   // scale2D_64to32: rounded mean of 2x2
        const int y = i >> 5, x = i & 31;
        const pixel* p = f + (intptr_t)(2 * y) * ss + 2 * x;
        fencS[(intptr_t)item * 1024 + i] = (pixel)((p[0] + p[1] + p[ss] + p[ss + 1] + 2) >> 2);
    }
    const pixel* r = nbRef + (intptr_t)item * nbPitch;
    pixel* o = nbS + (intptr_t)item * 129;
    if (tid == 0) o[0] = r[0];
    for (int i = tid; i < 128; i += 256)
    {   // scale1D_128to64 on both halves: dst[x] = (src[2x] + src[2x+1] + 1) >> 1 for the 64 above and the 64 left samples
        o[1 + i] = (pixel)((r[1 + 2 * i] + r[1 + 2 * i + 1] + 1) >> 1);
    }
}

// ---- the same scan with one WAVEFRONT per (mode, CU) pass: lane = pixel of an 8x8 sub-block ---------------------------------
// A workgroup owns one CU (neighbour arrays and source block staged in LDS once); its four wavefronts take the 35 modes
// round-robin.  Per 8x8 sub-block every lane predicts ITS pixel straight from the LDS neighbour arrays (the projected
// reference of negative angles is evaluated on the fly, intrapred.cpp:136-150) and the 64 differences go through a
// 64-point Walsh-Hadamard transform across the lanes (six butterfly stages: DPP quad_perm / bank-masked row shifts /
// row_ror:8, then v_permlane16_swap and v_permlane32_swap) -- the 2-D 8x8 Hadamard of sa8d up to coefficient order.
#define XI_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)
__device__ __forceinline__ int wht64_abs(int d, int lane)      // |coefficient| held by this lane; the caller sums over lanes
{
    const int s1 = (lane & 1) ? -1 : 1, s2 = (lane & 2) ? -1 : 1, s4 = (lane & 4) ? -1 : 1, s8 = (lane & 8) ? -1 : 1, s16 = (lane & 16) ? -1 : 1, s32 = (lane & 32) ? -1 : 1;
    d = XI_DPP(d, 0xB1) + __mul24(d, s1);                                     // lane ^ 1
    d = XI_DPP(d, 0x4E) + __mul24(d, s2);                                     // lane ^ 2
    {   // lane ^ 4: row_shl:4 into banks 0/2, row_shr:4 into banks 1/3
        int o = __builtin_amdgcn_update_dpp(0, d, 0x104, 0xF, 0x5, false);
        o = __builtin_amdgcn_update_dpp(o, d, 0x114, 0xF, 0xA, false);
        d = o + __mul24(d, s4);
    }
    d = XI_DPP(d, 0x128) + __mul24(d, s8);                                    // row_ror:8 = lane ^ 8
    { auto r = __builtin_amdgcn_permlane16_swap((unsigned)d, (unsigned)d, false, false); d = (int)((lane & 16) ? r[0] : r[1]) + __mul24(d, s16); }
    { auto r = __builtin_amdgcn_permlane32_swap((unsigned)d, (unsigned)d, false, false); d = (int)((lane & 32) ? r[0] : r[1]) + __mul24(d, s32); }
    return abs(d);
}
// neighbour sample i of the array a mode works on: the array itself, or "flipped" (above <-> left) for horizontal modes
__device__ __forceinline__ int nb_at(const pixel* a, int i, bool flip, int n2) { return a[flip && i >= 1 ? (i <= n2 ? i + n2 : i - n2) : i]; }

template<int N, int CPB>      // CPB = CUs per workgroup: 1 (the four wavefronts share one CU's modes) or 4 (one wavefront per CU, all 35 modes)
__global__ __launch_bounds__(256) void intra_scan_kernel(int origSize, const pixel* __restrict__ src, intptr_t ss, const int32_t* __restrict__ srcOff, intptr_t srcItemStride,
                                                         const pixel* __restrict__ nbRef, const pixel* __restrict__ nbFilt, int nbPitch,
                                                         int costShift, int32_t* __restrict__ costs, int n)
{
    constexpr int n2 = 2 * N, NB = 4 * N + 1, LG = N == 8 ? 3 : N == 16 ? 4 : 5, PER = N / 8;
    __shared__ pixel s_nbAll[CPB][2][NB + 3];
    __shared__ pixel s_fencAll[CPB][N * N];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int item = CPB == 1 ? (int)blockIdx.x : (int)blockIdx.x * CPB + wave;
    if (item >= n) return;                                      // wave-granular for CPB == 4 (no workgroup barrier below in that case)
    pixel (*s_nb)[NB + 3] = s_nbAll[CPB == 1 ? 0 : wave];
    pixel* s_fenc = s_fencAll[CPB == 1 ? 0 : wave];
    {
        constexpr int T = CPB == 1 ? 256 : 64;
        const int t = CPB == 1 ? tid : lane;
        const pixel* r = nbRef + (intptr_t)item * nbPitch; const pixel* fl = nbFilt + (intptr_t)item * nbPitch;
        for (int i = t; i < NB; i += T) { s_nb[0][i] = r[i]; s_nb[1][i] = fl[i]; }
        const pixel* f = src + (srcOff ? (intptr_t)srcOff[item] : (intptr_t)item * srcItemStride);
        for (int i = t; i < N * N; i += T) s_fenc[i] = f[(intptr_t)(i >> LG) * ss + (i & (N - 1))];
    }
    if (CPB == 1) __syncthreads(); else wave_sync();
    const int px = lane & 7, py = lane >> 3;
    for (int mode = CPB == 1 ? wave : 0; mode < 35; mode += CPB == 1 ? 4 : 1)
    {
        const bool hor = mode >= 2 && mode < 18;
        const bool useFilt = mode == 1 ? false : mode == 0 ? (origSize & (8 | 16 | 32)) != 0 : (k_intraFilterFlags[mode] & N) != 0;
        const bool bFilter = mode != 0 && N <= 16;
        const pixel* a = s_nb[useFilt ? 1 : 0];
        int dc = 0, angle = 0, invAngle = 0;
        if (mode == 1)
        {   // DC value (intrapred.cpp:58-64): N above + N left samples, one or two per lane
            int v = 0;
            for (int i = lane; i < n2; i += 64) v += i < N ? a[1 + i] : a[n2 + 1 + i - N];
            dc = (wave_sum(v) + N) / n2;
        }
        else if (mode >= 2)
        {
            const int angleOffset = hor ? 10 - mode : mode - 26;
            angle = k_angleTable[8 + angleOffset];
            if (angle < 0) invAngle = k_invAngleTable[-angleOffset - 1];
        }
        int total = 0, sum16 = 0;
        for (int sb = 0; sb < PER * PER; sb++)
        {
            // sub-block order: the four 8x8 of a 16x16 are consecutive (sa8d rounds per 16x16, pixel.cpp:330-345)
            const int q = sb >> 2, r4 = sb & 3;
            const int bx = PER == 1 ? 0 : ((q % (PER / 2 > 0 ? PER / 2 : 1)) * 2 + (r4 & 1)), by = PER == 1 ? 0 : ((q / (PER / 2 > 0 ? PER / 2 : 1)) * 2 + (r4 >> 1));
            const int x = bx * 8 + px, y = by * 8 + py;
            int p;
            if (mode == 0)
            {   // planar (intrapred.cpp:87-100)
                const int tr = a[1 + N], bl = a[n2 + 1 + N];
                p = ((N - 1 - x) * a[n2 + 1 + y] + (N - 1 - y) * a[1 + x] + (x + 1) * tr + (y + 1) * bl + N) >> (LG + 1);
            }
            else if (mode == 1)
            {   // DC + edge filter (intrapred.cpp:66-85)
                p = dc;
                if (bFilter)
                {
                    if (x == 0 && y == 0) p = (a[1] + a[n2 + 1] + 2 * dc + 2) >> 2;
                    else if (y == 0) p = (a[1 + x] + 3 * dc + 2) >> 2;
                    else if (x == 0) p = (a[n2 + 1 + y] + 3 * dc + 2) >> 2;
                }
            }
            else if (!angle)
            {   // pure vertical (or horizontal, on the flipped array): copy + optional edge filter (intrapred.cpp:152-166)
                p = nb_at(a, 1 + x, hor, n2);
                if (bFilter && x == 0) p = clip_pixel((int16_t)(nb_at(a, 1, hor, n2) + ((nb_at(a, n2 + 1 + y, hor, n2) - (int)a[0]) >> 1)));
            }
            else
            {   // 1/32-pel two-tap interpolation along the (possibly projected) reference row (intrapred.cpp:168-190)
                const int angleSum = (y + 1) * angle, off = angleSum >> 5, frac = angleSum & 31;
                auto ref = [&](int j) -> int {          // ref[j], j >= -1: the main array; j < -1: projected from the side array
                    return j >= -1 ? nb_at(a, 1 + j, hor, n2) : nb_at(a, n2 + ((128 + (-1 - j) * invAngle) >> 8), hor, n2);
                };
                p = frac ? ((32 - frac) * ref(off + x) + frac * ref(off + x + 1) + 16) >> 5 : ref(off + x);
            }
            const int f = hor ? s_fenc[x * N + y] : s_fenc[y * N + x];
            sum16 += wht64_abs(f - p, lane);                                   // per-lane partial of the current 8x8 / 16x16
            if (N == 8) total = (wave_sum(sum16) + 2) >> 2;
            else if (r4 == 3) { total += (wave_sum(sum16) + 2) >> 2; sum16 = 0; }
        }
        if (lane == 0) costs[(intptr_t)item * 35 + mode] = total << costShift;
    }
}

bool bad_n(int N) { return N != 4 && N != 8 && N != 16 && N != 32; }

} // namespace

extern "C" int x265hip_intra_filter_batch(void* stream, int N, const void* nb, const int32_t* nbOff, void* filt, const int32_t* filtOff, int n)
{
    if (n <= 0) return X265HIP_OK;
    if (bad_n(N)) { set_error("intra_filter_batch: N must be 4/8/16/32"); return X265HIP_EARG; }
    XH_KLAUNCH(intra_filter_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, N, (const pixel*)nb, nbOff, (pixel*)filt, filtOff, n);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
extern "C" int x265hip_intra_pred_batch(void* stream, int N, const void* nb, const int32_t* nbOff, void* dst, intptr_t dstStride,
                                        const int32_t* dstOff, const int32_t* modeFilter, int n)
{
    if (n <= 0) return X265HIP_OK;
    if (bad_n(N) || !modeFilter) { set_error("intra_pred_batch: bad arguments"); return X265HIP_EARG; }
    XH_KLAUNCH(intra_pred_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, N, (const pixel*)nb, nbOff, (pixel*)dst, dstStride, dstOff, modeFilter, n);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
extern "C" int x265hip_intra_allangs_batch(void* stream, int N, const void* ref, const int32_t* refOff, const void* filt, const int32_t* filtOff,
                                           void* dst, int bLuma, int n)
{
    if (n <= 0) return X265HIP_OK;
    if (bad_n(N)) { set_error("intra_allangs_batch: N must be 4/8/16/32"); return X265HIP_EARG; }
    XH_KLAUNCH(intra_allangs_kernel, dim3(33, n), dim3(256), 0, (hipStream_t)stream, N, (const pixel*)ref, refOff, (const pixel*)filt, filtOff, (pixel*)dst, bLuma, n);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

extern "C" size_t x265hip_intra_cost_workspace(int log2Size, int n)
{
    return log2Size == 6 ? (size_t)n * (1024 + 129 + 3) * sizeof(pixel) : 0;
}
extern "C" int x265hip_intra_cost_batch(void* stream, int log2Size, const void* srcPlane, intptr_t srcStride, const int32_t* srcOff,
                                        const void* nbRef, const void* nbFilt, int nbPitch, int n, int32_t* costs,
                                        void* workspace, size_t workspaceBytes)
{
    if (n <= 0) return X265HIP_OK;
    if (log2Size < 2 || log2Size > 6 || !srcPlane || !srcOff || !nbRef || !nbFilt || !costs || nbPitch < 4 * (1 << log2Size) + 1)
    { set_error("intra_cost_batch: bad arguments"); return X265HIP_EARG; }
    hipStream_t st = (hipStream_t)stream;
    const int size = 1 << log2Size;
    if (size == 64)
    {
        if (!workspace || workspaceBytes < x265hip_intra_cost_workspace(6, n)) { set_error("intra_cost_batch: workspace too small for 64x64 CUs"); return X265HIP_EARG; }
        pixel* fencS = (pixel*)workspace; pixel* nbS = fencS + (size_t)n * 1024;
        XH_KLAUNCH(intra_scale64_kernel, dim3(n), dim3(256), 0, st, (const pixel*)srcPlane, srcStride, srcOff, (const pixel*)nbRef, nbPitch, fencS, nbS, n);
        XH_LAUNCH_CHECK();
        // "we do not estimate filtering for downscaled samples": both neighbour arrays are the scaled unfiltered one
        XH_KLAUNCH((intra_scan_kernel<32, 1>), dim3(n), dim3(256), 0, st, 64, (const pixel*)fencS, (intptr_t)32, (const int32_t*)nullptr, (intptr_t)1024,
                           (const pixel*)nbS, (const pixel*)nbS, 129, 2, costs, n);
    }
    else if (size == 4)     // 16 pixels: the workgroup-per-(mode, CU) form
        XH_KLAUNCH(intra_cost_kernel, dim3(35, n), dim3(256), 0, st, size, size, (const pixel*)srcPlane, srcStride, srcOff, (intptr_t)0,
                           (const pixel*)nbRef, (const pixel*)nbFilt, nbPitch, 0, costs, n);
    else if (size == 8)
        XH_KLAUNCH((intra_scan_kernel<8, 4>), dim3((n + 3) / 4), dim3(256), 0, st, size, (const pixel*)srcPlane, srcStride, srcOff, (intptr_t)0,
                           (const pixel*)nbRef, (const pixel*)nbFilt, nbPitch, 0, costs, n);
    else if (size == 16)
        XH_KLAUNCH((intra_scan_kernel<16, 4>), dim3((n + 3) / 4), dim3(256), 0, st, size, (const pixel*)srcPlane, srcStride, srcOff, (intptr_t)0,
                           (const pixel*)nbRef, (const pixel*)nbFilt, nbPitch, 0, costs, n);
    else
        XH_KLAUNCH((intra_scan_kernel<32, 1>), dim3(n), dim3(256), 0, st, size, (const pixel*)srcPlane, srcStride, srcOff, (intptr_t)0,
                           (const pixel*)nbRef, (const pixel*)nbFilt, nbPitch, 0, costs, n);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
