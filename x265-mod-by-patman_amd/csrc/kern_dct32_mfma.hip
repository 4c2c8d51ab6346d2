// kern_dct32_mfma.hip -- 32x32 forward DCT as two exact integer matrix products on the gfx950
// matrix cores (v_mfma_i32_32x32x32_i8), one wavefront per TU, no LDS transposes.
//
// Reference arithmetic (dct.cpp:136-203,511-526): out = round2(T * round1(T * X^T)^T ... ) with
// round_s(v) = (int16)((v + (1 << (s-1))) >> s).  Both stages are 32x32x32 products of an int16
// matrix with the int8 DCT matrix T.  An int16 x is split EXACTLY into three int8 planes:
//     x = 256 * hi + lo_s + 256 * b,   hi = x >> 8 (floor, in [-128,127]),
//     lo_s = (int8)(x & 0xFF), b = (x >> 7) & 1
// so X*T = 256 * ((Hi + B) * T) + Lo_s * T with int32 accumulation -- exact for every int16 input
// (|acc| < 2^27), which is why 3 MFMAs per stage are issued instead of 2.
//
// Dataflow (lane l: r = l & 31, g = l >> 5):
//  stage 1  D1[n][j] = sum_m X[n][m] T[j][m]:  A = X  (lane: row n = r, k-slots m = 16g..16g+15,
//           one 32-byte load per lane), B = T^T (lane: col j = r, same k-slots -> T[r][16g..]).
//           C/D layout (dtype independent): lane holds col j = r, rows n = (i&3) + 8*(i>>2) + 4*g.
//  stage 2  D2[k][j] = sum_n T[k][n] D1'[n][j]: the rounded stage-1 registers ARE the B operand
//           (col j = r, k-slot i <-> n(i,g)); A = T with its columns permuted to the same
//           n(i,g) order.  MFMA pairs A's and B's k-slots positionally, so any consistent k
//           ordering gives the exact sum.
#include "xh_common.h"
#include "xh_internal.h"
#include "xh_dct32.h"
#include <cstdint>
#ifndef XH_LDS
#define XH_LDS __attribute__((address_space(3)))
#endif
#include <cstdlib>
using namespace xh;

namespace {

__global__ __launch_bounds__(256) void dct32_mfma_kernel(const int16_t* __restrict__ src, intptr_t ss, const int32_t* __restrict__ sOff,
                                                         int16_t* __restrict__ dst, const int32_t* __restrict__ dOff, int n)
{
    const int lane = threadIdx.x & 63, r = lane & 31, g = lane >> 5;
    const int wavesTotal = gridDim.x * 4;
    int tu = blockIdx.x * 4 + (threadIdx.x >> 6);
    v4i tB1, tA2;
    dct32_operands(r, g, tB1, tA2);       // built once per wave, amortised over the grid-stride loop
    const int shift2 = 11, add2 = 1 << (shift2 - 1);
    for (; tu < n; tu += wavesTotal)
    {
        const intptr_t so = sOff ? (intptr_t)sOff[tu] : (intptr_t)tu * 1024;
        const int16_t* p = src + so + (intptr_t)r * ss + 16 * g;
        int d[8];
        if (((uintptr_t)p & 15) == 0)
        {
            const int4 u0 = *(const int4*)p, u1 = *(const int4*)(p + 8);
            d[0] = u0.x; d[1] = u0.y; d[2] = u0.z; d[3] = u0.w; d[4] = u1.x; d[5] = u1.y; d[6] = u1.z; d[7] = u1.w;
        }
        else
        {
#pragma unroll
            for (int q = 0; q < 8; q++) d[q] = (int)(((unsigned)(uint16_t)p[2 * q]) | ((unsigned)(uint16_t)p[2 * q + 1] << 16));
        }
        v16i acc;
        dct32_forward(d, tB1, tA2, acc);
        int16_t* o = dst + (dOff ? (intptr_t)dOff[tu] : (intptr_t)tu * 1024);
#pragma unroll
        for (int i = 0; i < 16; i++)
        {
            int k = (i & 3) + 8 * (i >> 2) + 4 * g;
            o[k * 32 + r] = (int16_t)((acc[i] + add2) >> shift2);
        }
    }
}

// Inverse: one wavefront per TU; the coefficient block is read row by row (32 bytes per lane) and handed over through LDS so that every lane gets its column.
__global__ __launch_bounds__(256) void idct32_mfma_kernel(const int16_t* __restrict__ src, const int32_t* __restrict__ sOff, int16_t* __restrict__ dst, intptr_t ds,
                                                          const int32_t* __restrict__ dOff, int n)
{
    __shared__ __attribute__((aligned(16))) int16_t s_c[4][32 * 34];                        // rows of 34: the column reads of a wavefront spread over the banks
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, g = lane >> 5;
    const int wavesTotal = gridDim.x * 4;
    int tu = blockIdx.x * 4 + wave;
    v4i tB1, tA2;
    idct32_operands(r, g, tB1, tA2);
    const int shift2 = 12 - (X265_DEPTH - 8), add2 = 1 << (shift2 - 1);
    XH_LDS int16_t* tile = (XH_LDS int16_t*)s_c[wave];
    for (; tu < n; tu += wavesTotal)
    {
        const int16_t* p = src + (sOff ? (intptr_t)sOff[tu] : (intptr_t)tu * 1024) + r * 32 + 16 * g;
        int u[8];
        if (((uintptr_t)p & 15) == 0)
        {
            const int4 u0 = *(const int4*)p, u1 = *(const int4*)(p + 8);
            u[0] = u0.x; u[1] = u0.y; u[2] = u0.z; u[3] = u0.w; u[4] = u1.x; u[5] = u1.y; u[6] = u1.z; u[7] = u1.w;
        }
        else
        {
#pragma unroll
            for (int q = 0; q < 8; q++) u[q] = (int)(((unsigned)(uint16_t)p[2 * q]) | ((unsigned)(uint16_t)p[2 * q + 1] << 16));
        }
#pragma unroll
        for (int q = 0; q < 8; q++) *(XH_LDS int*)(tile + r * 34 + 16 * g + 2 * q) = u[q];
        wave_sync();
        int d[8];
#pragma unroll
        for (int q = 0; q < 8; q++)
        {
            const unsigned a = (uint16_t)tile[(16 * g + 2 * q) * 34 + r], b = (uint16_t)tile[(16 * g + 2 * q + 1) * 34 + r];
            d[q] = (int)(a | (b << 16));
        }
        wave_sync();                                                                          // the next TU of this wavefront overwrites the tile
        v16i acc;
        idct32_inverse(d, tB1, tA2, acc);
        int16_t* o = dst + (dOff ? (intptr_t)dOff[tu] : (intptr_t)tu * 1024) + (intptr_t)r * ds;
#pragma unroll
        for (int i4 = 0; i4 < 4; i4++)
        {   // columns q(i,g) = 8 * i4 + 4 * g + (0 .. 3): four neighbours per step
            int v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = min(max((acc[4 * i4 + e] + add2) >> shift2, -32768), 32767);
            int16_t* q = o + 8 * i4 + 4 * g;
            if (((uintptr_t)q & 7) == 0) { int2 w; w.x = __builtin_amdgcn_perm(v[1], v[0], 0x05040100); w.y = __builtin_amdgcn_perm(v[3], v[2], 0x05040100); *(int2*)q = w; }
            else { q[0] = (int16_t)v[0]; q[1] = (int16_t)v[1]; q[2] = (int16_t)v[2]; q[3] = (int16_t)v[3]; }
        }
    }
}

// ---- 16x16: two TUs per wavefront pass (xh_dct32.h) ----
__global__ __launch_bounds__(256) void dct16_mfma_kernel(const int16_t* __restrict__ src, intptr_t ss, const int32_t* __restrict__ sOff,
                                                         int16_t* __restrict__ dst, const int32_t* __restrict__ dOff, int n)
{
    const int lane = threadIdx.x & 63, r = lane & 31, g = lane >> 5, half = r >> 4;
    const int pairsTotal = gridDim.x * 4, nPairs = (n + 1) >> 1;
    v4i tB1, tA2;
    dct16_operands(r, g, tB1, tA2);
    for (int pair = blockIdx.x * 4 + (threadIdx.x >> 6); pair < nPairs; pair += pairsTotal)
    {
        const int tu = 2 * pair + half;
        int d[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        if (g == half && tu < n)
        {
            const int16_t* p = src + (sOff ? (intptr_t)sOff[tu] : (intptr_t)tu * 256) + (intptr_t)(r & 15) * ss;
            if (((uintptr_t)p & 15) == 0)
            {
                const int4 u0 = *(const int4*)p, u1 = *(const int4*)(p + 8);
                d[0] = u0.x; d[1] = u0.y; d[2] = u0.z; d[3] = u0.w; d[4] = u1.x; d[5] = u1.y; d[6] = u1.z; d[7] = u1.w;
            }
            else
            {
#pragma unroll
                for (int q = 0; q < 8; q++) d[q] = (int)(((unsigned)(uint16_t)p[2 * q]) | ((unsigned)(uint16_t)p[2 * q + 1] << 16));
            }
        }
        v16i acc;
        dct16_forward(d, tB1, tA2, acc);
        if (tu < n)
        {
            int16_t* o = dst + (dOff ? (intptr_t)dOff[tu] : (intptr_t)tu * 256);
#pragma unroll
            for (int i = 0; i < 16; i++)
            {
                const int k = (i & 3) + 8 * (i >> 2) + 4 * g;
                if ((k >> 4) == half) o[(k & 15) * 16 + (r & 15)] = (int16_t)((acc[i] + (1 << 9)) >> 10);
            }
        }
    }
}
__global__ __launch_bounds__(256) void idct16_mfma_kernel(const int16_t* __restrict__ src, const int32_t* __restrict__ sOff, int16_t* __restrict__ dst, intptr_t ds,
                                                          const int32_t* __restrict__ dOff, int n)
{
    __shared__ __attribute__((aligned(16))) int16_t s_c[4][2][16 * 18];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 31, g = lane >> 5, half = r >> 4;
    const int pairsTotal = gridDim.x * 4, nPairs = (n + 1) >> 1;
    v4i tB1, tA2;
    idct16_operands(r, g, tB1, tA2);
    const int shift2 = 12 - (X265_DEPTH - 8), add2 = 1 << (shift2 - 1);
    XH_LDS int16_t* tile = (XH_LDS int16_t*)s_c[wave][half];
    for (int pair = blockIdx.x * 4 + wave; pair < nPairs; pair += pairsTotal)
    {
        const int tu = 2 * pair + half;
        // the lanes with g == half bring in row (r & 15) of their TU; every lane of that TU then takes its column
        if (g == half && tu < n)
        {
            const int16_t* p = src + (sOff ? (intptr_t)sOff[tu] : (intptr_t)tu * 256) + (r & 15) * 16;
            int u[8];
            if (((uintptr_t)p & 15) == 0)
            {
                const int4 u0 = *(const int4*)p, u1 = *(const int4*)(p + 8);
                u[0] = u0.x; u[1] = u0.y; u[2] = u0.z; u[3] = u0.w; u[4] = u1.x; u[5] = u1.y; u[6] = u1.z; u[7] = u1.w;
            }
            else
            {
#pragma unroll
                for (int q = 0; q < 8; q++) u[q] = (int)(((unsigned)(uint16_t)p[2 * q]) | ((unsigned)(uint16_t)p[2 * q + 1] << 16));
            }
#pragma unroll
            for (int q = 0; q < 8; q++) *(XH_LDS int*)(tile + (r & 15) * 18 + 2 * q) = u[q];
        }
        wave_sync();
        int d[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        if (g == half && tu < n)
        {
#pragma unroll
            for (int q = 0; q < 8; q++)
                d[q] = (int)((unsigned)(uint16_t)tile[(2 * q) * 18 + (r & 15)] | ((unsigned)(uint16_t)tile[(2 * q + 1) * 18 + (r & 15)] << 16));
        }
        wave_sync();
        v16i acc;
        idct16_inverse(d, tB1, tA2, acc);
        if (tu < n)
        {
            int16_t* o = dst + (dOff ? (intptr_t)dOff[tu] : (intptr_t)tu * 256) + (intptr_t)(r & 15) * ds;
#pragma unroll
            for (int i4 = 0; i4 < 4; i4++)
            {   // columns q(i,g) = 8 * i4 + 4 * g + (0 .. 3): the lane's TU owns i4 = 2 * half, 2 * half + 1
                if ((i4 >> 1) != half) continue;
                int v[4];
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = min(max((acc[4 * i4 + e] + add2) >> shift2, -32768), 32767);
                int16_t* q = o + ((8 * i4 + 4 * g) & 15);
                if (((uintptr_t)q & 7) == 0) { int2 w; w.x = __builtin_amdgcn_perm(v[1], v[0], 0x05040100); w.y = __builtin_amdgcn_perm(v[3], v[2], 0x05040100); *(int2*)q = w; }
                else { q[0] = (int16_t)v[0]; q[1] = (int16_t)v[1]; q[2] = (int16_t)v[2]; q[3] = (int16_t)v[3]; }
            }
        }
    }
}

} // namespace

bool xh_dct32_mfma_enabled()
{
    static int on = -1;
    if (on < 0) { const char* e = xh_experiment("X265HIP_DCT32_VALU"); on = (e && e[0] == '1') ? 0 : 1; }
    return on == 1;
}

int xh_dct32_mfma(hipStream_t st, const int16_t* src, intptr_t ss, const int32_t* sOff, int16_t* dst, const int32_t* dOff, int n)
{
    int blocks = (n + 3) / 4;
    if (blocks > 4096) blocks = 4096;       // grid-stride: constant operands are amortised over many TUs
    XH_KLAUNCH(dct32_mfma_kernel, dim3(blocks), dim3(256), 0, st, src, ss, sOff, dst, dOff, n);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

int xh_idct32_mfma(hipStream_t st, const int16_t* src, const int32_t* sOff, int16_t* dst, intptr_t ds, const int32_t* dOff, int n)
{
    int blocks = (n + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    XH_KLAUNCH(idct32_mfma_kernel, dim3(blocks), dim3(256), 0, st, src, sOff, dst, ds, dOff, n);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

int xh_dct16_mfma(hipStream_t st, const int16_t* src, intptr_t ss, const int32_t* sOff, int16_t* dst, const int32_t* dOff, int n)
{
    int blocks = ((n + 1) / 2 + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    XH_KLAUNCH(dct16_mfma_kernel, dim3(blocks), dim3(256), 0, st, src, ss, sOff, dst, dOff, n);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
int xh_idct16_mfma(hipStream_t st, const int16_t* src, const int32_t* sOff, int16_t* dst, intptr_t ds, const int32_t* dOff, int n)
{
    int blocks = ((n + 1) / 2 + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    XH_KLAUNCH(idct16_mfma_kernel, dim3(blocks), dim3(256), 0, st, src, sOff, dst, ds, dOff, n);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
