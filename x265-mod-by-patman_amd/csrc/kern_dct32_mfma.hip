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
#include <cstdlib>
using namespace xh;

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int pack_lo(int d0, int d1) { return __builtin_amdgcn_perm(d1, d0, 0x06040200); }   // bytes 0,2 of d0, d1
__device__ __forceinline__ int pack_hi(int d0, int d1) { return __builtin_amdgcn_perm(d1, d0, 0x07050301); }   // bytes 1,3 of d0, d1

// three exact planes of 16 int16 values held as 8 dwords (2 per dword)
__device__ __forceinline__ void split_planes(const int* d, v4i& lo, v4i& hi, v4i& bb)
{
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
        int L = pack_lo(d[2 * q], d[2 * q + 1]);
        lo[q] = L;
        hi[q] = pack_hi(d[2 * q], d[2 * q + 1]);
        bb[q] = (int)(((unsigned)L >> 7) & 0x01010101u);
    }
}

__device__ __forceinline__ v16i mm3(const v4i& lo, const v4i& hi, const v4i& bb, const v4i& t, v16i& accLo)
{
    v16i z = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    return z; (void)lo; (void)hi; (void)bb; (void)t; (void)accLo;
}

template<bool XT_IS_A>
__device__ __forceinline__ void product(const v4i& lo, const v4i& hi, const v4i& bb, const v4i& t, v16i& out)
{
    v16i z = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    v16i aLo, aHb;
    if (XT_IS_A)
    {   // data matrix is the A operand, T the B operand
        aLo = __builtin_amdgcn_mfma_i32_32x32x32_i8(lo, t, z, 0, 0, 0);
        aHb = __builtin_amdgcn_mfma_i32_32x32x32_i8(hi, t, z, 0, 0, 0);
        aHb = __builtin_amdgcn_mfma_i32_32x32x32_i8(bb, t, aHb, 0, 0, 0);
    }
    else
    {
        aLo = __builtin_amdgcn_mfma_i32_32x32x32_i8(t, lo, z, 0, 0, 0);
        aHb = __builtin_amdgcn_mfma_i32_32x32x32_i8(t, hi, z, 0, 0, 0);
        aHb = __builtin_amdgcn_mfma_i32_32x32x32_i8(t, bb, aHb, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[i] = aHb[i] * 256 + aLo[i];
}

__global__ __launch_bounds__(256) void dct32_mfma_kernel(const int16_t* __restrict__ src, intptr_t ss, const int32_t* __restrict__ sOff,
                                                         int16_t* __restrict__ dst, const int32_t* __restrict__ dOff, int n)
{
    const int lane = threadIdx.x & 63, r = lane & 31, g = lane >> 5;
    const int wavesTotal = gridDim.x * 4;
    int tu = blockIdx.x * 4 + (threadIdx.x >> 6);

    // constant operands, built once per wave
    v4i tB1, tA2;   // stage-1 B = T[r][16g + s]; stage-2 A = T[r][n(s,g)]
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
        unsigned b1 = 0, a2 = 0;
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
            int s = 4 * q + e;
            b1 |= ((unsigned)dct_coef(r, 16 * g + s) & 0xFFu) << (8 * e);
            a2 |= ((unsigned)dct_coef(r, (s & 3) + 8 * (s >> 2) + 4 * g) & 0xFFu) << (8 * e);
        }
        tB1[q] = (int)b1; tA2[q] = (int)a2;
    }
    const int shift1 = 4 + X265_DEPTH - 8, add1 = 1 << (shift1 - 1);
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
        v4i lo, hi, bb;
        split_planes(d, lo, hi, bb);
        v16i acc;
        product<true>(lo, hi, bb, tB1, acc);          // D1[n][j], lane: col j = r, rows n(i,g)
        // round stage 1, cast to int16, repack as the stage-2 B operand (k-slot i <-> n(i,g))
        int t16[8];
#pragma unroll
        for (int q = 0; q < 8; q++)
        {
            int v0 = (acc[2 * q] + add1) >> shift1, v1 = (acc[2 * q + 1] + add1) >> shift1;
            t16[q] = __builtin_amdgcn_perm(v1, v0, 0x05040100);   // two int16 per dword
        }
        split_planes(t16, lo, hi, bb);
        product<false>(lo, hi, bb, tA2, acc);         // D2[k][j], lane: col j = r, rows k(i,g)
        int16_t* o = dst + (dOff ? (intptr_t)dOff[tu] : (intptr_t)tu * 1024);
#pragma unroll
        for (int i = 0; i < 16; i++)
        {
            int k = (i & 3) + 8 * (i >> 2) + 4 * g;
            o[k * 32 + r] = (int16_t)((acc[i] + add2) >> shift2);
        }
    }
}

} // namespace

bool xh_dct32_mfma_enabled()
{
    static int on = -1;
    if (on < 0) { const char* e = getenv("X265HIP_DCT32_VALU"); on = (e && e[0] == '1') ? 0 : 1; }
    return on == 1;
}

int xh_dct32_mfma(hipStream_t st, const int16_t* src, intptr_t ss, const int32_t* sOff, int16_t* dst, const int32_t* dOff, int n)
{
    int blocks = (n + 3) / 4;
    if (blocks > 4096) blocks = 4096;       // grid-stride: constant operands are amortised over many TUs
    hipLaunchKernelGGL(dct32_mfma_kernel, dim3(blocks), dim3(256), 0, st, src, ss, sOff, dst, dOff, n);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
