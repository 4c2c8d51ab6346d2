// xh_dct32.h -- exact 32x32x32 integer products on the gfx950 matrix cores (see kern_dct32_mfma.hip for the
// derivation): byte-plane split of int16 operands and the two constant T operands of the forward DCT.
#pragma once
#include "xh_common.h"

namespace xh {

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


// constant operands for lane (r = lane & 31, g = lane >> 5):
//   tB1: stage-1 B = T[r][16g + s]           (k-slot s = 0..15)
//   tA2: stage-2 A = T[r][n(s,g)],  n(s,g) = (s & 3) + 8 * (s >> 2) + 4 * g   (the C/D row order of stage 1)
__device__ __forceinline__ void dct32_operands(int r, int g, v4i& tB1, v4i& tA2)
{
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
}

// Forward 32x32 DCT of one TU held as 8 dwords per lane (row r, columns 16g..16g+15, two int16 per dword).
// Result: acc[i] = UNROUNDED stage-2 sum for coefficient (k = (i & 3) + 8 * (i >> 2) + 4 * g, j = r);
// the caller applies (acc + (1 << 10)) >> 11 (dct.cpp:511-526 shift_2nd = 11).
__device__ __forceinline__ void dct32_forward(const int* d, const v4i& tB1, const v4i& tA2, v16i& acc)
{
    const int shift1 = 4 + X265_DEPTH - 8, add1 = 1 << (shift1 - 1);
    v4i lo, hi, bb;
    split_planes(d, lo, hi, bb);
    product<true>(lo, hi, bb, tB1, acc);
    int t16[8];
#pragma unroll
    for (int q = 0; q < 8; q++)
    {
        int v0 = (acc[2 * q] + add1) >> shift1, v1 = (acc[2 * q + 1] + add1) >> shift1;
        t16[q] = __builtin_amdgcn_perm(v1, v0, 0x05040100);
    }
    split_planes(t16, lo, hi, bb);
    product<false>(lo, hi, bb, tA2, acc);
}

// ---- inverse 32x32 DCT (dct.cpp:242-416, 579-594): X = T^T C T with the reference's two rounding points --------------------------------------------------
// stage 1  D1T[j][k] = sum_m C[m][j] T[m][k]  (the reference's first pass, stored transposed there too), rounded (v + 64) >> 7 and clipped to int16:
//          A = C^T (lane: row j = r, k-slots m = 16g .. 16g+15 -> the lane needs a COLUMN of the coefficient block: one trip through LDS), B = T (lane: col k = r,
//          k-slot m -> T[16g + s][r]).  C/D: lane holds col k = r, rows j(i,g) = (i & 3) + 8 * (i >> 2) + 4 * g.
// stage 2  X[p][q] = sum_j D1T'[j][p] T[j][q]: the rounded stage-1 registers are the B operand as they stand (col p = r, k-slot i <-> j(i,g)); A = T^T with its
//          k-slots in the same order (lane: row q = r, k-slot s -> T[j(s,g)][r]).  C/D: lane holds col p = r, rows q(i,g): row r of the residual block.
__device__ __forceinline__ void idct32_operands(int r, int g, v4i& tB1, v4i& tA2)
{
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
        unsigned b1 = 0, a2 = 0;
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
            int s = 4 * q + e;
            b1 |= ((unsigned)dct_coef(16 * g + s, r) & 0xFFu) << (8 * e);
            a2 |= ((unsigned)dct_coef((s & 3) + 8 * (s >> 2) + 4 * g, r) & 0xFFu) << (8 * e);
        }
        tB1[q] = (int)b1; tA2[q] = (int)a2;
    }
}
// d: the lane's column of the coefficient block, C[16g + s][r] for s = 0 .. 15 (two per dword).  Result: acc[i] = UNROUNDED stage-2 sum of X[r][q(i,g)];
// the caller applies clip16((acc + add2) >> shift2), shift2 = 12 - (X265_DEPTH - 8).
__device__ __forceinline__ void idct32_inverse(const int* d, const v4i& tB1, const v4i& tA2, v16i& acc)
{
    v4i lo, hi, bb;
    split_planes(d, lo, hi, bb);
    product<true>(lo, hi, bb, tB1, acc);
    int t16[8];
#pragma unroll
    for (int q = 0; q < 8; q++)
    {
        const int v0 = min(max((acc[2 * q] + 64) >> 7, -32768), 32767), v1 = min(max((acc[2 * q + 1] + 64) >> 7, -32768), 32767);
        t16[q] = __builtin_amdgcn_perm(v1, v0, 0x05040100);
    }
    split_planes(t16, lo, hi, bb);
    product<false>(lo, hi, bb, tA2, acc);
}

// ---- 16x16 transforms on the same instruction: TWO TUs per 32x32x32 product, as the diagonal blocks of block-diagonal operands (rows / columns 0..15 = the first
// TU, 16..31 = the second; the off-diagonal blocks are exact zeros in, exact zeros out -- rounding keeps them zero).  A quarter of the multiplier array does useful work;
// the kernels are bound by memory either way.  T16[k][m] = dct_coef(2k, m).
// lane (r, g): `mine` = the half of the k-slots (16g .. 16g+15) that belongs to the lane's TU (r >> 4).
__device__ __forceinline__ void dct16_operands(int r, int g, v4i& tB1, v4i& tA2)
{
    const bool mine = g == (r >> 4);
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
        unsigned b1 = 0, a2 = 0;
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
            const int s = 4 * q + e, n = (s & 3) + 8 * (s >> 2) + 4 * g;
            if (mine) b1 |= ((unsigned)dct_coef(2 * (r & 15), s) & 0xFFu) << (8 * e);                        // stage-1 B = T16^T: col j = r, k-slot m = s
            if ((n >> 4) == (r >> 4)) a2 |= ((unsigned)dct_coef(2 * (r & 15), n & 15) & 0xFFu) << (8 * e);    // stage-2 A = T16: row k = r, k-slot n(s,g)
        }
        tB1[q] = (int)b1; tA2[q] = (int)a2;
    }
}
__device__ __forceinline__ void idct16_operands(int r, int g, v4i& tB1, v4i& tA2)
{
    const bool mine = g == (r >> 4);
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
        unsigned b1 = 0, a2 = 0;
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
            const int s = 4 * q + e, j = (s & 3) + 8 * (s >> 2) + 4 * g;
            if (mine) b1 |= ((unsigned)dct_coef(2 * s, r & 15) & 0xFFu) << (8 * e);                            // stage-1 B = T16: col k = r, k-slot m = s
            if ((j >> 4) == (r >> 4)) a2 |= ((unsigned)dct_coef(2 * (j & 15), r & 15) & 0xFFu) << (8 * e);    // stage-2 A = T16^T: row q = r, k-slot j(s,g)
        }
        tB1[q] = (int)b1; tA2[q] = (int)a2;
    }
}
// d: forward -- row (r & 15) of the lane's TU (zeros on the lanes whose k-slots are not the TU's); acc[i] = unrounded stage-2 sum of coefficient (k(i,g) & 15, r & 15)
// of TU r >> 4, valid where k(i,g) >> 4 == r >> 4; the caller applies (acc + (1 << 9)) >> 10 (dct.cpp:528-543).
__device__ __forceinline__ void dct16_forward(const int* d, const v4i& tB1, const v4i& tA2, v16i& acc)
{
    const int shift1 = 3 + X265_DEPTH - 8, add1 = 1 << (shift1 - 1);
    v4i lo, hi, bb;
    split_planes(d, lo, hi, bb);
    product<true>(lo, hi, bb, tB1, acc);
    int t16[8];
#pragma unroll
    for (int q = 0; q < 8; q++)
    {
        const int v0 = (acc[2 * q] + add1) >> shift1, v1 = (acc[2 * q + 1] + add1) >> shift1;
        t16[q] = __builtin_amdgcn_perm(v1, v0, 0x05040100);
    }
    split_planes(t16, lo, hi, bb);
    product<false>(lo, hi, bb, tA2, acc);
}
// d: inverse -- column (r & 15) of the lane's TU's coefficient block (zeros on the other lanes); acc[i] = unrounded stage-2 sum of residual (r & 15, q(i,g) & 15)
__device__ __forceinline__ void idct16_inverse(const int* d, const v4i& tB1, const v4i& tA2, v16i& acc)
{
    idct32_inverse(d, tB1, tA2, acc);                  // the same two products and the same rounding / clipping between them
}

} // namespace xh
