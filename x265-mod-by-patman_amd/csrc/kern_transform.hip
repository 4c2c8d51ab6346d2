// kern_transform.hip -- batched forward/inverse integer transforms and the quant family
// (reference dct.cpp:43-81 DST, :83-440 butterflies == exact matrix products, :443-611 wrappers,
// :614-757 quant/dequant/count/denoise).
//
// Generic path: every TU is staged in LDS; stage 1 and stage 2 are direct matrix products with the
// reference's two rounding points ((x + add) >> shift, int16 cast; inverse clips to int16).
// The 32x32 forward transform additionally has an MFMA path (kern_dct32_mfma.hip).
#include "xh_common.h"
#include "xh_internal.h"
using namespace xh;

namespace {

// DST-VII basis (what fastForwardDst / inversedst factor, dct.cpp:43-81)
__device__ const int8_t k_dst4[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };

template<int N> struct Log2 { static const int v = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5; };

// TPB TUs per 256-thread block, EPT elements per thread
template<int N, int OP>
__global__ __launch_bounds__(256) void transform_kernel(const int16_t* __restrict__ src, intptr_t ss, const int32_t* __restrict__ sOff,
                                                        int16_t* __restrict__ dst, intptr_t ds, const int32_t* __restrict__ dOff, int n)
{
    constexpr int NN = N * N;
    constexpr int TPB = NN >= 256 ? 1 : 256 / NN;
    constexpr int EPT = NN >= 256 ? NN / 256 : 1;
    constexpr bool FWD = (OP == X265HIP_TR_DCT || OP == X265HIP_TR_DST4);
    constexpr bool DST = (OP == X265HIP_TR_DST4 || OP == X265HIP_TR_IDST4);
    __shared__ int16_t sA[TPB][NN];
    __shared__ int16_t sB[TPB][NN];
    __shared__ int8_t sM[NN];

    for (int i = threadIdx.x; i < NN; i += 256)
        sM[i] = DST ? k_dst4[i] : (int8_t)dct_coef((i / N) * (32 / N), i % N);

    const int tuLocal = (EPT == 1) ? threadIdx.x / NN : 0;
    const int tu = blockIdx.x * TPB + tuLocal;
    const bool active = tu < n;
    const intptr_t so = active ? (sOff ? (intptr_t)sOff[tu] : (intptr_t)tu * NN) : 0;
    // load: forward reads a strided residual block; inverse reads a dense coefficient block
#pragma unroll
    for (int e = 0; e < EPT; e++)
    {
        int i = (EPT == 1) ? threadIdx.x % NN : threadIdx.x + e * 256;
        int r = i / N, c = i % N;
        if (active) sA[tuLocal][i] = FWD ? src[so + r * ss + c] : src[so + i];
    }
    __syncthreads();

    const int shift1 = FWD ? (DST ? 1 : Log2<N>::v - 1) + X265_DEPTH - 8 : 7;
    const int shift2 = FWD ? (DST ? 8 : Log2<N>::v + 6) : 12 - (X265_DEPTH - 8);
#pragma unroll
    for (int stage = 0; stage < 2; stage++)
    {
        const int shift = stage ? shift2 : shift1, add = 1 << (shift - 1);
        const int16_t* in = stage ? sB[tuLocal] : sA[tuLocal];
        int16_t* outp = stage ? sA[tuLocal] : sB[tuLocal];
#pragma unroll
        for (int e = 0; e < EPT; e++)
        {
            int i = (EPT == 1) ? threadIdx.x % NN : threadIdx.x + e * 256;
            int p = i / N, q = i % N;          // forward: out[k=p][j=q]; inverse: out[j=p][k=q]
            int s = 0;
            if (FWD)
            {
#pragma unroll
                for (int m = 0; m < N; m++) s += (int)sM[p * N + m] * (int)in[q * N + m];
                outp[i] = (int16_t)((s + add) >> shift);
            }
            else
            {
#pragma unroll
                for (int m = 0; m < N; m++) s += (int)sM[m * N + q] * (int)in[m * N + p];
                outp[i] = clip16((s + add) >> shift);
            }
        }
        __syncthreads();
    }
    // store: forward writes dense, inverse writes a strided block
    if (active)
    {
        const intptr_t dofs = dOff ? dOff[tu] : (intptr_t)tu * NN;
#pragma unroll
        for (int e = 0; e < EPT; e++)
        {
            int i = (EPT == 1) ? threadIdx.x % NN : threadIdx.x + e * 256;
            int r = i / N, c = i % N;
            if (FWD) dst[dofs + i] = sA[tuLocal][i];
            else dst[dofs + r * ds + c] = sA[tuLocal][i];
        }
    }
}


// lowPassDct8/16/32_c (lowpassdct.cpp:34-116): 2x2 means of the residual (int16 arithmetic like the reference), forward DCT
// of the half-size block, embedded top-left in a zeroed full-size block, DC replaced by the scaled block sum.
// One workgroup per TU; H = half size (4, 8, 16).
template<int H>
__global__ __launch_bounds__(256) void lowpass_kernel(const int16_t* __restrict__ src, intptr_t ss, const int32_t* __restrict__ sOff,
                                                      int16_t* __restrict__ dst, const int32_t* __restrict__ dOff, int n)
{
    constexpr int N = 2 * H, HH = H * H;
    __shared__ int16_t sA[HH], sB[HH];
    __shared__ int8_t sM[HH];
    __shared__ int sSum;
    const int tu = blockIdx.x, t = threadIdx.x;
    if (tu >= n) return;
    const intptr_t so = sOff ? (intptr_t)sOff[tu] : (intptr_t)tu * N * N;
    const intptr_t dofs = dOff ? (intptr_t)dOff[tu] : (intptr_t)tu * N * N;
    if (t == 0) sSum = 0;
    for (int i = t; i < HH; i += 256) sM[i] = (int8_t)dct_coef((i / H) * (32 / H), i % H);
    __syncthreads();
    for (int i = t; i < HH; i += 256)
    {
        const int r = i / H, c = i % H;
        const int16_t* p = src + so + (2 * r) * ss + 2 * c;
        const int16_t sum = (int16_t)(p[0] + p[1] + p[ss] + p[ss + 1]);
        sA[i] = (int16_t)(sum >> 2);
        atomicAdd(&sSum, (int)sum);
    }
    __syncthreads();
    constexpr int LG = H == 4 ? 2 : H == 8 ? 3 : 4;
    const int shift1 = LG - 1 + X265_DEPTH - 8, shift2 = LG + 6;
#pragma unroll
    for (int stage = 0; stage < 2; stage++)
    {
        const int shift = stage ? shift2 : shift1, add = 1 << (shift - 1);
        const int16_t* in = stage ? sB : sA;
        int16_t* outp = stage ? sA : sB;
        for (int i = t; i < HH; i += 256)
        {
            const int pp = i / H, q = i % H;
            int s = 0;
#pragma unroll
            for (int m = 0; m < H; m++) s += (int)sM[pp * H + m] * (int)in[q * H + m];
            outp[i] = (int16_t)((s + add) >> shift);
        }
        __syncthreads();
    }
    for (int i = t; i < N * N; i += 256)
    {
        const int r = i / N, c = i % N;
        int16_t v = (r < H && c < H) ? sA[r * H + c] : (int16_t)0;
        if (i == 0)
        {
            const int total = sSum;
            if (N == 8) v = X265_DEPTH == 8 ? (int16_t)((int16_t)total << 1) : (int16_t)((int16_t)total >> (X265_DEPTH - 9 > 0 ? X265_DEPTH - 9 : 0));   // int16 block sum (:37)
            else v = (int16_t)(total >> ((N == 16 ? 1 : 3) + (X265_DEPTH - 8)));
        }
        dst[dofs + i] = v;
    }
}

template<int N, int OP> int launch_tr(hipStream_t st, const int16_t* src, intptr_t ss, const int32_t* sOff,
                                      int16_t* dst, intptr_t ds, const int32_t* dOff, int n)
{
    constexpr int TPB = N * N >= 256 ? 1 : 256 / (N * N);
    XH_KLAUNCH((transform_kernel<N, OP>), dim3((n + TPB - 1) / TPB), dim3(256), 0, st, src, ss, sOff, dst, ds, dOff, n);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

// ---------------- quant family ----------------
// segment sum over aligned groups of G lanes (G = 16 or 64)
template<bool WITH_DELTA, bool ABS_OUT>
__global__ __launch_bounds__(256) void quant_kernel(const int16_t* __restrict__ coef, const int32_t* __restrict__ qc, int32_t* __restrict__ deltaU,
                                                    int16_t* __restrict__ qCoef, int qBits, int add, int numCoeff, int total, uint32_t* __restrict__ numSig)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    int nz = 0;
    if (i < total)
    {
        // dct.cpp:666-715 (int32 wrap-around arithmetic like the reference's `int`)
        int level = coef[i];
        const int sign = level < 0 ? -1 : 1;
        const int32_t tmplevel = (int32_t)((uint32_t)abs(level) * (uint32_t)qc[i % numCoeff]);
        level = (int32_t)((uint32_t)tmplevel + (uint32_t)add) >> qBits;
        if (WITH_DELTA && deltaU) deltaU[i] = (int32_t)((uint32_t)tmplevel - ((uint32_t)level << qBits)) >> (qBits - 8);
        nz = level != 0;
        level *= sign;
        level = clip3(-32768, 32767, level);
        qCoef[i] = ABS_OUT ? (int16_t)abs(level) : (int16_t)level;
    }
    if (!numSig) return;
    if (numCoeff >= 64)
    {   // whole wave inside one block (numCoeff is a multiple of 64): one atomic per wave
        unsigned long long m = __ballot(nz);
        if ((threadIdx.x & 63) == 0 && i < total) atomicAdd(&numSig[i / numCoeff], (uint32_t)__popcll(m));
    }
    else
    {   // 4x4 blocks: 16 lanes per block, exclusive owner -> plain store
        int v = nz;
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
        if ((threadIdx.x & 15) == 0 && i < total) numSig[i / numCoeff] = (uint32_t)v;
    }
}

__global__ __launch_bounds__(256) void dequant_kernel(const int16_t* __restrict__ q, const int32_t* __restrict__ deq, int16_t* __restrict__ coef,
                                                      int numCoeff, int total, int scaleOrPer, int shift)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    if (!deq)
    {   // dct.cpp:614-636 dequant_normal
        const int add = 1 << (shift - 1);
        coef[i] = clip16((int32_t)((uint32_t)((int)q[i] * scaleOrPer) + (uint32_t)add) >> shift);
    }
    else
    {   // dct.cpp:638-664 dequant_scaling
        const int per = scaleOrPer, sh = shift + 4;
        const int d = deq[i % numCoeff];
        if (sh > per)
        {
            const int add = 1 << (sh - per - 1);
            coef[i] = clip16((int32_t)((uint32_t)((int)q[i] * d) + (uint32_t)add) >> (sh - per));
        }
        else
        {
            int c = clip3(-32768, 32767, (int)q[i] * d);
            coef[i] = clip16((int32_t)((uint32_t)c * (1u << (per - sh))));
        }
    }
}

// count_nonzero (dct.cpp:716-728) / copy_count (:730-744): one wave per block
__global__ __launch_bounds__(256) void count_kernel(int N, const int16_t* __restrict__ src, intptr_t ss, int16_t* __restrict__ copyTo, int n, uint32_t* __restrict__ out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + wave;
    if (item >= n) return;
    const int16_t* s = src + (intptr_t)item * (copyTo ? 0 : N * N);
    int c = 0;
    for (int i = lane; i < N * N; i += 64)
    {
        int r = i / N, col = i - r * N;
        int16_t v = copyTo ? s[r * ss + col] : s[i];
        if (copyTo) copyTo[i] = v;
        c += v != 0;
    }
    c = wave_sum(c);
    if (lane == 0) out[item] = (uint32_t)c;
}

// denoiseDct (dct.cpp:746-757)
__global__ __launch_bounds__(256) void denoise_kernel(int16_t* __restrict__ coef, uint32_t* __restrict__ resSum, const uint16_t* __restrict__ offset, int num)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= num) return;
    int level = coef[i];
    const int sign = level >> 31;
    level = (level + sign) ^ sign;
    resSum[i] += (uint32_t)level;
    level -= offset[i];
    coef[i] = (int16_t)(level < 0 ? 0 : (level ^ sign) - sign);
}

} // namespace

extern "C" int x265hip_transform_batch(void* stream, int op, int N, const int16_t* src, intptr_t srcStride, const int32_t* srcOff,
                                       int16_t* dst, intptr_t dstStride, const int32_t* dstOff, int n)
{
    if (n <= 0) return X265HIP_OK;
    hipStream_t st = (hipStream_t)stream;
    if (op == X265HIP_TR_DST4 || op == X265HIP_TR_IDST4)
    {
        if (N != 4) { set_error("DST exists for 4x4 only"); return X265HIP_EARG; }
        return op == X265HIP_TR_DST4 ? launch_tr<4, X265HIP_TR_DST4>(st, src, srcStride, srcOff, dst, dstStride, dstOff, n)
                                     : launch_tr<4, X265HIP_TR_IDST4>(st, src, srcStride, srcOff, dst, dstStride, dstOff, n);
    }
    if (op == X265HIP_TR_LOWPASS)
    {
        if (N == 8) XH_KLAUNCH(lowpass_kernel<4>, dim3(n), dim3(256), 0, st, src, srcStride, srcOff, dst, dstOff, n);
        else if (N == 16) XH_KLAUNCH(lowpass_kernel<8>, dim3(n), dim3(256), 0, st, src, srcStride, srcOff, dst, dstOff, n);
        else if (N == 32) XH_KLAUNCH(lowpass_kernel<16>, dim3(n), dim3(256), 0, st, src, srcStride, srcOff, dst, dstOff, n);
        else { set_error("lowpass dct exists for 8, 16 and 32"); return X265HIP_EARG; }
        XH_LAUNCH_CHECK();
        return X265HIP_OK;
    }
    if (op != X265HIP_TR_DCT && op != X265HIP_TR_IDCT) { set_error("transform_batch: unknown op %d", op); return X265HIP_EARG; }
    const bool f = op == X265HIP_TR_DCT;
    switch (N)
    {
    case 4: return f ? launch_tr<4, X265HIP_TR_DCT>(st, src, srcStride, srcOff, dst, dstStride, dstOff, n) : launch_tr<4, X265HIP_TR_IDCT>(st, src, srcStride, srcOff, dst, dstStride, dstOff, n);
    case 8: return f ? launch_tr<8, X265HIP_TR_DCT>(st, src, srcStride, srcOff, dst, dstStride, dstOff, n) : launch_tr<8, X265HIP_TR_IDCT>(st, src, srcStride, srcOff, dst, dstStride, dstOff, n);
    case 16:
        if (xh_dct32_mfma_enabled()) return f ? xh_dct16_mfma(st, src, srcStride, srcOff, dst, dstOff, n) : xh_idct16_mfma(st, src, srcOff, dst, dstStride, dstOff, n);
        return f ? launch_tr<16, X265HIP_TR_DCT>(st, src, srcStride, srcOff, dst, dstStride, dstOff, n) : launch_tr<16, X265HIP_TR_IDCT>(st, src, srcStride, srcOff, dst, dstStride, dstOff, n);
    case 32:
        if (f && xh_dct32_mfma_enabled()) return xh_dct32_mfma(st, src, srcStride, srcOff, dst, dstOff, n);
        if (!f && xh_dct32_mfma_enabled()) return xh_idct32_mfma(st, src, srcOff, dst, dstStride, dstOff, n);
        return f ? launch_tr<32, X265HIP_TR_DCT>(st, src, srcStride, srcOff, dst, dstStride, dstOff, n) : launch_tr<32, X265HIP_TR_IDCT>(st, src, srcStride, srcOff, dst, dstStride, dstOff, n);
    }
    set_error("transform_batch: N must be 4, 8, 16 or 32");
    return X265HIP_EARG;
}

// VALU/LDS reference path of the 32x32 forward transform, kept callable for A/B tests against the MFMA path
int xh_dct32_valu(hipStream_t st, const int16_t* src, intptr_t ss, const int32_t* sOff, int16_t* dst, const int32_t* dOff, int n)
{ return launch_tr<32, X265HIP_TR_DCT>(st, src, ss, sOff, dst, 32, dOff, n); }

static int quant_common(bool nq, void* stream, const int16_t* coef, const int32_t* qc, int32_t* deltaU, int16_t* qCoef,
                        int qBits, int add, int numCoeff, int n, uint32_t* numSig)
{
    if (n <= 0) return X265HIP_OK;
    if (numCoeff < 16 || (numCoeff & 15) || qBits < 8 || (numCoeff > 16 && (numCoeff & 63))) { set_error("quant_batch: bad numCoeff/qBits"); return X265HIP_EARG; }
    hipStream_t st = (hipStream_t)stream;
    const int total = numCoeff * n;
    if (numSig && numCoeff >= 64) XH_HIP(hipMemsetAsync(numSig, 0, sizeof(uint32_t) * n, st));
    if (nq) XH_KLAUNCH((quant_kernel<false, true>), dim3((total + 255) / 256), dim3(256), 0, st, coef, qc, deltaU, qCoef, qBits, add, numCoeff, total, numSig);
    else XH_KLAUNCH((quant_kernel<true, false>), dim3((total + 255) / 256), dim3(256), 0, st, coef, qc, deltaU, qCoef, qBits, add, numCoeff, total, numSig);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
extern "C" int x265hip_quant_batch(void* stream, const int16_t* coef, const int32_t* qc, int32_t* deltaU, int16_t* qCoef,
                                   int qBits, int add, int numCoeff, int n, uint32_t* numSig)
{ return quant_common(false, stream, coef, qc, deltaU, qCoef, qBits, add, numCoeff, n, numSig); }
extern "C" int x265hip_nquant_batch(void* stream, const int16_t* coef, const int32_t* qc, int16_t* qCoef,
                                    int qBits, int add, int numCoeff, int n, uint32_t* numSig)
{ return quant_common(true, stream, coef, qc, nullptr, qCoef, qBits, add, numCoeff, n, numSig); }
extern "C" int x265hip_dequant_normal_batch(void* stream, const int16_t* q, int16_t* coef, int num, int scale, int shift)
{
    if (num <= 0) return X265HIP_OK;
    if (shift < 1) { set_error("dequant_normal: shift must be >= 1"); return X265HIP_EARG; }
    XH_KLAUNCH(dequant_kernel, dim3((num + 255) / 256), dim3(256), 0, (hipStream_t)stream, q, (const int32_t*)nullptr, coef, num, num, scale, shift);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
extern "C" int x265hip_dequant_scaling_batch(void* stream, const int16_t* q, const int32_t* deq, int16_t* coef, int numCoeff, int n, int per, int shift)
{
    if (n <= 0) return X265HIP_OK;
    if (!deq) { set_error("dequant_scaling: NULL table"); return X265HIP_EARG; }
    const int total = numCoeff * n;
    XH_KLAUNCH(dequant_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, q, deq, coef, numCoeff, total, per, shift);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
int xh_count_nonzero(hipStream_t st, int N, const int16_t* q, int n, uint32_t* out)
{
    XH_KLAUNCH(count_kernel, dim3((n + 3) / 4), dim3(256), 0, st, N, q, (intptr_t)N, (int16_t*)nullptr, n, out);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
int xh_copy_count(hipStream_t st, int N, const int16_t* resi, intptr_t rs, int16_t* coef, uint32_t* out)
{
    XH_KLAUNCH(count_kernel, dim3(1), dim3(256), 0, st, N, resi, rs, coef, 1, out);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
int xh_denoise(hipStream_t st, int16_t* coef, uint32_t* resSum, const uint16_t* offset, int num)
{
    XH_KLAUNCH(denoise_kernel, dim3((num + 255) / 256), dim3(256), 0, st, coef, resSum, offset, num);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
