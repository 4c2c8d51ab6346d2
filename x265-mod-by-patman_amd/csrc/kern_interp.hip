// kern_interp.hip -- batched HEVC interpolation (reference ipfilter.cpp:40-369): 8-tap luma and
// 4-tap chroma FIR in the pp / ps / sp / ss / hv flavours plus pixel->short conversion.
// One 256-thread workgroup per block; the hv path keeps its 14-bit intermediate in LDS
// (the reference's `immed[width * (height + N - 1)]`, ipfilter.cpp:362-369).
#include "xh_common.h"
using namespace xh;

namespace {

__device__ const int8_t k_lumaFilter[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
                                               { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };   // constants.cpp:250-256
__device__ const int8_t k_chromaFilter[8][4] = { { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
                                                 { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };   // :258-268

template<int TAPS> __device__ __forceinline__ void load_taps(int idx, int* c)
{
#pragma unroll
    for (int t = 0; t < TAPS; t++) c[t] = TAPS == 8 ? k_lumaFilter[idx & 3][t] : k_chromaFilter[idx & 7][t];
}
template<int TAPS, class T> __device__ __forceinline__ int fir(const T* p, intptr_t step, const int* c)
{
    int s = 0;
#pragma unroll
    for (int t = 0; t < TAPS; t++) s += (int)p[t * step] * c[t];
    return s;
}
__device__ __forceinline__ pixel to_pixel(int sum, int offset, int shift)
{
    int16_t v = (int16_t)((sum + offset) >> shift);
    return clip_pixel(v);
}

template<int TAPS>
__global__ __launch_bounds__(256) void interp_kernel(int op, int w, int h, const void* __restrict__ srcv, intptr_t ss, const int32_t* __restrict__ sOff,
                                                     void* __restrict__ dstv, intptr_t ds, const int32_t* __restrict__ dOff,
                                                     const int32_t* __restrict__ cIdx, int n)
{
    __shared__ int16_t immed[64 * (64 + 7)];
    const int item = blockIdx.x;
    const intptr_t so = sOff ? sOff[item] : 0, dofs = dOff ? dOff[item] : 0;
    const int ci = cIdx ? cIdx[item] : 0;
    const int idxX = ci & 0xFF, idxY = (ci >> 8) & 0xFF, rowExt = (ci >> 16) & 1;
    constexpr int HALF = TAPS / 2 - 1;
    const int headRoom = XH_IF_INTERNAL_PREC - X265_DEPTH;
    int c[TAPS];

    if (op == X265HIP_IP_HPP || op == X265HIP_IP_VPP)
    {   // ipfilter.cpp:79-118, 164-203
        load_taps<TAPS>(idxX, c);
        const pixel* src = (const pixel*)srcv + so; pixel* dst = (pixel*)dstv + dofs;
        const intptr_t step = op == X265HIP_IP_HPP ? 1 : ss;
        for (int i = threadIdx.x; i < w * h; i += 256)
        {
            int y = i / w, x = i - y * w;
            dst[y * ds + x] = to_pixel(fir<TAPS>(src + y * ss + x - HALF * step, step, c), 32, 6);
        }
    }
    else if (op == X265HIP_IP_HPS || op == X265HIP_IP_VPS)
    {   // ipfilter.cpp:120-162, 205-239
        load_taps<TAPS>(idxX, c);
        const pixel* src = (const pixel*)srcv + so; int16_t* dst = (int16_t*)dstv + dofs;
        const int shift = XH_IF_FILTER_PREC - headRoom, offset = (int)((unsigned)-XH_IF_INTERNAL_OFFS << shift);
        const bool hor = op == X265HIP_IP_HPS;
        const intptr_t step = hor ? 1 : ss;
        int rows = h;
        if (hor && rowExt) { src -= HALF * ss; rows += TAPS - 1; }
        for (int i = threadIdx.x; i < w * rows; i += 256)
        {
            int y = i / w, x = i - y * w;
            dst[y * ds + x] = (int16_t)((fir<TAPS>(src + y * ss + x - HALF * step, step, c) + offset) >> shift);
        }
    }
    else if (op == X265HIP_IP_VSP || op == X265HIP_IP_VSS)
    {   // ipfilter.cpp:241-317
        load_taps<TAPS>(idxX, c);
        const int16_t* src = (const int16_t*)srcv + so;
        const int shift = XH_IF_FILTER_PREC + headRoom, offset = (1 << (shift - 1)) + (XH_IF_INTERNAL_OFFS << XH_IF_FILTER_PREC);
        for (int i = threadIdx.x; i < w * h; i += 256)
        {
            int y = i / w, x = i - y * w;
            int sum = fir<TAPS>(src + (y - HALF) * ss + x, ss, c);
            if (op == X265HIP_IP_VSP) ((pixel*)dstv + dofs)[y * ds + x] = to_pixel(sum, offset, shift);
            else ((int16_t*)dstv + dofs)[y * ds + x] = (int16_t)(sum >> XH_IF_FILTER_PREC);
        }
    }
    else if (op == X265HIP_IP_P2S)
    {   // ipfilter.cpp:40-57
        const pixel* src = (const pixel*)srcv + so; int16_t* dst = (int16_t*)dstv + dofs;
        for (int i = threadIdx.x; i < w * h; i += 256)
        {
            int y = i / w, x = i - y * w;
            int16_t v = (int16_t)((int)src[y * ss + x] << headRoom);
            dst[y * ds + x] = (int16_t)(v - (int16_t)XH_IF_INTERNAL_OFFS);
        }
    }
    else // X265HIP_IP_HVPP: hps with row extension into LDS, then vertical sp (ipfilter.cpp:362-369)
    {
        load_taps<TAPS>(idxX, c);
        const pixel* src = (const pixel*)srcv + so - HALF * ss; pixel* dst = (pixel*)dstv + dofs;
        const int shift1 = XH_IF_FILTER_PREC - headRoom, offset1 = (int)((unsigned)-XH_IF_INTERNAL_OFFS << shift1);
        const int rows = h + TAPS - 1;
        for (int i = threadIdx.x; i < w * rows; i += 256)
        {
            int y = i / w, x = i - y * w;
            immed[i] = (int16_t)((fir<TAPS>(src + y * ss + x - HALF, 1, c) + offset1) >> shift1);
        }
        __syncthreads();
        load_taps<TAPS>(idxY, c);
        const int shift2 = XH_IF_FILTER_PREC + headRoom, offset2 = (1 << (shift2 - 1)) + (XH_IF_INTERNAL_OFFS << XH_IF_FILTER_PREC);
        for (int i = threadIdx.x; i < w * h; i += 256)
        {
            int y = i / w, x = i - y * w;
            dst[y * ds + x] = to_pixel(fir<TAPS>(immed + y * w + x, w, c), offset2, shift2);
        }
    }
}

} // namespace

extern "C" int x265hip_interp_batch(void* stream, int op, int taps, int w, int h, const void* src, intptr_t srcStride, const int32_t* srcOff,
                                    void* dst, intptr_t dstStride, const int32_t* dstOff, const int32_t* coeffIdx, int n)
{
    if (n <= 0) return X265HIP_OK;
    if (op < X265HIP_IP_HPP || op > X265HIP_IP_P2S || (taps != 8 && taps != 4) || w < 2 || h < 2 || w > 64 || h > 64)
    { set_error("interp_batch: bad op/taps/size"); return X265HIP_EARG; }
    hipStream_t st = (hipStream_t)stream;
    if (taps == 8) XH_KLAUNCH(interp_kernel<8>, dim3(n), dim3(256), 0, st, op, w, h, src, srcStride, srcOff, dst, dstStride, dstOff, coeffIdx, n);
    else XH_KLAUNCH(interp_kernel<4>, dim3(n), dim3(256), 0, st, op, w, h, src, srcStride, srcOff, dst, dstStride, dstOff, coeffIdx, n);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
