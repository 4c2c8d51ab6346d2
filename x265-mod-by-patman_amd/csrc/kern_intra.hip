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

bool bad_n(int N) { return N != 4 && N != 8 && N != 16 && N != 32; }

} // namespace

extern "C" int x265hip_intra_filter_batch(void* stream, int N, const void* nb, const int32_t* nbOff, void* filt, const int32_t* filtOff, int n)
{
    if (n <= 0) return X265HIP_OK;
    if (bad_n(N)) { set_error("intra_filter_batch: N must be 4/8/16/32"); return X265HIP_EARG; }
    hipLaunchKernelGGL(intra_filter_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, N, (const pixel*)nb, nbOff, (pixel*)filt, filtOff, n);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
extern "C" int x265hip_intra_pred_batch(void* stream, int N, const void* nb, const int32_t* nbOff, void* dst, intptr_t dstStride,
                                        const int32_t* dstOff, const int32_t* modeFilter, int n)
{
    if (n <= 0) return X265HIP_OK;
    if (bad_n(N) || !modeFilter) { set_error("intra_pred_batch: bad arguments"); return X265HIP_EARG; }
    hipLaunchKernelGGL(intra_pred_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, N, (const pixel*)nb, nbOff, (pixel*)dst, dstStride, dstOff, modeFilter, n);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
extern "C" int x265hip_intra_allangs_batch(void* stream, int N, const void* ref, const int32_t* refOff, const void* filt, const int32_t* filtOff,
                                           void* dst, int bLuma, int n)
{
    if (n <= 0) return X265HIP_OK;
    if (bad_n(N)) { set_error("intra_allangs_batch: N must be 4/8/16/32"); return X265HIP_EARG; }
    hipLaunchKernelGGL(intra_allangs_kernel, dim3(33, n), dim3(256), 0, (hipStream_t)stream, N, (const pixel*)ref, refOff, (const pixel*)filt, filtOff, (pixel*)dst, bLuma, n);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
