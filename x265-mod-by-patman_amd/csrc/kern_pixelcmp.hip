// kern_pixelcmp.hip -- batched SAD / SATD / sa8d / SSE / psy-cost (reference pixel.cpp:40-383,718-749).
// One 64-lane wavefront per block pair; 4 pairs per 256-thread workgroup.  Rounding points follow
// the reference slot map (pixel.cpp:1148-1184): SATD halves per 8x4 (or per 4x4 for widths 4/12),
// sa8d rounds per 8x8 (size 8) or once per 16x16 (sizes >= 16).
#include "xh_common.h"
using namespace xh;

namespace {

__device__ __forceinline__ void had4(int& a, int& b, int& c, int& d)
{
    int t0 = a + b, t1 = a - b, t2 = c + d, t3 = c - d;
    a = t0 + t2; c = t0 - t2; b = t1 + t3; d = t1 - t3;
}

// sum |H4x4(a-b)| (not halved)
template<class T>
__device__ __forceinline__ int had4x4_abs(const T* a, intptr_t sa, const T* b, intptr_t sb)
{
    int d[16];
#pragma unroll
    for (int y = 0; y < 4; y++)
#pragma unroll
        for (int x = 0; x < 4; x++)
            d[y * 4 + x] = (int)a[y * sa + x] - (b ? (int)b[y * sb + x] : 0);
#pragma unroll
    for (int y = 0; y < 4; y++) had4(d[4 * y], d[4 * y + 1], d[4 * y + 2], d[4 * y + 3]);
    int s = 0;
#pragma unroll
    for (int x = 0; x < 4; x++)
    {
        had4(d[x], d[4 + x], d[8 + x], d[12 + x]);
        s += abs(d[x]) + abs(d[4 + x]) + abs(d[8 + x]) + abs(d[12 + x]);
    }
    return s;
}

// sum |H8x8(a-b)| (raw)
template<class T>
__device__ __forceinline__ int had8x8_abs(const T* a, intptr_t sa, const T* b, intptr_t sb)
{
    int d[64];
#pragma unroll
    for (int y = 0; y < 8; y++)
#pragma unroll
        for (int x = 0; x < 8; x++)
            d[y * 8 + x] = (int)a[y * sa + x] - (b ? (int)b[y * sb + x] : 0);
#pragma unroll
    for (int y = 0; y < 8; y++)
    {
        int* r = d + 8 * y;
        had4(r[0], r[1], r[2], r[3]); had4(r[4], r[5], r[6], r[7]);
#pragma unroll
        for (int k = 0; k < 4; k++) { int u = r[k], v = r[k + 4]; r[k] = u + v; r[k + 4] = u - v; }
    }
    int s = 0;
#pragma unroll
    for (int x = 0; x < 8; x++)
    {
        had4(d[x], d[8 + x], d[16 + x], d[24 + x]); had4(d[32 + x], d[40 + x], d[48 + x], d[56 + x]);
#pragma unroll
        for (int k = 0; k < 4; k++) s += abs(d[8 * k + x] + d[8 * (k + 4) + x]) + abs(d[8 * k + x] - d[8 * (k + 4) + x]);
    }
    return s;
}

template<class T>
__device__ __forceinline__ int block_sum8x8(const T* a, intptr_t sa)
{
    int s = 0;
#pragma unroll
    for (int y = 0; y < 8; y++)
#pragma unroll
        for (int x = 0; x < 8; x++) s += (int)a[y * sa + x];
    return s;
}

__global__ __launch_bounds__(256) void pixelcmp_kernel(int op, int w, int h,
    const void* __restrict__ av, intptr_t sa, const int32_t* __restrict__ offA,
    const void* __restrict__ bv, intptr_t sb, const int32_t* __restrict__ offB, int n, void* __restrict__ outv)
{
    __shared__ int lds[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + wave;
    if (item >= n) return;      // whole wave exits together; no block-level barrier is used below

    if (op == X265HIP_CMP_SSE_SS || op == X265HIP_CMP_SSD_S)
    {
        const int16_t* a = (const int16_t*)av + offA[item];
        const int16_t* b = op == X265HIP_CMP_SSE_SS ? (const int16_t*)bv + offB[item] : nullptr;
        unsigned long long s = 0;
        for (int i = lane; i < w * h; i += 64)
        {
            int y = i / w, x = i - y * w;
            int t = (int)a[y * sa + x] - (b ? (int)b[y * sb + x] : 0);
            int32_t sq = (int32_t)((uint32_t)t * (uint32_t)t);
            s += (unsigned long long)(long long)sq;
        }
        s = wave_sum64(s);
        if (lane == 0) ((uint64_t*)outv)[item] = (uint64_t)(sse_t)s;
        return;
    }

    const pixel* a = (const pixel*)av + offA[item];
    const pixel* b = (const pixel*)bv + offB[item];

    if (op == X265HIP_CMP_SAD)
    {
        int s = 0;
        for (int i = lane; i < w * h; i += 64)
        {
            int y = i / w, x = i - y * w;
            s += abs((int)a[y * sa + x] - (int)b[y * sb + x]);
        }
        s = wave_sum(s);
        if (lane == 0) ((int32_t*)outv)[item] = s;
    }
    else if (op == X265HIP_CMP_SSE_PP)
    {
        unsigned long long s = 0;
        for (int i = lane; i < w * h; i += 64)
        {
            int y = i / w, x = i - y * w;
            int t = (int)a[y * sa + x] - (int)b[y * sb + x];
            s += (unsigned long long)(t * t);
        }
        s = wave_sum64(s);
        if (lane == 0) ((uint64_t*)outv)[item] = (uint64_t)(sse_t)s;
    }
    else if (op == X265HIP_CMP_SATD)
    {
        const bool use4 = (w == 4) || (w == 12);
        const int uw = use4 ? 4 : 8;
        const int ux = w / uw, nunits = ux * (h >> 2);
        int s = 0;
        for (int u = lane; u < nunits; u += 64)
        {
            int uy = u / ux, x0 = (u - uy * ux) * uw, y0 = uy * 4;
            const pixel* pa = a + y0 * sa + x0; const pixel* pb = b + y0 * sb + x0;
            int v = had4x4_abs(pa, sa, pb, sb);
            if (!use4) v += had4x4_abs(pa + 4, sa, pb + 4, sb);
            s += v >> 1;
        }
        s = wave_sum(s);
        if (lane == 0) ((int32_t*)outv)[item] = s;
    }
    else if (op == X265HIP_CMP_SA8D)
    {
        int res;
        if (w == 4)
            res = had4x4_abs(a, sa, b, sb) >> 1;       // every lane computes the same tiny block
        else
        {
            const int nb = w >> 3, n8 = nb * nb;
            int raw = 0;
            if (lane < n8)
            {
                int by = lane / nb, bx = lane - by * nb;
                raw = had8x8_abs(a + 8 * by * sa + 8 * bx, sa, b + 8 * by * sb + 8 * bx, sb);
            }
            if (w == 8)
                res = (__shfl(raw, 0, 64) + 2) >> 2;
            else
            {
                lds[wave][lane] = raw;
                wave_sync();
                int ng = nb >> 1, v = 0;
                if (lane < ng * ng)
                {
                    int gy = lane / ng, gx = lane - gy * ng;
                    int i0 = (2 * gy) * nb + 2 * gx;
                    v = (lds[wave][i0] + lds[wave][i0 + 1] + lds[wave][i0 + nb] + lds[wave][i0 + nb + 1] + 2) >> 2;
                }
                res = wave_sum(v);
            }
        }
        if (lane == 0) ((int32_t*)outv)[item] = res;
    }
    else // X265HIP_CMP_PSY_COST
    {
        int res;
        if (w == 4)
        {
            int se = (had4x4_abs(a, sa, (const pixel*)nullptr, 0) >> 1), re = (had4x4_abs(b, sb, (const pixel*)nullptr, 0) >> 1);
            int ss = 0, rs = 0;
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) { ss += a[y * sa + x]; rs += b[y * sb + x]; }
            res = abs((se - (ss >> 2)) - (re - (rs >> 2)));
        }
        else
        {
            const int nb = w >> 3, n8 = nb * nb;
            int v = 0;
            if (lane < n8)
            {
                int by = lane / nb, bx = lane - by * nb;
                const pixel* pa = a + 8 * by * sa + 8 * bx; const pixel* pb = b + 8 * by * sb + 8 * bx;
                int se = ((had8x8_abs(pa, sa, (const pixel*)nullptr, 0) + 2) >> 2) - (block_sum8x8(pa, sa) >> 2);
                int re = ((had8x8_abs(pb, sb, (const pixel*)nullptr, 0) + 2) >> 2) - (block_sum8x8(pb, sb) >> 2);
                v = abs(se - re);
            }
            res = wave_sum(v);
        }
        if (lane == 0) ((int32_t*)outv)[item] = res;
    }
}

} // namespace


namespace {
// pu[].ads (pixel.cpp:121-165): one wavefront scans the row of candidate x positions 64 at a time; survivors are appended
// in ascending order through a ballot prefix count.
__global__ __launch_bounds__(64) void ads_kernel(int parts, int half, const int32_t* __restrict__ encDC, const uint32_t* __restrict__ sums, int delta,
                                                 const uint16_t* __restrict__ costMvX, int16_t* __restrict__ mvs, int width, int thresh, int32_t* __restrict__ nmvOut)
{
    const int lane = threadIdx.x;
    int nmv = 0;
    const long long e0 = encDC[0], e1 = parts > 1 ? encDC[1] : 0, e2 = parts > 2 ? encDC[2] : 0, e3 = parts > 2 ? encDC[3] : 0;
    for (int base = 0; base < width; base += 64)
    {
        const int i = base + lane;
        bool keep = false;
        if (i < width)
        {
            long long a = llabs(e0 - (long long)sums[i]);
            if (parts == 2) a += llabs(e1 - (long long)sums[i + delta]);
            if (parts == 4) a += llabs(e1 - (long long)sums[i + half]) + llabs(e2 - (long long)sums[i + delta]) + llabs(e3 - (long long)sums[i + delta + half]);
            keep = (int)(a + costMvX[i]) < thresh;
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        if (keep) mvs[nmv + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (int16_t)i;
        nmv += __builtin_popcountll(m);
    }
    if (lane == 0) *nmvOut = nmv;
}
} // namespace
extern "C" int x265hip_ads(void* stream, int parts, int lx, const int32_t* encDC, const uint32_t* sums, int delta,
                           const uint16_t* costMvX, int16_t* mvs, int width, int thresh, int32_t* nmv)
{
    if ((parts != 1 && parts != 2 && parts != 4) || width < 0 || !encDC || !sums || !costMvX || !mvs || !nmv) { set_error("ads: bad arguments"); return X265HIP_EARG; }
    XH_KLAUNCH(ads_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, parts, lx >> 1, encDC, sums, delta, costMvX, mvs, width, thresh, nmv);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

extern "C" int x265hip_pixelcmp_batch(void* stream, int op, int w, int h,
                                      const void* a, intptr_t strideA, const int32_t* offA,
                                      const void* b, intptr_t strideB, const int32_t* offB, int n, void* out)
{
    if (n <= 0) return X265HIP_OK;
    if (op < X265HIP_CMP_SAD || op > X265HIP_CMP_SSD_S || w < 4 || h < 4 || w > 64 || h > 64 || ((w | h) & 3))
    { set_error("pixelcmp_batch: bad op/size %d %dx%d", op, w, h); return X265HIP_EARG; }
    if ((op == X265HIP_CMP_SA8D || op == X265HIP_CMP_PSY_COST) && (w != h || (w & (w - 1))))
    { set_error("pixelcmp_batch: sa8d/psy need square power-of-two blocks"); return X265HIP_EARG; }
    XH_KLAUNCH(pixelcmp_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       op, w, h, a, strideA, offA, b, strideB, offB, n, out);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
