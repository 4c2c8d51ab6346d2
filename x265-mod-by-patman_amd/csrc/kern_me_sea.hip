// kern_me_sea.hip -- the SEA-search instantiations of me_body.inc (reference motion.cpp:1438-1591) and the SEA integral planes
// (encoder/framefilter.cpp:38-139, 740-833)
#include "me_body.inc"

namespace {

// One thread per position of the padded picture: the six horizontal box widths by successive extension, stored as uint16 (32 pixels
// of at most 11 bits; uint32 in the 12-bit build), then per position the twelve (W, H) boxes as running vertical sums of those rows.
constexpr int SEA_NW = 6;
#if X265_DEPTH <= 11
typedef uint16_t rowsum_t;
#else
typedef uint32_t rowsum_t;
#endif
__device__ const int8_t k_seaWidths[SEA_NW] = { 4, 8, 12, 16, 24, 32 };
__global__ __launch_bounds__(256) void sea_rowsum_kernel(const pixel* __restrict__ pic, intptr_t stride, int rows, int cols, rowsum_t* __restrict__ rs, int64_t rsElems)
{   // pic / rs point at the first padded row and column; cols = stride (row sums near the right end run into the next row exactly like
    // integral_initNh_c's `x < stride - N` bound leaves them undefined -- those columns are never read)
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= cols) return;
    const pixel* p = pic + (intptr_t)y * stride + x;
    const int64_t o = (int64_t)y * stride + x;
    const bool last = y == rows - 1;
    int s = 0, i = 0;
    for (int k = 0; k < SEA_NW; k++)
    {
        const int W = k_seaWidths[k];
        for (; i < W; i++) s += (!last || x + i < cols) ? (int)p[i] : 0;       // stay inside the allocation on the last row
        rs[k * rsElems + o] = (rowsum_t)s;
    }
}
__device__ const int8_t k_seaW[12] = { 5, 5, 5, 4, 3, 3, 3, 2, 1, 1, 0, 0 };      // index into k_seaWidths: 32 32 32 24 16 16 16 12 8 8 4 4
__device__ const int8_t k_seaH[12] = { 32, 24, 8, 32, 16, 12, 4, 16, 32, 8, 16, 4 };
constexpr int SEA_STRIP = 32;          // rows per thread: a vertical sliding window costs (H + 2 * 31) reads per 32 outputs instead of 32 * H
__global__ __launch_bounds__(256) void sea_box_kernel(const rowsum_t* __restrict__ rs, int64_t rsElems, intptr_t stride, int rows, int cols,
                                                      uint32_t* __restrict__ out, int64_t outElems)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * SEA_STRIP, k = blockIdx.z;
    const int H = k_seaH[k];
    if (x >= cols || y0 + H > rows) return;                                        // boxes that leave the padded picture stay undefined
    const rowsum_t* r = rs + k_seaW[k] * rsElems + (int64_t)y0 * stride + x;
    uint32_t* o = out + k * outElems + (int64_t)y0 * stride + x;
    uint32_t s = 0;
    for (int j = 0; j < H; j++) s += r[(intptr_t)j * stride];
    o[0] = s;
    const int n = min(SEA_STRIP, rows - H + 1 - y0);                               // outputs of this strip
    for (int t = 1; t < n; t++)
    {
        s += (uint32_t)r[(intptr_t)(H - 1 + t) * stride] - (uint32_t)r[(intptr_t)(t - 1) * stride];
        o[(intptr_t)t * stride] = s;
    }
}

// the row-granular primitives integral_initNh / integral_initNv themselves (framefilter.cpp:38-139), for the table slots
__global__ __launch_bounds__(256) void integral_h_kernel(uint32_t* __restrict__ sum, const uint32_t* __restrict__ above, const pixel* __restrict__ pix, int W, int n)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= n) return;                               // n = stride - W positions
    uint32_t v = 0;
    for (int i = 0; i < W; i++) v += pix[x + i];
    sum[x] = v + above[x];
}
__global__ __launch_bounds__(256) void integral_v_kernel(uint32_t* __restrict__ top, const uint32_t* __restrict__ below, int n)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x < n) top[x] = below[x] - top[x];
}

} // namespace

extern "C" int x265hip_integral_init_h(void* stream, uint32_t* sum, const uint32_t* above, const void* pix, int boxWidth, int positions)
{
    if (positions <= 0) return X265HIP_OK;
    XH_KLAUNCH(integral_h_kernel, dim3((positions + 255) / 256), dim3(256), 0, (hipStream_t)stream, sum, above, (const pixel*)pix, boxWidth, positions);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
extern "C" int x265hip_integral_init_v(void* stream, uint32_t* top, const uint32_t* below, int positions)
{
    if (positions <= 0) return X265HIP_OK;
    XH_KLAUNCH(integral_v_kernel, dim3((positions + 255) / 256), dim3(256), 0, (hipStream_t)stream, top, below, positions);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

extern "C" size_t x265hip_sea_integral_workspace(intptr_t stride, int rows) { return sizeof(rowsum_t) * (size_t)SEA_NW * (size_t)stride * (size_t)rows; }

extern "C" int x265hip_sea_integral_planes(void* stream, const void* picPadded, intptr_t stride, int rows, uint32_t* planes, int64_t planeElems,
                                           void* workspace, size_t workspaceBytes)
{
    if (!picPadded || !planes || stride < 32 || rows < 32 || planeElems < (int64_t)stride * rows || !workspace ||
        workspaceBytes < x265hip_sea_integral_workspace(stride, rows))
    { set_error("sea_integral_planes: bad arguments"); return X265HIP_EARG; }
    hipStream_t st = (hipStream_t)stream;
    const int64_t rsElems = (int64_t)stride * rows;
    XH_KLAUNCH(sea_rowsum_kernel, dim3((unsigned)((stride + 255) / 256), rows), dim3(256), 0, st, (const pixel*)picPadded, stride, rows, (int)stride,
                       (rowsum_t*)workspace, rsElems);
    XH_LAUNCH_CHECK();
    XH_KLAUNCH(sea_box_kernel, dim3((unsigned)((stride + 255) / 256), (rows + SEA_STRIP - 1) / SEA_STRIP, 12), dim3(256), 0, st, (const rowsum_t*)workspace, rsElems, stride, rows, (int)stride,
                       planes, planeElems);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

int xh_me_sea(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
              const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
              int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
              const void* subpelPlanes, int64_t planeElems, const uint32_t* integral, int64_t integralElems)
{
    return dispatch_me<3>(stream, w, h, curPlane, curStride, refPlane, refStride, tasks, n, costRow, costHalfRange, merange, method, subpelRefine, results, mvpSource,
                          subpelPlanes, planeElems, integral, integralElems);
}
