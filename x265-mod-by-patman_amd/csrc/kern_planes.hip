// kern_planes.hip -- quarter-pel phase planes of a reference picture.
//
// The reference interpolates a fresh block for every sub-pel candidate of every PU (motion.cpp:1797-1801:
// luma_hpp | luma_vpp | luma_hvpp into a 64x64 stack buffer).  The interpolated value at a position is a pure
// function of the reference pixels around it, so on a device with 288 GB of HBM we compute each of the 15 phases
// ONCE per reference picture -- the idea x265 itself uses for its lookahead's half-pel planes (lowres.h:104-124)
// extended to quarter-pel -- and every sub-pel candidate of the motion search becomes a plain SATD against
// plane[yFrac*4 + xFrac] at an integer offset.  Arithmetic per phase is exactly the primitive's:
//   yFrac == 0: luma_hpp (ipfilter.cpp:79-118)     xFrac == 0: luma_vpp (:164-203)
//   else:       luma_hvpp = hps (row extended, :120-162) -> vsp (:319-369)
// Slot 0 of the output receives a copy of the reference plane, so that the motion search addresses every candidate --
// integer or sub-pel -- as one buffer base plus a 32-bit offset.
// One workgroup produces a 64x16 tile of all 15 planes: source tile + apron in LDS, the three horizontal 14-bit
// intermediates in LDS, then nine vertical passes.  Streaming kernel: 1 plane read, 15 written.
#include "xh_mc.h"
#include "../../include/x265hip_frame.h"
using namespace xh;

namespace {

constexpr int TW = 64, TH = 16, SROWS = TH + 7, SCOLS = 76;          // source tile: rows y0-3 .. y0+TH+3, cols x0-4 .. x0+71
constexpr int SSTRIDE = sizeof(pixel) == 1 ? 76 : 78;               // odd number of dwords per row

__device__ __forceinline__ void store4g(pixel* p, const int* v)      // 4 pixels, p 4-pixel aligned
{
#if X265_DEPTH == 8
    int v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
    asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));      // see xh_mc.h store4(): v_ashr_pk_u8_i32 miscompile
    *(uint32_t*)p = (uint32_t)v0 | ((uint32_t)v1 << 8) | ((uint32_t)v2 << 16) | ((uint32_t)v3 << 24);
#else
    uint2 a; a.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16); a.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
    *(uint2*)p = a;
#endif
}

__global__ __launch_bounds__(256) void subpel_planes_kernel(const pixel* __restrict__ ref, intptr_t stride, int rows,
                                                            pixel* __restrict__ out, int64_t planeElems)
{
    __shared__ __attribute__((aligned(16))) pixel s_src[SROWS * SSTRIDE];
    __shared__ __attribute__((aligned(16))) int16_t s_im[3][SROWS * TW];
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH, t = threadIdx.x;
    const lpixel* src = (const lpixel*)s_src;
    const int headRoom = XH_IF_INTERNAL_PREC - X265_DEPTH;

    // ---- source tile (coordinates clamped into the allocation; border tiles only feed margin pixels) ----
    for (int i = t; i < SROWS * (SCOLS / 4); i += 256)
    {
        const int rr = i / (SCOLS / 4), cq = i - rr * (SCOLS / 4);
        const int gy = min(max(y0 - 3 + rr, 0), rows - 1);
        int v[4];
#pragma unroll
        for (int e = 0; e < 4; e++)
        {
            const int gx = min(max(x0 - 4 + cq * 4 + e, 0), (int)stride - 1);
            v[e] = ref[(intptr_t)gy * stride + gx];
        }
#if X265_DEPTH == 8
        *(lu32*)((lpixel*)s_src + rr * SSTRIDE + cq * 4) = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
#else
        *(lu32*)((lpixel*)s_src + rr * SSTRIDE + cq * 4) = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
        *(lu32*)((lpixel*)s_src + rr * SSTRIDE + cq * 4 + 2) = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
#endif
    }
    __syncthreads();

    // ---- stage 1: horizontal 8-tap for the three fractions; 14-bit intermediates + the yFrac == 0 planes ----
    {
        const int shift1 = XH_IF_FILTER_PREC - headRoom, offset1 = (int)((unsigned)-XH_IF_INTERNAL_OFFS << shift1);
        for (int i = t; i < SROWS * (TW / 4); i += 256)
        {
            const int rr = i >> 4, x4 = (i & 15) * 4;
            int a[4], b[4], c4[4], px[11];
            load4u(src + rr * SSTRIDE + x4, a); load4u(src + rr * SSTRIDE + x4 + 4, b); load4u(src + rr * SSTRIDE + x4 + 8, c4);
            // pixels x-3 .. x+7 for x = x0 + x4: tile columns x4+1 .. x4+11
            px[0] = a[1]; px[1] = a[2]; px[2] = a[3]; px[3] = b[0]; px[4] = b[1]; px[5] = b[2]; px[6] = b[3];
            px[7] = c4[0]; px[8] = c4[1]; px[9] = c4[2]; px[10] = c4[3];
            const int gy = y0 - 3 + rr;
            const bool inTile = rr >= 3 && rr < 3 + TH && gy < rows && x0 + x4 < stride;
#pragma unroll
            for (int xf = 1; xf < 4; xf++)
            {
                int o[4], im[4];
#pragma unroll
                for (int e = 0; e < 4; e++)
                {
                    int s = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) s += px[e + k] * k_lumaTaps[xf][k];
                    o[e] = clip3(0, XH_PIXEL_MAX, (int)(int16_t)((s + 32) >> 6));
                    im[e] = (s + offset1) >> shift1;
                }
                u32x2 pk;
                pk.x = (uint32_t)(uint16_t)im[0] | ((uint32_t)(uint16_t)im[1] << 16);
                pk.y = (uint32_t)(uint16_t)im[2] | ((uint32_t)(uint16_t)im[3] << 16);
                *(lu2*)((lshort*)s_im[xf - 1] + rr * TW + x4) = pk;
                if (inTile) store4g(out + (int64_t)xf * planeElems + (intptr_t)gy * stride + x0 + x4, o);
            }
        }
    }
    __syncthreads();

    // ---- stage 2: vertical 8-tap: xFrac == 0 from the pixels, xFrac != 0 from the intermediates ----
    const int y = t >> 4, x4 = (t & 15) * 4;
    const int gy = y0 + y;
    if (gy >= rows || x0 + x4 >= stride) return;
    pixel* o0 = out + (intptr_t)gy * stride + x0 + x4;
    {
        int col[8][4];
#pragma unroll
        for (int k = 0; k < 8; k++) load4u(src + (y + k) * SSTRIDE + x4 + 4, col[k]);
        store4g(o0, col[3]);                                          // slot 0: the reference plane itself (one base for every candidate)
#pragma unroll
        for (int yf = 1; yf < 4; yf++)
        {
            int o[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
            {
                int s = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) s += col[k][e] * k_lumaTaps[yf][k];
                o[e] = clip3(0, XH_PIXEL_MAX, (int)(int16_t)((s + 32) >> 6));
            }
            store4g(o0 + (int64_t)(yf * 4) * planeElems, o);
        }
    }
    const int shift2 = XH_IF_FILTER_PREC + headRoom, offset2 = (1 << (shift2 - 1)) + (XH_IF_INTERNAL_OFFS << XH_IF_FILTER_PREC);
#pragma unroll
    for (int xf = 1; xf < 4; xf++)
    {
        int col[8][4];
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            u32x2 a = *(const lu2*)((const lshort*)s_im[xf - 1] + (y + k) * TW + x4);
            col[k][0] = (int)(int16_t)(a.x & 0xFFFF); col[k][1] = (int)(int16_t)(a.x >> 16);
            col[k][2] = (int)(int16_t)(a.y & 0xFFFF); col[k][3] = (int)(int16_t)(a.y >> 16);
        }
#pragma unroll
        for (int yf = 1; yf < 4; yf++)
        {
            int o[4];
#pragma unroll
            for (int e = 0; e < 4; e++)
            {
                int s = 0;
#pragma unroll
                for (int k = 0; k < 8; k++) s += col[k][e] * k_lumaTaps[yf][k];
                o[e] = clip3(0, XH_PIXEL_MAX, (int)(int16_t)((s + offset2) >> shift2));
            }
            store4g(o0 + (int64_t)(yf * 4 + xf) * planeElems, o);
        }
    }
}

} // namespace

extern "C" int x265hip_subpel_planes(void* stream, const void* refPlane, intptr_t stride, int rows, void* outPlanes, int64_t planeElems)
{
    if (!refPlane || !outPlanes || stride < 16 || rows < 8 || (stride & 3) || planeElems < (int64_t)stride * rows || (planeElems & 3))
    { set_error("subpel_planes: bad arguments (stride and planeElems must be multiples of 4)"); return X265HIP_EARG; }
    if (((uintptr_t)refPlane | (uintptr_t)outPlanes) & 7) { set_error("subpel_planes: planes must be 8-byte aligned"); return X265HIP_EARG; }
    dim3 grid((unsigned)((stride + TW - 1) / TW), (unsigned)((rows + TH - 1) / TH));
    hipLaunchKernelGGL(subpel_planes_kernel, grid, dim3(256), 0, (hipStream_t)stream,
                       (const pixel*)refPlane, stride, rows, (pixel*)outPlanes, planeElems);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
