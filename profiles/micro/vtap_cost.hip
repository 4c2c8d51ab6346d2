// vtap_cost.hip -- what the round-3 review's "4 planes + the vertical 8-tap inside the search kernel" costs per sub-pel candidate, measured on the instruction side alone
// (all data resident in L1 / L2: no HBM term), for the 8x8 PU at 10 bit on 8 lanes, the layout of the size-specialised kernels (me_body.inc: lane = (row r = l >> 1,
// column half l & 1), two 8-byte units per lane, rows r and r + 4):
//   form A (today): the candidate is a block of a pre-interpolated phase plane -- per lane 2 aligned 12-byte loads + 2 x 2 funnel shifts -> SATD (8 lanes, DPP butterflies);
//   form B (the proposal): the candidate is a block of a horizontally filtered int16 plane (interp_horiz_ps, ipfilter.cpp:120-162); the lane filters its two output rows
//           vertically (filterVertical_sp, ipfilter.cpp:205-262: 8 taps, (sum + 512 + (8192 << 6)) >> 10, clip) from the 12 input rows r - 3 .. r + 8 of its column -- per
//           lane 12 aligned 12-byte loads + 12 x 2 funnel shifts + 2 rows x 4 pixels x 8 multiply-adds + round / clip / pack -> the same SATD;
//   form C: B with the rows shared inside the DPP quad of a column half (every lane loads 4 of the 15 rows its four lanes need and fetches the others with DPP moves).
// Build: hipcc --offload-arch=gfx950 -O3 -save-temps vtap_cost.hip -o vtap_cost   (the .s shows the instruction mix; the program times the three forms)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3a4 __attribute__((aligned(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
struct unit { uint32_t x, y; };                                 // 4 pixels / 4 int16 intermediates

__device__ __forceinline__ unit ld_unit(const char* base, uint32_t byteOff)
{   // 8 bytes at any 2-byte offset: the aligned 12 bytes around them + two funnel shifts (me_body.inc ldq_a)
    const uint32_t m = byteOff & 3u;
    const u32x3 w = *(const u32x3a4*)(base + (byteOff - m));
    unit u; u.x = __builtin_amdgcn_alignbyte(w.y, w.x, m); u.y = __builtin_amdgcn_alignbyte(w.z, w.y, m);
    return u;
}
__device__ __forceinline__ s16x2 pk(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
#define DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)
// SATD of the 8x8 on its 8 lanes (the XH_SATD8 form of me_body.inc, condensed: same instruction classes and counts)
__device__ __forceinline__ int satd8(const unit (&src)[2], const unit (&ref)[2], int lane)
{
    const int r = lane >> 1;
    const s16x2 sg1 = (r & 1) ? (s16x2)(-1) : (s16x2)(1), sg2 = (r & 2) ? (s16x2)(-1) : (s16x2)(1);
    int tot = 0;
#pragma unroll
    for (int i = 0; i < 2; i++)
    {
        // even / odd pixel split of the 4-pixel unit, differences in packed 16-bit lanes
        const uint32_t ae = __builtin_amdgcn_perm(src[i].y, src[i].x, 0x05040100), ao = __builtin_amdgcn_perm(src[i].y, src[i].x, 0x07060302);
        const uint32_t be = __builtin_amdgcn_perm(ref[i].y, ref[i].x, 0x05040100), bo = __builtin_amdgcn_perm(ref[i].y, ref[i].x, 0x07060302);
        s16x2 d0 = pk(ae) - pk(be), d1 = pk(ao) - pk(bo);
        d0 = pk((uint32_t)DPP(__builtin_bit_cast(int, d0), 0x4E)) + d0 * sg1; d1 = pk((uint32_t)DPP(__builtin_bit_cast(int, d1), 0x4E)) + d1 * sg1;
        d0 = pk((uint32_t)DPP(__builtin_bit_cast(int, d0), 0x141)) + d0 * sg2; d1 = pk((uint32_t)DPP(__builtin_bit_cast(int, d1), 0x141)) + d1 * sg2;
        const s16x2 s = d0 + d1, t = d0 - d1;
        const s16x2 as = __builtin_elementwise_max(s, -s), at = __builtin_elementwise_max(t, -t);
        const s16x2 acc = as + at;
        int v = 2 * ((int)acc.x + (int)acc.y);
        v += DPP(v, 0x4E); v += DPP(v, 0xB1); v += DPP(v, 0x141);
        tot += v >> 1;
    }
    return tot;
}
// filterVertical_sp of one output row (4 pixels) from 8 input rows of int16 units; coefficient row `c` of g_lumaFilter
__device__ __forceinline__ unit vfilter(const unit (&in)[12], int first, const int (&c)[8])
{
    int s0 = 512 + (8192 << 6), s1 = s0, s2 = s0, s3 = s0;
#pragma unroll
    for (int k = 0; k < 8; k++)
    {
        const uint32_t a = in[first + k].x, b = in[first + k].y;
        s0 += c[k] * (int)(short)(a & 0xFFFF); s1 += c[k] * (int)(short)(a >> 16);
        s2 += c[k] * (int)(short)(b & 0xFFFF); s3 += c[k] * (int)(short)(b >> 16);
    }
    auto fin = [](int v) { v >>= 10; return (uint32_t)min(max(v, 0), 1023); };
    unit o; o.x = fin(s0) | (fin(s1) << 16); o.y = fin(s2) | (fin(s3) << 16);
    return o;
}

template<int FORM> __global__ __launch_bounds__(256) void cost_kernel(const char* plane, uint32_t strideB, const uint32_t* cand, int ncand, int* out, int yfrac)
{
    const int lane = threadIdx.x & 7, grp = (blockIdx.x * 256 + threadIdx.x) >> 3;
    const int r = lane >> 1, half = lane & 1;
    const int coef[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
    int c[8];
#pragma unroll
    for (int k = 0; k < 8; k++) c[k] = coef[yfrac & 3][k];
    const uint32_t org = (uint32_t)(8 + (grp & 63) * 16) * strideB + 64;                  // the PU's origin inside the (small, cache-resident) plane
    unit src[2];
    src[0] = ld_unit(plane, org + (uint32_t)r * strideB + half * 8); src[1] = ld_unit(plane, org + (uint32_t)(r + 4) * strideB + half * 8);
    int best = 1 << 30;
    for (int k = 0; k < ncand; k++)
    {
        const uint32_t e = org + cand[k];                                                   // byte offset of the candidate's top-left pixel (any 2-byte alignment)
        unit ref[2];
        if (FORM == 0)
        {
            ref[0] = ld_unit(plane, e + (uint32_t)r * strideB + half * 8); ref[1] = ld_unit(plane, e + (uint32_t)(r + 4) * strideB + half * 8);
        }
        else if (FORM == 1)
        {
            unit in[12];
#pragma unroll
            for (int j = 0; j < 12; j++) in[j] = ld_unit(plane, e + (uint32_t)(r - 3 + j) * strideB + half * 8);
            ref[0] = vfilter(in, 0, c); ref[1] = vfilter(in, 4, c);
        }
        else
        {   // the four lanes of a column half (lanes l, l ^ 2, l ^ 4, l ^ 6: rows r = 0 .. 3) need input rows -3 .. 11; lane r loads rows r - 3 + 4 j (j = 0 .. 3) and
            // takes row r - 3 + k (k = 4 j + i, i = 1 .. 3) from lane (r + i) & 3 -- its j-th load, or its (j + 1)-th when r + i wrapped
            unit mine[4];
#pragma unroll
            for (int j = 0; j < 4; j++) mine[j] = ld_unit(plane, e + (uint32_t)(r - 3 + 4 * j) * strideB + half * 8);
            unit in[12];
#pragma unroll
            for (int k2 = 0; k2 < 12; k2++)
            {
                const int j = k2 >> 2, i = k2 & 3;
                if (i == 0) { in[k2] = mine[j]; continue; }
                const int srcLane = (lane & 1) | ((((r + i) & 3)) << 1) | (lane & ~7);
                const bool wrap = r + i > 3;
                const unit a = mine[j], b = mine[j < 3 ? j + 1 : 3];
                unit t;
                t.x = (uint32_t)__builtin_amdgcn_ds_bpermute(srcLane << 2, (int)(wrap ? b.x : a.x)); t.y = (uint32_t)__builtin_amdgcn_ds_bpermute(srcLane << 2, (int)(wrap ? b.y : a.y));
                in[k2] = t;
            }
            ref[0] = vfilter(in, 0, c); ref[1] = vfilter(in, 4, c);
        }
        const int cost = satd8(src, ref, lane);
        best = min(best, cost + k);
    }
    if (lane == 0) out[grp] = best;
}

template<int FORM> float time_form(const char* plane, uint32_t strideB, const uint32_t* cand, int ncand, int* out, int blocks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(cost_kernel<FORM>, dim3(blocks), dim3(256), 0, 0, plane, strideB, cand, ncand, out, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL(cost_kernel<FORM>, dim3(blocks), dim3(256), 0, 0, plane, strideB, cand, ncand, out, 2);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    const uint32_t strideB = 512; const int rows = 64 * 16 + 64, ncand = 256, blocks = 256 * 16;      // 64 PU origins, a 0.5 MB plane: every load hits L1 / L2
    std::vector<uint16_t> h((size_t)strideB / 2 * rows);
    uint32_t s = 12345; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (uint16_t)((s >> 20) & 1023); }
    std::vector<uint32_t> hc((size_t)ncand);
    for (int k = 0; k < ncand; k++) { s = s * 1664525u + 1013904223u; hc[(size_t)k] = ((s >> 8) % 5) * strideB + ((s >> 16) % 9) * 2; }      // +0 .. 4 rows, +0 .. 8 pixels
    char* d; uint32_t* dc; int* dout;
    hipMalloc(&d, h.size() * 2); hipMalloc(&dc, hc.size() * 4); hipMalloc(&dout, (size_t)blocks * 32 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    const double cands = (double)blocks * 32 * ncand;                                   // PU-candidates per launch
    const float a = time_form<0>(d, strideB, dc, ncand, dout, blocks), b = time_form<1>(d, strideB, dc, ncand, dout, blocks), c = time_form<2>(d, strideB, dc, ncand, dout, blocks);
    printf("8x8 PU, 10 bit, 8 lanes, cache-resident data, %d blocks x 256 threads, %d candidates per PU\n", blocks, ncand);
    printf("A phase-plane block            : %.3f ms  %.2f ns per PU-candidate  (x1.00)\n", a, a * 1e6 / cands);
    printf("B vertical 8-tap in the lane   : %.3f ms  %.2f ns per PU-candidate  (x%.2f)\n", b, b * 1e6 / cands, b / a);
    printf("C B with rows shared in a quad : %.3f ms  %.2f ns per PU-candidate  (x%.2f)\n", c, c * 1e6 / cands, c / a);
    return 0;
}
