// kern_me_pyr.hip -- the three lower levels of the CU pyramid (32x32 -> 16x16 -> 8x8 2Nx2N searches, STAR) of a 32x32 quadrant in ONE wavefront of ONE launch.
//
// x265hip_batch_step searches the pyramid level by level (reference: the 2Nx2N inter predictions of Analysis::compressInterCU_*, each seeded with its parent CU's MV
// the way Analysis::deriveMVsForCTU seeds PUs from m_areaBestMV, analysis.cpp:248-306; every search = MotionEstimate::motionEstimate, motion.cpp:923-1773).  A launch per
// level streams the 16 phase planes of the whole batch (2.4 GB at 4K 10 bit, F = 8) through the caches once per level, and every launch ends in the tail of its slowest
// searches.  The levels below 64x64 only depend on each other inside a 32x32 quadrant: the 32x32 PU seeds its four 16x16 PUs, each of those its four 8x8 PUs.  Here a
// wavefront owns a quadrant: 1 PU on 64 lanes, then 4 PUs on 16 lanes each, then 2 x 8 PUs on 8 lanes each -- every lane busy at every level, no workgroup barrier, the
// parents' MVs handed down through LDS, and the children's sub-pel candidates land on plane lines their parent touched microseconds ago.
// The searches are me_one of me_body.inc with the template arguments dispatch_me picks for these sizes: same code, same results.
#ifndef XH_LWIN
#define XH_LWIN 0
#endif
#include "me_body.inc"
#include "xh_internal.h"

namespace {

struct PyrArgs
{
    const pixel* cur; intptr_t cs; const pixel* ref; intptr_t rs;
    const x265hip_me_task* tasks[3]; x265hip_me_result* results[3];      // levels 32, 16, 8 (picture-major, raster), the whole batch
    const x265hip_me_result* parent64;                                   // results of the 64x64 level (tasks[0][i].mvpFrom indexes them)
    int g0, ctuPerRow, width;                                            // first global CTU row of the range; CTUs per row; picture width
    const uint16_t* costCentre; int chr, merange, method, subme;
    const pixel* planes; int64_t planeElems;
};

#if X265_DEPTH == 8
constexpr int G32 = 32;
#else
constexpr int G32 = 64;
#endif

template<int STARK, bool WITH32>
__global__ __launch_bounds__(256, 4) void me_pyr_kernel(PyrArgs A)
{
    __shared__ uint16_t s_cost[2 * XH_COST_R + 2];
    __shared__ x265hip_me_result s_r32[4], s_r16[16];
    for (int i = threadIdx.x; i < 2 * XH_COST_R + 1; i += 256) s_cost[i] = A.costCentre[i - XH_COST_R];
    __syncthreads();                                                     // the only workgroup barrier: the MVD cost slice

    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int row = (int)blockIdx.x / A.ctuPerRow, cx = (int)blockIdx.x - row * A.ctuPerRow, g = A.g0 + row;
    const int qy = wv >> 1, qx = wv & 1;                                 // the wavefront's quadrant of the CTU
    const ChromaArgs noChroma{};
    const lu16* lcost = (const lu16*)s_cost;

    if constexpr (WITH32)
    {   // ---- 32x32: the quadrant's PU (16 bit: on the whole wavefront; 8 bit: on its first 32 lanes, as dispatch_me does) ----
        const int nx = A.width >> 5, item = __builtin_amdgcn_readfirstlane(g * (nx * 2) + qy * nx + cx * 2 + qx);       // wave-uniform: the task record and the decisions of a one-PU wavefront sit in scalar registers
        if (G32 == 64 || lane < 32)
            me_one<G32, 1024, 32, 32, 4, false, STARK, true, 32, 32, false>(32, 32, A.cur, A.cs, A.ref, A.rs, A.tasks[0] + item, &s_r32[wv], item, wv * (64 / G32), lcost, A.costCentre, A.chr,
                                                                               A.merange, A.method, A.subme, A.parent64, A.planes, A.planeElems, 0, 0, nullptr, 0, 0, noChroma);
        wave_sync();
        if (lane == 0) A.results[0][item] = s_r32[wv];
    }
    {   // ---- 16x16: its four PUs, 16 lanes each ----
        const int j = lane >> 4, py = qy * 2 + (j >> 1), px = qx * 2 + (j & 1);
        const int nx = A.width >> 4, item = g * (nx * 4) + py * nx + cx * 4 + px, gi = tid >> 4;
        const x265hip_me_task* tp = A.tasks[1] + item;
        // the parent's record is in LDS: me_one reads mvpSource[tp->mvpFrom]
        const x265hip_me_result* par = WITH32 ? (const x265hip_me_result*)&s_r32[wv] - tp->mvpFrom : A.results[0];
        me_one<16, 256, 16, 16, 4, false, STARK, true, 16, 16, false>(16, 16, A.cur, A.cs, A.ref, A.rs, tp, &s_r16[gi], item, gi, lcost, A.costCentre, A.chr,
                                                                         A.merange, A.method, A.subme, par, A.planes, A.planeElems, 0, 0, nullptr, 0, 0, noChroma);
        wave_sync();
        if ((lane & 15) == 0) A.results[1][item] = s_r16[gi];
    }
#pragma unroll 1
    for (int round = 0; round < 2; round++)
    {   // ---- 8x8: sixteen PUs, 8 lanes each, two rows of four per round ----
        const int j = lane >> 3, py = qy * 4 + round * 2 + (j >> 2), px = qx * 4 + (j & 3);
        const int nx = A.width >> 3, item = g * (nx * 8) + py * nx + cx * 8 + px, gi = tid >> 3;
        const x265hip_me_task* tp = A.tasks[2] + item;
        const x265hip_me_result* par = (const x265hip_me_result*)&s_r16[wv * 4 + (((py >> 1) & 1) << 1) + ((px >> 1) & 1)] - tp->mvpFrom;
        me_one<8, 64, 8, 8, 4, false, STARK, true, 8, 8, false>(8, 8, A.cur, A.cs, A.ref, A.rs, tp, A.results[2] + item, item, gi, lcost, A.costCentre, A.chr,
                                                                   A.merange, A.method, A.subme, par, A.planes, A.planeElems, 0, 0, nullptr, 0, 0, noChroma);
    }
}

} // namespace

// Can the three lower levels of this call run fused?  STAR with phase planes addressed by 32-bit byte offsets (the size-specialised kernels' requirement).
bool xh_me_pyr_ok(int method, int64_t planeElems, int costHalfRange)
{
    return method == X265HIP_ME_STAR && (uint64_t)planeElems * 16u * sizeof(pixel) < (1ull << 32) && costHalfRange >= XH_COST_R;
}

int xh_me_pyr(void* stream, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
              const x265hip_me_task* const* tasks /* 32, 16, 8 */, x265hip_me_result* const* results, const x265hip_me_result* parent64,
              int firstCtuRow, int ctuRows, int width, const uint16_t* costRow, int costHalfRange, int merange, int method, int subpelRefine,
              const void* subpelPlanes, int64_t planeElems)
{
    PyrArgs A;
    A.cur = (const pixel*)curPlane; A.cs = curStride; A.ref = (const pixel*)refPlane; A.rs = refStride;
    for (int i = 0; i < 3; i++) { A.tasks[i] = tasks[i]; A.results[i] = results[i]; }
    A.parent64 = parent64; A.g0 = firstCtuRow; A.ctuPerRow = width / 64; A.width = width;
    A.costCentre = costRow + costHalfRange; A.chr = costHalfRange; A.merange = merange; A.method = method; A.subme = subpelRefine;
    A.planes = (const pixel*)subpelPlanes; A.planeElems = planeElems;
    const int n = ctuRows * A.ctuPerRow;
    if (n <= 0) return X265HIP_OK;
    if (parent64) XH_KLAUNCH((me_pyr_kernel<1, true>), dim3(n), dim3(256), 0, (hipStream_t)stream, A);
    else XH_KLAUNCH((me_pyr_kernel<1, false>), dim3(n), dim3(256), 0, (hipStream_t)stream, A);        // the 32x32 level was searched by its own launch: results[0] seeds the 16x16 PUs
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
