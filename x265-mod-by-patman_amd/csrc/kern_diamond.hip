// kern_diamond.hip -- MotionEstimate::diamondSearch (reference encoder/motion.cpp:631-773) for a batch of PUs: the full-pel predictor search of
// ThreadedME's first stage (Search::puMotionEstimation with isMVP, search.cpp:355-363: the CTU and its four sub-CUs, range 32, MVP (0,0)), whose
// results seed m_areaBestMV for the PU searches that follow (analysis.cpp:248-306).
//
// One wavefront per PU.  The source PU is cached in LDS; a step costs four points (or one) against the reference plane, every lane summing its
// share of 4-pixel row pieces, and the decision sequence -- strict `<` in the reference's point order -- runs uniformly in all lanes.
// Kept from the reference: the search starts from (0,0) with bcost = INT_MAX without costing (0,0); the first loop (distances 1, 2, 4) never moves
// its centre; away from the window's edge the points go through COST_MV_X4 (:307-328), which ADDS its arguments to omv although diamondSearch
// hands it absolute coordinates -- in the second loop the measured positions are omv + coordinate, tested against the vertical bounds only.  Such
// positions lie up to twice the search range from the PU: the plane's padding has to cover them, as it has to in the reference.
#include "xh_mc.h"
#include "../../include/x265hip_frame.h"
#include <climits>
using namespace xh;

namespace {

struct DCtx { const lpixel* fenc; const pixel* ref; intptr_t rs; int w, h, qpr, nquads, lane; const uint16_t* centre; int chr, mvpx, mvpy; };

__device__ __forceinline__ int dmvcost(const DCtx& c, int qx, int qy)
{   // bitcost.h:57 (uint16 sum of two row entries); the row is clamped to its extent like every kernel's
    const int dx = min(max(qx - c.mvpx, -c.chr), c.chr), dy = min(max(qy - c.mvpy, -c.chr), c.chr);
    return (uint16_t)(c.centre[dx] + c.centre[dy]);
}
// SAD + MV cost of K full-pel positions
template<int K> __device__ __forceinline__ void cost_points(const DCtx& c, const int (&px)[K], const int (&py)[K], int (&out)[K])
{
    unsigned p[K];
#pragma unroll
    for (int k = 0; k < K; k++) p[k] = 0;
    for (int q = c.lane; q < c.nquads; q += 64)
    {
        const int y = q / c.qpr, x4 = (q - y * c.qpr) * 4;
#pragma unroll
        for (int k = 0; k < K; k++) p[k] = sad4(c.fenc + y * c.w + x4, c.ref + (intptr_t)(py[k] + y) * c.rs + px[k] + x4, p[k]);
    }
#pragma unroll
    for (int k = 0; k < K; k++) out[k] = wsum_u((int)p[k]) + dmvcost(c, px[k] * 4, py[k] * 4);
}

__global__ __launch_bounds__(256) void diamond_kernel(int w, int h, const pixel* __restrict__ cur, intptr_t cs, const pixel* __restrict__ ref, intptr_t rs,
                                                      const x265hip_me_task* __restrict__ tasks, int n, const uint16_t* __restrict__ costCentre, int chr,
                                                      x265hip_me_result* __restrict__ results, int64_t rowStride)
{
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) char, smem)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, item = blockIdx.x * 4 + wave;
    if (item >= n) return;
    lpixel* fenc = (lpixel*)smem + wave * w * h;
    const x265hip_me_task tk = tasks[item];
    DCtx c; c.fenc = fenc; c.ref = ref + tk.refOff; c.rs = rs; c.w = w; c.h = h; c.qpr = w >> 2; c.nquads = (w >> 2) * h; c.lane = lane;
    // rowStride != 0: costCentre is row 0 of a table of MVD cost rows, rowStride entries apart, and the task names its row in mvpFrom (PUs of different qp in one launch:
    // the ThreadedME producer's first stage, xh_tme.cpp)
    c.centre = costCentre + (rowStride ? (int64_t)tk.mvpFrom * rowStride : 0); c.chr = chr; c.mvpx = tk.qmvp[0]; c.mvpy = tk.qmvp[1];
    for (int q = lane; q < c.nquads; q += 64)
    {
        const int y = q / c.qpr, x4 = (q - y * c.qpr) * 4;
        int v[4]; load4u(cur + tk.curOff + (intptr_t)y * cs + x4, v); store4(fenc + y * w + x4, v);
    }
    wave_sync();
    const int mnx = tk.mvmin[0], mny = tk.mvmin[1], mxx = tk.mvmax[0], mxy = tk.mvmax[1];
    int bcost = INT_MAX, bx = 0, by = 0, ox = 0, oy = 0;
    // COST_MV_X4 (:307-328): positions omv + argument, the update guarded by the vertical bounds only
    auto x4 = [&](int a0, int a1, int b0, int b1, int c0, int c1, int d0, int d1) {
        const int X[4] = { ox + a0, ox + b0, ox + c0, ox + d0 }, Y[4] = { oy + a1, oy + b1, oy + c1, oy + d1 };
        int C[4];
        cost_points<4>(c, X, Y, C);
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (Y[k] >= mny && Y[k] <= mxy && C[k] < bcost) { bcost = C[k]; bx = X[k]; by = Y[k]; }
    };
    // COST_MV (:263-269) for the points of an edge step: all of them costed together, updates in the reference's order
    auto edge = [&](const int (&X)[16], const int (&Y)[16], const bool (&on)[16], int npts) {
        for (int base = 0; base < npts; base += 4)
        {
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; k++) any |= (base + k < npts) && on[base + k];
            if (!any) continue;
            int PX[4], PY[4], C[4];
#pragma unroll
            for (int k = 0; k < 4; k++) { const bool ok = (base + k < npts) && on[base + k]; PX[k] = ok ? X[base + k] : 0; PY[k] = ok ? Y[base + k] : 0; }
            cost_points<4>(c, PX, PY, C);
#pragma unroll
            for (int k = 0; k < 4; k++)
                if ((base + k < npts) && on[base + k] && C[k] < bcost) { bcost = C[k]; bx = PX[k]; by = PY[k]; }
        }
    };
    for (int dist = 1; dist <= 4; dist <<= 1)
    {
        const int bx0 = bx, by0 = by;
        const int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        const int top2 = oy - (dist >> 1), bottom2 = oy + (dist >> 1), left2 = ox - (dist >> 1), right2 = ox + (dist >> 1);
        if (top >= mny && left >= mnx && right <= mxx && bottom <= mxy)
        {
            x4(ox, top, ox, bottom, left, oy, right, oy);
            x4(left2, top2, right2, top2, left2, bottom2, right2, bottom2);
        }
        else
        {   // :654-693, in its order
            const int X[16] = { ox, left2, right2, left, right, left2, right2, ox }, Y[16] = { top, top2, top2, oy, oy, bottom2, bottom2, bottom };
            const bool on[16] = { top >= mny, top2 >= mny && left2 >= mnx, top2 >= mny && right2 <= mxx, left >= mnx, right <= mxx,
                                  bottom2 <= mxy && left2 >= mnx, bottom2 <= mxy && right2 <= mxx, bottom <= mxy };
            edge(X, Y, on, 8);
        }
        if (bx == bx0 && by == by0) break;
    }
    ox = bx; oy = by;
    for (int dist = 8; dist <= 64; dist += 8)
    {
        const int bx0 = bx, by0 = by;
        const int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        if (top >= mny && left >= mnx && right <= mxx && bottom <= mxy)
        {
            x4(ox, top, left, oy, right, oy, ox, bottom);
            for (int index = 1; index < 4; index++)
            {
                const int q = (dist >> 2) * index;
                const int posYT = top + q, posYB = bottom - q, posXL = ox - q, posXR = ox + q;
                x4(posXL, posYT, posXR, posYT, posXL, posYB, posXR, posYB);
            }
        }
        else
        {   // :712-764, in its order
            int X[16], Y[16]; bool on[16];
            X[0] = ox; Y[0] = top; on[0] = top >= mny;
            X[1] = left; Y[1] = oy; on[1] = left >= mnx;
            X[2] = right; Y[2] = oy; on[2] = right <= mxx;
            X[3] = ox; Y[3] = bottom; on[3] = bottom <= mxy;
#pragma unroll
            for (int index = 1; index < 4; index++)
            {
                const int q = (dist >> 2) * index;
                const int posYT = top + q, posYB = bottom - q, posXL = ox - q, posXR = ox + q, b = 4 * index;
                X[b] = posXL; Y[b] = posYT; on[b] = posYT >= mny && posXL >= mnx;
                X[b + 1] = posXR; Y[b + 1] = posYT; on[b + 1] = posYT >= mny && posXR <= mxx;
                X[b + 2] = posXL; Y[b + 2] = posYB; on[b + 2] = posYB <= mxy && posXL >= mnx;
                X[b + 3] = posXR; Y[b + 3] = posYB; on[b + 3] = posYB <= mxy && posXR <= mxx;
            }
            edge(X, Y, on, 16);
        }
        if (bx == bx0 && by == by0) break;
        ox = bx; oy = by;
    }
    if (lane == 0)
    {
        x265hip_me_result r;
        r.mv[0] = (int16_t)bx; r.mv[1] = (int16_t)by; r.cost = bcost; r.mvcost = dmvcost(c, bx * 4, by * 4); r.reserved = 0;
        results[item] = r;
    }
}

} // namespace

extern "C" int x265hip_diamond_batch(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                                     const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange, x265hip_me_result* results)
{
    if (n <= 0) return X265HIP_OK;
    if (!curPlane || !refPlane || !tasks || !costRow || !results || w < 4 || h < 4 || w > 64 || h > 64 || (w & 3) || costHalfRange < 1) return X265HIP_EARG;
    const size_t lds = 4 * (size_t)w * h * sizeof(pixel);
    XH_KLAUNCH(diamond_kernel, dim3((n + 3) / 4), dim3(256), lds, (hipStream_t)stream, w, h, (const pixel*)curPlane, curStride, (const pixel*)refPlane, refStride,
                       tasks, n, costRow + costHalfRange, costHalfRange, results, (int64_t)0);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

// the same with an MVD cost row PER TASK: costTable = rows of 2 * costHalfRange + 1 entries, tasks[i].mvpFrom = the row of task i (internal: x265hip_tme_picture)
int xh_diamond_rows(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                    const x265hip_me_task* tasks, int n, const uint16_t* costTable, int costHalfRange, x265hip_me_result* results)
{
    if (n <= 0) return X265HIP_OK;
    if (!curPlane || !refPlane || !tasks || !costTable || !results || w < 4 || h < 4 || w > 64 || h > 64 || (w & 3) || costHalfRange < 1) return X265HIP_EARG;
    const size_t lds = 4 * (size_t)w * h * sizeof(pixel);
    XH_KLAUNCH(diamond_kernel, dim3((n + 3) / 4), dim3(256), lds, (hipStream_t)stream, w, h, (const pixel*)curPlane, curStride, (const pixel*)refPlane, refStride,
                       tasks, n, costTable + costHalfRange, costHalfRange, results, (int64_t)(2 * costHalfRange + 1));
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
