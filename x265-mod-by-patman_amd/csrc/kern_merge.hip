// kern_merge.hip -- the per-PU choice among references and the bidirectional candidate: the tail of Search::puMotionEstimation / predInterSearch
// (reference encoder/search.cpp:258-556) after the per-reference searches of x265hip_me_batch, for every 2Nx2N PU of a batch in one launch:
//   bits and cost of each (list, reference):  listSelBits + MVP_IDX_BITS + getTUBits(ref) + BitCost::bitcost(mv - mvp); (satd - mvcost) + RDCost::getCost(bits)
//   best reference per list (strict `<` in reference order),
//   bidirectional candidate (B slices): predInterLumaPixel of both bests -> pixelavg_pp -> SATD (search.cpp:436-446), and the same with both MVs zero
//   (:449-502), final choice (:504-555) -> a record like MEData (encoder/threadedme.h:122-130).
// One wavefront per PU: the decision arithmetic is uniform (every lane computes it), the two averaged predictions are built from the references'
// phase planes (a prediction at any quarter-pel MV is a block of plane 4 * yFrac + xFrac) into LDS and compared with the cached source PU at SATD.
#include "xh_bidir.h"
#include "../../include/x265hip_frame.h"
#include <cmath>
using namespace xh;

namespace {

struct MergeArgs
{
    int w, h, n, isP, bidir, sourceMaxDim, numRef[2];
    const pixel* cur; intptr_t cs; intptr_t rs;
    const x265hip_me_task* tasks;
    const x265hip_me_result* res[2 * X265HIP_MAX_REF]; const x265hip_me_result* mvpSrc[2 * X265HIP_MAX_REF];      // [list * X265HIP_MAX_REF + ref]
    const pixel* planes[2 * X265HIP_MAX_REF]; int64_t planeElems;                                // 16-slot phase-plane buffer of each reference
    const float* bitsCentre; int bitsHalf; unsigned long long lambda;
    x265hip_inter_choice* out;
};

__device__ __forceinline__ uint32_t bitcost(const MergeArgs& a, int mvx, int mvy, int px, int py)
{
    const int dx = min(max(mvx - px, -a.bitsHalf), a.bitsHalf), dy = min(max(mvy - py, -a.bitsHalf), a.bitsHalf);
    return (uint32_t)(a.bitsCentre[dx] + a.bitsCentre[dy] + 0.5f);               // bitcost.h:66-70 (float sum, truncation)
}
__device__ __forceinline__ uint32_t getcost(const MergeArgs& a, uint32_t bits) { return (uint32_t)(((unsigned long long)bits * a.lambda + 128) >> 8); }   // rdcost.h:164-169

__device__ int bidir_satd(const MergeArgs& a, const lpixel* fenc, lpixel* avg, int refOff, const pixel* pa, int ax, int ay, const pixel* pb, int bx, int by, int lane)
{
    return bidir_satd_core(a.w, a.h, a.rs, a.planeElems, fenc, avg, refOff, pa, ax, ay, pb, bx, by, lane);
}

// P slices / no bidirectional candidate: the choice is arithmetic on the search results only -- one THREAD per PU
__global__ __launch_bounds__(256) void inter_merge_uni_kernel(MergeArgs a)
{
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= a.n) return;
    const x265hip_me_task* tp = a.tasks + item;
    const uint32_t listSelBits[2] = { a.isP ? 1u : 3u, 3u };
    const int qx = tp->qmvp[0], qy = tp->qmvp[1], from = tp->mvpFrom;
    uint32_t bcost[2] = { 0xFFFFFFFFu, 0xFFFFFFFFu }, bbits[2] = { 0, 0 }, bmvc[2] = { 0, 0 };
    int bmvx[2] = { 0, 0 }, bmvy[2] = { 0, 0 }, bpx[2] = { 0, 0 }, bpy[2] = { 0, 0 }, bref[2] = { -1, -1 };
#pragma unroll
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < a.numRef[l]; r++)
        {
            const int k = l * X265HIP_MAX_REF + r;
            const x265hip_me_result m = a.res[k][item];
            int px = qx, py = qy;
            if (from >= 0 && a.mvpSrc[k]) { px = a.mvpSrc[k][from].mv[0]; py = a.mvpSrc[k][from].mv[1]; }
            const uint32_t bits = listSelBits[l] + 1u + (uint32_t)(r + (r < a.numRef[l] - 1)) + bitcost(a, m.mv[0], m.mv[1], px, py);
            const uint32_t c = (uint32_t)(m.cost - m.mvcost) + getcost(a, bits);
            if (c < bcost[l]) { bcost[l] = c; bbits[l] = bits; bmvc[l] = (uint32_t)m.mvcost; bref[l] = r; bmvx[l] = m.mv[0]; bmvy[l] = m.mv[1]; bpx[l] = px; bpy[l] = py; }
        }
    const int l = bcost[0] <= bcost[1] ? 0 : 1;                                  // search.cpp:530-555
    x265hip_inter_choice o;
    o.mv[0][0] = o.mv[0][1] = o.mv[1][0] = o.mv[1][1] = 0; o.mvp[0][0] = o.mvp[0][1] = o.mvp[1][0] = o.mvp[1][1] = 0;
    o.mvCost[0] = o.mvCost[1] = 0; o.ref[0] = o.ref[1] = -1; o.reserved = 0;
    o.mv[l][0] = (int16_t)bmvx[l]; o.mv[l][1] = (int16_t)bmvy[l]; o.mvp[l][0] = (int16_t)bpx[l]; o.mvp[l][1] = (int16_t)bpy[l];
    o.mvCost[l] = bmvc[l]; o.ref[l] = (int8_t)bref[l]; o.bits = (int32_t)bbits[l]; o.cost = bcost[l];
    a.out[item] = o;
}

__global__ __launch_bounds__(256) void inter_merge_kernel(MergeArgs a)
{
    __shared__ __attribute__((aligned(16))) pixel s_fenc[4][64 * 64];
    __shared__ __attribute__((aligned(16))) pixel s_avg[4][64 * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.n) return;
    const x265hip_me_task* tp = a.tasks + item;
    const uint32_t listSelBits[3] = { a.isP ? 1u : 3u, 3u, 5u };                // getBlkBits, SIZE_2Nx2N (search.cpp:4896-4901)
    struct Best { int mvx, mvy, px, py, ref; uint32_t cost, bits, mvCost; } best[2];
    best[0].cost = best[1].cost = 0xFFFFFFFFu; best[0].ref = best[1].ref = -1;
#pragma unroll
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < a.numRef[l]; r++)
        {
            const int k = l * X265HIP_MAX_REF + r;
            const x265hip_me_result m = a.res[k][item];
            int px = tp->qmvp[0], py = tp->qmvp[1];
            if (tp->mvpFrom >= 0 && a.mvpSrc[k]) { px = a.mvpSrc[k][tp->mvpFrom].mv[0]; py = a.mvpSrc[k][tp->mvpFrom].mv[1]; }
            uint32_t bits = listSelBits[l] + 1u + (uint32_t)(r + (r < a.numRef[l] - 1));      // MVP_IDX_BITS, getTUBits (search.h:252-255)
            bits += bitcost(a, m.mv[0], m.mv[1], px, py);
            const uint32_t c = (uint32_t)(m.cost - m.mvcost) + getcost(a, bits);               // search.cpp:372-374
            if (c < best[l].cost) { best[l].cost = c; best[l].bits = bits; best[l].mvCost = (uint32_t)m.mvcost; best[l].ref = r; best[l].mvx = m.mv[0]; best[l].mvy = m.mv[1]; best[l].px = px; best[l].py = py; }
        }
    uint32_t bidirCost = 0xFFFFFFFFu; int bidirBits = 0, b0x = 0, b0y = 0, b1x = 0, b1y = 0;
    if (!a.isP && a.bidir && best[0].cost != 0xFFFFFFFFu && best[1].cost != 0xFFFFFFFFu)
    {   // search.cpp:420-503
        lpixel* fenc = (lpixel*)s_fenc[wave]; lpixel* avg = (lpixel*)s_avg[wave];
        const int qpr = a.w >> 2, nquads = qpr * a.h;
        for (int q = lane; q < nquads; q += 64)
        {
            const int y = q / qpr, x4 = (q - y * qpr) * 4;
            int v[4]; load4u(a.cur + tp->curOff + (intptr_t)y * a.cs + x4, v); store4(fenc + y * a.w + x4, v);
        }
        wave_sync();
        b0x = best[0].mvx; b0y = best[0].mvy; b1x = best[1].mvx; b1y = best[1].mvy;
        const pixel* pa = a.planes[best[0].ref]; const pixel* pb = a.planes[X265HIP_MAX_REF + best[1].ref];
        int satd = bidir_satd(a, fenc, avg, tp->refOff, pa, b0x, b0y, pb, b1x, b1y, lane);
        bidirBits = (int)(best[0].bits + best[1].bits + listSelBits[2] - (listSelBits[0] + listSelBits[1]));
        bidirCost = (uint32_t)satd + getcost(a, (uint32_t)bidirBits);
        bool tryZero = (b0x | b0y | b1x | b1y) != 0;
        if (tryZero)
        {   // setSearchRange(cu, mvzero, max(sourceWidth, sourceHeight)) (search.cpp:4969-5021), mvmax.y += 2, << 2: both MVPs inside
            const int cx0 = tp->mvmin[0], cy0 = tp->mvmin[1], cx1 = tp->mvmax[0], cy1 = tp->mvmax[1], d = a.sourceMaxDim << 2;
            int mnx = min(max(-d, cx0), cx1) >> 2, mny = min(max(-d, cy0), cy1) >> 2, mxx = min(max(d, cx0), cx1) >> 2, mxy = min(max(d, cy0), cy1) >> 2;
            mxy = max(mxy, mny) + 2;
            mnx <<= 2; mny <<= 2; mxx <<= 2; mxy <<= 2;
#pragma unroll
            for (int l = 0; l < 2; l++) tryZero &= best[l].px >= mnx && best[l].px <= mxx && best[l].py >= mny && best[l].py <= mxy;
        }
        if (tryZero)
        {
            satd = bidir_satd(a, fenc, avg, tp->refOff, pa, 0, 0, pb, 0, 0, lane);
            const uint32_t bits0 = best[0].bits - bitcost(a, best[0].mvx, best[0].mvy, best[0].px, best[0].py) + bitcost(a, 0, 0, best[0].px, best[0].py);
            const uint32_t bits1 = best[1].bits - bitcost(a, best[1].mvx, best[1].mvy, best[1].px, best[1].py) + bitcost(a, 0, 0, best[1].px, best[1].py);
            const uint32_t c = (uint32_t)satd + getcost(a, bits0) + getcost(a, bits1);
            if (c < bidirCost) { b0x = b0y = b1x = b1y = 0; bidirCost = c; bidirBits = (int)(bits0 + bits1 + listSelBits[2] - (listSelBits[0] + listSelBits[1])); }
        }
    }
    if (lane == 0)
    {   // search.cpp:504-555
        x265hip_inter_choice o;
        o.mv[0][0] = o.mv[0][1] = o.mv[1][0] = o.mv[1][1] = 0; o.mvp[0][0] = o.mvp[0][1] = o.mvp[1][0] = o.mvp[1][1] = 0;
        o.mvCost[0] = o.mvCost[1] = 0; o.ref[0] = o.ref[1] = -1; o.reserved = 0;
        if (bidirCost < best[0].cost && bidirCost < best[1].cost)
        {
            o.mv[0][0] = (int16_t)b0x; o.mv[0][1] = (int16_t)b0y; o.mv[1][0] = (int16_t)b1x; o.mv[1][1] = (int16_t)b1y;
#pragma unroll
            for (int l = 0; l < 2; l++) { o.mvp[l][0] = (int16_t)best[l].px; o.mvp[l][1] = (int16_t)best[l].py; o.mvCost[l] = best[l].mvCost; o.ref[l] = (int8_t)best[l].ref; }
            o.bits = bidirBits; o.cost = bidirCost;
        }
        else
        {
            const int l = best[0].cost <= best[1].cost ? 0 : 1;
            o.mv[l][0] = (int16_t)best[l].mvx; o.mv[l][1] = (int16_t)best[l].mvy; o.mvp[l][0] = (int16_t)best[l].px; o.mvp[l][1] = (int16_t)best[l].py;
            o.mvCost[l] = best[l].mvCost; o.ref[l] = (int8_t)best[l].ref; o.bits = (int32_t)best[l].bits; o.cost = best[l].cost;
        }
        a.out[item] = o;
    }
}

// the bidirectional candidate's distortion alone (search.cpp:436-446) for a batch of PUs: predInterLumaPixel of both references at the given MVs -> pixelavg_pp -> SATD
__global__ __launch_bounds__(256) void bidir_satd_kernel(MergeArgs a, const x265hip_bidir_task* __restrict__ bt, int32_t* __restrict__ satd, const int8_t* __restrict__ ref0, const int8_t* __restrict__ ref1)
{
    __shared__ __attribute__((aligned(16))) pixel s_fenc[4][64 * 64];
    __shared__ __attribute__((aligned(16))) pixel s_avg[4][64 * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, item = blockIdx.x * 4 + wave;
    if (item >= a.n) return;
    const x265hip_bidir_task t = bt[item];
    lpixel* fenc = (lpixel*)s_fenc[wave]; lpixel* avg = (lpixel*)s_avg[wave];
    const int qpr = a.w >> 2, nquads = qpr * a.h;
    for (int q = lane; q < nquads; q += 64)
    {
        const int y = q / qpr, x4 = (q - y * qpr) * 4;
        int v[4]; load4u(a.cur + t.curOff + (intptr_t)y * a.cs + x4, v); store4(fenc + y * a.w + x4, v);
    }
    wave_sync();
    // per-task references (ref0 / ref1 given): the planes of the list-0 / list-1 reference this PU chose
    const pixel* pa = a.planes[ref0 ? uni((int)ref0[item]) & (X265HIP_MAX_REF - 1) : 0]; const pixel* pb = a.planes[X265HIP_MAX_REF + (ref1 ? uni((int)ref1[item]) & (X265HIP_MAX_REF - 1) : 0)];
    const int s = bidir_satd(a, fenc, avg, t.refOff, pa, t.mv0[0], t.mv0[1], pb, t.mv1[0], t.mv1[1], lane);
    if (lane == 0) satd[item] = s;
}

} // namespace

extern "C" int x265hip_mvbits_row(int halfRange, float* out)
{   // BitCost::CalculateLogs (bitcost.cpp:72-86): s_bitsizes, out[halfRange + d]; pure host code
    if (halfRange < 0 || !out) { set_error("mvbits_row: bad arguments"); return X265HIP_EARG; }
    const float log2_2 = (float)(2.0f / std::log((double)2.0f));
    for (int i = 0; i <= halfRange; i++)
        out[halfRange + i] = out[halfRange - i] = i ? (float)(std::log((double)(float)(i + 1)) * log2_2 + 1.718f) : 0.718f;
    return X265HIP_OK;
}
extern "C" uint64_t x265hip_rd_lambda(int qp)
{   // RDCost::setLambda (rdcost.h:88-92) on x265_lambda_tab[qp] (constants.cpp:28-116)
    const double lambda = std::floor(std::pow(2.0, (double)qp / 6.0 - 2.0) * (double)(1 << (X265_DEPTH - 8)) * 10000.0 + 0.5) / 10000.0;
    return (uint64_t)std::floor(256.0 * lambda);
}

extern "C" int x265hip_inter_merge_batch(void* stream, int w, int h, const void* curPlane, intptr_t curStride, intptr_t refStride,
                                         const x265hip_me_task* tasks, int n, const x265hip_merge_params* p, x265hip_inter_choice* out)
{
    if (n <= 0) return X265HIP_OK;
    if (w < 4 || h < 4 || w > 64 || h > 64 || ((w | h) & 3) || !tasks || !p || !out || !curPlane || !p->bitsRow || p->bitsHalfRange < 1)
    { set_error("inter_merge_batch: bad arguments"); return X265HIP_EARG; }
    if (p->numRef[0] < 1 || p->numRef[0] > X265HIP_MAX_REF || p->numRef[1] < 0 || p->numRef[1] > X265HIP_MAX_REF) { set_error("inter_merge_batch: 1..%d references in list 0, 0..%d in list 1", X265HIP_MAX_REF, X265HIP_MAX_REF); return X265HIP_EARG; }
    MergeArgs a{};
    a.w = w; a.h = h; a.n = n; a.isP = p->numRef[1] == 0; a.bidir = p->bidir; a.sourceMaxDim = p->sourceMaxDim; a.numRef[0] = p->numRef[0]; a.numRef[1] = p->numRef[1];
    a.cur = (const pixel*)curPlane; a.cs = curStride; a.rs = refStride; a.tasks = tasks; a.planeElems = p->planeElems;
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < p->numRef[l]; r++)
        {
            if (!p->results[l][r]) { set_error("inter_merge_batch: results of list %d reference %d missing", l, r); return X265HIP_EARG; }
            if (p->bidir && p->numRef[1] && !p->subpelPlanes[l][r]) { set_error("inter_merge_batch: the bidirectional candidate needs the phase planes of every reference"); return X265HIP_EARG; }
            a.res[l * X265HIP_MAX_REF + r] = p->results[l][r]; a.mvpSrc[l * X265HIP_MAX_REF + r] = p->mvpSource[l][r]; a.planes[l * X265HIP_MAX_REF + r] = (const pixel*)p->subpelPlanes[l][r];
        }
    a.bitsCentre = p->bitsRow + p->bitsHalfRange; a.bitsHalf = p->bitsHalfRange; a.lambda = p->lambda; a.out = out;
    if (a.isP || !a.bidir) XH_KLAUNCH(inter_merge_uni_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    else XH_KLAUNCH(inter_merge_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

extern "C" int x265hip_bidir_satd_batch(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* subpelPlanes0, const void* subpelPlanes1, int64_t planeElems,
                                        intptr_t refStride, const x265hip_bidir_task* tasks, int n, int32_t* satd)
{
    if (n <= 0) return X265HIP_OK;
    if (w < 4 || h < 4 || w > 64 || h > 64 || ((w | h) & 3) || !curPlane || !subpelPlanes0 || !subpelPlanes1 || !tasks || !satd) { set_error("bidir_satd_batch: bad arguments"); return X265HIP_EARG; }
    MergeArgs a{};
    a.w = w; a.h = h; a.n = n; a.cur = (const pixel*)curPlane; a.cs = curStride; a.rs = refStride; a.planeElems = planeElems;
    a.planes[0] = (const pixel*)subpelPlanes0; a.planes[X265HIP_MAX_REF] = (const pixel*)subpelPlanes1;
    XH_KLAUNCH(bidir_satd_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, tasks, satd, (const int8_t*)nullptr, (const int8_t*)nullptr);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

// the same with the references chosen per task: subpelPlanes0[r] / subpelPlanes1[r] = phase planes of reference r of list 0 / 1, ref0[i] / ref1[i] = task i's choice
extern "C" int x265hip_bidir_satd_batch_refs(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* const* subpelPlanes0, const void* const* subpelPlanes1,
                                             int64_t planeElems, intptr_t refStride, const x265hip_bidir_task* tasks, const int8_t* ref0, const int8_t* ref1, int n, int32_t* satd)
{
    if (n <= 0) return X265HIP_OK;
    if (w < 4 || h < 4 || w > 64 || h > 64 || ((w | h) & 3) || !curPlane || !subpelPlanes0 || !subpelPlanes1 || !tasks || !satd || !ref0 || !ref1) { set_error("bidir_satd_batch_refs: bad arguments"); return X265HIP_EARG; }
    MergeArgs a{};
    a.w = w; a.h = h; a.n = n; a.cur = (const pixel*)curPlane; a.cs = curStride; a.rs = refStride; a.planeElems = planeElems;
    for (int r = 0; r < X265HIP_MAX_REF; r++) { a.planes[r] = (const pixel*)(subpelPlanes0[r] ? subpelPlanes0[r] : subpelPlanes0[0]); a.planes[X265HIP_MAX_REF + r] = (const pixel*)(subpelPlanes1[r] ? subpelPlanes1[r] : subpelPlanes1[0]); }
    XH_KLAUNCH(bidir_satd_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, tasks, satd, ref0, ref1);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
