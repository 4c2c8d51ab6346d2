// kern_amvp.hip -- CUData::getPMV (reference common/cudata.cpp:1806-1990, the build without multiview / SCC) for a batch of (PU, list, reference):
// the two AMVP candidates and the motion-candidate list (mvc) Search::puMotionEstimation / predInterSearch hand to motionEstimate (search.cpp:308, 2683).
// Pure integer logic on the PU's six neighbour records (MVP_DIR order, cudata.h:67-75) and the slice's POC lists: direct candidates (same reference picture
// in either list, :2110-2123), indirect ones scaled by POC distance (:2126-2156, scaleMvByPOCDist :2229-2244, scaleMv :104-110), the left / above selection
// order, the duplicate rule, the temporal candidate.  One thread per task.
#include "xh_amvp.h"
using namespace xh;

namespace {

__global__ __launch_bounds__(256) void amvp_kernel(const x265hip_amvp_task* __restrict__ tasks, int n, x265hip_amvp_params p, x265hip_amvp_result* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = get_pmv(tasks[i], p);
}

} // namespace

extern "C" int x265hip_amvp_batch(void* stream, const x265hip_amvp_task* tasks, int n, const x265hip_amvp_params* params, x265hip_amvp_result* out)
{
    if (n <= 0) return X265HIP_OK;
    if (!tasks || !params || !out) return X265HIP_EARG;
    XH_KLAUNCH(amvp_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, tasks, n, *params, out);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

// ---- Search::selectMVP (reference encoder/search.cpp:2347-2382) for a batch of PUs: each of the two AMVP candidates is clipped like CUData::clipMv
// (cudata.cpp:2094-2107), the PU is motion compensated there (predInterLumaPixel = a block of the reference's phase plane 4 * yFrac + xFrac) and compared
// with the source PU at SAD; the cheaper candidate wins, ties go to candidate 0.  One wavefront per PU. ----
#include "xh_mc.h"
namespace {
__global__ __launch_bounds__(256) void select_mvp_kernel(int w, int h, const pixel* __restrict__ cur, intptr_t cs, const pixel* __restrict__ planes, int64_t planeElems, intptr_t rs,
                                                         const x265hip_select_task* __restrict__ tasks, int n, x265hip_select_result* __restrict__ out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, item = blockIdx.x * 4 + wave;
    if (item >= n) return;
    const x265hip_select_task t = tasks[item];
    x265hip_select_result r; r.mvpIdx = 0; r.cost[0] = r.cost[1] = 0;
    if (t.amvp[0][0] != t.amvp[1][0] || t.amvp[0][1] != t.amvp[1][1])
    {
        const int qpr = w >> 2, nquads = qpr * h;
        const pixel* src[2];
#pragma unroll
        for (int i = 0; i < 2; i++)
        {
            const int mx = min(max((int)t.amvp[i][0], t.clip[0]), t.clip[2]), my = min(max((int)t.amvp[i][1], t.clip[1]), t.clip[3]);
            src[i] = planes + (int64_t)((my & 3) * 4 + (mx & 3)) * planeElems + t.refOff + (intptr_t)(my >> 2) * rs + (mx >> 2);
        }
        unsigned p0 = 0, p1 = 0;
        for (int q = lane; q < nquads; q += 64)
        {
            const int y = q / qpr, x4 = (q - y * qpr) * 4;
            int a[4], b0[4], b1[4];
            load4u(cur + t.curOff + (intptr_t)y * cs + x4, a); load4u(src[0] + (intptr_t)y * rs + x4, b0); load4u(src[1] + (intptr_t)y * rs + x4, b1);
#pragma unroll
            for (int e = 0; e < 4; e++) { p0 += (unsigned)abs(a[e] - b0[e]); p1 += (unsigned)abs(a[e] - b1[e]); }
        }
        const int c0 = wsum_u((int)p0), c1 = wsum_u((int)p1);
        r.cost[0] = c0; r.cost[1] = c1; r.mvpIdx = c0 <= c1 ? 0 : 1;
    }
    if (lane == 0) out[item] = r;
}

// Search::checkBestMVP / updateMVP (search.cpp:4947-4967): the bit and cost bookkeeping around the MVP of a finished search.  One thread per record.
__device__ __forceinline__ uint32_t bits_of(const float* centre, int half, int mvx, int mvy, int px, int py)
{
    const int dx = min(max(mvx - px, -half), half), dy = min(max(mvy - py, -half), half);
    return (uint32_t)(centre[dx] + centre[dy] + 0.5f);                            // bitcost.h:60-70
}
__global__ __launch_bounds__(256) void mvp_bits_kernel(x265hip_mvp_bits* __restrict__ rec, int n, const float* __restrict__ centre, int half, unsigned long long lambda)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    x265hip_mvp_bits r = rec[i];
    auto getcost = [&](uint32_t bits) { return (uint32_t)(((unsigned long long)bits * lambda + 128) >> 8); };      // rdcost.h:164-169
    if (r.useAlter)
    {   // updateMVP: the bits were counted against `alter`; re-base them to amvp[mvpIdx]
        const int diff = (int)bits_of(centre, half, r.mv[0], r.mv[1], r.amvp[r.mvpIdx][0], r.amvp[r.mvpIdx][1]) - (int)bits_of(centre, half, r.mv[0], r.mv[1], r.alter[0], r.alter[1]);
        const uint32_t orig = r.bits;
        r.bits = orig + diff; r.cost = (r.cost - getcost(orig)) + getcost(r.bits);
    }
    {   // checkBestMVP
        const int o = !r.mvpIdx;
        const int diff = (int)bits_of(centre, half, r.mv[0], r.mv[1], r.amvp[o][0], r.amvp[o][1]) - (int)bits_of(centre, half, r.mv[0], r.mv[1], r.amvp[r.mvpIdx][0], r.amvp[r.mvpIdx][1]);
        if (diff < 0)
        {
            const uint32_t orig = r.bits;
            r.mvpIdx = (int16_t)o; r.bits = orig + diff; r.cost = (r.cost - getcost(orig)) + getcost(r.bits);
        }
    }
    rec[i] = r;
}
} // namespace

extern "C" int x265hip_select_mvp_batch(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* subpelPlanes, int64_t planeElems, intptr_t refStride,
                                        const x265hip_select_task* tasks, int n, x265hip_select_result* out)
{
    if (n <= 0) return X265HIP_OK;
    if (!curPlane || !subpelPlanes || !tasks || !out || w < 4 || h < 4 || w > 64 || h > 64 || (w & 3)) { set_error("select_mvp_batch: bad arguments (%dx%d)", w, h); return X265HIP_EARG; }
    XH_KLAUNCH(select_mvp_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, w, h, (const pixel*)curPlane, curStride, (const pixel*)subpelPlanes, planeElems, refStride, tasks, n, out);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
extern "C" int x265hip_mvp_bits_batch(void* stream, x265hip_mvp_bits* records, int n, const float* bitsRow, int bitsHalfRange, uint64_t lambda)
{
    if (n <= 0) return X265HIP_OK;
    if (!records || !bitsRow || bitsHalfRange < 1) return X265HIP_EARG;
    XH_KLAUNCH(mvp_bits_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, records, n, bitsRow + bitsHalfRange, bitsHalfRange, (unsigned long long)lambda);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
