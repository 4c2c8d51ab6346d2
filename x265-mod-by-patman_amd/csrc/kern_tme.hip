// kern_tme.hip -- the PU stage of ThreadedME for all CTUs of a picture: Analysis::computeMVForPUs -> Search::puMotionEstimation (reference encoder/analysis.cpp:161-246,
// search.cpp:226-556) stepped through x265hip_tme_schedule's entries.  Entry k is the same PU shape at the same place of every CTU, so each stage of the PU's chain
// is ONE launch over all CTUs:
//     gather   neighbour records out of the CTU's MEData table (+ the host's temporal neighbour) -> CUData::getPMV -> the predictor-choice task      (this file)
//     x265hip_select_mvp_batch                                                                                                               (kern_amvp.hip)
//     build    predictor (AMVP choice | m_areaBestMV | the reference frame's record), the lookahead's MV as candidate -> the two search tasks          (this file)
//     x265hip_me_batch twice (the search from the predictor; the search from the lookahead's MV where the reference runs it)                    (me_body.inc)
//     cost     bits / cost bookkeeping, updateMVP, checkBestMVP, best reference per list (bestME lives across the partitions of an entry)               (this file)
//     bidir    the bidirectional candidate's two task lists -> x265hip_bidir_satd_batch twice -> finish: the MEData record                             (this file)
// The glue is the one tests/tme_pu.py restates (and pins to recorded puMotionEstimation calls); the kernels here are one thread per CTU.
#include "xh_amvp.h"
#include <cstdint>
using namespace xh;

namespace {

struct TmeState
{
    int32_t bestMv[2][2], bestMvp[2][2], bestRef[2];
    uint32_t bestCost[2], bestBits[2], bestMvCost[2];
    int32_t lastMode, selBits[3];
    int32_t amvp[2][2], mvpIdx, numMvc, mvpBase[2], lowres[2], hasLowres, ranB, mvpA[2];
    int16_t mvc[12][2];
    int32_t bidirOn, tryZero;
    unsigned long long lambda;               // of the CU's qp (Analysis::setLambdaFromQP per CU: AQ / cuTree move the qp inside a picture)
};

struct Lambdas { unsigned long long v[64]; };

struct Slice
{
    int isP, numRef[2], searchRange, picW, picH, ctuSize, numCtuX, lowresBlocksX;
    x265hip_amvp_params amvp;
    intptr_t stride; int64_t origin;
};

__device__ __forceinline__ void blk_bits(int part, bool isP, int partIdx, int lastMode, int32_t (&b)[3])
{   // Search::getBlkBits (search.cpp:4893-4944)
    if (part == 0 || part == 3) { b[0] = isP ? 1 : 3; b[1] = 3; b[2] = 5; return; }
    if (isP) { b[0] = 3; b[1] = 0; b[2] = 0; return; }
    const bool horizontal = part == 1 || part == 4 || part == 5;
    const int h[2][3][3] = { { { 0, 0, 3 }, { 0, 0, 0 }, { 0, 0, 0 } }, { { 5, 7, 7 }, { 7, 5, 7 }, { 6, 6, 6 } } };
    const int v[2][3][3] = { { { 0, 2, 3 }, { 0, 0, 0 }, { 0, 0, 0 } }, { { 5, 7, 7 }, { 5, 5, 7 }, { 6, 6, 6 } } };
#pragma unroll
    for (int k = 0; k < 3; k++) b[k] = horizontal ? h[partIdx][lastMode][k] : v[partIdx][lastMode][k];
}
__device__ __forceinline__ void clip_limits(const Slice& s, int cuAbsX, int cuAbsY, int32_t (&c)[4])
{   // CUData::clipMv (cudata.cpp:2094-2107)
    c[0] = -((s.ctuSize + 8 + cuAbsX - 1) << 2); c[1] = -((s.ctuSize + 8 + cuAbsY - 1) << 2);
    c[2] = (s.picW + 8 - cuAbsX - 1) << 2; c[3] = (s.picH + 8 - cuAbsY - 1) << 2;
}
__device__ __forceinline__ uint32_t bits_of(const float* centre, int half, int mvx, int mvy, int px, int py)
{
    const int dx = min(max(mvx - px, -half), half), dy = min(max(mvy - py, -half), half);
    return (uint32_t)(centre[dx] + centre[dy] + 0.5f);
}
__device__ __forceinline__ uint32_t getcost(unsigned long long lambda, uint32_t bits) { return (uint32_t)(((unsigned long long)bits * lambda + 128) >> 8); }

// ---- gather: search.cpp:250-312 for partition pi, list l, reference r of the entry ----
__global__ __launch_bounds__(256) void tme_gather_kernel(Slice s, x265hip_tme_step st, int stepIdx, int nSteps, int pi, int l, int r, int nCtu,
                                                         const x265hip_inter_choice* __restrict__ table, const int16_t* __restrict__ areaBest,
                                                         const x265hip_tme_temporal* __restrict__ temporal, const x265hip_inter_choice* __restrict__ refTable,
                                                         const int16_t* __restrict__ lowresMv, TmeState* __restrict__ state, x265hip_select_task* __restrict__ sel)
{
    const int ctu = blockIdx.x * 256 + threadIdx.x;
    if (ctu >= nCtu) return;
    TmeState& S = state[ctu];
    const int ctuX = (ctu % s.numCtuX) * s.ctuSize, ctuY = (ctu / s.numCtuX) * s.ctuSize;
    const int cuAbsX = ctuX + st.cuX, cuAbsY = ctuY + st.cuY;
    if (pi == 0 && l == 0 && r == 0)
    {   // a new puMotionEstimation call: bestME and lastMode start over (search.cpp:241-246)
        S.bestCost[0] = S.bestCost[1] = 0xFFFFFFFFu; S.bestRef[0] = S.bestRef[1] = -1; S.lastMode = 0;
    }
    const int area = st.cuSize == s.ctuSize ? 0 : (cuAbsX >= (s.ctuSize >> 1)) + 2 * (cuAbsY >= (s.ctuSize >> 1)) + 1;     // analysis.cpp:175-179 (absolute position, as there)
    const int16_t* ab = areaBest + ((((int64_t)ctu * 5 + area) * 2 + l) * 4 + r) * 2;
    S.mvpBase[0] = ab[0]; S.mvpBase[1] = ab[1];
    x265hip_amvp_task t;
#pragma unroll
    for (int d = 0; d < 5; d++)
    {
        const int slot = st.neighbor[d];
        if (slot >= 0)
        {
            const x265hip_inter_choice n = table[(int64_t)ctu * 593 + slot];
            t.nb[d].mv[0][0] = n.mv[0][0]; t.nb[d].mv[0][1] = n.mv[0][1]; t.nb[d].mv[1][0] = n.mv[1][0]; t.nb[d].mv[1][1] = n.mv[1][1];
            t.nb[d].refIdx[0] = n.ref[0]; t.nb[d].refIdx[1] = n.ref[1]; t.nb[d].available = (n.ref[0] >= 0 || n.ref[1] >= 0);
        }
        else { t.nb[d].mv[0][0] = t.nb[d].mv[0][1] = t.nb[d].mv[1][0] = t.nb[d].mv[1][1] = 0; t.nb[d].refIdx[0] = t.nb[d].refIdx[1] = -1; t.nb[d].available = 0; }
        t.nb[d].reserved = 0;
    }
    const x265hip_tme_temporal tp = temporal[((int64_t)ctu * nSteps + stepIdx) * 2 + pi];
    t.nb[5] = tp.nb; t.list = (int8_t)l; t.refIdx = (int8_t)r; t.reserved = 0; t.colPOC = tp.colPOC[l]; t.colRefPOC = tp.colRefPOC[l];
    const x265hip_amvp_result a = get_pmv(t, s.amvp);
    S.numMvc = a.numMvc;
#pragma unroll
    for (int k = 0; k < 11; k++) { S.mvc[k][0] = a.mvc[k][0]; S.mvc[k][1] = a.mvc[k][1]; }
    S.mvc[11][0] = S.mvc[11][1] = 0;
    if (a.numMvc > 0) { S.amvp[0][0] = a.amvp[0][0]; S.amvp[0][1] = a.amvp[0][1]; S.amvp[1][0] = a.amvp[1][0]; S.amvp[1][1] = a.amvp[1][1]; }
    else
    {   // no candidate: amvp = zeroMV (search.cpp:271-272); the predictor falls back to the reference frame's own record at this slot (:313-330)
        S.amvp[0][0] = S.amvp[0][1] = S.amvp[1][0] = S.amvp[1][1] = 0;
        if (refTable)
        {
            const x265hip_inter_choice m = refTable[(int64_t)ctu * 593 + st.finalIdx + pi * st.puOffset];
            if (m.ref[0] >= 0 && m.ref[1] < 0) { S.mvpBase[0] = m.mv[0][0]; S.mvpBase[1] = m.mv[0][1]; }
            else if (m.ref[1] >= 0 && m.ref[0] < 0) { S.mvpBase[0] = m.mv[1][0]; S.mvpBase[1] = m.mv[1][1]; }
            else if (m.ref[0] >= 0 && m.ref[1] >= 0) { S.mvpBase[0] = m.mv[l][0]; S.mvpBase[1] = m.mv[l][1]; }
        }
    }
    // the lookahead's MV of the 16x16 block under the PU's centre (Search::getLowresMV, search.cpp:2323-2343; lowresMv == NULL: not estimated / out of range)
    const int px = ctuX + st.pu[pi][0], py = ctuY + st.pu[pi][1], pw = st.pu[pi][2], ph = st.pu[pi][3];
    S.hasLowres = 0; S.lowres[0] = S.lowres[1] = 0;
    if (lowresMv && px + (pw >> 1) < s.picW && py + (ph >> 1) < s.picH)
    {
        const int idx = ((py + ph / 2) >> 4) * s.lowresBlocksX + ((px + pw / 2) >> 4);
        S.lowres[0] = (int)lowresMv[2 * idx] * 2; S.lowres[1] = (int)lowresMv[2 * idx + 1] * 2;
        S.hasLowres = (S.lowres[0] | S.lowres[1]) != 0;
    }
    x265hip_select_task q;
    q.curOff = (int32_t)(s.origin + (int64_t)py * s.stride + px); q.refOff = q.curOff;
    q.amvp[0][0] = (int16_t)S.amvp[0][0]; q.amvp[0][1] = (int16_t)S.amvp[0][1]; q.amvp[1][0] = (int16_t)S.amvp[1][0]; q.amvp[1][1] = (int16_t)S.amvp[1][1];
    clip_limits(s, cuAbsX, cuAbsY, q.clip);
    sel[ctu] = q;
}

// ---- build: the predictor and the two search tasks (search.cpp:309-390) ----
__global__ __launch_bounds__(256) void tme_build_kernel(Slice s, x265hip_tme_step st, int pi, int nCtu, const x265hip_select_result* __restrict__ selRes, TmeState* __restrict__ state,
                                                        x265hip_me_task* __restrict__ taskA, x265hip_me_task* __restrict__ taskB, const uint8_t* __restrict__ qpIndex, int stepIdx, int nSteps)
{
    const int ctu = blockIdx.x * 256 + threadIdx.x;
    if (ctu >= nCtu) return;
    TmeState& S = state[ctu];
    const int q = qpIndex ? qpIndex[(int64_t)ctu * nSteps + stepIdx] : 0;                  // the CU's qp: the row of the cost table this PU's searches price MVDs with
    const int ctuX = (ctu % s.numCtuX) * s.ctuSize, ctuY = (ctu / s.numCtuX) * s.ctuSize;
    int mvp[2] = { S.mvpBase[0], S.mvpBase[1] };
    S.mvpIdx = 0;
    if (S.numMvc > 0) { S.mvpIdx = selRes[ctu].mvpIdx; mvp[0] = S.amvp[S.mvpIdx][0]; mvp[1] = S.amvp[S.mvpIdx][1]; }
    S.mvpA[0] = mvp[0]; S.mvpA[1] = mvp[1];
    int numCand = S.numMvc;
    if (S.hasLowres) { S.mvc[numCand][0] = (int16_t)S.lowres[0]; S.mvc[numCand][1] = (int16_t)S.lowres[1]; numCand++; }
    S.ranB = S.hasLowres && (S.lowres[0] != mvp[0] || S.lowres[1] != mvp[1]);
    x265hip_me_task a;
    const int px = ctuX + st.pu[pi][0], py = ctuY + st.pu[pi][1];
    a.curOff = (int32_t)(s.origin + (int64_t)py * s.stride + px); a.refOff = a.curOff;
    int32_t c[4]; clip_limits(s, ctuX + st.cuX, ctuY + st.cuY, c);
    a.mvmin[0] = (int16_t)c[0]; a.mvmin[1] = (int16_t)c[1]; a.mvmax[0] = (int16_t)c[2]; a.mvmax[1] = (int16_t)c[3];
    a.qmvp[0] = (int16_t)mvp[0]; a.qmvp[1] = (int16_t)mvp[1];
#pragma unroll
    for (int k = 0; k < 12; k++) { a.mvc[2 * k] = S.mvc[k][0]; a.mvc[2 * k + 1] = S.mvc[k][1]; }
    a.numCand = (int16_t)numCand; a.flags = (int16_t)(X265HIP_ME_WINDOW | X265HIP_ME_ROWS | (q << 8)); a.mvpFrom = -1;
    x265hip_me_task b = a;
    taskA[ctu] = a;
    if (S.ranB) { b.qmvp[0] = (int16_t)S.lowres[0]; b.qmvp[1] = (int16_t)S.lowres[1]; }
    else
    {   // no second search for this PU: a search that costs next to nothing (window of one position, no candidates); its result is not read
        b.mvmin[0] = b.mvmin[1] = b.mvmax[0] = b.mvmax[1] = 0; b.qmvp[0] = b.qmvp[1] = 0; b.numCand = 0;
    }
    taskB[ctu] = b;
}

// ---- cost: search.cpp:392-416 ----
__global__ __launch_bounds__(256) void tme_cost_kernel(Slice s, x265hip_tme_step st, int pi, int l, int r, int nCtu, const x265hip_me_result* __restrict__ resA,
                                                       const x265hip_me_result* __restrict__ resB, const uint16_t* __restrict__ costTable, int costHalf,
                                                       const float* __restrict__ bitsCentre, int bitsHalf, TmeState* __restrict__ state, const uint8_t* __restrict__ qpIndex, int stepIdx, int nSteps,
                                                       Lambdas lambdas)
{
    const int ctu = blockIdx.x * 256 + threadIdx.x;
    if (ctu >= nCtu) return;
    const int q = qpIndex ? qpIndex[(int64_t)ctu * nSteps + stepIdx] : 0;
    const uint16_t* costCentre = costTable + (size_t)q * (size_t)(2 * costHalf + 1) + costHalf;
    const unsigned long long lambda = lambdas.v[q];
    TmeState& S = state[ctu];
    S.lambda = lambda;
    blk_bits(st.part, s.isP != 0, pi, S.lastMode, S.selBits);
    uint32_t bits = (uint32_t)S.selBits[l] + 1u + (uint32_t)(r + (r < s.numRef[l] - 1));
    x265hip_me_result m = resA[ctu];
    bool bLow = S.hasLowres != 0;
    int lastMvp[2] = { S.mvpA[0], S.mvpA[1] };
    if (S.ranB)
    {
        bLow = false;
        lastMvp[0] = S.lowres[0]; lastMvp[1] = S.lowres[1];
        const x265hip_me_result mb = resB[ctu];
        if (mb.cost < m.cost) { m = mb; bLow = true; }
    }
    const int outx = m.mv[0], outy = m.mv[1];
    bits += bits_of(bitsCentre, bitsHalf, outx, outy, lastMvp[0], lastMvp[1]);
    const int dx = min(max(outx - lastMvp[0], -costHalf), costHalf), dy = min(max(outy - lastMvp[1], -costHalf), costHalf);
    const uint32_t mvCost = (uint16_t)(costCentre[dx] + costCentre[dy]);               // m_me.mvcost(outmv): against the LAST predictor the ME object was given (:393)
    uint32_t cost = (uint32_t)(m.cost - (int)mvCost) + getcost(lambda, bits);
    int idx = S.mvpIdx;
    if (bLow)
    {   // updateMVP(mvp, outmv, bits, cost, mvp_lowres) (:395-396, 4961-4967)
        const int diff = (int)bits_of(bitsCentre, bitsHalf, outx, outy, S.mvpA[0], S.mvpA[1]) - (int)bits_of(bitsCentre, bitsHalf, outx, outy, S.lowres[0], S.lowres[1]);
        const uint32_t orig = bits;
        bits = orig + diff; cost = (cost - getcost(lambda, orig)) + getcost(lambda, bits);
    }
    {   // checkBestMVP (:398, 4947-4958)
        const int o = !idx;
        const int diff = (int)bits_of(bitsCentre, bitsHalf, outx, outy, S.amvp[o][0], S.amvp[o][1]) - (int)bits_of(bitsCentre, bitsHalf, outx, outy, S.amvp[idx][0], S.amvp[idx][1]);
        if (diff < 0)
        {
            const uint32_t orig = bits;
            idx = o; bits = orig + diff; cost = (cost - getcost(lambda, orig)) + getcost(lambda, bits);
        }
    }
    if (cost < S.bestCost[l])
    {
        S.bestCost[l] = cost; S.bestBits[l] = bits; S.bestMvCost[l] = mvCost; S.bestRef[l] = r;
        S.bestMv[l][0] = outx; S.bestMv[l][1] = outy; S.bestMvp[l][0] = S.amvp[idx][0]; S.bestMvp[l][1] = S.amvp[idx][1];
    }
}

// ---- the bidirectional candidate's tasks (search.cpp:418-450) ----
__global__ __launch_bounds__(256) void tme_bidir_kernel(Slice s, x265hip_tme_step st, int pi, int nCtu, TmeState* __restrict__ state, x265hip_bidir_task* __restrict__ t0,
                                                        x265hip_bidir_task* __restrict__ t1, int8_t* __restrict__ ref0, int8_t* __restrict__ ref1)
{
    const int ctu = blockIdx.x * 256 + threadIdx.x;
    if (ctu >= nCtu) return;
    TmeState& S = state[ctu];
    const int ctuX = (ctu % s.numCtuX) * s.ctuSize, ctuY = (ctu / s.numCtuX) * s.ctuSize;
    const bool restricted = st.cuSize == 8 && st.part != 0;                            // CUData::isBipredRestriction
    S.bidirOn = !s.isP && !restricted && st.part != 0 && S.bestCost[0] != 0xFFFFFFFFu && S.bestCost[1] != 0xFFFFFFFFu;
    x265hip_bidir_task a;
    const int px = ctuX + st.pu[pi][0], py = ctuY + st.pu[pi][1];
    a.curOff = (int32_t)(s.origin + (int64_t)py * s.stride + px); a.refOff = a.curOff;
    a.mv0[0] = a.mv0[1] = a.mv1[0] = a.mv1[1] = 0;
    t1[ctu] = a;
    S.tryZero = 0;
    if (S.bidirOn)
    {
        a.mv0[0] = (int16_t)S.bestMv[0][0]; a.mv0[1] = (int16_t)S.bestMv[0][1]; a.mv1[0] = (int16_t)S.bestMv[1][0]; a.mv1[1] = (int16_t)S.bestMv[1][1];
        bool tz = (S.bestMv[0][0] | S.bestMv[0][1] | S.bestMv[1][0] | S.bestMv[1][1]) != 0;
        if (tz)
        {   // setSearchRange(cu, mvzero, max(sourceWidth, sourceHeight)), mvmax.y += 2, << 2: both MVPs inside (:452-462)
            int32_t c[4]; clip_limits(s, ctuX + st.cuX, ctuY + st.cuY, c);
            const int d = max(s.picW, s.picH) << 2;
            int mnx = min(c[2], max(c[0], -d)) >> 2, mny = min(c[3], max(c[1], -d)) >> 2, mxx = min(c[2], max(c[0], d)) >> 2, mxy = min(c[3], max(c[1], d)) >> 2;
            mxy = max(mxy, mny) + 2;
            mnx <<= 2; mny <<= 2; mxx <<= 2; mxy <<= 2;
#pragma unroll
            for (int l = 0; l < 2; l++) tz = tz && S.bestMvp[l][0] >= mnx && S.bestMvp[l][0] <= mxx && S.bestMvp[l][1] >= mny && S.bestMvp[l][1] <= mxy;
        }
        S.tryZero = tz;
    }
    t0[ctu] = a;
    ref0[ctu] = (int8_t)(S.bidirOn ? S.bestRef[0] : 0); ref1[ctu] = (int8_t)(S.bidirOn ? S.bestRef[1] : 0);
}

// ---- finish: the bidirectional decision and the MEData record (search.cpp:440-556) ----
__global__ __launch_bounds__(256) void tme_finish_kernel(Slice s, x265hip_tme_step st, int pi, int nCtu, const int32_t* __restrict__ satd, const int32_t* __restrict__ satdZero,
                                                         const float* __restrict__ bitsCentre, int bitsHalf, TmeState* __restrict__ state, x265hip_inter_choice* __restrict__ table)
{
    const int ctu = blockIdx.x * 256 + threadIdx.x;
    if (ctu >= nCtu) return;
    TmeState& S = state[ctu];
    uint32_t bidirCost = 0xFFFFFFFFu; int bidirBits = 0;
    int bmv[2][2] = { { S.bestMv[0][0], S.bestMv[0][1] }, { S.bestMv[1][0], S.bestMv[1][1] } };
    if (S.bidirOn)
    {
        bidirBits = (int)(S.bestBits[0] + S.bestBits[1]) + S.selBits[2] - (S.selBits[0] + S.selBits[1]);
        bidirCost = (uint32_t)satd[ctu] + getcost(S.lambda, (uint32_t)bidirBits);
        if (S.tryZero)
        {
            const uint32_t b0 = S.bestBits[0] - bits_of(bitsCentre, bitsHalf, S.bestMv[0][0], S.bestMv[0][1], S.bestMvp[0][0], S.bestMvp[0][1]) + bits_of(bitsCentre, bitsHalf, 0, 0, S.bestMvp[0][0], S.bestMvp[0][1]);
            const uint32_t b1 = S.bestBits[1] - bits_of(bitsCentre, bitsHalf, S.bestMv[1][0], S.bestMv[1][1], S.bestMvp[1][0], S.bestMvp[1][1]) + bits_of(bitsCentre, bitsHalf, 0, 0, S.bestMvp[1][0], S.bestMvp[1][1]);
            const uint32_t c = (uint32_t)satdZero[ctu] + getcost(S.lambda, b0) + getcost(S.lambda, b1);
            if (c < bidirCost) { bmv[0][0] = bmv[0][1] = bmv[1][0] = bmv[1][1] = 0; bidirCost = c; bidirBits = (int)(b0 + b1) + S.selBits[2] - (S.selBits[0] + S.selBits[1]); }
        }
    }
    x265hip_inter_choice& o = table[(int64_t)ctu * 593 + st.finalIdx + pi * st.puOffset];
    // the reference writes only the fields of the chosen list(s); the others keep what the slot held
    if (bidirCost < S.bestCost[0] && bidirCost < S.bestCost[1])
    {
        S.lastMode = 2;
#pragma unroll
        for (int l = 0; l < 2; l++)
        {
            o.mv[l][0] = (int16_t)bmv[l][0]; o.mv[l][1] = (int16_t)bmv[l][1]; o.mvp[l][0] = (int16_t)S.bestMvp[l][0]; o.mvp[l][1] = (int16_t)S.bestMvp[l][1];
            o.mvCost[l] = S.bestMvCost[l]; o.ref[l] = (int8_t)S.bestRef[l];
        }
        o.bits = bidirBits; o.cost = bidirCost;
    }
    else
    {
        const int l = S.bestCost[0] <= S.bestCost[1] ? 0 : 1;
        S.lastMode = l;
        o.mv[l][0] = (int16_t)S.bestMv[l][0]; o.mv[l][1] = (int16_t)S.bestMv[l][1]; o.mvp[l][0] = (int16_t)S.bestMvp[l][0]; o.mvp[l][1] = (int16_t)S.bestMvp[l][1];
        o.mvCost[l] = S.bestMvCost[l]; o.ref[l] = (int8_t)S.bestRef[l]; o.ref[l ^ 1] = -1;
        o.bits = (int32_t)S.bestBits[l]; o.cost = S.bestCost[l];
    }
}

// Entries of different PU shapes never read each other's slots (a PU's neighbours are PUs of its own shape, search.cpp:283-305), so every shape is its own chain of
// dependent launches: the chains run side by side on their own streams, each with its own slice of the workspace.
constexpr int XH_TME_CHAINS = 24;

} // namespace

extern "C" size_t x265hip_tme_workspace(int nCtu)
{
    const size_t per = sizeof(TmeState) + sizeof(x265hip_select_task) + sizeof(x265hip_select_result) + 2 * sizeof(x265hip_me_task) + 2 * sizeof(x265hip_me_result) +
                       2 * sizeof(x265hip_bidir_task) + 2 * sizeof(int32_t) + 2;
    return ((size_t)nCtu * per + 16 * 256) * XH_TME_CHAINS;
}

extern "C" int x265hip_tme_frame(void* stream, const x265hip_tme_args* a)
{
    if (!a || !a->steps || a->nSteps < 1 || !a->curPlane || !a->table || !a->areaBest || !a->temporal || !a->bitsRow || !a->workspace || a->nQp < 1 || a->nQp > 64 || (a->nQp > 1 && !a->qpIndex)) { set_error("tme_frame: missing arguments"); return X265HIP_EARG; }
    if (!a->costRows) { set_error("tme_frame: cost table missing"); return X265HIP_EARG; }
    if (a->ctuSize < 16 || a->picWidth < 8 || a->picHeight < 8) { set_error("tme_frame: bad picture / CTU size"); return X265HIP_EARG; }
    // CTUs cut by the picture edge run the whole schedule like the others -- computeMVForPUs does not look at the picture size; PUs beyond the edge read the planes' padding
    const int nCtuX = (a->picWidth + a->ctuSize - 1) / a->ctuSize, nCtuY = (a->picHeight + a->ctuSize - 1) / a->ctuSize, nCtu = nCtuX * nCtuY;
    if (a->workspaceBytes < x265hip_tme_workspace(nCtu)) { set_error("tme_frame: workspace too small"); return X265HIP_EARG; }
    const int nl = a->isP ? 1 : 2;
    for (int l = 0; l < nl; l++)
    {
        if (a->numRef[l] < 1 || a->numRef[l] > 4) { set_error("tme_frame: %d references in list %d", a->numRef[l], l); return X265HIP_EARG; }
        for (int r = 0; r < a->numRef[l]; r++) if (!a->refs[l][r].mePlane || !a->refs[l][r].mePhase || !a->refs[l][r].reconPhase) { set_error("tme_frame: planes of list %d reference %d missing", l, r); return X265HIP_EARG; }
    }
    hipStream_t mainStream = (hipStream_t)stream;
    // the chains: entries grouped by shape (CU size, partition type), order kept
    int chainOf[24 * 4]; int nChains = 0; int key[XH_TME_CHAINS];
    (void)chainOf;
    static thread_local hipStream_t side[XH_TME_CHAINS] = {};
    static thread_local hipEvent_t evFork = nullptr, evJoin[XH_TME_CHAINS] = {};
    if (!evFork) XH_HIP(hipEventCreateWithFlags(&evFork, hipEventDisableTiming));
    for (int k = 0; k < a->nSteps; k++)
    {
        const int kk = a->steps[k].cuSize * 8 + a->steps[k].part;
        int c = 0; while (c < nChains && key[c] != kk) c++;
        if (c == nChains) { if (nChains == XH_TME_CHAINS) { set_error("tme_frame: more than %d PU shapes", XH_TME_CHAINS); return X265HIP_EARG; } key[nChains++] = kk; }
    }
    XH_HIP(hipEventRecord(evFork, mainStream));
    const size_t chainBytes = x265hip_tme_workspace(nCtu) / XH_TME_CHAINS;
    for (int chain = 0; chain < nChains; chain++)
    {
    if (!side[chain]) { XH_HIP(hipStreamCreateWithFlags(&side[chain], hipStreamNonBlocking)); XH_HIP(hipEventCreateWithFlags(&evJoin[chain], hipEventDisableTiming)); }
    hipStream_t st = side[chain];
    void* stream = (void*)st;                                                         // the batch entry points of this chain launch on its stream
    XH_HIP(hipStreamWaitEvent(st, evFork, 0));
    char* w = (char*)a->workspace + (size_t)chain * chainBytes;
    auto take = [&](size_t bytes) { char* p = w; w += (bytes + 255) & ~(size_t)255; return p; };
    TmeState* state = (TmeState*)take(sizeof(TmeState) * nCtu);
    x265hip_select_task* sel = (x265hip_select_task*)take(sizeof(x265hip_select_task) * nCtu);
    x265hip_select_result* selRes = (x265hip_select_result*)take(sizeof(x265hip_select_result) * nCtu);
    x265hip_me_task* tA = (x265hip_me_task*)take(sizeof(x265hip_me_task) * nCtu); x265hip_me_task* tB = (x265hip_me_task*)take(sizeof(x265hip_me_task) * nCtu);
    x265hip_me_result* rA = (x265hip_me_result*)take(sizeof(x265hip_me_result) * nCtu); x265hip_me_result* rB = (x265hip_me_result*)take(sizeof(x265hip_me_result) * nCtu);
    x265hip_bidir_task* b0 = (x265hip_bidir_task*)take(sizeof(x265hip_bidir_task) * nCtu); x265hip_bidir_task* b1 = (x265hip_bidir_task*)take(sizeof(x265hip_bidir_task) * nCtu);
    int32_t* s0 = (int32_t*)take(4 * nCtu); int32_t* s1 = (int32_t*)take(4 * nCtu); int8_t* br0 = (int8_t*)take(nCtu); int8_t* br1 = (int8_t*)take(nCtu);
    const void* ph0[4] = {}; const void* ph1[4] = {};
    for (int r = 0; r < 4; r++) { ph0[r] = r < a->numRef[0] ? a->refs[0][r].reconPhase : nullptr; ph1[r] = (!a->isP && r < a->numRef[1]) ? a->refs[1][r].reconPhase : nullptr; }
    Slice s{};
    s.isP = a->isP; s.numRef[0] = a->numRef[0]; s.numRef[1] = a->isP ? 0 : a->numRef[1]; s.searchRange = a->searchRange; s.picW = a->picWidth; s.picH = a->picHeight;
    s.ctuSize = a->ctuSize; s.numCtuX = nCtuX; s.lowresBlocksX = a->lowresBlocksX; s.stride = a->stride; s.origin = a->origin;
    s.amvp.curPOC = a->curPOC; s.amvp.temporalMvp = a->temporalMvp;
    for (int l = 0; l < 2; l++) for (int r = 0; r < 16; r++) s.amvp.refPOC[l][r] = a->refPOC[l][r];
    const dim3 grid((nCtu + 255) / 256), block(256);
    const float* bitsCentre = a->bitsRow + a->bitsHalfRange;
    Lambdas lambdas{};
    for (int q = 0; q < a->nQp; q++) lambdas.v[q] = a->lambdas[q];
    for (int k = 0; k < a->nSteps; k++)
    {
        const x265hip_tme_step& e = a->steps[k];
        if (e.cuSize * 8 + e.part != key[chain]) continue;
        for (int pi = 0; pi < e.numPart; pi++)
        {
            const int pw = e.pu[pi][2], ph = e.pu[pi][3];
            for (int l = 0; l < nl; l++)
                for (int r = 0; r < a->numRef[l]; r++)
                {
                    const x265hip_tme_ref& R = a->refs[l][r];
                    hipLaunchKernelGGL(tme_gather_kernel, grid, block, 0, st, s, e, k, a->nSteps, pi, l, r, nCtu, a->table, a->areaBest, a->temporal, R.refTable, R.lowresMv, state, sel);
                    int rc = x265hip_select_mvp_batch(stream, pw, ph, a->curPlane, a->stride, R.reconPhase, a->planeElems, a->stride, sel, nCtu, selRes);
                    if (rc) { set_error("tme_frame: select_mvp_batch %dx%d failed", pw, ph); return rc; }
                    hipLaunchKernelGGL(tme_build_kernel, grid, block, 0, st, s, e, pi, nCtu, selRes, state, tA, tB, a->qpIndex, k, a->nSteps);
                    rc = x265hip_me_batch(stream, pw, ph, a->curPlane, a->stride, R.mePlane, a->stride, tA, nCtu, a->costRows, a->costHalfRange, a->searchRange, a->searchMethod,
                                          a->subpelRefine, rA, nullptr, R.mePhase, a->planeElems);
                    if (rc) return rc;
                    rc = x265hip_me_batch(stream, pw, ph, a->curPlane, a->stride, R.mePlane, a->stride, tB, nCtu, a->costRows, a->costHalfRange, a->searchRange, a->searchMethod,
                                          a->subpelRefine, rB, nullptr, R.mePhase, a->planeElems);
                    if (rc) return rc;
                    hipLaunchKernelGGL(tme_cost_kernel, grid, block, 0, st, s, e, pi, l, r, nCtu, rA, rB, a->costRows, a->costHalfRange, bitsCentre, a->bitsHalfRange, state,
                                       a->qpIndex, k, a->nSteps, lambdas);
                }
            hipLaunchKernelGGL(tme_bidir_kernel, grid, block, 0, st, s, e, pi, nCtu, state, b0, b1, br0, br1);
            if (!a->isP && e.part != 0 && e.cuSize != 8)
            {   // the bidirectional candidate exists for this shape: its two distortions, each PU against the references it chose per list
                int rc = x265hip_bidir_satd_batch_refs(stream, pw, ph, a->curPlane, a->stride, ph0, ph1, a->planeElems, a->stride, b0, br0, br1, nCtu, s0);
                if (rc) return rc;
                rc = x265hip_bidir_satd_batch_refs(stream, pw, ph, a->curPlane, a->stride, ph0, ph1, a->planeElems, a->stride, b1, br0, br1, nCtu, s1);
                if (rc) return rc;
            }
            hipLaunchKernelGGL(tme_finish_kernel, grid, block, 0, st, s, e, pi, nCtu, s0, s1, bitsCentre, a->bitsHalfRange, state, a->table);
        }
    }
    XH_HIP(hipEventRecord(evJoin[chain], st));
    XH_HIP(hipStreamWaitEvent(mainStream, evJoin[chain], 0));
    }
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
