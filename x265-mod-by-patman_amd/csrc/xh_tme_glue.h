// xh_tme_glue.h -- the bookkeeping of Search::puMotionEstimation around its searches (reference encoder/search.cpp:226-556), one PU of one CTU per call: used by the
// one-launch-per-stage kernels of kern_tme.hip (one thread per CTU) and by the chain kernels of kern_tme_chain*.hip (lane 0 of the CTU's lane group).
#pragma once
#include "xh_amvp.h"
#include <cstdint>

namespace xh {
namespace tme {

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
    int refLag, frameParallel;               // Search::m_refLagPixels (full pel; search.cpp:96) and m_bFrameParallel
    int pirStartCol, pirSafeX;               // --intra-refresh: CUs of CTU columns < pirStartCol keep their windows left of 4 * (pirSafeX - cuX) (search.cpp:4987-4996); 0 = off
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
__device__ __forceinline__ void tme_gather(const Slice& s, const x265hip_tme_step& st, int stepIdx, int nSteps, int pi, int l, int r, int ctu,
                                           const x265hip_inter_choice* __restrict__ table, const int16_t* __restrict__ areaBest, const x265hip_tme_temporal* __restrict__ temporal,
                                           const x265hip_inter_choice* __restrict__ refTable, const int16_t* __restrict__ lowresMv, TmeState& S, x265hip_select_task& selOut,
                                           const x265hip_inter_choice* __restrict__ tableInit = nullptr, int laterMask = 0)
{
    const int ctuX = (ctu % s.numCtuX) * s.ctuSize, ctuY = (ctu / s.numCtuX) * s.ctuSize;
    const int cuAbsX = ctuX + st.cuX, cuAbsY = ctuY + st.cuY;
    if (pi == 0 && l == 0 && r == 0)
    {   // a new puMotionEstimation call: bestME and lastMode start over (search.cpp:241-246)
        S.bestCost[0] = S.bestCost[1] = 0xFFFFFFFFu; S.bestRef[0] = S.bestRef[1] = -1; S.lastMode = 0;
    }
    const int area = st.cuSize == s.ctuSize ? 0 : (cuAbsX >= (s.ctuSize >> 1)) + 2 * (cuAbsY >= (s.ctuSize >> 1)) + 1;     // analysis.cpp:175-179 (absolute position, as there)
    const int16_t* ab = areaBest + ((((int64_t)ctu * 5 + area) * 2 + l) * X265HIP_MAX_REF + r) * 2;
    S.mvpBase[0] = ab[0]; S.mvpBase[1] = ab[1];
    x265hip_amvp_task t;
#pragma unroll
    for (int d = 0; d < 5; d++)
    {
        const int slot = st.neighbor[d];
        if (slot >= 0)
        {
            // a neighbour that comes later in the schedule still holds what the table held when the picture started (the reference reads it before it gets there): kernels that
            // do not walk the entries in schedule order read those from a copy of the picture-start table
            const x265hip_inter_choice n = ((laterMask >> d) & 1) ? tableInit[(int64_t)ctu * 593 + slot] : table[(int64_t)ctu * 593 + slot];
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
    selOut = q;
}

// ---- build: the predictor and the two search tasks (search.cpp:309-390) ----
__device__ __forceinline__ void tme_build(const Slice& s, const x265hip_tme_step& st, int pi, int ctu, const x265hip_select_result& selRes, TmeState& S,
                                          x265hip_me_task& taskA, x265hip_me_task& taskB, const uint8_t* __restrict__ qpIndex, int stepIdx, int nSteps)
{
    const int q = qpIndex ? qpIndex[(int64_t)ctu * nSteps + stepIdx] : 0;                  // the CU's qp: the row of the cost table this PU's searches price MVDs with
    const int ctuX = (ctu % s.numCtuX) * s.ctuSize, ctuY = (ctu / s.numCtuX) * s.ctuSize;
    int mvp[2] = { S.mvpBase[0], S.mvpBase[1] };
    S.mvpIdx = 0;
    if (S.numMvc > 0)
    {
        S.mvpIdx = selRes.mvpIdx;
        if (s.frameParallel && (S.amvp[0][0] != S.amvp[1][0] || S.amvp[0][1] != S.amvp[1][1]))
        {   // selectMVP with frame threads: a candidate pointing below the rows the reference has finished is not costed (COST_MAX; search.cpp:2360-2365)
            const int lim = (s.searchRange + 1) * 4;
            const bool skip0 = S.amvp[0][1] >= lim, skip1 = S.amvp[1][1] >= lim;          // a costed candidate always beats COST_MAX; two skipped ones tie -> 0
            if (skip0 || skip1) S.mvpIdx = (skip0 && !skip1) ? 1 : 0;
        }
        mvp[0] = S.amvp[S.mvpIdx][0]; mvp[1] = S.amvp[S.mvpIdx][1];
    }
    S.mvpA[0] = mvp[0]; S.mvpA[1] = mvp[1];
    int numCand = S.numMvc;
    if (S.hasLowres) { S.mvc[numCand][0] = (int16_t)S.lowres[0]; S.mvc[numCand][1] = (int16_t)S.lowres[1]; numCand++; }
    S.ranB = S.hasLowres && (S.lowres[0] != mvp[0] || S.lowres[1] != mvp[1]);
    x265hip_me_task a;
    const int px = ctuX + st.pu[pi][0], py = ctuY + st.pu[pi][1];
    a.curOff = (int32_t)(s.origin + (int64_t)py * s.stride + px); a.refOff = a.curOff;
    int32_t c[4]; clip_limits(s, ctuX + st.cuX, ctuY + st.cuY, c);
    // the window is derived on the device as clamp(mvp -+ range, limits) >> 2; setSearchRange's last clamp, min(full-pel y, m_refLagPixels) on both ends
    // (search.cpp:5017-5018), is the same as an upper quarter-pel limit of 4 * lag + 3
    c[3] = min(c[3], (s.refLag << 2) + 3);
    // --intra-refresh: mvmax.x = min(mvmax.x, safe), mvmin.x = min(mvmin.x, safe) after clipMv (search.cpp:4993-4995) -- the window's clamp has its min outermost, so
    // that is an upper quarter-pel limit as well (clip_limits itself stays CUData::clipMv: selectMVP's candidates are not touched by the refresh)
    if ((ctuX + st.cuX) / s.ctuSize < s.pirStartCol) c[2] = min(c[2], (s.pirSafeX - (ctuX + st.cuX)) * 4);
    a.mvmin[0] = (int16_t)c[0]; a.mvmin[1] = (int16_t)c[1]; a.mvmax[0] = (int16_t)c[2]; a.mvmax[1] = (int16_t)c[3];
    a.qmvp[0] = (int16_t)mvp[0]; a.qmvp[1] = (int16_t)mvp[1];
#pragma unroll
    for (int k = 0; k < 12; k++) { a.mvc[2 * k] = S.mvc[k][0]; a.mvc[2 * k + 1] = S.mvc[k][1]; }
    a.numCand = (int16_t)numCand; a.flags = (int16_t)(X265HIP_ME_WINDOW | X265HIP_ME_ROWS | (q << 8)); a.mvpFrom = -1;
    x265hip_me_task b = a;
    taskA = a;
    if (S.ranB) { b.qmvp[0] = (int16_t)S.lowres[0]; b.qmvp[1] = (int16_t)S.lowres[1]; }
    else
    {   // no second search for this PU: a search that costs next to nothing (window of one position, no candidates); its result is not read
        b.mvmin[0] = b.mvmin[1] = b.mvmax[0] = b.mvmax[1] = 0; b.qmvp[0] = b.qmvp[1] = 0; b.numCand = 0;
    }
    taskB = b;
}

// ---- cost: search.cpp:392-416 ----
// S: the PU's running best (and lastMode, selBits, lambda); P: the state tme_gather / tme_build left for THIS search (predictors, candidates, lowres MV) -- the same object
// when the searches of a PU run one after the other, the state of another lane group when its references are searched side by side (tme_chain.inc)
__device__ __forceinline__ void tme_cost(const Slice& s, const x265hip_tme_step& st, int pi, int l, int r, int ctu, const x265hip_me_result& resA, const x265hip_me_result& resB,
                                         const uint16_t* __restrict__ costTable, int costHalf, const float* __restrict__ bitsCentre, int bitsHalf, TmeState& S,
                                         const uint8_t* __restrict__ qpIndex, int stepIdx, int nSteps, const Lambdas& lambdas, const TmeState& P)
{
    const int q = qpIndex ? qpIndex[(int64_t)ctu * nSteps + stepIdx] : 0;
    const uint16_t* costCentre = costTable + (size_t)q * (size_t)(2 * costHalf + 1) + costHalf;
    const unsigned long long lambda = lambdas.v[q];
    S.lambda = lambda;
    blk_bits(st.part, s.isP != 0, pi, S.lastMode, S.selBits);
    uint32_t bits = (uint32_t)S.selBits[l] + 1u + (uint32_t)(r + (r < s.numRef[l] - 1));
    x265hip_me_result m = resA;
    bool bLow = P.hasLowres != 0;
    int lastMvp[2] = { P.mvpA[0], P.mvpA[1] };
    if (P.ranB)
    {
        bLow = false;
        lastMvp[0] = P.lowres[0]; lastMvp[1] = P.lowres[1];
        const x265hip_me_result mb = resB;
        if (mb.cost < m.cost) { m = mb; bLow = true; }
    }
    const int outx = m.mv[0], outy = m.mv[1];
    bits += bits_of(bitsCentre, bitsHalf, outx, outy, lastMvp[0], lastMvp[1]);
    const int dx = min(max(outx - lastMvp[0], -costHalf), costHalf), dy = min(max(outy - lastMvp[1], -costHalf), costHalf);
    const uint32_t mvCost = (uint16_t)(costCentre[dx] + costCentre[dy]);               // m_me.mvcost(outmv): against the LAST predictor the ME object was given (:393)
    uint32_t cost = (uint32_t)(m.cost - (int)mvCost) + getcost(lambda, bits);
    int idx = P.mvpIdx;
    if (bLow)
    {   // updateMVP(mvp, outmv, bits, cost, mvp_lowres) (:395-396, 4961-4967)
        const int diff = (int)bits_of(bitsCentre, bitsHalf, outx, outy, P.mvpA[0], P.mvpA[1]) - (int)bits_of(bitsCentre, bitsHalf, outx, outy, P.lowres[0], P.lowres[1]);
        const uint32_t orig = bits;
        bits = orig + diff; cost = (cost - getcost(lambda, orig)) + getcost(lambda, bits);
    }
    {   // checkBestMVP (:398, 4947-4958)
        const int o = !idx;
        const int diff = (int)bits_of(bitsCentre, bitsHalf, outx, outy, P.amvp[o][0], P.amvp[o][1]) - (int)bits_of(bitsCentre, bitsHalf, outx, outy, P.amvp[idx][0], P.amvp[idx][1]);
        if (diff < 0)
        {
            const uint32_t orig = bits;
            idx = o; bits = orig + diff; cost = (cost - getcost(lambda, orig)) + getcost(lambda, bits);
        }
    }
    if (cost < S.bestCost[l])
    {
        S.bestCost[l] = cost; S.bestBits[l] = bits; S.bestMvCost[l] = mvCost; S.bestRef[l] = r;
        S.bestMv[l][0] = outx; S.bestMv[l][1] = outy; S.bestMvp[l][0] = P.amvp[idx][0]; S.bestMvp[l][1] = P.amvp[idx][1];
    }
}

__device__ __forceinline__ void tme_cost(const Slice& s, const x265hip_tme_step& st, int pi, int l, int r, int ctu, const x265hip_me_result& resA, const x265hip_me_result& resB,
                                         const uint16_t* __restrict__ costTable, int costHalf, const float* __restrict__ bitsCentre, int bitsHalf, TmeState& S,
                                         const uint8_t* __restrict__ qpIndex, int stepIdx, int nSteps, const Lambdas& lambdas)
{
    tme_cost(s, st, pi, l, r, ctu, resA, resB, costTable, costHalf, bitsCentre, bitsHalf, S, qpIndex, stepIdx, nSteps, lambdas, S);
}

// ---- the bidirectional candidate's tasks (search.cpp:418-450) ----
__device__ __forceinline__ void tme_bidir(const Slice& s, const x265hip_tme_step& st, int pi, int ctu, TmeState& S, x265hip_bidir_task& t0, x265hip_bidir_task& t1, int8_t& ref0, int8_t& ref1)
{
    const int ctuX = (ctu % s.numCtuX) * s.ctuSize, ctuY = (ctu / s.numCtuX) * s.ctuSize;
    const bool restricted = st.cuSize == 8 && st.part != 0;                            // CUData::isBipredRestriction
    S.bidirOn = !s.isP && !restricted && st.part != 0 && S.bestCost[0] != 0xFFFFFFFFu && S.bestCost[1] != 0xFFFFFFFFu;
    x265hip_bidir_task a;
    const int px = ctuX + st.pu[pi][0], py = ctuY + st.pu[pi][1];
    a.curOff = (int32_t)(s.origin + (int64_t)py * s.stride + px); a.refOff = a.curOff;
    a.mv0[0] = a.mv0[1] = a.mv1[0] = a.mv1[1] = 0;
    t1 = a;
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
            mny = min(mny, s.refLag); mxy = min(mxy, s.refLag);
            mxy = max(mxy, mny) + 2;
            mnx <<= 2; mny <<= 2; mxx <<= 2; mxy <<= 2;
#pragma unroll
            for (int l = 0; l < 2; l++) tz = tz && S.bestMvp[l][0] >= mnx && S.bestMvp[l][0] <= mxx && S.bestMvp[l][1] >= mny && S.bestMvp[l][1] <= mxy;
        }
        S.tryZero = tz;
    }
    t0 = a;
    ref0 = (int8_t)(S.bidirOn ? S.bestRef[0] : 0); ref1 = (int8_t)(S.bidirOn ? S.bestRef[1] : 0);
}

// ---- finish: the bidirectional decision and the MEData record (search.cpp:440-556) ----
__device__ __forceinline__ void tme_finish(const Slice& s, const x265hip_tme_step& st, int pi, int ctu, int satd, int satdZero, const float* __restrict__ bitsCentre, int bitsHalf,
                                           TmeState& S, x265hip_inter_choice* __restrict__ table)
{
    uint32_t bidirCost = 0xFFFFFFFFu; int bidirBits = 0;
    int bmv[2][2] = { { S.bestMv[0][0], S.bestMv[0][1] }, { S.bestMv[1][0], S.bestMv[1][1] } };
    if (S.bidirOn)
    {
        bidirBits = (int)(S.bestBits[0] + S.bestBits[1]) + S.selBits[2] - (S.selBits[0] + S.selBits[1]);
        bidirCost = (uint32_t)satd + getcost(S.lambda, (uint32_t)bidirBits);
        if (S.tryZero)
        {
            const uint32_t b0 = S.bestBits[0] - bits_of(bitsCentre, bitsHalf, S.bestMv[0][0], S.bestMv[0][1], S.bestMvp[0][0], S.bestMvp[0][1]) + bits_of(bitsCentre, bitsHalf, 0, 0, S.bestMvp[0][0], S.bestMvp[0][1]);
            const uint32_t b1 = S.bestBits[1] - bits_of(bitsCentre, bitsHalf, S.bestMv[1][0], S.bestMv[1][1], S.bestMvp[1][0], S.bestMvp[1][1]) + bits_of(bitsCentre, bitsHalf, 0, 0, S.bestMvp[1][0], S.bestMvp[1][1]);
            const uint32_t c = (uint32_t)satdZero + getcost(S.lambda, b0) + getcost(S.lambda, b1);
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

} // namespace tme
} // namespace xh
