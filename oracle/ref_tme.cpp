/*
 * ref_tme.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * The reference encoder run with --threaded-me (decoupled motion estimation: ThreadedME::findJob -> Analysis::deriveMVsForCTU ->
 * Search::puMotionEstimation, encoder/threadedme.cpp:207-261, analysis.cpp:161-306, search.cpp:226-560) on a synthetic clip, with every
 * call of MotionEstimate::motionEstimate on a full-resolution reference RECORDED: the PU's source pixels, the reference plane, the search
 * window, predictor and candidates exactly as puMotionEstimation / predInterSearch built them, and what the reference returned.  The records
 * are the fixture of tests/golden/tme_*.npz: the same task list goes through x265hip_me_batch (GPU) and the oracle (CPU) and must give the
 * same MV and cost -- the data-level form of SURVEY 8(f1).
 *
 * How the calls are seen without touching the reference's sources: oracle/Makefile compiles encoder/motion.cpp a second time with
 * -DmotionEstimate=motionEstimate_ref (the member keeps its body, under another name) and links that object instead of the regular one;
 * the definition of MotionEstimate::motionEstimate below is what every caller in the encoder then reaches: it forwards to the renamed member
 * and writes the record.
 *
 * usage: x265tme_<depth> <width> <height> <frames> <preset> <out.bin> [option=value ...]
 * record stream (little endian):  int32 kind, int32 nInts, int32 ints[nInts], then kind-specific pixels as uint16
 *   kind 1 (reference plane snapshot): ints = { id, stride, rows, originOffset, width, height }, pixels = stride * rows
 *   kind 2 (call): ints = { planeId, w, h, blockOffset, mvmin.x, .y, mvmax.x, .y, qmvp.x, .y, numCand, merange, searchMethod, subpelRefine, qp,
 *                           bChromaSATD, maxSlices, vertRestriction, srcPlaneGiven, out.x, out.y, cost, mvcost(out),
 *                           cbPlaneId, crPlaneId, chromaOffset, chromaStride, cw, ch  (-1, -1, 0, 0, 0, 0 without chroma SATD),  mvc[2 * numCand] },
 *                  pixels = w * h (source PU) [+ cw * ch Cb + cw * ch Cr of the source PU]
 *   kind 3 (chroma plane snapshot of a reference picture, 4:2:0): ints and pixels as kind 1
 *   kind 4 (MotionEstimate::diamondSearch call, motion.cpp:631-773 -- the predictor stage of ThreadedME, search.cpp:362; recorded through the same
 *           renaming trick, -DdiamondSearch=diamondSearch_ref): ints = { planeId, w, h, blockOffset, mvmin.x, .y, mvmax.x, .y, mvp.x, .y (what setMVP
 *           was given: the MVD origin of mvcost), qp, out.x, out.y, cost }, pixels = w * h (source PU)
 *   kind 5 (CUData::getPMV call, cudata.cpp:1806-1990 -- the AMVP candidates and the motion-candidate list of a PU; recorded through -DgetPMV=getPMV_ref on
 *           common/cudata.cpp; at most X265TME_PMV calls, default 0 = none): ints = { list, refIdx, curPOC, temporalMvpEnabled, numRefIdx[2], refPOCList[2][16],
 *           6 neighbours x { mv0.x, mv0.y, mv1.x, mv1.y, refIdx0, refIdx1, cuAddr0, cuAddr1, isAvailable }, colPOC, colRefPOC (of the temporal candidate, 0 when unused),
 *           amvp0.x, .y, amvp1.x, .y, numMvc, mvc[2 * numMvc] }, no pixels
 *   kind 6 (Search::selectMVP call, search.cpp:2347-2382; at most X265TME_SEL calls): ints = { planeId, w, h, blockOffset, amvp0.x, .y, amvp1.x, .y, CUData::clipMv's xmin, ymin,
 *           xmax, ymax for this CU, frameParallel, result }, pixels = w * h (source PU as setSourcePU cached it)
 *   kind 7 (Search::checkBestMVP, search.cpp:4947-4958): ints = { amvp0.x, .y, amvp1.x, .y, mv.x, .y, mvpIdx, bits, cost, lambda lo, lambda hi, -> mvpIdx, bits, cost }
 *   kind 8 (Search::updateMVP, search.cpp:4961-4967): ints = { amvp.x, .y, mv.x, .y, alter.x, .y, bits, cost, lambda lo, lambda hi, -> bits, cost }
 *   (kinds 6-8: the callers live in search.cpp itself, so instead of a renamed second compile the regular search.o gets its three definitions weakened and
 *   aliased with objcopy -- oracle/Makefile -- and the strong definitions below take every call)
 *   kind 9 / 10: a whole Search::puMotionEstimation call with everything under it (see below; at most X265TME_PU calls)
 * With threaded-me=0 on the command line the calls are those of Search::predInterSearch (search.cpp:2582-2700), whose setSourcePU overload enables
 * the chroma SATD terms of subpelCompare (motion.cpp:218-247, 1805-1865) at subme >= 3.
 */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#define protected public
#define private public
#include "x265.h"
#include "common.h"
#include "primitives.h"
#include "picyuv.h"
#include "motion.h"
#include "cudata.h"
#include "slice.h"
#include "frame.h"
#include "framedata.h"
#include "search.h"
#include "threadedme.h"
#undef protected
#undef private

using namespace X265_NS;

static FILE* g_out;
static std::mutex g_lock;
struct Snap { int id; uint64_t sum; };
static std::map<const pixel*, Snap> g_planes;
static int g_nextPlane, g_calls, g_skipped;
/* heights of enum LumaPU (primitives.h:52-64), in its order */
static const int g_lumaH[25] = { 4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 12, 16, 4, 16, 24, 32, 8, 32, 48, 64, 16, 64 };

/* a Search::puMotionEstimation call being recorded on this thread (kind 9): what the functions under it record is collected here and written behind the
 * call's own record, in call order */
struct Composite { std::vector<int> kinds; std::vector<std::vector<int32_t>> ints; std::vector<std::vector<uint16_t>> px; };
static thread_local Composite* t_cur;
static void put(int kind, const std::vector<int32_t>& ints, const std::vector<uint16_t>& px)
{
    if (t_cur && kind != 1 && kind != 3) { t_cur->kinds.push_back(kind); t_cur->ints.push_back(ints); t_cur->px.push_back(px); return; }
    const int32_t hdr[2] = { kind, (int32_t)ints.size() };
    fwrite(hdr, 4, 2, g_out); fwrite(ints.data(), 4, ints.size(), g_out); fwrite(px.data(), 2, px.size(), g_out);
}


/* id of the luma plane of a reference picture, snapshotting it at first sight (or when its buffer holds another picture now); g_lock held */
static int luma_plane_id(const ReferencePlanes* ref)
{
    const PicYuv* rp = ref->reconPic;
    const intptr_t stride = ref->lumaStride;
    const int rows = rp->m_picHeight + 2 * rp->m_lumaMarginY;
    const pixel* top = ref->fpelPlane[0] - (intptr_t)rp->m_lumaMarginY * stride - rp->m_lumaMarginX;
    uint64_t sum = 1469598103934665603ull;
    for (intptr_t i = 0; i < stride * rows; i++) sum = (sum ^ top[i]) * 1099511628211ull;
    auto it = g_planes.find(ref->fpelPlane[0]);
    if (it == g_planes.end() || it->second.sum != sum)
    {
        Snap s = { g_nextPlane++, sum };
        g_planes[ref->fpelPlane[0]] = s;
        std::vector<uint16_t> px((size_t)stride * rows);
        for (size_t i = 0; i < px.size(); i++) px[i] = top[i];
        put(1, { s.id, (int32_t)stride, rows, (int32_t)(rp->m_lumaMarginY * stride + rp->m_lumaMarginX), (int32_t)rp->m_picWidth, (int32_t)rp->m_picHeight }, px);
        return s.id;
    }
    return it->second.id;
}
static int g_diamonds;

/* the reference's own body, compiled from encoder/motion.cpp as the member motionEstimate_ref (see the header comment); a member cannot be declared
 * outside its class, so it is reached as a function with `this` as first argument under the member's mangled name (Itanium ABI) */
#if X265_DEPTH == 8
#define XTME_REAL "_ZN4x26514MotionEstimate18motionEstimate_refEPNS_15ReferencePlanesERKNS_2MVES5_S5_iPS4_iRS3_jbPh"
#else
#define XTME_REAL "_ZN4x26514MotionEstimate18motionEstimate_refEPNS_15ReferencePlanesERKNS_2MVES5_S5_iPS4_iRS3_jbPt"
#endif
int motionEstimate_ref(MotionEstimate* self, ReferencePlanes* ref, const MV& mvmin, const MV& mvmax, const MV& qmvp, int numCandidates, const MV* mvc, int merange,
                       MV& outQMv, uint32_t maxSlices, bool m_vertRestriction, pixel* srcReferencePlane) __asm__(XTME_REAL);

namespace X265_NS {
int MotionEstimate::motionEstimate(ReferencePlanes* ref, const MV& mvmin, const MV& mvmax, const MV& qmvp, int numCandidates, const MV* mvc, int merange,
                                   MV& outQMv, uint32_t maxSlices, bool m_vertRestriction, pixel* srcReferencePlane)
{
    const int cost = ::motionEstimate_ref(this, ref, mvmin, mvmax, qmvp, numCandidates, mvc, merange, outQMv, maxSlices, m_vertRestriction, srcReferencePlane);
    if (!g_out || ref->isLowres || ref->isHMELowres) return cost;         /* the lookahead's searches on half-resolution pictures are another path */
    std::lock_guard<std::mutex> guard(g_lock);
    if (srcReferencePlane || !ref->reconPic) { g_skipped++; return cost; }
    const PicYuv* rp = ref->reconPic;
    const int lumaId = luma_plane_id(ref);
    int cbId = -1, crId = -1, chromaOff = 0, strideC = 0, cw = 0, ch = 0;
    if (bChromaSATD)
    {   /* the chroma planes of the reference picture, snapshot like the luma plane */
        strideC = (int)rp->m_strideC; cw = blockwidth >> fencPUYuv.m_hChromaShift; ch = g_lumaH[partEnum] >> fencPUYuv.m_vChromaShift;
        chromaOff = (int)rp->getChromaAddrOffset(ctuAddr, absPartIdx);
        const int rowsC = (rp->m_picHeight >> rp->m_vChromaShift) + 2 * rp->m_chromaMarginY;
        for (int c = 1; c <= 2; c++)
        {
            const pixel* topC = ref->fpelPlane[c] - (intptr_t)rp->m_chromaMarginY * strideC - rp->m_chromaMarginX;
            uint64_t sc = 1469598103934665603ull;
            for (intptr_t i = 0; i < (intptr_t)strideC * rowsC; i++) sc = (sc ^ topC[i]) * 1099511628211ull;
            auto ic = g_planes.find(ref->fpelPlane[c]);
            if (ic == g_planes.end() || ic->second.sum != sc)
            {
                Snap sn = { g_nextPlane++, sc };
                g_planes[ref->fpelPlane[c]] = sn;
                std::vector<uint16_t> pc((size_t)strideC * rowsC);
                for (size_t i = 0; i < pc.size(); i++) pc[i] = topC[i];
                put(3, { sn.id, strideC, rowsC, (int32_t)(rp->m_chromaMarginY * strideC + rp->m_chromaMarginX), (int32_t)(rp->m_picWidth >> rp->m_hChromaShift),
                         (int32_t)(rp->m_picHeight >> rp->m_vChromaShift) }, pc);
                ic = g_planes.find(ref->fpelPlane[c]);
            }
            (c == 1 ? cbId : crId) = ic->second.id;
        }
    }
    int qp = -1;
    for (int q = 0; q < BC_MAX_QP; q++)
        if (s_costs[q] && s_costs[q] == m_cost) qp = q;
    const int blockh = (int)(g_lumaH[partEnum]);     /* setSourcePU never sets blockheight (motion.cpp:167-247): the height follows from the partition enum */
    std::vector<int32_t> ints = { lumaId, blockwidth, blockh, (int32_t)blockOffset, mvmin.x, mvmin.y, mvmax.x, mvmax.y, qmvp.x, qmvp.y, numCandidates, merange,
                                  searchMethod, subpelRefine, qp, (int32_t)bChromaSATD, (int32_t)maxSlices, (int32_t)m_vertRestriction, srcReferencePlane ? 1 : 0,
                                  outQMv.x, outQMv.y, cost, (int32_t)mvcost(outQMv), cbId, crId, chromaOff, strideC, cw, ch };
    for (int i = 0; i < numCandidates; i++) { ints.push_back(mvc[i].x); ints.push_back(mvc[i].y); }
    std::vector<uint16_t> px((size_t)blockwidth * blockh);
    for (int y = 0; y < blockh; y++)
        for (int x = 0; x < blockwidth; x++) px[(size_t)y * blockwidth + x] = fencPUYuv.m_buf[0][y * FENC_STRIDE + x];
    for (int c = 1; c <= 2 && bChromaSATD; c++)
        for (int y = 0; y < ch; y++)
            for (int x = 0; x < cw; x++) px.push_back(fencPUYuv.m_buf[c][y * fencPUYuv.m_csize + x]);
    put(2, ints, px);
    g_calls++;
    return cost;
}
}


#define XTME_DIAMOND "_ZN4x26514MotionEstimate17diamondSearch_refEPNS_15ReferencePlanesERKNS_2MVES5_RS3_"
int diamondSearch_ref(MotionEstimate* self, ReferencePlanes* ref, const MV& mvmin, const MV& mvmax, MV& outMV) __asm__(XTME_DIAMOND);
namespace X265_NS {
int MotionEstimate::diamondSearch(ReferencePlanes* ref, const MV& mvmin, const MV& mvmax, MV& outMV)
{
    const int cost = ::diamondSearch_ref(this, ref, mvmin, mvmax, outMV);
    if (!g_out || ref->isLowres || ref->isHMELowres || !ref->reconPic) return cost;
    std::lock_guard<std::mutex> guard(g_lock);
    const int lumaId = luma_plane_id(ref);
    int qp = -1;
    for (int q = 0; q < BC_MAX_QP; q++)
        if (s_costs[q] && s_costs[q] == m_cost) qp = q;
    const int blockh = (int)(g_lumaH[partEnum]);
    std::vector<int32_t> ints = { lumaId, blockwidth, blockh, (int32_t)blockOffset, mvmin.x, mvmin.y, mvmax.x, mvmax.y, m_mvp.x, m_mvp.y, qp, outMV.x, outMV.y, cost };
    std::vector<uint16_t> px((size_t)blockwidth * blockh);
    for (int y = 0; y < blockh; y++)
        for (int x = 0; x < blockwidth; x++) px[(size_t)y * blockwidth + x] = fencPUYuv.m_buf[0][y * FENC_STRIDE + x];
    put(4, ints, px);
    g_diamonds++;
    return cost;
}
}

/* CUData::getPMV under the recorder (common/cudata.cpp compiled with -DgetPMV=getPMV_ref, oracle/Makefile) */
static int g_pmvCalls, g_pmvMax;
int getPMV_ref(const CUData* self, InterNeighbourMV* neighbours, uint32_t picList, uint32_t refIdx, MV* amvpCand, MV* pmv) __asm__("_ZNK4x2656CUData10getPMV_refEPNS_16InterNeighbourMVEjjPNS_2MVES4_");
namespace X265_NS {
int CUData::getPMV(InterNeighbourMV* neighbours, uint32_t picList, uint32_t refIdx, MV* amvpCand, MV* pmv) const
{
    InterNeighbourMV in[6];
    memcpy(in, neighbours, sizeof(in));
    const int numMvc = ::getPMV_ref(this, neighbours, picList, refIdx, amvpCand, pmv);
    if (!g_out || (g_pmvCalls >= g_pmvMax && !t_cur)) return numMvc;
    std::lock_guard<std::mutex> guard(g_lock);
    std::vector<int32_t> ints = { (int32_t)picList, (int32_t)refIdx, m_slice->m_poc, (int32_t)m_slice->m_sps->bTemporalMVPEnabled, m_slice->m_numRefIdx[0], m_slice->m_numRefIdx[1] };
    for (int l = 0; l < 2; l++)
        for (int r = 0; r < 16; r++) ints.push_back(m_slice->m_refPOCList[l][r]);
    for (int d = 0; d < 6; d++)
    {
        ints.push_back(in[d].mv[0].x); ints.push_back(in[d].mv[0].y); ints.push_back(in[d].mv[1].x); ints.push_back(in[d].mv[1].y);
        ints.push_back(in[d].refIdx[0]); ints.push_back(in[d].refIdx[1]); ints.push_back((int32_t)in[d].cuAddr[0]); ints.push_back((int32_t)in[d].cuAddr[1]);
        ints.push_back(in[d].isAvailable ? 1 : 0);
    }
    int colPOC = 0, colRefPOC = 0;
    const int tempRefIdx = in[MD_COLLOCATED].refIdx[picList];
    if (m_slice->m_sps->bTemporalMVPEnabled && tempRefIdx != -1)
    {   /* what :1962-1970 reads for the scaling of the temporal candidate */
        const Frame* colPic = m_slice->m_refFrameList[m_slice->isInterB() && !m_slice->m_colFromL0Flag][m_slice->m_colRefIdx];
        const CUData* colCU = colPic->m_encData->getPicCTU(in[MD_COLLOCATED].cuAddr[picList]);
        colRefPOC = colCU->m_slice->m_refPOCList[tempRefIdx >> 4][tempRefIdx & 0xf];
        colPOC = colCU->m_slice->m_poc;
    }
    ints.push_back(colPOC); ints.push_back(colRefPOC);
    ints.push_back(amvpCand[0].x); ints.push_back(amvpCand[0].y); ints.push_back(amvpCand[1].x); ints.push_back(amvpCand[1].y);
    ints.push_back(numMvc);
    for (int i = 0; i < numMvc; i++) { ints.push_back(pmv[i].x); ints.push_back(pmv[i].y); }
    put(5, ints, {});
    g_pmvCalls++;
    return numMvc;
}
}

/* id of a reconstructed picture's luma plane (as luma_plane_id, keyed by the plane pointer) */
static int recon_plane_id(const PicYuv* rp)
{
    const intptr_t stride = rp->m_stride;
    const int rows = rp->m_picHeight + 2 * rp->m_lumaMarginY;
    const pixel* org = rp->m_picOrg[0];
    const pixel* top = org - (intptr_t)rp->m_lumaMarginY * stride - rp->m_lumaMarginX;
    uint64_t sum = 1469598103934665603ull;
    for (intptr_t i = 0; i < stride * rows; i++) sum = (sum ^ top[i]) * 1099511628211ull;
    auto it = g_planes.find(org);
    if (it == g_planes.end() || it->second.sum != sum)
    {
        Snap s = { g_nextPlane++, sum };
        g_planes[org] = s;
        std::vector<uint16_t> px((size_t)stride * rows);
        for (size_t i = 0; i < px.size(); i++) px[i] = top[i];
        put(1, { s.id, (int32_t)stride, rows, (int32_t)(rp->m_lumaMarginY * stride + rp->m_lumaMarginX), (int32_t)rp->m_picWidth, (int32_t)rp->m_picHeight }, px);
        return s.id;
    }
    return it->second.id;
}
static int g_selCalls, g_selMax, g_chkCalls, g_updCalls, g_dbgMask = 7;
int selectMVP_ref(Search* self, const CUData& cu, const PredictionUnit& pu, const MV* amvp, int list, int ref) __asm__("xtme_selectMVP");
const MV& checkBestMVP_ref(const Search* self, const MV* amvpCand, const MV& mv, int& mvpIdx, uint32_t& outBits, uint32_t& outCost) __asm__("xtme_checkBestMVP");
void updateMVP_ref(Search* self, const MV amvp, const MV& mv, uint32_t& outBits, uint32_t& outCost, const MV& alterMVP) __asm__("xtme_updateMVP");
namespace X265_NS {
int Search::selectMVP(const CUData& cu, const PredictionUnit& pu, const MV amvp[AMVP_NUM_CANDS], int list, int ref)
{
    const int idx = ::selectMVP_ref(this, cu, pu, amvp, list, ref);
    if (!g_out || (g_selCalls >= g_selMax && !t_cur) || amvp[0] == amvp[1] || !(g_dbgMask & 1)) return idx;
    std::lock_guard<std::mutex> guard(g_lock);
    const PicYuv* rp = m_slice->m_refReconPicList[list][ref];
    const int planeId = recon_plane_id(rp);
    const int blockOffset = (int)(rp->getLumaAddr(pu.ctuAddr, pu.cuAbsPartIdx + pu.puAbsPartIdx) - rp->getLumaAddr(0));
    MV lo(-0x40000000, -0x40000000), hi(0x3fffffff, 0x3fffffff);
    cu.clipMv(lo); cu.clipMv(hi);                       /* the limits CUData::clipMv applies for this CU */
    std::vector<int32_t> ints = { planeId, pu.width, pu.height, blockOffset, amvp[0].x, amvp[0].y, amvp[1].x, amvp[1].y, lo.x, lo.y, hi.x, hi.y, (int32_t)m_bFrameParallel, idx };
    std::vector<uint16_t> px((size_t)pu.width * pu.height);
    for (int y = 0; y < pu.height; y++)
        for (int x = 0; x < pu.width; x++) px[(size_t)y * pu.width + x] = m_me.fencPUYuv.m_buf[0][y * FENC_STRIDE + x];
    put(6, ints, px);
    g_selCalls++;
    return idx;
}
const MV& Search::checkBestMVP(const MV* amvpCand, const MV& mv, int& mvpIdx, uint32_t& outBits, uint32_t& outCost) const
{
    const int i0 = mvpIdx; const uint32_t b0 = outBits, c0 = outCost;
    const MV& r = ::checkBestMVP_ref(this, amvpCand, mv, mvpIdx, outBits, outCost);
    if (g_out && (g_chkCalls < g_selMax || t_cur) && (g_dbgMask & 2))
    {
        std::lock_guard<std::mutex> guard(g_lock);
        put(7, { amvpCand[0].x, amvpCand[0].y, amvpCand[1].x, amvpCand[1].y, mv.x, mv.y, i0, (int32_t)b0, (int32_t)c0, (int32_t)(m_rdCost.m_lambda & 0xffffffffu), (int32_t)(m_rdCost.m_lambda >> 32),
                 mvpIdx, (int32_t)outBits, (int32_t)outCost }, {});
        g_chkCalls++;
    }
    return r;
}
void Search::updateMVP(const MV amvp, const MV& mv, uint32_t& outBits, uint32_t& outCost, const MV& alterMVP)
{
    const uint32_t b0 = outBits, c0 = outCost;
    ::updateMVP_ref(this, amvp, mv, outBits, outCost, alterMVP);
    if (g_out && (g_updCalls < g_selMax || t_cur) && (g_dbgMask & 4))
    {
        std::lock_guard<std::mutex> guard(g_lock);
        put(8, { amvp.x, amvp.y, mv.x, mv.y, alterMVP.x, alterMVP.y, (int32_t)b0, (int32_t)c0, (int32_t)(m_rdCost.m_lambda & 0xffffffffu), (int32_t)(m_rdCost.m_lambda >> 32), (int32_t)outBits, (int32_t)outCost }, {});
        g_updCalls++;
    }
}
}

/* ---- kind 9: a whole Search::puMotionEstimation call of the PU stage (isMVP = false), search.cpp:226-556 ----
 * ints = { 1 (layout version), isInterP, numRefIdx[2], curPOC, temporalMvpEnabled, refPOCList[2][16], part, log2CUSize, cuPelX, cuPelY (the CU), puOffset, areaIdx, finalIdx,
 *          neighborIdx[5], searchRange, searchMethod, subpelRefine, lambda lo, lambda hi, picWidth, picHeight, maxCUSize, numPart, maxSlices,
 *          m_areaBestMV[areaIdx][list][ref] x, y for list 0..1, ref 0..3 (16 ints),
 *          the 5 neighbour MEData records as the call finds them: mv[0].x, .y, mv[1].x, .y, ref[0], ref[1] (30 ints; -2 in ref[0] = no such neighbour),
 *          per partition: pos, then the MEData the call left at pos: mv[0], mv[1], mvp[0], mvp[1] (8 ints), mvCost[2], ref[2], bits, cost,
 *          per partition, list, reference 0..3: the reference FRAME's MEData at pos (mv[0].x, .y, mv[1].x, .y, ref[0], ref[1]; ref[0] = -3: that frame is intra / absent),
 *          per partition: x, y (luma position in the picture), w, h;  per list, reference 0..3: id of the plane motionEstimate searches (slice->m_mref[l][r], weighted or not) and of the
 *          reconstructed picture selectMVP / the bidirectional candidate read (m_refReconPicList[l][r]); -1 = no such reference,
 *          number of records that follow }, pixels = the partitions' source blocks (w * h each)
 * followed by that many records of kinds 2, 5, 6, 7, 8, 10 in call order -- everything the call did.  kind 10 = Search::getLowresMV: { list, ref, result.x, .y } */
static int g_puCalls, g_puMax;
void puMotionEstimation_real(Search* self, const Slice* slice, const CUGeom& cuGeom, CUData& cu, PicYuv* fencPic, int puOffset, PartSize part, int areaIdx, int finalIdx, bool isMVP,
                             const int* neighborIdx) __asm__("xtme_puMotionEstimation");
MV getLowresMV_real(Search* self, const CUData& cu, const PredictionUnit& pu, int list, int ref) __asm__("xtme_getLowresMV");
namespace X265_NS {
MV Search::getLowresMV(const CUData& cu, const PredictionUnit& pu, int list, int ref)
{
    const MV r = ::getLowresMV_real(this, cu, pu, list, ref);
    if (t_cur) { std::lock_guard<std::mutex> guard(g_lock); put(10, { list, ref, r.x, r.y }, {}); }
    return r;
}
void Search::puMotionEstimation(const Slice* slice, const CUGeom& cuGeom, CUData& cu, PicYuv* fencPic, int puOffset, PartSize part, int areaIdx, int finalIdx, bool isMVP, const int* neighborIdx)
{
    if (isMVP || !g_out || g_puCalls >= g_puMax) { ::puMotionEstimation_real(this, slice, cuGeom, cu, fencPic, puOffset, part, areaIdx, finalIdx, isMVP, neighborIdx); return; }
    const int numCols = m_slice->m_sps->numCuInWidth;
    const int slotIdx = (cu.m_cuAddr / numCols) * numCols + cu.m_cuAddr % numCols;
    std::vector<int32_t> ints = { 1, (int32_t)slice->isInterP(), slice->m_numRefIdx[0], slice->m_numRefIdx[1], slice->m_poc, (int32_t)slice->m_sps->bTemporalMVPEnabled };
    for (int l = 0; l < 2; l++) for (int r = 0; r < 16; r++) ints.push_back(slice->m_refPOCList[l][r]);
    const int numPart = (int)cu.getNumPartInter(0);
    for (int32_t v : { (int32_t)part, (int32_t)cuGeom.log2CUSize, (int32_t)(cu.m_cuPelX), (int32_t)(cu.m_cuPelY), puOffset, areaIdx, finalIdx,
                       neighborIdx[0], neighborIdx[1], neighborIdx[2], neighborIdx[3], neighborIdx[4], m_param->searchRange, m_param->searchMethod, m_param->subpelRefine,
                       (int32_t)(m_rdCost.m_lambda & 0xffffffffu), (int32_t)(m_rdCost.m_lambda >> 32), (int32_t)m_slice->m_sps->picWidthInLumaSamples, (int32_t)m_slice->m_sps->picHeightInLumaSamples,
                       (int32_t)m_param->maxCUSize, numPart, (int32_t)m_param->maxSlices }) ints.push_back(v);
    for (int l = 0; l < 2; l++) for (int r = 0; r < 4; r++) { ints.push_back(m_areaBestMV[areaIdx][l][r].x); ints.push_back(m_areaBestMV[areaIdx][l][r].y); }
    for (int d = 0; d < 5; d++)
    {
        if (neighborIdx[d] >= 0)
        {
            const MEData& nd = slice->m_ctuMV[slotIdx * MAX_NUM_PUS_PER_CTU + neighborIdx[d]];
            for (int32_t v : { nd.mv[0].x, nd.mv[0].y, nd.mv[1].x, nd.mv[1].y, nd.ref[0], nd.ref[1] }) ints.push_back(v);
        }
        else for (int32_t v : { 0, 0, 0, 0, -2, -2 }) ints.push_back(v);
    }
    /* the reference frames' own records at the partitions' slots (the fallback predictor of :313-330), read before the call */
    std::vector<int32_t> refRec;
    for (int pi = 0; pi < numPart; pi++)
        for (int l = 0; l < 2; l++)
            for (int r = 0; r < 4; r++)
            {
                const Frame* rf = (l < (slice->isInterP() ? 1 : 2) && r < slice->m_numRefIdx[l]) ? slice->m_refFrameList[l][r] : NULL;
                if (rf && rf->m_encData->m_slice->m_sliceType != I_SLICE && rf->m_encData->m_slice->m_ctuMV)
                {
                    const MEData& md = rf->m_encData->m_slice->m_ctuMV[slotIdx * MAX_NUM_PUS_PER_CTU + finalIdx + pi * puOffset];
                    for (int32_t v : { md.mv[0].x, md.mv[0].y, md.mv[1].x, md.mv[1].y, md.ref[0], md.ref[1] }) refRec.push_back(v);
                }
                else for (int32_t v : { 0, 0, 0, 0, -3, -3 }) refRec.push_back(v);
            }
    std::vector<int32_t> geo; std::vector<uint16_t> blocks;
    for (int pi = 0; pi < numPart; pi++)
    {
        PredictionUnit pu(cu, cuGeom, pi);
        const int x = cu.m_cuPelX + g_zscanToPelX[pu.puAbsPartIdx], y = cu.m_cuPelY + g_zscanToPelY[pu.puAbsPartIdx];
        for (int32_t v : { x, y, pu.width, pu.height }) geo.push_back(v);
        const pixel* src = fencPic->m_picOrg[0] + (intptr_t)y * fencPic->m_stride + x;
        for (int yy = 0; yy < pu.height; yy++) for (int xx = 0; xx < pu.width; xx++) blocks.push_back(src[(intptr_t)yy * fencPic->m_stride + xx]);
    }
    {
        std::lock_guard<std::mutex> guard(g_lock);
        for (int l = 0; l < 2; l++)
            for (int r = 0; r < 4; r++)
            {
                const bool on = l < (slice->isInterP() ? 1 : 2) && r < slice->m_numRefIdx[l];
                geo.push_back(on ? luma_plane_id(&slice->m_mref[l][r]) : -1);
                geo.push_back(on ? recon_plane_id(slice->m_refReconPicList[l][r]) : -1);
            }
    }
    Composite c;
    t_cur = &c;
    ::puMotionEstimation_real(this, slice, cuGeom, cu, fencPic, puOffset, part, areaIdx, finalIdx, isMVP, neighborIdx);
    t_cur = nullptr;
    for (int pi = 0; pi < numPart; pi++)
    {
        const int pos = finalIdx + pi * puOffset;
        const MEData& o = slice->m_ctuMV[slotIdx * MAX_NUM_PUS_PER_CTU + pos];
        for (int32_t v : { pos, o.mv[0].x, o.mv[0].y, o.mv[1].x, o.mv[1].y, o.mvp[0].x, o.mvp[0].y, o.mvp[1].x, o.mvp[1].y, (int32_t)o.mvCost[0], (int32_t)o.mvCost[1], o.ref[0], o.ref[1], o.bits, (int32_t)o.cost })
            ints.push_back(v);
    }
    ints.insert(ints.end(), refRec.begin(), refRec.end());
    ints.insert(ints.end(), geo.begin(), geo.end());
    ints.push_back((int32_t)c.kinds.size());
    std::lock_guard<std::mutex> guard(g_lock);
    if (g_puCalls >= g_puMax) return;
    put(9, ints, blocks);
    for (size_t i = 0; i < c.kinds.size(); i++) put(c.kinds[i], c.ints[i], c.px[i]);
    g_puCalls++;
}
}

static void synth(std::vector<pixel>& y, std::vector<pixel>& u, std::vector<pixel>& v, int w, int h, int f)
{   /* textured picture in (not purely translational) motion + deterministic noise: predictors, candidates and search paths vary from PU to PU */
    uint32_t s = 4242u + 733u * (uint32_t)f;
    const int pm = (1 << X265_DEPTH) - 1;
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++)
        {
            const int x = i + 5 * f + ((j >> 5) & 1) * f, yy = j + 3 * f;
            const int t = (((x * x) / 9 + yy * 7 + (x * yy) / 13 + ((x >> 3) ^ (yy >> 3)) * 11) & 255) * (pm + 1) / 256;
            s = s * 1664525u + 1013904223u;
            const int val = t + (int)((s >> 24) & 7) - 3;
            y[(size_t)j * w + i] = (pixel)(val < 0 ? 0 : val > pm ? pm : val);
        }
    for (int j = 0; j < h / 2; j++)
        for (int i = 0; i < w / 2; i++)
        {
            u[(size_t)j * (w / 2) + i] = (pixel)((((i + f) * 3 + j) & 127) * (pm + 1) / 256 + (pm + 1) / 4);
            v[(size_t)j * (w / 2) + i] = (pixel)((((j + 2 * f) * 5 + i) & 127) * (pm + 1) / 256 + (pm + 1) / 4);
        }
}

int main(int argc, char** argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s width height frames preset out.bin [option=value ...]\n", argv[0]); return 2; }
    const int w = atoi(argv[1]), h = atoi(argv[2]), frames = atoi(argv[3]);
    x265_param* p = x265_param_alloc();
    if (x265_param_default_preset(p, argv[4], NULL) < 0) { fprintf(stderr, "bad preset\n"); return 2; }
    p->sourceWidth = w; p->sourceHeight = h; p->fpsNum = 25; p->fpsDenom = 1; p->internalCsp = X265_CSP_I420;
    p->totalFrames = frames; p->logLevel = X265_LOG_WARNING; p->bRepeatHeaders = 1;
    p->frameNumThreads = 1; p->bEnableWavefront = 0; p->lookaheadSlices = 0;
    x265_param_parse(p, "pools", "32");                /* ThreadedME needs a worker pool of at least MIN_TME_THREADS threads (encoder.cpp:273-306, threadpool.cpp:288-296) */
    x265_param_parse(p, "threaded-me", "1");
    for (int i = 6; i < argc; i++)
    {
        char* eq = strchr(argv[i], '=');
        if (eq) *eq = 0;
        if (x265_param_parse(p, argv[i], eq ? eq + 1 : NULL) < 0) { fprintf(stderr, "bad option %s\n", argv[i]); return 2; }
    }
    g_puMax = getenv("X265TME_PU") ? atoi(getenv("X265TME_PU")) : 0;
    g_selMax = getenv("X265TME_SEL") ? atoi(getenv("X265TME_SEL")) : 0;
    g_dbgMask = getenv("X265TME_DBG") ? atoi(getenv("X265TME_DBG")) : 7;
    g_pmvMax = getenv("X265TME_PMV") ? atoi(getenv("X265TME_PMV")) : 0;
    g_out = fopen(argv[5], "wb");
    if (!g_out) { fprintf(stderr, "cannot write %s\n", argv[5]); return 2; }
    x265_encoder* enc = x265_encoder_open(p);
    if (!enc) { fprintf(stderr, "encoder_open failed\n"); return 2; }
    x265_picture* pic = x265_picture_alloc();
    x265_picture_init(p, pic);
    std::vector<pixel> Y((size_t)w * h), U((size_t)w * h / 4), V((size_t)w * h / 4);
    pic->planes[0] = Y.data(); pic->planes[1] = U.data(); pic->planes[2] = V.data();
    pic->stride[0] = w * (int)sizeof(pixel); pic->stride[1] = pic->stride[2] = (w / 2) * (int)sizeof(pixel);
    pic->bitDepth = X265_DEPTH; pic->colorSpace = X265_CSP_I420;
    x265_nal* nal; uint32_t nnal;
    for (int f = 0; f < frames; f++)
    {
        synth(Y, U, V, w, h, f);
        pic->pts = f;
        if (x265_encoder_encode(enc, &nal, &nnal, pic, NULL) < 0) { fprintf(stderr, "encode failed\n"); return 2; }
    }
    while (x265_encoder_encode(enc, &nal, &nnal, NULL, NULL) > 0) {}
    x265_param* live = x265_param_alloc();
    x265_encoder_parameters(enc, live);                /* the encoder's own copy: threaded-me is switched off there when the pool is too small */
    const int tme = live->bThreadedME;
    x265_param_free(live);
    x265_encoder_close(enc); x265_picture_free(pic); x265_param_free(p);
    fclose(g_out);
    printf("{\"calls\": %d, \"diamond_calls\": %d, \"pmv_calls\": %d, \"select_calls\": %d, \"check_calls\": %d, \"update_calls\": %d, \"pu_calls\": %d, \"planes\": %d, \"skipped\": %d, \"threaded_me\": %d}\n", g_calls, g_diamonds, g_pmvCalls, g_selCalls, g_chkCalls, g_updCalls, g_puCalls, g_nextPlane, g_skipped, tme);
    return 0;
}
