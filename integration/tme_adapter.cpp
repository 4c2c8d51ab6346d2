/*
 * tme_adapter.cpp -- Analysis::deriveMVsForCTU with the GPU as ThreadedME's producer (see tme_adapter.h).
 *
 * What the adapter reads out of the encoder's state for the picture, and how: the planes (PicYuv allocations), the table as FrameData::reinit left it, per
 * reference the reference picture's own table and the lookahead's MVs (Lowres::lowresMvs), and -- through the encoder's own member functions, on its own CUData
 * objects, in computeMVForPUs' order -- the qp of every CU (Analysis::calculateQpforCuSize), the collocated neighbour of every PU (CUData::getNeighbourMV), the
 * collocated median of every CTU (CUData::getMedianColMV).
 *
 * Frame threads (the encoder's default, encoder.cpp:285): a picture starts while its references are still being coded.  The frame encoder releases a picture's CTU rows to
 * ThreadedME one by one, row r once every reference has reconstructed (and weighted) its rows up to r + m_refLagRows (frameencoder.cpp:166-171, 975-990, 1029-1043;
 * threadedme.cpp:121-150).  The adapter follows that protocol: a JOB is a band of CTU rows of one picture -- from the first row the picture has no records for up to the last row
 * whose reference rows are final now (Frame::m_reconRowFlag, MotionReference::numSliceWeightedRows) -- and the producer gets every reference with the count of rows that are
 * final (x265hip_tme_host_ref::reconRowsValid / meRowsValid): it uploads and phase-interpolates a keyed plane incrementally.  The window and selectMVP restrictions of
 * m_bFrameParallel / m_refLagPixels are the producer's (desc.frameThreads).  With one frame thread every reference is complete and the band is the whole picture.
 *
 * Preconditions (checked): numRefIdx <= X265HIP_MAX_REF = MAX_NUM_REF.  Handed back to the encoder's own body (with a line on stderr): --slices with several frame threads (the
 * reference's ThreadedME reads uninitialised slice MV bounds there: nothing defined to reproduce) and --me sea (the producer keeps no integral planes).
 */
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <thread>
#include <vector>
#include "x265.h"
#include "common.h"
#include "primitives.h"
#include "picyuv.h"
#include "frame.h"
#include "framedata.h"
#include "slice.h"
#include "cudata.h"
#include "lowres.h"
#include "search.h"
#include "analysis.h"
#include "threadedme.h"
#include "../include/x265hip_ctx.h"
#include "tme_adapter.h"

using namespace X265_NS;

namespace {
struct Api
{
    int (*ctx_create)(int, x265hip_ctx**);
    void (*ctx_destroy)(x265hip_ctx*);
    int (*tme_create)(x265hip_ctx*, int, int, int, int, int, int, x265hip_tme**);
    void (*tme_destroy)(x265hip_tme*);
    int (*tme_entries)(const x265hip_tme*, const x265hip_tme_step**);
    int (*tme_picture)(x265hip_tme*, const x265hip_tme_picture_desc*);
    const char* (*last_error)();
} g_api;
void* g_lib;
/* ONE producer: jobs -- bands of CTU rows, of whatever picture -- take turns on it.  (A producer per picture in flight was built and measured in round 5: no faster, the
   producer calls of different host threads slow each other down in the HIP runtime -- profiles/r05_m2_lanes.txt.) */
x265hip_ctx* g_ctx;
x265hip_tme* g_tme;
int g_useGpu, g_device, g_pictures, g_weighted, g_keepPlanes = 1;
double g_sec[4];      /* per encode: [0] job set-up seconds (incl. creating the producer on the first picture), [1] wall seconds up to the producer call (set-up + harvest: qps,
                         collocated neighbours, medians, table conversions -- spread over the workers), [2] CTUs harvested by workers other than the leader, [3] write-back seconds */
double g_createSeconds;                     /* creating the producer (context, streams, device buffers) on the first picture */
double g_gpuSeconds, g_pictureSeconds;      /* inside x265hip_tme_picture; the whole producer call incl. the adapter's harvest and write-back */
double g_gpuSecondsWarm; int g_callsWarm;   /* the same without the first four calls (code objects are loaded by the first launch of each kernel) */
int g_calls;
std::mutex g_lock, g_statLock;
struct PicState { int poc1 = 0, rowsDone = 0, claimed = 0; };      /* POC + 1 of the picture the Frame object holds now; its CTU rows [0, rowsDone) have their records, [rowsDone, claimed) belong to jobs in flight */
std::map<const Frame*, PicState> g_pics;
int g_bands;                                 /* jobs (bands of CTU rows) run; == g_pictures with one frame thread */
int g_trace;                                 /* X265TME_TRACE=1: a line per job on stderr (where a stalled encode stands) */
/* Frame threads: a band that would hold fewer than g_minRows rows (0 = half the picture's CTU rows) waits up to g_waitUs for the references to release another row.  A producer call costs the host a job set-up, a harvest and three
   rounds of wake-ups whatever it holds, and the row that asked cannot start before the rows above it have anyway: fewer, larger calls.  Measured at 1080p medium, five frame
   threads, WPP (profiles/r05_min_rows_ab.txt, five runs each): no wait 7.0 fps, 4 rows / 6 ms 7.9, 6 / 10 8.25, 8 / 16 8.4, 12 / 30 8.3 -- the encoder without --threaded-me 8.0.
   X265TME_MIN_ROWS / X265TME_WAIT_US override (1 / 0: every ready row at once, the first form) */
int g_minRows = 0, g_waitUs = 16000;
int g_help = -1;                             /* workers that arrive for a running band take CTUs of its host passes: -1 = with one frame thread only (there the ThreadedME workers have
                                                nothing else to do and the picture's harvest is 8 ms on one of them); with frame threads they sleep until the job ends -- waking them
                                                for a band's millisecond of host work costs more than it gives (profiles/r05_queues_ab.txt: 9.40 -> 9.74 fps).  X265TME_HELP=0 / 1 override */
int g_ahead;                                 /* X265TME_AHEAD=1: a band may be set up and harvested while the band before it is in its producer call (two jobs in flight; the producer still
                                                sees one call at a time, in the jobs' order).  Off: one job at a time, the form every figure of round 5 was measured with.  The host half
                                                is checked in both forms on the CPU (tests/test_tme_adapter_cpu.py); what it does to the encode's speed is for the GPU box to say */
int g_waitRefs;                              /* X265TME_WAIT_REFS=1 (diagnosis; unweighted references only): a picture waits for its references to be complete and goes through the
                                                producer whole -- separates the frame-parallel window rules from the band protocol */

/* a producer call failed (or the schedule does not match): the encode cannot go on and must not go on quietly.  The encoder's pool threads are running: exit() would run the
   static destructors under them (seen: a hang until the caller's timeout) -- leave at once, the message is on stderr */
[[noreturn]] void die() { fflush(stdout); fflush(stderr); _Exit(3); }

void to_choice(const MEData& m, x265hip_inter_choice& o)
{
    for (int l = 0; l < 2; l++)
    {
        o.mv[l][0] = (int16_t)m.mv[l].x; o.mv[l][1] = (int16_t)m.mv[l].y; o.mvp[l][0] = (int16_t)m.mvp[l].x; o.mvp[l][1] = (int16_t)m.mvp[l].y;
        o.mvCost[l] = m.mvCost[l]; o.ref[l] = (int8_t)m.ref[l];
    }
    o.reserved = 0; o.bits = m.bits; o.cost = m.cost;
}
void from_choice(const x265hip_inter_choice& o, MEData& m)
{
    for (int l = 0; l < 2; l++)
    {
        m.mv[l] = MV(o.mv[l][0], o.mv[l][1]); m.mvp[l] = MV(o.mvp[l][0], o.mvp[l][1]); m.mvCost[l] = o.mvCost[l]; m.ref[l] = o.ref[l];
    }
    m.bits = o.bits; m.cost = o.cost;
}
} // namespace

void deriveMVsForCTU_cpu(Analysis* self, CUData& ctu, const CUGeom& cuGeom, Frame& frame) __asm__("_ZN4x2658Analysis19deriveMVsForCTU_cpuERNS_6CUDataERKNS_6CUGeomERNS_5FrameE");      /* the encoder's own body, under its second name */

namespace X265_NS {
void Analysis::deriveMVsForCTU(CUData& ctu, const CUGeom& cuGeom, Frame& frame)
{
    if (!g_useGpu) { ::deriveMVsForCTU_cpu(this, ctu, cuGeom, frame); return; }
    if (frame.m_param->frameNumThreads > 1 && frame.m_param->maxSlices > 1)
    {   /* --slices with several frame threads: setSearchRange and selectMVP then read Search::m_sliceMinY / m_sliceMaxY (search.cpp:2367-2369, 4999-5003), which only
           FrameEncoder::processRowEncoder ever sets, and only on the FRAME ENCODER's Analysis objects (frameencoder.cpp:1624-1629).  ThreadedME's workers have their own
           (threadedme.cpp:69-73), whose two members no constructor and no caller initialises: the reference's own producer searches with whatever the allocation held.  There is
           no defined behaviour to reproduce, so the encoder's own body runs (with this worker's members, exactly as without the binding) and the binding says so */
        static std::atomic<int> told{0};
        if (!told.exchange(1)) fprintf(stderr, "tme_adapter: --slices with several frame threads: the encoder's own ThreadedME producer runs (its slice MV bounds are uninitialised members)\n");
        ::deriveMVsForCTU_cpu(this, ctu, cuGeom, frame);
        return;
    }
    if (frame.m_param->searchMethod == X265_SEA)
    {   /* the producer searches DIA / HEX / UMH / STAR / FULL (x265hip_me_batch_rows; the chain kernels).  SEA needs the reference's twelve integral planes, which the encoder
           builds row by row in its filter pipeline (FrameFilter::computeMEIntegral, framefilter.cpp:793-833) and the producer does not keep (the batched SEA search itself
           exists: x265hip_me_batch_sea, csrc/kern_me_sea.hip) -- the encoder's own body runs, under any threading */
        static std::atomic<int> told{0};
        if (!told.exchange(1)) fprintf(stderr, "tme_adapter: --me sea: the encoder's own ThreadedME producer runs\n");
        ::deriveMVsForCTU_cpu(this, ctu, cuGeom, frame);
        return;
    }
    /* local classes of a member function have the member's access: calculateQpforCuSize (protected) is reached without touching analysis.h */
    struct Harvest
    {
        Analysis& an; const Slice* slice; Frame& frame; const x265hip_tme_step* steps; int nSteps;
        std::vector<x265hip_tme_temporal>& temporal; std::vector<int>& entryQp;
        int ctuAddr, k;
        void collocated(const CUData& cu, int puIdx, uint32_t puAbsPartIdx, x265hip_tme_temporal& t)
        {
            InterNeighbourMV nb[6];
            cu.getNeighbourMV(puIdx, puAbsPartIdx, nb);
            memset(&t, 0, sizeof(t));
            t.nb.refIdx[0] = t.nb.refIdx[1] = -1;
            if (nb[MD_COLLOCATED].unifiedRef == -1) return;
            for (int l = 0; l < 2; l++)
            {
                t.nb.mv[l][0] = (int16_t)nb[MD_COLLOCATED].mv[l].x; t.nb.mv[l][1] = (int16_t)nb[MD_COLLOCATED].mv[l].y;
                const int tempRefIdx = nb[MD_COLLOCATED].refIdx[l];
                t.nb.refIdx[l] = (int8_t)tempRefIdx;
                if (tempRefIdx != -1)
                {   /* what CUData::getPMV looks up to scale the candidate (cudata.cpp:1962-1967) */
                    const Frame* colPic = slice->m_refFrameList[slice->isInterB() && !slice->m_colFromL0Flag][slice->m_colRefIdx];
                    const CUData* colCU = colPic->m_encData->getPicCTU(nb[MD_COLLOCATED].cuAddr[l]);
                    t.colRefPOC[l] = colCU->m_slice->m_refPOCList[tempRefIdx >> 4][tempRefIdx & 0xf];
                    t.colPOC[l] = colCU->m_slice->m_poc;
                }
            }
        }
        /* Analysis::computeMVForPUs' walk (analysis.cpp:161-246): sub-CUs first, then the CU's own PU shapes -- here only to harvest per entry the CU's qp and the partitions' collocated neighbours */
        void walk(CUData& ctu, const CUGeom& geom, int qp)
        {
            const uint32_t cuSize = 1u << geom.log2CUSize;
            if (cuSize > an.m_param->minCUSize)
            {
                int nextQP = qp;
                for (uint32_t sub = 0; sub < 4; sub++)
                {
                    const CUGeom& child = *(&geom + geom.childOffset + sub);
                    if (slice->m_pps->bUseDQP && geom.depth + 1 <= slice->m_pps->maxCuDQPDepth)
                        nextQP = x265_clip3(QP_MIN, QP_MAX_SPEC, an.calculateQpforCuSize(ctu, child));
                    walk(ctu, child, nextQP);
                }
            }
            CUData& cu = an.m_modeDepth[geom.depth].pred[Analysis::PRED_2Nx2N].cu;
            bool inited = false;
            while (k < nSteps && steps[k].cuSize == (int)cuSize && steps[k].cuX == (int)g_zscanToPelX[geom.absPartIdx] && steps[k].cuY == (int)g_zscanToPelY[geom.absPartIdx])
            {
                const x265hip_tme_step& e = steps[k];
                entryQp[(size_t)ctuAddr * nSteps + k] = qp;
                if (!inited) { cu.initSubCU(ctu, geom, qp); inited = true; }          /* once per CU: the entries of a CU differ in the partition size only */
                cu.setPartSizeSubParts((PartSize)e.part);
                for (int pi = 0; pi < e.numPart; pi++)
                {
                    PredictionUnit pu(cu, geom, pi);
                    collocated(cu, pi, pu.puAbsPartIdx, temporal[((size_t)ctuAddr * nSteps + k) * 2 + pi]);
                }
                k++;
            }
        }
    };
    /* One band of CTU rows (one frame thread: the whole picture) = one job.  The first worker that arrives for a row without records sets the job up; with one frame thread every
       worker that arrives while it runs (ThreadedME's workers are waiting for this picture's records anyway) takes CTUs of the host-side passes -- the harvest before the producer
       call, the write-back after it -- with its OWN Analysis object, so the per-picture host work is spread over the encoder's ThreadedME workers instead of sitting on one of
       them; under frame threads they sleep until the job ends (g_help). */
    struct Job
    {
        Frame* frame; int poc, nCtu, nCtuX, nCtuY, nS, nl;
        int row0 = 0, row1 = 0, c0 = 0, c1 = 0;                    /* the band: CTU rows [row0, row1) = CTUs [c0, c1) of the picture */
        const x265hip_tme_step* steps;
        std::vector<int> used;                                     /* the MEData slots of a CTU the schedule writes (and reads) */
        std::vector<int> sliceOfRow;
        std::vector<x265hip_tme_temporal> temporal; std::vector<int> entryQp, areaQp; std::vector<int16_t> median;
        std::vector<x265hip_inter_choice> table; std::vector<std::vector<x265hip_inter_choice>> refTables; std::vector<const MEData*> refSrc;
        std::vector<std::vector<int16_t>> lowres; size_t nRefTables = 0, nLowres = 0;      /* (the outer vectors only grow: their inner buffers are reused) */
        std::vector<uint8_t> qpIndex, areaQpIndex;
        x265hip_tme_picture_desc d;
        std::atomic<int> nextA{0}, doneA{0}, nextB{0}, doneB{0}, failed{0}, helped{0};
        int phase = 0, users = 0;                                  /* 0 harvest, 1 producer call, 2 write-back, 3 done; workers inside the job (both under g_lock) */

        /* FrameEncoder::m_refLagRows (frameencoder.cpp:166-171): how many CTU rows of its references a row waits for */
        static int ref_lag_rows(const x265_param* p)
        {
            int range = p->searchRange;
            range += !!(p->searchMethod < 2);
            range += NTAPS_LUMA / 2;
            range += 2 + (MotionEstimate::hpelIterationCount(p->subpelRefine) + 1) / 2;
            return 1 + ((range + p->maxCUSize - 1) / p->maxCUSize);
        }
        /* CTU rows of the picture, from `from` on, whose reference rows are final now (the frame encoder's own test per row, frameencoder.cpp:1029-1036); `from` itself was
           released by the frame encoder, or this call would not have come */
        static int ready_rows(const Analysis& an, int from, int nCtuY)
        {
            const Slice* slice = an.m_slice;
            const x265_param* p = an.m_param;
            if (p->frameNumThreads <= 1) return nCtuY;
            const int lag = ref_lag_rows(p);
            int r = from + 1;                                        /* rows [from, r) are ready */
            for (; r < nCtuY; r++)
            {
                const int idx = X265_MIN(nCtuY - 1, r + lag);
                bool ok = true;
                for (int l = 0; ok && l < (slice->isInterP() ? 1 : 2); l++)
                    for (int ref = 0; ok && ref < slice->m_numRefIdx[l]; ref++)
                    {
                        ok = slice->m_refFrameList[l][ref]->m_reconRowFlag[idx].get() != 0;
                        const MotionReference& mr = slice->m_mref[l][ref];
                        /* applyWeight(idx, ...) has run (:1035-1036).  The frame encoder writes the counter with a plain store after the plane's rows (reference.cpp:119-185): an
                           acquire load keeps the compiler and this core from reading plane rows ahead of it; the WRITER's ordering is what x86's store order gives (the reference
                           has no release there, and the file is not ours to change) -- on a weakly ordered host take weighted readiness from m_reconRowFlag alone */
                        if (ok && mr.isWeighted) ok = (int)__atomic_load_n(&mr.numSliceWeightedRows[0], __ATOMIC_ACQUIRE) >= idx;
                    }
                if (!ok) break;
            }
            return r;
        }

        static Job* create(Job* j, Analysis& an, Frame& frame, int row0, int row1)
        {
            const Slice* slice = an.m_slice;
            const x265_param* p = an.m_param;
            const int W = slice->m_sps->picWidthInLumaSamples, H = slice->m_sps->picHeightInLumaSamples, ctuSize = p->maxCUSize;
            if (!g_tme)
            {
                const auto t0 = std::chrono::steady_clock::now();
                if (g_api.ctx_create(g_device, &g_ctx) || g_api.tme_create(g_ctx, W, H, ctuSize, p->minCUSize, p->bEnableRectInter, p->bEnableAMP, &g_tme))
                { fprintf(stderr, "x265hip_tme_create: %s\n", g_api.last_error()); return nullptr; }
                g_createSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
            /* (the job's arrays are kept from job to job -- j is one of two static objects: fresh 10 MB vectors per picture cost more in page faults than the work on them) */
            j->nextA = 0; j->doneA = 0; j->nextB = 0; j->doneB = 0; j->failed = 0; j->helped = 0; j->phase = 0; j->users = 0;
            j->used.clear(); j->refSrc.clear(); j->nRefTables = 0; j->nLowres = 0;
            j->frame = &frame; j->poc = slice->m_poc;
            j->nCtuX = slice->m_sps->numCuInWidth; j->nCtuY = slice->m_sps->numCuInHeight; j->nCtu = j->nCtuX * j->nCtuY;
            j->nS = g_api.tme_entries(g_tme, &j->steps);
            j->nl = slice->isInterP() ? 1 : 2;
            j->row0 = row0; j->row1 = row1; j->c0 = row0 * j->nCtuX; j->c1 = row1 * j->nCtuX;
            const int nCtu = j->nCtu, nS = j->nS, nl = j->nl;
            for (int l = 0; l < nl; l++)
                if (slice->m_numRefIdx[l] < 1 || slice->m_numRefIdx[l] > X265HIP_MAX_REF) { fprintf(stderr, "tme_adapter: %d references in list %d (1..%d)\n", slice->m_numRefIdx[l], l, X265HIP_MAX_REF); return nullptr; }
            /* --slices: the rows of a slice as FrameEncoder::init deals them (frameencoder.cpp:131-146).  What a slice changes on this path with one frame thread: the
               first / last-row flags initCTU gets (threadedme.cpp:298-308).  Search::setSearchRange and selectMVP look at slice bounds only with several frame threads
               (search.cpp:2367-2369, 4999-5003), and motionEstimate's slice check (motion.cpp:1655-1659) tests an MV against the window it was searched in. */
            j->sliceOfRow.assign((size_t)j->nCtuY, 0);
            if (p->maxSlices > 1)
            {
                const uint32_t accu = ((uint32_t)j->nCtuY << 8) / p->maxSlices;
                uint32_t rowSum = accu, sidx = 0;
                for (uint32_t i = 0; i < (uint32_t)j->nCtuY; i++)
                {
                    if ((i >= (rowSum >> 8)) & (sidx != (uint32_t)p->maxSlices - 1)) { rowSum += accu; ++sidx; }
                    j->sliceOfRow[i] = (int)sidx;
                }
            }
            {
                std::vector<char> mark(593, 0);
                for (int k = 0; k < nS; k++) for (int pi = 0; pi < j->steps[k].numPart; pi++) { const int sl = j->steps[k].finalIdx + pi * j->steps[k].puOffset; if (sl >= 0 && sl < 593) mark[sl] = 1; }
                for (int sl = 0; sl < 593; sl++) if (mark[sl]) j->used.push_back(sl);
            }
            x265hip_tme_picture_desc& d = j->d;
            memset(&d, 0, sizeof(d));
            d.isP = slice->isInterP(); d.numRef[0] = slice->m_numRefIdx[0]; d.numRef[1] = nl > 1 ? slice->m_numRefIdx[1] : 0; d.curPOC = slice->m_poc;
            d.temporalMvp = slice->m_sps->bTemporalMVPEnabled;
            for (int l = 0; l < 2; l++) for (int r = 0; r < 16; r++) d.refPOC[l][r] = slice->m_refPOCList[l][r];
            d.searchRange = p->searchRange; d.searchMethod = p->searchMethod; d.subpelRefine = p->subpelRefine;
            d.flags = getenv("X265TME_PROF") ? X265HIP_TME_PROFILE : 0;
            d.width = W; d.height = H; d.sourceHeight = p->sourceHeight; d.frameThreads = p->frameNumThreads;
            /* --intra-refresh: Search::setSearchRange's test (search.cpp:4987-4996) -- a P picture whose first reference has not finished its sweep keeps the windows of the
               CTU columns left of its own refresh column out of the reference's not yet refreshed columns; the per-CU part of the test (cuPelX) is the producer's */
            if (p->bIntraRefresh && slice->m_sliceType == P_SLICE && slice->m_refFrameList[0][0]->m_encData->m_pir.pirEndCol < slice->m_sps->numCuInWidth)
            {
                d.pirStartCol = (int)frame.m_encData->m_pir.pirStartCol;
                d.pirSafeX = (int)(slice->m_refFrameList[0][0]->m_encData->m_pir.pirEndCol * p->maxCUSize) - 3;
            }
            const PicYuv* fenc = frame.m_fencPic;
            d.curPlane = fenc->m_picBuf[0]; d.stride = fenc->m_stride; d.origin = fenc->m_picOrg[0] - fenc->m_picBuf[0];
            d.planeElems = (int64_t)fenc->m_stride * (fenc->m_picHeight + 2 * fenc->m_lumaMarginY);
            j->temporal.resize((size_t)nCtu * nS * 2); j->entryQp.resize((size_t)nCtu * nS); j->areaQp.resize((size_t)nCtu * 5); j->median.assign((size_t)nCtu * 2 * X265HIP_MAX_REF * 3, 0);
            j->table.resize((size_t)nCtu * 593);
            /* references: planes, the lookahead's MVs; their own tables are converted CTU by CTU in the harvest */
            d.lowresBlocksX = frame.m_lowres.maxBlocksInRow;
            for (int l = 0; l < nl; l++)
                for (int r = 0; r < slice->m_numRefIdx[l]; r++)
                {
                    x265hip_tme_host_ref& R = d.refs[l][r];
                    const MotionReference& mr = slice->m_mref[l][r];
                    const bool parallel = p->frameNumThreads > 1;
                    if (mr.isWeighted && !parallel)
                    {   /* the frame encoder weights the reference's rows as it releases them to the row encoders (frameencoder.cpp:1029-1036); with one frame thread the
                           picture is complete and the producer takes it at once: finish the plane now (same values, the later calls find nothing left to do) */
                        const_cast<MotionReference&>(mr).applyWeight(j->nCtuY - 1, j->nCtuY, j->nCtuY, 0);
                    }
                    if (mr.isWeighted) g_weighted++;
                    const PicYuv* rec = slice->m_refReconPicList[l][r];
                    R.mePlane = mr.fpelPlane[0] - d.origin;
                    R.reconPlane = rec->m_picBuf[0];
                    if (parallel)
                    {   /* what is final for the band's last row: reconstruction rows up to CTU row idx (its flag is set: ready_rows), weighted rows below idx (applyWeight(idx)
                           weights [0, idx), everything when idx is the last row, reference.cpp:119-185) */
                        const int idx = X265_MIN(j->nCtuY - 1, row1 - 1 + ref_lag_rows(p));
                        const int allRows = (int)(d.planeElems / d.stride), top = (int)(d.origin / d.stride);
                        const bool last = idx == j->nCtuY - 1;
                        R.reconRowsValid = last ? 0 : top + (idx + 1) * ctuSize;
                        R.meRowsValid = !mr.isWeighted ? R.reconRowsValid : (last ? 0 : top + idx * ctuSize);
                        if (R.reconRowsValid > allRows) R.reconRowsValid = 0;
                        if (R.meRowsValid > allRows) R.meRowsValid = 0;
                        /* the weighted plane belongs to (this picture, list, reference): keyed so that its rows go up once */
                        if (mr.isWeighted) R.meKey = ((uint64_t)1 << 62) | (((uint64_t)frame.m_encodeOrder + 1) << 8) | ((uint64_t)l << 5) | (uint64_t)r;
                    }
                    const Frame* rf = slice->m_refFrameList[l][r];
                    R.reconKey = g_keepPlanes ? (uint64_t)rf->m_encodeOrder + 1 : 0;      /* a finished picture: its planes stay on the device for the pictures that reference it */
                    if (rf->m_encData->m_slice->m_sliceType != I_SLICE)
                    {   /* only the slots the schedule names are ever read */
                        if (j->refTables.size() <= j->nRefTables) j->refTables.emplace_back();
                        j->refTables[j->nRefTables++].resize((size_t)nCtu * 593);
                        j->refSrc.push_back(rf->m_encData->m_slice->m_ctuMV);
                    }
                    const int diffPoc = abs(slice->m_poc - slice->m_refPOCList[l][r]);
                    if (diffPoc <= p->bframes + 1)
                    {
                        const MV* mvs = frame.m_lowres.lowresMvs[l][diffPoc];
                        if (mvs[0].x != 0x7FFF)
                        {
                            const size_t nb = (size_t)frame.m_lowres.maxBlocksInRow * ((H + 15) / 16);
                            if (j->lowres.size() <= j->nLowres) j->lowres.emplace_back();
                            std::vector<int16_t>& lm = j->lowres[j->nLowres++];
                            lm.resize(nb * 2);
                            for (size_t i = 0; i < nb; i++) { lm[2 * i] = (int16_t)mvs[i].x; lm[2 * i + 1] = (int16_t)mvs[i].y; }
                        }
                    }
                }
            /* (the vectors above do not move any more: hand their storage to the descriptor) */
            size_t rt = 0, lw = 0;
            for (int l = 0; l < nl; l++)
                for (int r = 0; r < slice->m_numRefIdx[l]; r++)
                {
                    const Frame* rf = slice->m_refFrameList[l][r];
                    if (rf->m_encData->m_slice->m_sliceType != I_SLICE) d.refs[l][r].refTable = j->refTables[rt++].data();
                    const int diffPoc = abs(slice->m_poc - slice->m_refPOCList[l][r]);
                    if (diffPoc <= p->bframes + 1 && frame.m_lowres.lowresMvs[l][diffPoc][0].x != 0x7FFF) d.refs[l][r].lowresMv = j->lowres[lw++].data();
                }
            return j;
        }

        /* per CTU, by whichever worker takes it: what findJob sets up before the call (threadedme.cpp:238-246), the qps, the collocated neighbours and medians, and this
           CTU's part of the table conversions */
        void harvest(Analysis& an, const CUGeom& ctuGeom, int c)
        {
            const Slice* slice = an.m_slice;
            Frame& fr = *frame;
            CUData* ctu = fr.m_encData->getPicCTU(c);
            ctu->m_slice = fr.m_encData->m_slice;
            const int row = c / nCtuX, col = c % nCtuX;
            fr.m_encData->m_cuStat[c].baseQp = fr.m_encData->m_avgQpRc;
            const bool firstRow = row == 0 || sliceOfRow[row - 1] != sliceOfRow[row], lastRow = row == nCtuY - 1 || sliceOfRow[row + 1] != sliceOfRow[row];
            ctu->initCTU(fr, c, slice->m_sliceQp, firstRow, lastRow, lastRow && col == nCtuX - 1);
            const int rawBase = slice->m_pps->bUseDQP ? an.calculateQpforCuSize(*ctu, ctuGeom) : slice->m_sliceQp;
            areaQp[c * 5] = rawBase;
            for (int sub = 0; sub < 4; sub++)
                areaQp[c * 5 + 1 + sub] = slice->m_pps->bUseDQP ? an.calculateQpforCuSize(*ctu, *(&ctuGeom + ctuGeom.childOffset + sub)) : slice->m_sliceQp;
            Harvest h{ an, slice, fr, steps, nS, temporal, entryQp, c, 0 };
            h.walk(*ctu, ctuGeom, x265_clip3(QP_MIN, QP_MAX_SPEC, rawBase));
            if (h.k != nS) { fprintf(stderr, "schedule mismatch: %d of %d entries\n", h.k, nS); failed = 1; return; }
            const Frame* colPic = slice->m_refFrameList[slice->isInterB() && !slice->m_colFromL0Flag][slice->m_colRefIdx];
            const CUData* colCU = colPic->m_encData->getPicCTU(c);
            for (int l = 0; l < nl; l++)
                for (int r = 0; r < slice->m_numRefIdx[l]; r++)
                {
                    MV m;
                    if (ctu->getMedianColMV(colCU, colPic, l, r, m)) { int16_t* o = &median[(((size_t)c * 2 + l) * X265HIP_MAX_REF + r) * 3]; o[0] = 1; o[1] = (int16_t)m.x; o[2] = (int16_t)m.y; }
                }
            const MEData* dst = fr.m_encData->m_slice->m_ctuMV;
            for (int sl : used) to_choice(dst[(size_t)c * 593 + sl], table[(size_t)c * 593 + sl]);
            for (size_t t = 0; t < nRefTables; t++)
                for (int sl : used) to_choice(refSrc[t][(size_t)c * 593 + sl], refTables[t][(size_t)c * 593 + sl]);
        }

        int call()
        {   /* the distinct qps, then the producer */
            std::vector<int> qps;
            auto qidx = [&](int qp) { if (qp > QP_MAX_SPEC) { fprintf(stderr, "qp %d above 51: not handled\n", qp); die(); }
                                      for (size_t i = 0; i < qps.size(); i++) if (qps[i] == qp) return (int)i; qps.push_back(qp); return (int)qps.size() - 1; };
            qpIndex.resize(entryQp.size()); areaQpIndex.resize(areaQp.size());
            for (size_t i = (size_t)c0 * nS; i < (size_t)c1 * nS; i++) qpIndex[i] = (uint8_t)qidx(entryQp[i]);
            for (size_t i = (size_t)c0 * 5; i < (size_t)c1 * 5; i++) areaQpIndex[i] = (uint8_t)qidx(areaQp[i]);
            if (qps.size() > 64) { fprintf(stderr, "more than 64 distinct qps\n"); return -1; }
            d.nQp = (int)qps.size();
            for (int i = 0; i < d.nQp; i++) d.qps[i] = qps[i];
            d.qpIndex = qpIndex.data(); d.areaQpIndex = areaQpIndex.data(); d.temporal = temporal.data(); d.median = median.data(); d.table = table.data();
            if (row0 > 0 || row1 < nCtuY) { d.ctuRowFirst = row0; d.ctuRowCount = row1 - row0; }
            const auto t0 = std::chrono::steady_clock::now();
            const int rc = g_api.tme_picture(g_tme, &d);
            if (rc) { fprintf(stderr, "x265hip_tme_picture (POC %d, %s slice, refs %d / %d): %d %s\n", poc, d.isP ? "P" : "B", d.numRef[0], d.numRef[1], rc, g_api.last_error()); return -1; }
            const double dtCall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::lock_guard<std::mutex> sg(g_statLock);
            g_gpuSeconds += dtCall;
            if (++g_calls > 4) { g_gpuSecondsWarm += dtCall; g_callsWarm++; }
            return 0;
        }

        void writeback(int c)
        {
            MEData* dst = frame->m_encData->m_slice->m_ctuMV;
            for (int sl : used) from_choice(table[(size_t)c * 593 + sl], dst[(size_t)c * 593 + sl]);
        }
    };
    /* A job is first set up and harvested (s_prep), then in its producer call and write-back (s_call).  With X265TME_AHEAD the next band may enter the first stage while a job
       is in the second -- the host work of a band in the shadow of the previous band's call instead of queueing behind it; the producer still sees one call at a time, in the
       order the jobs were created.  Without it a band starts when no job is in flight. */
    static Job s_store[2];
    static Job* s_prep = nullptr;
    static Job* s_call = nullptr;
    /* queues: s_cv -- workers whose row belongs to a job in flight sleep until a job ENDS; s_cvSlot -- workers that want to open a job while another is being set up;
       s_cvCall -- the leader of the harvested job, until the producer is free; s_cvJob -- a job's own helpers and its leader follow its phases.  (One queue for all woke every
       waiting ThreadedME worker several times per band: with a hundred pool threads on a host that grants them 16 CPUs that is scheduling work of its own) */
    static std::condition_variable s_cv, s_cvJob, s_cvSlot, s_cvCall;

    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    std::unique_lock<std::mutex> lk(g_lock);
    const int poc = ctu.m_slice->m_poc;
    m_slice = ctu.m_slice; m_frame = &frame; m_param = m_frame->m_param;      /* as the encoder's body starts (analysis.cpp:250-252) */
    const int nCtuX = m_slice->m_sps->numCuInWidth, nCtuY = m_slice->m_sps->numCuInHeight, row = (int)ctu.m_cuAddr / nCtuX;
    bool leader = false;
    const double tStart = now();
    Job* job = nullptr;
    for (;;)
    {
        PicState& ps = g_pics[&frame];
        if (ps.poc1 != poc + 1) { ps.poc1 = poc + 1; ps.rowsDone = 0; ps.claimed = 0; }       /* the Frame object holds a new picture */
        if (row < ps.rowsDone) return;                                        /* this CTU's records are there already */
        if (row < ps.claimed)
        {   /* the row belongs to a job in flight */
            if (g_help > 0 || (g_help < 0 && m_param->frameNumThreads <= 1))
                for (Job* j : { s_prep, s_call })
                    if (j && j->frame == &frame && j->poc == poc && row >= j->row0 && row < j->row1) job = j;
            if (job) break;                                                   /* help with its host passes */
            s_cv.wait(lk);                                                    /* sleep until a job ends */
            continue;
        }
        if (!s_prep && (g_ahead || !s_call))
        {   /* a new band: from the first row no job has claimed to the last one whose reference rows are final (one frame thread: the whole picture) */
            int row1 = Job::ready_rows(*this, row, nCtuY);
            const int minRows = g_minRows > 0 ? g_minRows : (nCtuY + 1) / 2;
            if (minRows > 1 && g_waitUs > 0 && m_param->frameNumThreads > 1 && row1 < nCtuY && row1 - ps.claimed < minRows)
            {   /* few rows are ready: give the references a moment to release more (the lock is open meanwhile; whoever comes back first opens the job) */
                const int have = row1;
                lk.unlock();
                const double tw = now();
                while ((now() - tw) * 1e6 < g_waitUs && Job::ready_rows(*this, row, nCtuY) == have) std::this_thread::sleep_for(std::chrono::microseconds(150));
                lk.lock();
                PicState& ps2 = g_pics[&frame];
                if (ps2.poc1 != poc + 1 || row < ps2.claimed || s_prep || (!g_ahead && s_call)) continue;       /* somebody else got there: look again */
                row1 = Job::ready_rows(*this, row, nCtuY);
            }
            if (g_waitRefs && row1 < nCtuY)
            {
                lk.unlock();
                for (int l = 0; l < (m_slice->isInterP() ? 1 : 2); l++)
                    for (int ref = 0; ref < m_slice->m_numRefIdx[l]; ref++)
                    {
                        Frame* rf = m_slice->m_refFrameList[l][ref];
                        while (rf->m_reconRowFlag[nCtuY - 1].get() == 0) rf->m_reconRowFlag[nCtuY - 1].waitForChange(0);
                    }
                lk.lock();
                continue;                                                     /* (the world may have changed while the lock was open: look again -- the rows are all ready now) */
            }
            if (g_trace) fprintf(stderr, "tme_adapter: POC %d rows %d..%d (asked for row %d of %d)\n", poc, ps.claimed, row1 - 1, row, nCtuY);
            s_prep = Job::create(s_call == &s_store[0] ? &s_store[1] : &s_store[0], *this, frame, ps.claimed, row1);
            if (!s_prep) die();
            ps.claimed = row1;
            g_sec[0] += now() - tStart;                                       /* job set-up (the first picture also creates the producer: context, streams, code objects) */
            leader = true;
            job = s_prep;
            break;
        }
        s_cvSlot.wait(lk);                                                    /* another job is being set up (or, without X265TME_AHEAD, running): wait for the slot */
    }
    job->users++;
    lk.unlock();
    const int nBand = job->c1 - job->c0;
    for (;;)
    {   /* pass A */
        const int c = job->c0 + job->nextA.fetch_add(1);
        if (c >= job->c1) break;
        job->harvest(*this, cuGeom, c);
        job->doneA.fetch_add(1);
        if (!leader) job->helped.fetch_add(1);
    }
    double tBack = 0;
    if (leader)
    {
        while (job->doneA.load() < nBand) std::this_thread::yield();          /* the helpers' last CTUs */
        const double tCall = now();
        { std::lock_guard<std::mutex> sg(g_statLock); g_sec[2] += job->helped.load(); g_sec[1] += tCall - tStart; }      /* [2] CTUs other workers harvested; [1] wall time up to the producer call: set-up + harvest (qps, collocated neighbours, medians, table conversions) */
        /* the producer takes one call at a time: wait for the job before this one (its call and its write-back), then hand the set-up slot to the next band */
        lk.lock();
        while (s_call) s_cvCall.wait(lk);
        s_call = job; s_prep = nullptr;
        lk.unlock();
        if (g_ahead) s_cvSlot.notify_all();
        if (job->failed.load() || job->call()) die();
        tBack = now();                                                        /* write-back wall time, closed below */
        lk.lock(); job->phase = 2; lk.unlock();
        s_cvJob.notify_all();
    }
    else
    {
        lk.lock();
        while (job->phase < 2) s_cvJob.wait(lk);
        lk.unlock();
    }
    for (;;)
    {   /* pass B */
        const int c = job->c0 + job->nextB.fetch_add(1);
        if (c >= job->c1) break;
        job->writeback(c);
        job->doneB.fetch_add(1);
    }
    lk.lock();
    if (leader)
    {
        while (job->doneB.load() < nBand) { lk.unlock(); std::this_thread::yield(); lk.lock(); }
        g_sec[3] += now() - tBack;
        g_pics[&frame].rowsDone = job->row1;
        g_bands++;
        if (g_trace) fprintf(stderr, "tme_adapter: POC %d rows %d..%d done (%.1f ms, %d helpers' CTUs)\n", poc, job->row0, job->row1 - 1, 1e3 * (now() - tStart), job->helped.load());
        if (job->row1 == nCtuY) g_pictures++;
        g_pictureSeconds += now() - tStart;
        job->phase = 3;
        s_cvJob.notify_all();
        while (job->users > 1) s_cvJob.wait(lk);                              /* every helper has left the job */
        s_call = nullptr;
        s_cvCall.notify_all();
        s_cvSlot.notify_all();
        s_cv.notify_all();
    }
    else
    {
        while (job->phase < 3) s_cvJob.wait(lk);
        job->users--;
        s_cvJob.notify_all();
    }
}
}

extern "C" int x265hip_tme_adapter_load(const char* libraryPath, int device)
{
    if (getenv("X265TME_NOKEEP")) g_keepPlanes = 0;
    g_trace = getenv("X265TME_TRACE") && atoi(getenv("X265TME_TRACE"));
    g_waitRefs = getenv("X265TME_WAIT_REFS") && atoi(getenv("X265TME_WAIT_REFS"));
    if (getenv("X265TME_HELP")) g_help = atoi(getenv("X265TME_HELP"));
    g_ahead = getenv("X265TME_AHEAD") && atoi(getenv("X265TME_AHEAD"));
    if (getenv("X265TME_MIN_ROWS")) g_minRows = atoi(getenv("X265TME_MIN_ROWS"));
    if (getenv("X265TME_WAIT_US")) g_waitUs = atoi(getenv("X265TME_WAIT_US"));
    g_lib = dlopen(libraryPath, RTLD_NOW | RTLD_LOCAL);
    if (!g_lib) { fprintf(stderr, "tme_adapter: dlopen: %s\n", dlerror()); return -1; }
    g_api.ctx_create = (int (*)(int, x265hip_ctx**))dlsym(g_lib, "x265hip_ctx_create");
    g_api.ctx_destroy = (void (*)(x265hip_ctx*))dlsym(g_lib, "x265hip_ctx_destroy");
    g_api.tme_create = (int (*)(x265hip_ctx*, int, int, int, int, int, int, x265hip_tme**))dlsym(g_lib, "x265hip_tme_create");
    g_api.tme_destroy = (void (*)(x265hip_tme*))dlsym(g_lib, "x265hip_tme_destroy");
    g_api.tme_entries = (int (*)(const x265hip_tme*, const x265hip_tme_step**))dlsym(g_lib, "x265hip_tme_entries");
    g_api.tme_picture = (int (*)(x265hip_tme*, const x265hip_tme_picture_desc*))dlsym(g_lib, "x265hip_tme_picture");
    g_api.last_error = (const char* (*)())dlsym(g_lib, "x265hip_last_error");
    if (!g_api.ctx_create || !g_api.ctx_destroy || !g_api.tme_create || !g_api.tme_destroy || !g_api.tme_entries || !g_api.tme_picture || !g_api.last_error)
    { fprintf(stderr, "tme_adapter: %s lacks the producer's entry points\n", libraryPath); return -1; }
    g_device = device; g_useGpu = 1;
    return 0;
}
extern "C" void x265hip_tme_adapter_enable(int on) { g_useGpu = on && g_lib; }
extern "C" void x265hip_tme_adapter_close(void)
{
    std::lock_guard<std::mutex> guard(g_lock);
    if (g_tme) { g_api.tme_destroy(g_tme); g_tme = nullptr; }
    if (g_ctx) { g_api.ctx_destroy(g_ctx); g_ctx = nullptr; }
    g_pics.clear(); g_useGpu = 0;
}
extern "C" void x265hip_tme_adapter_get_stats(x265hip_tme_adapter_stats* o)
{
    o->pictures = g_pictures; o->bands = g_bands; o->producerSecondsWarm = g_gpuSecondsWarm; o->callsWarm = g_callsWarm; o->weightedRefs = g_weighted; o->producerSeconds = g_gpuSeconds; o->adapterSeconds = g_pictureSeconds; o->createSeconds = g_createSeconds;
    for (int i = 0; i < 4; i++) o->sections[i] = g_sec[i];
}
