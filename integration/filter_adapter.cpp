/*
 * filter_adapter.cpp -- FrameFilter::processRow, Deblock::deblockCTU and SAO::calcSaoStatsCTU with the GPU as the producer of the deblocked picture and of the SAO
 * statistics (see filter_adapter.h).
 *
 * Why the whole picture at its last row is the same thing as the reference's row pipeline: the deblocking of a CTU row touches no sample the analysis of later rows
 * reads (that is what FrameEncoder::m_filterRowDelay arranges), the SAO statistics of a CTU leave out the samples later deblocking still changes (skipB / skipR of
 * calcSaoStatsCTU) and are taken before the SAO of the neighbours is applied (the column delays of ParallelFilter::processTasks), and the bitstream is written after the
 * last filter row (FrameEncoder::encodeSlice).  So deblocking everything first, then collecting all statistics, then deciding and applying SAO row by row in the
 * reference's order gives the reference's picture and the reference's SAO parameters.
 */
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "x265.h"
#include "common.h"
#include "frame.h"
#include "framedata.h"
#include "cudata.h"
#include "picyuv.h"
#include "slice.h"
#include "deblock.h"
#include "sao.h"
#include "framefilter.h"
#include "../include/x265hip_ctx.h"
#include "filter_adapter.h"

using namespace X265_NS;

namespace {
/* a producer call failed: the encode must not go on quietly.  The encoder's pool threads are running -- exit() would run the static destructors under them (a hang, seen in
   the ThreadedME binding): leave at once, the message is on stderr */
[[noreturn]] void die() { fflush(stdout); fflush(stderr); _Exit(3); }
struct Api
{
    int (*ctx_create)(int, x265hip_ctx**);
    void (*ctx_destroy)(x265hip_ctx*);
    int (*ff_create)(x265hip_ctx*, int, int, int, intptr_t, intptr_t, x265hip_ff**);
    void (*ff_destroy)(x265hip_ff*);
    int (*ff_picture)(x265hip_ff*, const x265hip_ff_picture_desc*);
    const char* (*last_error)();
} g_api;
void* g_lib;
x265hip_ctx* g_ctx;
x265hip_ff* g_ff;
int g_on, g_device, g_deferOnly;
std::mutex g_lock;                        /* one picture at a time through the producer and its staging arrays */
x265hip_ff_adapter_stats g_stats;
std::mutex g_statLock;

/* the picture whose rows are being replayed: deblocked already, statistics in the table.  Keyed on the picture's FrameData, not on the replaying thread: with WPP
   another worker may run ParallelFilter::processTasks of one of the picture's rows while the replay is going on (frameencoder.cpp:2072-2076) and must find the same state.
   The state stays published until the next picture's replay replaces it (a straggler of this picture still finds it; the staging arrays behind `stats` live as long) */
struct Replay { const FrameData* data; bool deblocked; const int32_t* stats[3]; std::atomic<long long> skipped, served; };
Replay g_replayState;
std::atomic<Replay*> g_replay{nullptr};
inline Replay* replay_of(const FrameData* data)
{
    Replay* r = g_replay.load(std::memory_order_acquire);
    return (r && r->data == data) ? r : nullptr;
}

/* staging: CUData's per-partition arrays of all CTUs back to back, the statistics of the three planes */
struct Staging
{
    std::vector<uint8_t> log2CUSize, partSize, tuDepth, predMode, cbf, tqBypass;
    std::vector<int8_t> qp, refIdx[2];
    std::vector<int32_t> mv[2], stats[3];
} g_st;

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

/* ---- band mode (several frame threads): a picture in flight per FrameFilter (= per frame encoder), its rows filtered in bands as they arrive ---- */
int g_bandRows;                              /* X265FF_BAND_ROWS: rows a band waits for (0 = default 4); with one frame thread 0 = the whole picture at its last row, > 0 forces bands */
struct Band
{
    const FrameData* data = nullptr;
    std::mutex mu;                           /* the slices of a picture finish their rows side by side (WPP): the state below and the staging arrays' sizes change under it */
    std::vector<int> sliceOfRow, sliceFirst, sliceDone;      /* --slices: the slice of every CTU row, its first row, and how far its rows are filtered and replayed (one slice: [0 ...], [0], [done]) */
    int rowsThrough = 0;                     /* rows of the picture filtered and replayed so far (all slices) */
    Staging st;                              /* picture-addressed; a band fills its rows (+ the row above) */
    Replay replay;
};
std::mutex g_bandLock;
std::unordered_map<const FrameFilter*, std::unique_ptr<Band>> g_bands;
Band& band_of(const FrameFilter* ff)
{
    std::lock_guard<std::mutex> g(g_bandLock);
    std::unique_ptr<Band>& b = g_bands[ff];
    if (!b) b.reset(new Band());
    return *b;
}
/* the band being replayed by THIS thread: the encoder's row loop runs on the thread that made the producer call, and nobody else touches the picture's filter rows meanwhile
   (a picture's filter rows follow each other, frameencoder.cpp:1504-1508; the row encoders' early start is switched off in band mode, see processTasks below) */
thread_local Replay* t_replay = nullptr;

/* does the binding filter this FrameFilter's pictures (a function of the encoder's parameters and the picture format only: the same answer for every picture of an encode) */
bool takes(const FrameFilter& ff, const x265_param& p, const Frame* frame, bool useSao)
{
    if (!(g_on && (p.bEnableLoopFilter || useSao) && ff.m_parallelFilter && p.internalCsp != X265_CSP_I400 && frame)) return false;
    const PicYuv& rp = *frame->m_reconPic[0]; const PicYuv& fp = *frame->m_fencPic;
    return fp.m_picCsp == p.internalCsp && rp.m_stride == fp.m_stride && rp.m_strideC == fp.m_strideC && !(p.sourceWidth & 7) && !(p.sourceHeight & 7);
}
inline bool band_mode(const x265_param& p) { return p.frameNumThreads > 1 || g_bandRows > 0; }

x265hip_ff* producer(const x265_param& p, const PicYuv& recon)
{
    if (g_ff) return g_ff;
    if (g_api.ctx_create(g_device, &g_ctx) || g_api.ff_create(g_ctx, p.sourceWidth, p.sourceHeight, (int)p.maxCUSize, recon.m_stride, recon.m_strideC, &g_ff))
    {
        fprintf(stderr, "filter_adapter: x265hip_ff_create: %s -- the encoder's own filters run\n", g_api.last_error());
        g_on = 0; g_ff = nullptr;
    }
    return g_ff;
}
void processBand(FrameFilter* ff, int row, int layer);
}

void processRow_cpu(FrameFilter* self, int row, int layer) __asm__("xff_processRow_cpu");
void deblockCTU_cpu(const CUData* ctu, const CUGeom& cuGeom, int32_t dir) __asm__("xff_deblockCTU_cpu");
void calcSaoStatsCTU_cpu(SAO* self, int addr, int plane) __asm__("xff_calcSaoStatsCTU_cpu");
void processTasks_cpu(FrameFilter::ParallelFilter* self, int workerThreadId) __asm__("xff_processTasks_cpu");

namespace {
/* --slices: the slice of every CTU row and every slice's first row (FrameEncoder::init, frameencoder.cpp:131-146) */
void slice_rows(const x265_param& p, int numRows, std::vector<int>& of, std::vector<int>& first)
{
    of.assign((size_t)numRows, 0); first.assign(1, 0);
    if (p.maxSlices <= 1) return;
    const uint32_t accu = ((uint32_t)numRows << 8) / p.maxSlices;
    uint32_t rowSum = accu, sidx = 0;
    for (uint32_t i = 0; i < (uint32_t)numRows; i++)
    {
        if ((i >= (rowSum >> 8)) & (sidx != (uint32_t)p.maxSlices - 1)) { rowSum += accu; ++sidx; first.push_back((int)i); }
        of[i] = (int)sidx;
    }
}

/* CUData's per-partition arrays of the CTU rows [rowA, row1) into the picture-addressed staging arrays, and the description of the call (the whole picture: rowA = 0,
   row1 = the picture's rows; a band: the row above it included -- its CUs are the P side of the band's top edge) */
void describe(FrameFilter& ff, Staging& S, int rowA, int row1, x265hip_ff_picture_desc& d, std::vector<uint8_t>& sliceFirstRow)
{
    const x265_param& p = *ff.m_param;
    FrameData& encData = *ff.m_frame->m_encData;
    Slice* slice = encData.m_slice;
    PicYuv* recon = ff.m_frame->m_reconPic[0];
    PicYuv* fenc = ff.m_frame->m_fencPic;
    const uint32_t nctu = slice->m_sps->numCUsInFrame, np = encData.getPicCTU(0)->m_numPartitions, ncols = slice->m_sps->numCuInWidth;
    const size_t n = (size_t)nctu * np;
    const bool isB = slice->m_sliceType == B_SLICE, bypass = slice->m_pps->bTransquantBypassEnabled;
    S.log2CUSize.resize(n); S.partSize.resize(n); S.tuDepth.resize(n); S.predMode.resize(n); S.cbf.resize(n); S.tqBypass.resize(n); S.qp.resize(n);
    for (int l = 0; l < 2; l++) { S.refIdx[l].resize(n); S.mv[l].resize(2 * n); }
    if (p.bEnableLoopFilter)
        for (uint32_t a = (uint32_t)rowA * ncols; a < (uint32_t)row1 * ncols; a++)
        {
            const CUData* c = encData.getPicCTU(a);
            const size_t o = (size_t)a * np;
            memcpy(&S.log2CUSize[o], c->m_log2CUSize, np); memcpy(&S.partSize[o], c->m_partSize, np); memcpy(&S.tuDepth[o], c->m_tuDepth, np);
            memcpy(&S.predMode[o], c->m_predMode, np); memcpy(&S.cbf[o], c->m_cbf[0], np); memcpy(&S.qp[o], c->m_qp, np);
            if (bypass) memcpy(&S.tqBypass[o], c->m_tqBypass, np);
            for (int l = 0; l < (isB ? 2 : 1); l++)
            {
                memcpy(&S.refIdx[l][o], c->m_refIdx[l], np);
                static_assert(sizeof(MV) == 2 * sizeof(int32_t), "MV is a pair of 32-bit components");
                memcpy(&S.mv[l][2 * o], c->m_mv[l], np * sizeof(MV));
            }
        }
    memset(&d, 0, sizeof(d));
    d.pic.width = p.sourceWidth; d.pic.height = p.sourceHeight; d.pic.ctuSize = (int)p.maxCUSize; d.pic.sliceIsP = !isB;
    d.pic.betaOffsetDiv2 = slice->m_pps->deblockingFilterBetaOffsetDiv2; d.pic.tcOffsetDiv2 = slice->m_pps->deblockingFilterTcOffsetDiv2;
    d.pic.cbQpOffset = slice->m_pps->chromaQpOffset[0]; d.pic.crQpOffset = slice->m_pps->chromaQpOffset[1]; d.pic.tqBypassEnabled = bypass;
    d.pic.chromaFormat = p.internalCsp;                       /* X265_CSP_I420 / I422 / I444: the chroma planes' subsampling */
    d.pic.log2CUSize = S.log2CUSize.data(); d.pic.partSize = S.partSize.data(); d.pic.tuDepth = S.tuDepth.data(); d.pic.predMode = S.predMode.data();
    d.pic.cbfLuma = S.cbf.data(); d.pic.tqBypass = bypass ? S.tqBypass.data() : NULL; d.pic.qp = S.qp.data();
    d.pic.refIdx0 = S.refIdx[0].data(); d.pic.mv0 = S.mv[0].data(); d.pic.refIdx1 = isB ? S.refIdx[1].data() : NULL; d.pic.mv1 = isB ? S.mv[1].data() : NULL;
    /* the reference compares the Frame behind (list, refIdx) (deblock.cpp:getBoundaryStrength); the POC identifies it */
    for (int l = 0; l < 2; l++) for (int i = 0; i < 16; i++) d.pic.refPic[l][i] = i < slice->m_numRefIdx[l] ? slice->m_refPOCList[l][i] : -1 - i - 16 * l;
    d.reconY = recon->m_picOrg[0]; d.reconCb = recon->m_picOrg[1]; d.reconCr = recon->m_picOrg[2];
    d.fencY = fenc->m_picOrg[0]; d.fencCb = fenc->m_picOrg[1]; d.fencCr = fenc->m_picOrg[2];
    /* --slices: the CTU rows that begin a slice, as the encoder flagged their CTUs (CUData::initCTU, frameencoder.cpp:1669) */
    if (p.maxSlices > 1)
    {   /* (the rows of a slice as FrameEncoder::init deals them, frameencoder.cpp:131-146 -- the flags of CTUs whose rows have not started yet are not looked at) */
        std::vector<int> of, first;
        slice_rows(p, ff.m_numRows, of, first);
        sliceFirstRow.assign((size_t)ff.m_numRows + 1, 0);
        for (int r = 1; r < ff.m_numRows; r++) sliceFirstRow[r] = of[r] != of[r - 1];
        d.pic.sliceFirstRow = sliceFirstRow.data();
    }
    d.deblock = p.bEnableLoopFilter;
    const SAOParam* sp = encData.m_saoParam;
    d.saoStats = (ff.m_useSao && sp) ? (sp->bSaoFlag[0] ? 1 : 0) | (sp->bSaoFlag[1] ? 2 : 0) : 0;
    d.saoNonDeblocked = p.bSaoNonDeblocked;
    for (int k = 0; k < 3; k++) { S.stats[k].resize((size_t)nctu * 320); d.stats[k] = S.stats[k].data(); }
}

/* Band mode: rows arrive in order (one slice); they pass until a band is due -- g_bandRows rows waiting, or the picture's last row -- then the band goes through the producer and
   the encoder's own row loop runs over its rows (SAO decision and SAO of the row above each, border extension, PSNR / SSIM, Frame::m_reconRowFlag: what the next pictures wait for) */
void processBand(FrameFilter* ff, int row, int layer)
{
    const x265_param& p = *ff->m_param;
    Band& B = band_of(ff);
    FrameData& encData = *ff->m_frame->m_encData;
    const int numRows = ff->m_numRows, wait = g_bandRows > 0 ? g_bandRows : 4;
    int r0, r1;
    x265hip_ff_picture_desc d;
    std::vector<uint8_t> sliceFirstRow;
    const double t0 = now();
    {
        std::lock_guard<std::mutex> bg(B.mu);
        if (B.data != &encData || (int)B.sliceOfRow.size() != numRows)
        {   /* another picture geometry / FrameData behind this FrameFilter */
            B.data = &encData; B.rowsThrough = 0;
            slice_rows(p, numRows, B.sliceOfRow, B.sliceFirst);
            B.sliceDone = B.sliceFirst;
        }
        const int s = B.sliceOfRow[row], sFirst = B.sliceFirst[s], sEnd = s + 1 < (int)B.sliceFirst.size() ? B.sliceFirst[s + 1] : numRows;
        if (row == sFirst) B.sliceDone[s] = sFirst;                                  /* the slice's first row: a new picture's rows begin (a slice's rows come once, in order) */
        if (row < B.sliceDone[s]) return;
        if (row != sEnd - 1 && row + 1 - B.sliceDone[s] < wait) return;              /* the row waits for its band (a band stays inside its slice) */
        r0 = B.sliceDone[s]; r1 = row + 1;
        B.sliceDone[s] = r1;                                                         /* claimed: the rows above it in this slice cannot call again */
        describe(*ff, B.st, (r0 > sFirst) ? r0 - 1 : r0, r1, d, sliceFirstRow);      /* (the row above the band is the P side of its top edge -- not across a slice's first row) */
        if (r0 > 0 || r1 < numRows) { d.ctuRowFirst = r0; d.ctuRowCount = r1 - r0; }
    }
    const double t1 = now();
    double t2;
    {
        std::lock_guard<std::mutex> guard(g_lock);                                  /* one call at a time on the one producer; bands of the pictures in flight take turns */
        x265hip_ff* prod = producer(p, *ff->m_frame->m_reconPic[0]);
        if (!prod) { fprintf(stderr, "filter_adapter: no producer under frame threads\n"); die(); }      /* (rows of this picture may have passed already: there is no way back to the encoder's own filters) */
        const int rc = g_api.ff_picture(prod, &d);
        t2 = now();
        if (rc) { fprintf(stderr, "filter_adapter: x265hip_ff_picture (POC %d, CTU rows %d..%d): %d %s\n", encData.m_slice->m_poc, r0, r1 - 1, rc, g_api.last_error()); die(); }
    }
    Replay& rp = B.replay;                                                           /* (the same values whichever slice's band writes them) */
    rp.data = &encData; rp.deblocked = p.bEnableLoopFilter != 0;
    rp.stats[0] = (d.saoStats & 1) ? B.st.stats[0].data() : NULL;
    rp.stats[1] = (d.saoStats & 2) ? B.st.stats[1].data() : NULL; rp.stats[2] = (d.saoStats & 2) ? B.st.stats[2].data() : NULL;
    t_replay = &rp;
    for (int r = r0; r < r1; r++) ::processRow_cpu(ff, r, layer);
    t_replay = nullptr;
    const double t3 = now();
    bool whole;
    { std::lock_guard<std::mutex> bg(B.mu); B.rowsThrough += r1 - r0; whole = B.rowsThrough == numRows; if (whole) B.rowsThrough = 0; }
    std::lock_guard<std::mutex> sg(g_statLock);
    g_stats.bands++;
    if (whole) { g_stats.pictures++; g_stats.deblockSkipped += rp.skipped.exchange(0); g_stats.statsServed += rp.served.exchange(0); }
    g_stats.gatherSeconds += t1 - t0; g_stats.producerSeconds += t2 - t1; g_stats.replaySeconds += t3 - t2;
}
} // namespace

namespace X265_NS {

void Deblock::deblockCTU(const CUData* ctu, const CUGeom& cuGeom, int32_t dir)
{
    if (t_replay) { if (t_replay->data == ctu->m_encData && t_replay->deblocked) { t_replay->skipped++; return; } }
    else if (Replay* rp = replay_of(ctu->m_encData)) if (rp->deblocked) { rp->skipped++; return; }
    ::deblockCTU_cpu(ctu, cuGeom, dir);
}

/* The row encoders' early start of the row above (frameencoder.cpp:2067-2076: "Processing left Deblock block with current threading") -- the only caller of this member outside
   framefilter.cpp; processRow's own call is bound to the encoder's body inside its object.  In band mode the row is deblocked with its band: the early start would deblock it on
   the CPU ahead of that.  It is an early start of work processRow does in any case (m_allowedCol = all columns, then processTasks), so leaving it out changes no result. */
void FrameFilter::ParallelFilter::processTasks(int workerThreadId)
{
    const FrameFilter* ff = m_frameFilter;
    if (ff && ff->m_param && band_mode(*ff->m_param) && takes(*ff, *ff->m_param, ff->m_frame, ff->m_useSao)) return;
    ::processTasks_cpu(this, workerThreadId);
}

void SAO::calcSaoStatsCTU(int addr, int plane)
{
    Replay* rp = (t_replay && t_replay->data == m_frame->m_encData) ? t_replay : replay_of(m_frame->m_encData);
    const int32_t* tab = rp ? rp->stats[plane] : NULL;
    if (!tab) { ::calcSaoStatsCTU_cpu(this, addr, plane); return; }
    /* per CTU [2][5][32]: m_offsetOrg then m_count of the plane; the body ADDS to what rdoSaoUnitCu left there (zero, or the pre-deblock sums of --sao-non-deblock) */
    const int32_t* s = tab + (size_t)addr * (2 * MAX_NUM_SAO_TYPE * MAX_NUM_SAO_CLASS);
    /* --limit-sao (sao.cpp:854-855): the two diagonal classes are only collected where the reference collects them -- never in B slices (and, as the reference writes
       the test, in every P or I slice).  The producer has them for every CTU; the ones the reference leaves at zero are not added */
    const Slice* slice = m_frame->m_encData->m_slice;
    const CUData* cu = m_frame->m_encData->getPicCTU(addr);
    const bool diagonals = !m_param->bLimitSAO || ((slice->m_sliceType == P_SLICE && !cu->isSkipped(0)) || (slice->m_sliceType != B_SLICE));
    for (int t = 0; t < MAX_NUM_SAO_TYPE; t++)
        for (int c = 0; c < MAX_NUM_SAO_CLASS; c++)
        {
            if (!diagonals && (t == SAO_EO_2 || t == SAO_EO_3)) continue;
            m_offsetOrg[plane][t][c] += s[t * MAX_NUM_SAO_CLASS + c];
            m_count[plane][t][c] += s[(MAX_NUM_SAO_TYPE + t) * MAX_NUM_SAO_CLASS + c];
        }
    rp->served++;
}

void FrameFilter::processRow(int row, int layer)
{
    const x265_param& p = *m_param;
    /* one picture at a time goes through the producer and the replay (g_lock) and the replay state of a picture stays published until the next one's: several frame
       threads (pictures in flight together, FrameData objects changing hands) keep the encoder's own filters */
    const bool mine = takes(*this, p, m_frame, m_useSao);
    if (mine && band_mode(p) && !g_deferOnly) { processBand(this, row, layer); return; }
    if (!mine)
    {
        if (replay_of(m_frame->m_encData)) { std::lock_guard<std::mutex> guard(g_lock); if (replay_of(m_frame->m_encData)) g_replay.store(nullptr, std::memory_order_release); }      /* (a recycled FrameData) */
        if (g_on && row == m_numRows - 1) { std::lock_guard<std::mutex> guard(g_statLock); g_stats.cpuPictures++; }
        ::processRow_cpu(this, row, layer);
        return;
    }
    /* the rows wait for the picture.  One slice: a row's filter runs behind the row above it, the last row's call is the last one.  --slices (WPP, param.cpp:1760): the slices'
       rows finish in any order (framefilter.cpp:633), the picture is complete when every row has called -- one picture at a time (one frame thread), so a counter tells */
    if (p.maxSlices > 1)
    {   /* counted per FrameFilter (= per frame encoder of an encoder instance): several encoders may live in one process */
        static std::mutex rowLock; static std::unordered_map<const FrameFilter*, int> rowsIn;
        std::lock_guard<std::mutex> rg(rowLock);
        int& n = rowsIn[this];
        if (++n != m_numRows) return;
        n = 0;
    }
    else if (row != m_numRows - 1) return;
    if (g_deferOnly)
    {   /* X265FF_DEFER_ONLY: the deferral alone, filters by the encoder's own bodies (separates the two things the binding changes; needs no GPU) */
        const double t0 = now();
        for (int r = 0; r < m_numRows; r++) ::processRow_cpu(this, r, layer);
        std::lock_guard<std::mutex> sg(g_statLock); g_stats.cpuPictures++; g_stats.replaySeconds += now() - t0;
        return;
    }

    std::lock_guard<std::mutex> guard(g_lock);
    g_replay.store(nullptr, std::memory_order_release);        /* the previous picture is through (one frame thread): its table is about to be overwritten */
    FrameData& encData = *m_frame->m_encData;
    Slice* slice = encData.m_slice;
    PicYuv* recon = m_frame->m_reconPic[0];
    PicYuv* fenc = m_frame->m_fencPic;
    x265hip_ff* ff = producer(p, *recon);
    if (!ff)
    {
        for (int r = 0; r < m_numRows; r++) ::processRow_cpu(this, r, layer);
        return;
    }
    const double t0 = now();
    Staging& S = g_st;
    x265hip_ff_picture_desc d;
    std::vector<uint8_t> sliceFirstRow;
    describe(*this, S, 0, m_numRows, d, sliceFirstRow);
    const double t1 = now();
    const int rc = g_api.ff_picture(ff, &d);
    const double t2 = now();
    if (rc) { fprintf(stderr, "filter_adapter: x265hip_ff_picture (POC %d): %d %s\n", slice->m_poc, rc, g_api.last_error()); die(); }
    Replay& rp = g_replayState;
    rp.data = &encData; rp.deblocked = p.bEnableLoopFilter != 0; rp.skipped = 0; rp.served = 0;
    rp.stats[0] = (d.saoStats & 1) ? S.stats[0].data() : NULL;
    rp.stats[1] = (d.saoStats & 2) ? S.stats[1].data() : NULL; rp.stats[2] = (d.saoStats & 2) ? S.stats[2].data() : NULL;
    g_replay.store(&rp, std::memory_order_release);
    for (int r = 0; r < m_numRows; r++) ::processRow_cpu(this, r, layer);
    const double t3 = now();
    std::lock_guard<std::mutex> sg(g_statLock);
    g_stats.pictures++; g_stats.bands++; g_stats.deblockSkipped += rp.skipped.load(); g_stats.statsServed += rp.served.load();
    g_stats.gatherSeconds += t1 - t0; g_stats.producerSeconds += t2 - t1; g_stats.replaySeconds += t3 - t2;
}

}

extern "C" int x265hip_ff_adapter_load(const char* libraryPath, int device)
{
    g_deferOnly = getenv("X265FF_DEFER_ONLY") && atoi(getenv("X265FF_DEFER_ONLY"));
    g_bandRows = getenv("X265FF_BAND_ROWS") ? atoi(getenv("X265FF_BAND_ROWS")) : 0;
    g_lib = dlopen(libraryPath, RTLD_NOW | RTLD_LOCAL);
    if (!g_lib) { fprintf(stderr, "filter_adapter: dlopen: %s\n", dlerror()); return -1; }
#define SYM(field, name) *(void**)&g_api.field = dlsym(g_lib, name); if (!g_api.field) { fprintf(stderr, "filter_adapter: %s lacks %s\n", libraryPath, name); return -1; }
    SYM(ctx_create, "x265hip_ctx_create") SYM(ctx_destroy, "x265hip_ctx_destroy") SYM(ff_create, "x265hip_ff_create") SYM(ff_destroy, "x265hip_ff_destroy")
    SYM(ff_picture, "x265hip_ff_picture") SYM(last_error, "x265hip_last_error")
#undef SYM
    g_device = device; g_on = 1;
    return 0;
}
extern "C" void x265hip_ff_adapter_enable(int on) { g_on = on && g_lib; }
extern "C" void x265hip_ff_adapter_close(void)
{
    std::lock_guard<std::mutex> guard(g_lock);
    if (g_ff) { g_api.ff_destroy(g_ff); g_ff = nullptr; }
    if (g_ctx) { g_api.ctx_destroy(g_ctx); g_ctx = nullptr; }
    g_on = 0;
    { std::lock_guard<std::mutex> bg(g_bandLock); g_bands.clear(); }        /* the FrameFilter objects the band states were keyed on go with their encoder */
}
extern "C" void x265hip_ff_adapter_get_stats(x265hip_ff_adapter_stats* o) { std::lock_guard<std::mutex> guard(g_statLock); *o = g_stats; }
