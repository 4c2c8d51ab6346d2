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
#include "xh_tme_chain.h"
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <map>
#include <string>
#include <memory>
#include <mutex>
using namespace xh;

namespace {
using namespace xh::tme;

__global__ __launch_bounds__(256) void tme_gather_kernel(Slice s, x265hip_tme_step st, int stepIdx, int nSteps, int pi, int l, int r, int nCtu,
                                                         const x265hip_inter_choice* __restrict__ table, const int16_t* __restrict__ areaBest,
                                                         const x265hip_tme_temporal* __restrict__ temporal, const x265hip_inter_choice* __restrict__ refTable,
                                                         const int16_t* __restrict__ lowresMv, TmeState* __restrict__ state, x265hip_select_task* __restrict__ sel)
{
    const int ctu = blockIdx.x * 256 + threadIdx.x;
    if (ctu >= nCtu) return;
    tme_gather(s, st, stepIdx, nSteps, pi, l, r, ctu, table, areaBest, temporal, refTable, lowresMv, state[ctu], sel[ctu]);
}
__global__ __launch_bounds__(256) void tme_build_kernel(Slice s, x265hip_tme_step st, int pi, int nCtu, const x265hip_select_result* __restrict__ selRes, TmeState* __restrict__ state,
                                                        x265hip_me_task* __restrict__ taskA, x265hip_me_task* __restrict__ taskB, const uint8_t* __restrict__ qpIndex, int stepIdx, int nSteps)
{
    const int ctu = blockIdx.x * 256 + threadIdx.x;
    if (ctu >= nCtu) return;
    tme_build(s, st, pi, ctu, selRes[ctu], state[ctu], taskA[ctu], taskB[ctu], qpIndex, stepIdx, nSteps);
}
__global__ __launch_bounds__(256) void tme_cost_kernel(Slice s, x265hip_tme_step st, int pi, int l, int r, int nCtu, const x265hip_me_result* __restrict__ resA,
                                                       const x265hip_me_result* __restrict__ resB, const uint16_t* __restrict__ costTable, int costHalf,
                                                       const float* __restrict__ bitsCentre, int bitsHalf, TmeState* __restrict__ state, const uint8_t* __restrict__ qpIndex, int stepIdx, int nSteps,
                                                       Lambdas lambdas)
{
    const int ctu = blockIdx.x * 256 + threadIdx.x;
    if (ctu >= nCtu) return;
    tme_cost(s, st, pi, l, r, ctu, resA[ctu], resB[ctu], costTable, costHalf, bitsCentre, bitsHalf, state[ctu], qpIndex, stepIdx, nSteps, lambdas);
}
__global__ __launch_bounds__(256) void tme_bidir_kernel(Slice s, x265hip_tme_step st, int pi, int nCtu, TmeState* __restrict__ state, x265hip_bidir_task* __restrict__ t0,
                                                        x265hip_bidir_task* __restrict__ t1, int8_t* __restrict__ ref0, int8_t* __restrict__ ref1)
{
    const int ctu = blockIdx.x * 256 + threadIdx.x;
    if (ctu >= nCtu) return;
    tme_bidir(s, st, pi, ctu, state[ctu], t0[ctu], t1[ctu], ref0[ctu], ref1[ctu]);
}
__global__ __launch_bounds__(256) void tme_finish_kernel(Slice s, x265hip_tme_step st, int pi, int nCtu, const int32_t* __restrict__ satd, const int32_t* __restrict__ satdZero,
                                                         const float* __restrict__ bitsCentre, int bitsHalf, TmeState* __restrict__ state, x265hip_inter_choice* __restrict__ table)
{
    const int ctu = blockIdx.x * 256 + threadIdx.x;
    if (ctu >= nCtu) return;
    tme_finish(s, st, pi, ctu, satd[ctu], satdZero[ctu], bitsCentre, bitsHalf, state[ctu], table);
}

// Entries of different PU shapes never read each other's slots (a PU's neighbours are PUs of its own shape, search.cpp:283-305), so every shape is its own chain of
// dependent launches: the chains run side by side on their own streams, each with its own slice of the workspace.
constexpr int XH_TME_CHAINS = 24;

} // namespace

namespace {
// the chain kernels' part of the workspace: schedule copy (<= 1024 entries), level lists, later-masks, the picture-start table
constexpr size_t kStepsBytes = 1024 * sizeof(x265hip_tme_step), kSchedBytes = 32 * XH_CHAIN_LEVELS * XH_CHAIN_WIDTH * sizeof(int16_t), kLaterBytes = 1024;
size_t chain_workspace(int nCtu) { return kStepsBytes + kSchedBytes + kLaterBytes + (size_t)nCtu * 593 * sizeof(x265hip_inter_choice) + 1024; }
size_t launch_workspace(int nCtu)
{
    const size_t per = sizeof(TmeState) + sizeof(x265hip_select_task) + sizeof(x265hip_select_result) + 2 * sizeof(x265hip_me_task) + 2 * sizeof(x265hip_me_result) +
                       2 * sizeof(x265hip_bidir_task) + 2 * sizeof(int32_t) + 2;
    return ((size_t)nCtu * per + 16 * 256) * XH_TME_CHAINS;
}
// Side streams of the chain launches of one caller stream (the kernel configurations run side by side)
struct ChainStreams { hipStream_t cs[8]; hipEvent_t fork, join[8]; };
std::mutex g_chainMu;
std::map<std::pair<int, hipStream_t>, std::unique_ptr<ChainStreams>> g_chainSets;      // keyed by (device, caller's stream); dropped by x265hip_tme_release_stream
ChainStreams* chain_streams(int device, hipStream_t mainStream)
{
    std::mutex& mu = g_chainMu;
    auto& sets = g_chainSets;
    const std::pair<int, hipStream_t> key(device, mainStream);
    std::lock_guard<std::mutex> lock(mu);
    auto it = sets.find(key);
    if (it != sets.end()) return it->second.get();
    std::unique_ptr<ChainStreams> up(new ChainStreams());
    if (hipEventCreateWithFlags(&up->fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    for (int i = 0; i < 8; i++)
        if (hipStreamCreateWithFlags(&up->cs[i], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&up->join[i], hipEventDisableTiming) != hipSuccess) return nullptr;
    return (sets[key] = std::move(up)).get();
}
// The level lists of a schedule (built once per schedule, kept: the copies to the device read them asynchronously)
struct ChainPlan { std::vector<int16_t> sched; std::vector<uint8_t> later; std::vector<int> shapeKeys, nLevels, firstStep; };
const ChainPlan* chain_plan(const x265hip_tme_step* steps, int nSteps)
{
    static std::mutex mu;
    static std::map<std::string, std::unique_ptr<ChainPlan>> plans;
    const std::string key((const char*)steps, (size_t)nSteps * sizeof(x265hip_tme_step));
    std::lock_guard<std::mutex> lock(mu);
    auto it = plans.find(key);
    if (it != plans.end()) return it->second.get();
    std::unique_ptr<ChainPlan> up(new ChainPlan());
    ChainPlan& P = *up;
    {
        std::vector<int> owner(593 * 2, -1), level(nSteps, 0), keyOf(nSteps);
        std::vector<uint8_t>& later = P.later; later.assign(nSteps, 0);
        std::vector<int>& shapeKeys = P.shapeKeys;                                          // row of `sched` = index here
        for (int k = 0; k < nSteps; k++)
        {
            const x265hip_tme_step& e = steps[k];
            keyOf[k] = e.cuSize * 8 + e.part;
            for (int pi = 0; pi < e.numPart; pi++) { const int slot = e.finalIdx + pi * e.puOffset; if (slot >= 0 && slot < (int)owner.size()) owner[slot] = k; }
        }
        std::vector<int16_t>& sched = P.sched; sched.assign((size_t)32 * XH_CHAIN_LEVELS * XH_CHAIN_WIDTH, (int16_t)-1);
        std::vector<int>& nLevels = P.nLevels; std::vector<int>& firstStep = P.firstStep; std::vector<int> widthOf(32, 1);
        nLevels.assign(32, 0); firstStep.assign(32, -1);
        for (int k = 0; k < nSteps; k++)
        {
            const x265hip_tme_step& e = steps[k];
            int row = 0; while (row < (int)shapeKeys.size() && shapeKeys[row] != keyOf[k]) row++;
            if (row == (int)shapeKeys.size())
            {
                if (row == 32) { set_error("tme_frame: more than 32 PU shapes"); return nullptr; }
                shapeKeys.push_back(keyOf[k]); firstStep[row] = k; widthOf[row] = xh_chain_width(xh_chain_config(e.cuSize, e.part));
            }
            int lv = 0;
            for (int d = 0; d < 5; d++)
            {
                const int slot = e.neighbor[d];
                if (slot < 0 || slot >= (int)owner.size() || owner[slot] < 0) continue;
                const int o = owner[slot];
                if (o > k) { later[k] |= (uint8_t)(1 << d); continue; }
                if (o == k) continue;
                if (keyOf[o] != keyOf[k]) { set_error("tme_frame: entry %d reads a record of another PU shape", k); return nullptr; }
                lv = std::max(lv, level[o] + 1);
            }
            // a level holds as many entries as the shape's kernel runs side by side; a full one pushes the entry down (never up: its neighbours stay below it)
            for (;; lv++)
            {
                if (lv >= XH_CHAIN_LEVELS) { set_error("tme_frame: more than %d levels in a PU shape's chain", XH_CHAIN_LEVELS); return nullptr; }
                int16_t* slots = &sched[((size_t)row * XH_CHAIN_LEVELS + lv) * XH_CHAIN_WIDTH];
                int j = 0; while (j < widthOf[row] && slots[j] >= 0) j++;
                if (j < widthOf[row]) { slots[j] = (int16_t)k; break; }
            }
            level[k] = lv; nLevels[row] = std::max(nLevels[row], lv + 1);
        }
    }
    return (plans[key] = std::move(up)).get();
}
}

extern "C" size_t x265hip_tme_workspace(int nCtu)
{
    const size_t per = sizeof(TmeState) + sizeof(x265hip_select_task) + sizeof(x265hip_select_result) + 2 * sizeof(x265hip_me_task) + 2 * sizeof(x265hip_me_result) +
                       2 * sizeof(x265hip_bidir_task) + 2 * sizeof(int32_t) + 2;
    return ((size_t)nCtu * per + 16 * 256) * XH_TME_CHAINS + chain_workspace(nCtu);
}

// the side streams x265hip_tme_frame keeps for a caller's stream: to be released before that stream is destroyed (x265hip_ctx_destroy does it), so that a recycled
// handle never finds another stream's set
extern "C" void x265hip_tme_release_stream(void* stream)
{
    std::lock_guard<std::mutex> lock(g_chainMu);
    for (auto it = g_chainSets.begin(); it != g_chainSets.end();)
        if (it->first.second == (hipStream_t)stream)
        {
            for (int i = 0; i < 8; i++) { (void)hipStreamSynchronize(it->second->cs[i]); (void)hipStreamDestroy(it->second->cs[i]); (void)hipEventDestroy(it->second->join[i]); }
            (void)hipEventDestroy(it->second->fork);
            it = g_chainSets.erase(it);
        }
        else ++it;
}

extern "C" int x265hip_tme_frame(void* stream, const x265hip_tme_args* a)
{
    if (!a || !a->steps || a->nSteps < 1 || !a->curPlane || !a->table || !a->areaBest || !a->temporal || !a->bitsRow || !a->workspace || a->nQp < 1 || a->nQp > 64 || (a->nQp > 1 && !a->qpIndex)) { set_error("tme_frame: missing arguments"); return X265HIP_EARG; }
    if (!a->costRows) { set_error("tme_frame: cost table missing"); return X265HIP_EARG; }
    if (a->ctuSize < 16 || a->picWidth < 8 || a->picHeight < 8) { set_error("tme_frame: bad picture / CTU size"); return X265HIP_EARG; }
    // CTUs cut by the picture edge run the whole schedule like the others -- computeMVForPUs does not look at the picture size; PUs beyond the edge read the planes' padding
    const int nCtuX = (a->picWidth + a->ctuSize - 1) / a->ctuSize, nCtuY = (a->picHeight + a->ctuSize - 1) / a->ctuSize, nCtu = nCtuX * nCtuY;
    if (a->workspaceBytes < x265hip_tme_workspace(nCtu)) { set_error("tme_frame: workspace too small"); return X265HIP_EARG; }
    if (a->ctuFirst < 0 || a->ctuCount < 0 || a->ctuFirst + a->ctuCount > nCtu || (a->ctuFirst && !a->ctuCount)) { set_error("tme_frame: CTUs %d + %d of %d", a->ctuFirst, a->ctuCount, nCtu); return X265HIP_EARG; }
    const int c0 = a->ctuFirst, nBand = a->ctuCount ? a->ctuCount : nCtu;                 // the CTUs of this call (per-CTU arrays keep the picture's addressing)
    const int nl = a->isP ? 1 : 2;
    for (int l = 0; l < nl; l++)
    {
        if (a->numRef[l] < 1 || a->numRef[l] > X265HIP_MAX_REF) { set_error("tme_frame: %d references in list %d", a->numRef[l], l); return X265HIP_EARG; }
        for (int r = 0; r < a->numRef[l]; r++) if (!a->refs[l][r].mePlane || !a->refs[l][r].mePhase || !a->refs[l][r].reconPhase) { set_error("tme_frame: planes of list %d reference %d missing", l, r); return X265HIP_EARG; }
    }
    // the launches, side streams and events below belong to the device of the caller's stream, whatever device the calling thread had selected
    int device = 0;
    if (stream) { hipDevice_t dv = 0; XH_HIP(hipStreamGetDevice((hipStream_t)stream, &dv)); device = (int)dv; } else XH_HIP(hipGetDevice(&device));
    XH_HIP(hipSetDevice(device));
    hipStream_t mainStream = (hipStream_t)stream;
    const bool launches = (a->flags & X265HIP_TME_LAUNCH_PER_STAGE) != 0;               // the diagnostic form: every stage of every entry its own launch (below)
    if (!launches && (a->searchMethod == X265HIP_ME_DIA || a->searchMethod == X265HIP_ME_HEX || a->searchMethod == X265HIP_ME_STAR || a->searchMethod == X265HIP_ME_FULL || a->searchMethod == X265HIP_ME_UMH))
    {   // ---- the chains inside the kernels (tme_chain.inc): one launch per kernel configuration, side by side on their own streams ----
        const bool packed = (a->flags & X265HIP_TME_PACKED_GROUPS) != 0;
        ChainStreams* side = chain_streams(device, mainStream);                                 // one set per caller's stream, whichever thread calls
        if (!side) { set_error("tme_frame: could not create the chain streams"); return X265HIP_EDEVICE; }
        hipStream_t* cs = side->cs; hipEvent_t cFork = side->fork; hipEvent_t* cJoin = side->join;
        if (a->nSteps > 1024) { set_error("tme_frame: more than 1024 schedule entries"); return X265HIP_EARG; }
        char* ws = (char*)a->workspace;
        x265hip_tme_step* dSteps = (x265hip_tme_step*)ws; int16_t* dSched = (int16_t*)(ws + kStepsBytes); uint8_t* dLater = (uint8_t*)(ws + kStepsBytes + kSchedBytes);
        x265hip_inter_choice* dInit = (x265hip_inter_choice*)(ws + kStepsBytes + kSchedBytes + kLaterBytes);
        // ---- levels: an entry reads the records of its neighbours of the same shape; those that precede it in the schedule must be finished (-> a lower level), those that
        //      follow it are read as the picture found them (-> bit in `later`, read from the picture-start copy).  Entries of one level run side by side. ----
        const ChainPlan* plan = chain_plan(a->steps, a->nSteps);
        if (!plan) return X265HIP_EARG;
        const std::vector<int16_t>& sched = plan->sched; const std::vector<uint8_t>& later = plan->later;
        const std::vector<int>& shapeKeys = plan->shapeKeys; const std::vector<int>& nLevels = plan->nLevels; const std::vector<int>& firstStep = plan->firstStep;
        XH_HIP(hipMemcpyAsync(dSteps, a->steps, (size_t)a->nSteps * sizeof(x265hip_tme_step), hipMemcpyHostToDevice, mainStream));
        XH_HIP(hipMemcpyAsync(dSched, sched.data(), sched.size() * sizeof(int16_t), hipMemcpyHostToDevice, mainStream));
        XH_HIP(hipMemcpyAsync(dLater, later.data(), later.size(), hipMemcpyHostToDevice, mainStream));
        XH_HIP(hipMemcpyAsync(dInit + (size_t)c0 * 593, a->table + (size_t)c0 * 593, (size_t)nBand * 593 * sizeof(x265hip_inter_choice), hipMemcpyDeviceToDevice, mainStream));
        xh_chain_args A{};
        A.s.isP = a->isP; A.s.numRef[0] = a->numRef[0]; A.s.numRef[1] = a->isP ? 0 : a->numRef[1]; A.s.searchRange = a->searchRange; A.s.picW = a->picWidth; A.s.picH = a->picHeight;
        A.s.ctuSize = a->ctuSize; A.s.numCtuX = nCtuX; A.s.lowresBlocksX = a->lowresBlocksX; A.s.stride = a->stride; A.s.origin = a->origin;
        A.s.frameParallel = a->frameParallel != 0; A.s.refLag = a->refLagPixels > 0 ? a->refLagPixels : a->picHeight;
        A.s.pirStartCol = a->pirStartCol; A.s.pirSafeX = a->pirSafeX;
        A.s.amvp.curPOC = a->curPOC; A.s.amvp.temporalMvp = a->temporalMvp;
        for (int l = 0; l < 2; l++) for (int r = 0; r < 16; r++) A.s.amvp.refPOC[l][r] = a->refPOC[l][r];
        for (int q = 0; q < a->nQp; q++) A.lambdas.v[q] = a->lambdas[q];
        A.sched = dSched; A.later = dLater; A.tableInit = dInit;
        A.steps = dSteps; A.nSteps = a->nSteps; A.nCtu = nBand; A.ctuFirst = c0; A.cur = (const pixel*)a->curPlane; A.planeElems = a->planeElems;
        for (int l = 0; l < nl; l++)
            for (int r = 0; r < a->numRef[l]; r++)
            {
                const x265hip_tme_ref& R = a->refs[l][r];
                A.refs[l][r].mePlane = (const pixel*)R.mePlane; A.refs[l][r].mePhase = (const pixel*)R.mePhase; A.refs[l][r].reconPhase = (const pixel*)R.reconPhase;
                A.refs[l][r].refTable = R.refTable; A.refs[l][r].lowresMv = R.lowresMv;
            }
        A.table = a->table; A.areaBest = a->areaBest; A.temporal = a->temporal; A.qpIndex = a->qpIndex;
        A.costRows = a->costRows; A.costHalf = a->costHalfRange; A.bitsCentre = a->bitsRow + a->bitsHalfRange; A.bitsHalf = a->bitsHalfRange;
        A.searchRange = a->searchRange; A.method = a->searchMethod; A.subme = a->subpelRefine;
        XH_HIP(hipEventRecord(cFork, mainStream));
        for (int config = 0; config < 8; config++)
        {
            int nKeys = 0;
            for (int row = 0; row < (int)shapeKeys.size(); row++)
            {
                if (xh_chain_config(shapeKeys[row] >> 3, shapeKeys[row] & 7) != config) continue;
                if (nKeys == 8) { set_error("tme_frame: more than 8 PU shapes in one kernel configuration"); return X265HIP_EARG; }
                A.keys[nKeys] = shapeKeys[row]; A.keyRow[nKeys] = row; A.nLevels[nKeys] = nLevels[row]; A.firstStep[nKeys] = firstStep[row]; nKeys++;
            }
            if (!nKeys) continue;
            XH_HIP(hipStreamWaitEvent(cs[config], cFork, 0));
            const int rc = a->searchMethod == X265HIP_ME_STAR ? xh_tme_chain_star(cs[config], config, &A, nKeys, packed) : a->searchMethod == X265HIP_ME_UMH ? xh_tme_chain_umh(cs[config], config, &A, nKeys, packed)
                                                              : xh_tme_chain_hex(cs[config], config, &A, nKeys, packed);
            if (rc) return rc;
            XH_HIP(hipEventRecord(cJoin[config], cs[config]));
            XH_HIP(hipStreamWaitEvent(mainStream, cJoin[config], 0));
        }
        return X265HIP_OK;
    }
    if (a->ctuCount) { set_error("tme_frame: a band of CTUs is offered by the chain kernels only (not with X265HIP_TME_LAUNCH_PER_STAGE / SEA)"); return X265HIP_EARG; }
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
    const size_t chainBytes = launch_workspace(nCtu) / XH_TME_CHAINS;
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
    const void* ph0[X265HIP_MAX_REF] = {}; const void* ph1[X265HIP_MAX_REF] = {};
    for (int r = 0; r < X265HIP_MAX_REF; r++) { ph0[r] = r < a->numRef[0] ? a->refs[0][r].reconPhase : nullptr; ph1[r] = (!a->isP && r < a->numRef[1]) ? a->refs[1][r].reconPhase : nullptr; }
    Slice s{};
    s.isP = a->isP; s.numRef[0] = a->numRef[0]; s.numRef[1] = a->isP ? 0 : a->numRef[1]; s.searchRange = a->searchRange; s.picW = a->picWidth; s.picH = a->picHeight;
    s.ctuSize = a->ctuSize; s.numCtuX = nCtuX; s.lowresBlocksX = a->lowresBlocksX; s.stride = a->stride; s.origin = a->origin;
    s.frameParallel = a->frameParallel != 0; s.refLag = a->refLagPixels > 0 ? a->refLagPixels : a->picHeight;
    s.pirStartCol = a->pirStartCol; s.pirSafeX = a->pirSafeX;
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
                    XH_KLAUNCH(tme_gather_kernel, grid, block, 0, st, s, e, k, a->nSteps, pi, l, r, nCtu, a->table, a->areaBest, a->temporal, R.refTable, R.lowresMv, state, sel);
                    int rc = x265hip_select_mvp_batch(stream, pw, ph, a->curPlane, a->stride, R.reconPhase, a->planeElems, a->stride, sel, nCtu, selRes);
                    if (rc) { set_error("tme_frame: select_mvp_batch %dx%d failed", pw, ph); return rc; }
                    XH_KLAUNCH(tme_build_kernel, grid, block, 0, st, s, e, pi, nCtu, selRes, state, tA, tB, a->qpIndex, k, a->nSteps);
                    rc = x265hip_me_batch_rows(stream, pw, ph, a->curPlane, a->stride, R.mePlane, a->stride, tA, nCtu, a->costRows, a->costHalfRange, a->searchRange, a->searchMethod,
                                          a->subpelRefine, rA, nullptr, R.mePhase, a->planeElems);
                    if (rc) return rc;
                    rc = x265hip_me_batch_rows(stream, pw, ph, a->curPlane, a->stride, R.mePlane, a->stride, tB, nCtu, a->costRows, a->costHalfRange, a->searchRange, a->searchMethod,
                                          a->subpelRefine, rB, nullptr, R.mePhase, a->planeElems);
                    if (rc) return rc;
                    XH_KLAUNCH(tme_cost_kernel, grid, block, 0, st, s, e, pi, l, r, nCtu, rA, rB, a->costRows, a->costHalfRange, bitsCentre, a->bitsHalfRange, state,
                                       a->qpIndex, k, a->nSteps, lambdas);
                }
            XH_KLAUNCH(tme_bidir_kernel, grid, block, 0, st, s, e, pi, nCtu, state, b0, b1, br0, br1);
            if (!a->isP && e.part != 0 && e.cuSize != 8)
            {   // the bidirectional candidate exists for this shape: its two distortions, each PU against the references it chose per list
                int rc = x265hip_bidir_satd_batch_refs(stream, pw, ph, a->curPlane, a->stride, ph0, ph1, a->planeElems, a->stride, b0, br0, br1, nCtu, s0);
                if (rc) return rc;
                rc = x265hip_bidir_satd_batch_refs(stream, pw, ph, a->curPlane, a->stride, ph0, ph1, a->planeElems, a->stride, b1, br0, br1, nCtu, s1);
                if (rc) return rc;
            }
            XH_KLAUNCH(tme_finish_kernel, grid, block, 0, st, s, e, pi, nCtu, s0, s1, bitsCentre, a->bitsHalfRange, state, a->table);
        }
    }
    XH_HIP(hipEventRecord(evJoin[chain], st));
    XH_HIP(hipStreamWaitEvent(mainStream, evJoin[chain], 0));
    }
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

// ---- table transfers of the host producer: a schedule touches some of the 593 slots of a CTU (85 with preset medium's partitions); only those cross the bus ----
namespace {
__global__ __launch_bounds__(256) void tme_slots_kernel(x265hip_inter_choice* __restrict__ table, x265hip_inter_choice* __restrict__ packed, const int32_t* __restrict__ slots, int nUsed, int total, int toTable)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ctu = i / nUsed, k = i - ctu * nUsed;
    x265hip_inter_choice* a = table + (int64_t)ctu * 593 + slots[k];
    if (toTable) *a = packed[i]; else packed[i] = *a;
}
}
int xh_tme_slots(void* stream, x265hip_inter_choice* table, x265hip_inter_choice* packed, const int32_t* slots, int nUsed, int nCtu, int toTable)
{
    const int total = nUsed * nCtu;
    if (total <= 0) return X265HIP_OK;
    XH_KLAUNCH(tme_slots_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, packed, slots, nUsed, total, toTable);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

// ---- m_areaBestMV from the diamond stage's results, on the device (analysis.cpp:262-299): result i of (list l, reference r) belongs to area where[i] = ctu * 5 + area;
//      the collocated median, where the caller found one, replaces the five areas of its CTU ----
namespace {
__global__ __launch_bounds__(256) void tme_area_kernel(const x265hip_me_result* __restrict__ res, const int32_t* __restrict__ where, int nTasks, int nl, int numRef0, int numRef1,
                                                       const int16_t* __restrict__ median, int16_t* __restrict__ areaBest)
{
    const int i = blockIdx.x * 256 + threadIdx.x, lr = blockIdx.y, l = lr / X265HIP_MAX_REF, r = lr % X265HIP_MAX_REF;
    if (i >= nTasks || l >= nl || r >= (l ? numRef1 : numRef0)) return;
    const int ca = where[i], c = ca / 5;
    int mx = res[(int64_t)lr * nTasks + i].mv[0], my = res[(int64_t)lr * nTasks + i].mv[1];                 // the full-pel MV as the reference stores it (search.cpp:363)
    if (median) { const int16_t* m = median + (((int64_t)c * 2 + l) * X265HIP_MAX_REF + r) * 3; if (m[0]) { mx = m[1]; my = m[2]; } }
    int16_t* o = areaBest + (((int64_t)ca * 2 + l) * X265HIP_MAX_REF + r) * 2;
    o[0] = (int16_t)mx; o[1] = (int16_t)my;
}
}
int xh_tme_area(void* stream, const x265hip_me_result* res, const int32_t* where, int nTasks, int nl, int numRef0, int numRef1, const int16_t* median, int16_t* areaBest)
{
    if (nTasks <= 0) return X265HIP_OK;
    XH_KLAUNCH(tme_area_kernel, dim3((nTasks + 255) / 256, 2 * X265HIP_MAX_REF), dim3(256), 0, (hipStream_t)stream, res, where, nTasks, nl, numRef0, numRef1, median, areaBest);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
