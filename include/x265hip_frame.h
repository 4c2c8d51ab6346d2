/*
 * x265hip_frame.h -- frame/CTU-granular batched entry points (device-resident planes).
 *
 * These are the throughput path: what the per-call table slots of x265hip.h compute one block at a
 * time, computed for every PU / TU of a frame in one launch.  The contracts mirror the reference's
 * own decoupled-ME seam (encoder/threadedme.h:122-130 `MEData`, search.cpp:226-560
 * `puMotionEstimation`): per (PU, reference) a search window, a predictor and candidates go in, a
 * quarter-pel MV and its cost come out in a flat array.
 *
 * All pointers are DEVICE memory; `stream` is a hipStream_t.  Planes are padded like the reference's
 * PicYuv (picyuv.cpp:91-111): every address the search can touch -- block + mv range + 8 pixels of
 * interpolation / pattern overshoot -- must be inside the allocation.
 */
#ifndef X265HIP_FRAME_H
#define X265HIP_FRAME_H
#include "x265hip.h"

#define X265HIP_MAX_REF 16                     /* references per list: MAX_NUM_REF (common/common.h:329; m_areaBestMV[5][2][MAX_NUM_REF], encoder/search.h:297) */

/* the library is built with -fvisibility=hidden: what these headers declare is its whole exported surface */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif
#ifdef __cplusplus
extern "C" {
#endif

/* search methods, numbered like X265_*_SEARCH (reference x265.h:511-518) */
enum x265hip_me_method { X265HIP_ME_DIA = 0, X265HIP_ME_HEX = 1, X265HIP_ME_UMH = 2, X265HIP_ME_STAR = 3,
                         X265HIP_ME_SEA = 4, X265HIP_ME_FULL = 5 };

/* one (PU, reference) search; all PUs of one x265hip_me_batch call share (w, h).
 * replaces: MotionEstimate::setSourcePU + motionEstimate (motion.cpp:203-231, 923-1773) */
typedef struct x265hip_me_task {
    int32_t curOff;              /* element offset of the PU's top-left pixel in the source plane   */
    int32_t refOff;              /* element offset of the co-located pixel in the reference plane   */
    int16_t mvmin[2], mvmax[2];  /* full-pel search bounds (x, y), inclusive (motion.cpp:925-926);
                                    with X265HIP_ME_WINDOW: QUARTER-pel clip limits (CUData::clipMv)   */
    int16_t qmvp[2];             /* quarter-pel MV predictor (ignored when mvpFrom >= 0)             */
    int16_t mvc[24];             /* up to 12 quarter-pel candidates (x, y): the mvc[(MD_ABOVE_LEFT + 1) * 2 + 2] list of
                                    Search::puMotionEstimation / predInterSearch (search.cpp:237, 2599)              */
    int16_t numCand;             /* 0..12                                                            */
    int16_t flags;               /* X265HIP_ME_*                                                     */
    int32_t mvpFrom;             /* >= 0: predictor = mvpSource[mvpFrom].mv (e.g. the parent CU's MV, the way
                                    Analysis::deriveMVsForCTU seeds PUs from m_areaBestMV, analysis.cpp:248-306) */
} x265hip_me_task;               /* 76 bytes */

/* flags: derive the search window on the device the way Search::setSearchRange does (search.cpp:4969-5021):
 * [mvp - 4*merange, mvp + 4*merange] clipped to the task's quarter-pel limits, >> 2, mvmax.y >= mvmin.y */
#define X265HIP_ME_WINDOW 1
/* flags, honoured by x265hip_me_batch_rows only: the task has its own MVD cost row (CUs of different qp in one launch: Analysis::setLambdaFromQP runs per CU).
 * costRows of that call is a TABLE of rows, 2 * costHalfRange + 1 entries each, and bits 8..15 of flags are the task's row index. */
#define X265HIP_ME_ROWS 2

typedef struct x265hip_me_result {
    int16_t mv[2];               /* chosen quarter-pel MV (outQMv)                                   */
    int32_t cost;                /* return value of motionEstimate: distortion + lambda * MVD bits   */
    int32_t mvcost;              /* the lambda-scaled MVD cost of mv (BitCost::mvcost)               */
    int32_t reserved;
} x265hip_me_result;             /* 16 bytes */

/* Host helper: the lambda-scaled MVD cost row of BitCost::setQP / CalculateLogs (bitcost.cpp:30-105) for this
 * library's bit depth: out[halfRange + d] = min(uint16(bits(|d|) * lambda(qp) + 0.5), 32767), d in quarter-pels,
 * bits(0) = 0.718, bits(i) = log(i+1) * 2/log(2) + 1.718 in the reference's float/double mix, lambda =
 * x265_lambda_tab[qp] (constants.cpp:28-116).  Pure host code (no GPU needed); upload the row for x265hip_me_batch. */
int x265hip_mvcost_row(int qp, int halfRange, uint16_t* out /* 2*halfRange+1 */);

/* costRow: device uint16 table of 2*costHalfRange+1 entries, entry [costHalfRange + d] = lambda-scaled cost
 * of an MVD component d in quarter-pels -- the row BitCost::setQP builds (bitcost.cpp:30-56); it is an
 * INPUT so that host and device use the very same table.  costHalfRange must cover |mv - mvp| (and
 * 8 * mv for the STAR raster quirk, motion.cpp:1392). */
int x265hip_me_batch(void* stream, int w, int h,
                     const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                     const x265hip_me_task* tasks, int n,
                     const uint16_t* costRow, int costHalfRange,
                     int merange, int method, int subpelRefine, x265hip_me_result* results,
                     const x265hip_me_result* mvpSource /* may be NULL */,
                     const void* subpelPlanes /* may be NULL: interpolate inside the kernel */, int64_t planeElems);
/* The same search with a cost row per task (X265HIP_ME_ROWS above; DIA / HEX / UMH / STAR / FULL).  Its own kernels: such tasks read their row from memory instead of the
 * workgroup's LDS slice, and the plain entry point does not carry that test (it costs its small-PU kernels a third of their speed). */
int x265hip_me_batch_rows(void* stream, int w, int h,
                          const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                          const x265hip_me_task* tasks, int n,
                          const uint16_t* costRows, int costHalfRange,
                          int merange, int method, int subpelRefine, x265hip_me_result* results,
                          const x265hip_me_result* mvpSource, const void* subpelPlanes, int64_t planeElems);
/* subpelPlanes, when given, must be the 16-slot buffer x265hip_subpel_planes produced from refPlane (slot 0 == refPlane,
 * same stride / offsets).  method: DIA, HEX, UMH, STAR or FULL (SEA needs integral planes: x265hip_me_batch_sea). */

/* x265hip_me_batch with the chroma SATD terms of MotionEstimate::subpelCompare (motion.cpp:1805-1865): the search Search::predInterSearch runs
 * (setSourcePU's Yuv overload with bChroma, motion.cpp:218-247; search.cpp:2582).  4:2:0.  At subpelRefine >= 3, for PUs whose chroma block is a
 * multiple of 4x4 (the reference's chromaSatd exists: primitives.cpp:213-234), EVERY sub-pel cost -- clipped MVP and candidates, half / quarter-pel
 * refinement, zero-MV check -- carries SATD(Cb) + SATD(Cr) of the prediction at the eighth-pel chroma MV (4-tap filters); otherwise the result
 * equals x265hip_me_batch's.  subpelPlanes is required.  Methods DIA, HEX, STAR, FULL. */
typedef struct x265hip_me_chroma {
    const void *curCb, *curCr; intptr_t curStrideC;   /* source chroma planes                                                              */
    const void *refCb, *refCr; intptr_t refStrideC;   /* reference chroma planes, padded like the luma plane (half the margins)            */
    const int32_t* curOffC;                           /* device array, per task: element offset of the PU's chroma block in curCb / curCr  */
    const int32_t* refOffC;                           /* device array, per task: the co-located offset in refCb / refCr                    */
} x265hip_me_chroma;
int x265hip_me_batch_chroma(void* stream, int w, int h,
                            const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                            const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
                            int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
                            const void* subpelPlanes, int64_t planeElems, const x265hip_me_chroma* chroma);

/* ---- the order of ThreadedME's PU stage --------------------------------------------------------------------------------------------------------------------
 * x265hip_tme_schedule (host, no GPU): the calls Analysis::computeMVForPUs makes for one CTU (analysis.cpp:161-246), in its order -- sub-CUs first, then every PU
 * shape of the CU in g_puLookup's order (threadedme.h:67-92) -- each with the slot of its first partition in the CTU's MEData table (finalIdx, from
 * ThreadedME::initPuStartIdx, threadedme.cpp:86-107), the slot distance of the second partition, the neighbour slots Search::puMotionEstimation reads
 * (MVP_DIR order; -1 = none) and the partitions' rectangles inside the CTU.  The schedule is the same for every CTU (CTUs cut by the picture edge skip the
 * entries whose CU lies outside).  The area index of an entry is position dependent in the reference (analysis.cpp:175-179 compares the CU's ABSOLUTE
 * position with half a CTU): areaIdx = cuSize == ctuSize ? 0 : (ctuX + cuX >= ctuSize / 2) + 2 * (ctuY + cuY >= ctuSize / 2) + 1.
 * Returns the number of entries (also when maxSteps is smaller: call with 0 to size the array). */
typedef struct x265hip_tme_step {
    int16_t part, cuSize, cuX, cuY;        /* enum PartSize (2Nx2N 0, 2NxN 1, Nx2N 2, 2NxnU 4, 2NxnD 5, nLx2N 6, nRx2N 7); CU size and position inside the CTU */
    int16_t puOffset, finalIdx;            /* MEData slot of partition i = finalIdx + i * puOffset                                                           */
    int16_t neighbor[5];                   /* slots of the left, above, above-right, below-left (never), above-left neighbour of the same shape              */
    int16_t numPart;
    int16_t pu[2][4];                      /* per partition: x, y inside the CTU, width, height                                                              */
} x265hip_tme_step;                        /* 40 bytes */
int x265hip_tme_schedule(int ctuSize, int minCuSize, int rect, int amp, x265hip_tme_step* steps, int maxSteps);

/* The distortion of the bidirectional candidate alone (search.cpp:436-446): predInterLumaPixel of the list-0 and the list-1 reference at the given quarter-pel MVs (blocks
 * of their phase planes), pixelavg_pp, SATD against the source PU.  What x265hip_inter_merge_batch computes inside, for callers that keep the bit bookkeeping. */
typedef struct x265hip_bidir_task { int32_t curOff, refOff; int16_t mv0[2], mv1[2]; } x265hip_bidir_task;   /* 16 bytes */
int x265hip_bidir_satd_batch(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* subpelPlanes0, const void* subpelPlanes1, int64_t planeElems,
                             intptr_t refStride, const x265hip_bidir_task* tasks, int n, int32_t* satd);
/* ... with the references chosen per task: subpelPlanes0[r] / subpelPlanes1[r] (r = 0..X265HIP_MAX_REF-1, unused entries NULL) and device arrays ref0[i] / ref1[i] */
int x265hip_bidir_satd_batch_refs(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* const* subpelPlanes0, const void* const* subpelPlanes1,
                                  int64_t planeElems, intptr_t refStride, const x265hip_bidir_task* tasks, const int8_t* ref0, const int8_t* ref1, int n, int32_t* satd);

/* ---- AMVP: CUData::getPMV (common/cudata.cpp:1806-1990) for a batch of (PU, list, reference) --------------------------------------------------------------
 * In: the PU's neighbour records in MVP_DIR order (cudata.h:67-75: LEFT, ABOVE, ABOVE_RIGHT, BELOW_LEFT, ABOVE_LEFT, COLLOCATED) as CUData::getNeighbourMV /
 * Search::puMotionEstimation (search.cpp:283-305) fill InterNeighbourMV: per list an MV and a reference index (-1 = none); for COLLOCATED refIdx[list] is the
 * unified index (bit 4 = list of the collocated block's reference, low bits = index; -1 = no temporal candidate) and colPOC / colRefPOC are the POCs
 * :1962-1967 looks up in the collocated picture (the host resolves them).  Out: the two AMVP candidates (zero-filled) and the motion-candidate list handed to
 * motionEstimate (at most 11 entries).  Replaces getPMV + getDirectPMV + getIndirectPMV + scaleMvByPOCDist. */
typedef struct x265hip_amvp_neighbour { int16_t mv[2][2]; int8_t refIdx[2]; int8_t available; int8_t reserved; } x265hip_amvp_neighbour;   /* 12 bytes */
typedef struct x265hip_amvp_task {
    x265hip_amvp_neighbour nb[6];
    int8_t  list, refIdx; int16_t reserved;
    int32_t colPOC, colRefPOC;
} x265hip_amvp_task;                /* 84 bytes */
typedef struct x265hip_amvp_params { int curPOC; int temporalMvp; int refPOC[2][16]; } x265hip_amvp_params;   /* Slice::m_poc, SPS::bTemporalMVPEnabled, Slice::m_refPOCList */
typedef struct x265hip_amvp_result { int16_t amvp[2][2]; int16_t numMvc; int16_t mvc[11][2]; int16_t reserved; } x265hip_amvp_result;   /* 56 bytes */
int x265hip_amvp_batch(void* stream, const x265hip_amvp_task* tasks, int n, const x265hip_amvp_params* params, x265hip_amvp_result* out);

/* Search::selectMVP (search.cpp:2347-2382) for n PUs of one size: both AMVP candidates clipped to `clip` (CUData::clipMv's limits of the CU: xmin, ymin, xmax, ymax in
 * quarter-pels, cudata.cpp:2094-2107), motion compensated out of the reference's phase planes and compared with the source PU at SAD; mvpIdx = 0 on a tie or when the
 * candidates are equal (then no cost is measured).  The m_bFrameParallel exclusions (:2360-2371) are the caller's (they do not depend on pixels). */
typedef struct x265hip_select_task { int32_t curOff, refOff; int16_t amvp[2][2]; int32_t clip[4]; } x265hip_select_task;        /* 32 bytes */
typedef struct x265hip_select_result { int32_t mvpIdx; int32_t cost[2]; } x265hip_select_result;
int x265hip_select_mvp_batch(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* subpelPlanes, int64_t planeElems, intptr_t refStride,
                             const x265hip_select_task* tasks, int n, x265hip_select_result* out);
/* Search::updateMVP + checkBestMVP (search.cpp:4947-4967) on n records, in place: with useAlter the bits / cost (counted against `alter`, e.g. the lookahead's MV the
 * search started from, search.cpp:395-398) are re-based to amvp[mvpIdx]; then the other AMVP candidate takes over if it codes mv in fewer bits. */
typedef struct x265hip_mvp_bits { int16_t amvp[2][2]; int16_t mv[2]; int16_t alter[2]; int16_t mvpIdx; int16_t useAlter; uint32_t bits; uint32_t cost; } x265hip_mvp_bits;   /* 28 bytes */
int x265hip_mvp_bits_batch(void* stream, x265hip_mvp_bits* records, int n, const float* bitsRow /* x265hip_mvbits_row, device */, int bitsHalfRange, uint64_t lambda);

/* x265hip_tme_frame: the PU stage of ThreadedME for every CTU of one picture (Analysis::computeMVForPUs -> Search::puMotionEstimation, analysis.cpp:161-246,
 * search.cpp:226-556), stepped through the schedule: per entry, partition, list and reference one launch sequence over all CTUs -- neighbour records out of the
 * MEData table, getPMV, selectMVP, the search (and the second one from the lookahead's MV), the bit / cost bookkeeping with updateMVP / checkBestMVP, the
 * bidirectional candidate, the MEData record written back into the table (kern_tme.hip).  The table is read and written in the reference's order, so entries that a
 * PU reads before this picture wrote them (the reference reads whatever the FrameData held) are the caller's: pass the table as it was.
 * Host-supplied per PU: the temporal (collocated) neighbour CUData::getNeighbourMV finds in the collocated picture's motion (cudata.cpp:1992-2075) with the two POCs
 * getPMV scales it by.  All planes share stride, origin and planeElems; CTUs cut by the picture edge run the
 * whole schedule (as in the reference), their PUs beyond the edge read the planes' padding. */
typedef struct x265hip_tme_temporal { x265hip_amvp_neighbour nb; int32_t colPOC[2], colRefPOC[2]; } x265hip_tme_temporal;      /* 28 bytes; per (ctu, entry, partition) */
typedef struct x265hip_tme_ref {
    const void* mePlane;                       /* the plane motionEstimate searches (slice->m_mref[l][r].fpelPlane[0]: weighted or not), first element of the padded allocation */
    const void* mePhase;                       /* its 16-slot phase planes (x265hip_subpel_planes)                                              */
    const void* reconPhase;                    /* phase planes of the reconstructed reference picture (selectMVP, bidirectional candidate)      */
    const struct x265hip_inter_choice* refTable;   /* that reference picture's own MEData table, or NULL (intra picture / none): the fallback predictor of search.cpp:313-330 */
    const int16_t* lowresMv;                   /* the lookahead's MVs of this (list, distance), x / y per 16x16 block of the picture (Lowres::lowresMvs), or NULL = not estimated / out of range */
} x265hip_tme_ref;
#define X265HIP_TME_LAUNCH_PER_STAGE 1         /* every stage of every entry as its own launch over all CTUs (the first implementation; kept as the form the chain kernels are checked against) */
#define X265HIP_TME_PROFILE 4                  /* x265hip_tme_picture: time its sections (with a stream synchronisation after each) and print the averages when the producer is destroyed */
#define X265HIP_TME_PACKED_GROUPS 2            /* chain kernels with several PUs per wavefront for the small shapes                              */
typedef struct x265hip_tme_args {
    int isP, numRef[2], curPOC, temporalMvp, refPOC[2][16];
    int searchRange, searchMethod, subpelRefine;
    int picWidth, picHeight, ctuSize, lowresBlocksX;
    const void* curPlane; intptr_t stride; int64_t origin /* element offset of pixel (0,0) in every plane */, planeElems;
    x265hip_tme_ref refs[2][X265HIP_MAX_REF];
    struct x265hip_inter_choice* table;        /* [numCtu][593] MEData records, in / out                                                         */
    const int16_t* areaBest;                   /* [numCtu][5][2][X265HIP_MAX_REF][2]: m_areaBestMV after deriveMVsForCTU's first stage (x265hip_diamond_batch + the median of the collocated MVs) */
    const x265hip_tme_temporal* temporal;      /* [numCtu][nSteps][2]                                                                            */
    int nQp;                                   /* distinct qps of the picture's CUs, 1..64 (Analysis::setLambdaFromQP runs per CU: AQ / cuTree)   */
    const uint8_t* qpIndex;                    /* [numCtu][nSteps]: which of them the CU of an entry uses; NULL with nQp == 1                     */
    const uint16_t* costRows; int costHalfRange;   /* device table [nQp][2 * costHalfRange + 1]: row q = x265hip_mvcost_row(qp q) (X265HIP_ME_ROWS)  */
    uint64_t lambdas[64];                      /* per qp: x265hip_rd_lambda(qp)                                                                  */
    const float* bitsRow; int bitsHalfRange;   /* device x265hip_mvbits_row                                                                      */
    const x265hip_tme_step* steps; int nSteps; /* HOST array (x265hip_tme_schedule)                                                              */
    void* workspace; size_t workspaceBytes;    /* device scratch of x265hip_tme_workspace(numCtu) bytes                                          */
    int refLagPixels;                          /* Search::m_refLagPixels (search.cpp:96): param.sourceHeight with one frame thread, param.searchRange with several; full-pel upper
                                                  bound of both ends of every search window (search.cpp:5017-5018).  0 = picHeight                                                 */
    int flags;                                 /* X265HIP_TME_* below; 0 = the chain kernels                                                     */
    int frameParallel;                         /* != 0: m_bFrameParallel -- selectMVP does not cost a candidate with y >= (searchRange + 1) * 4 (search.cpp:2360-2365)           */
    int ctuFirst, ctuCount;                    /* only the CTUs ctuFirst .. ctuFirst + ctuCount - 1 (a band of whole CTU rows: ThreadedME under frame threads, threadedme.cpp:121-150);
                                                  every per-CTU array keeps the picture's addressing.  ctuCount 0 = the whole picture.  Chain kernels only               */
    int pirStartCol, pirSafeX;                 /* x265hip_tme_picture_desc's (x265hip_ctx.h): the intra-refresh limit of the windows' right edge; 0 = none */
} x265hip_tme_args;
size_t x265hip_tme_workspace(int nCtu);
int x265hip_tme_frame(void* stream, const x265hip_tme_args* args);
void x265hip_tme_release_stream(void* stream);      /* drops the side streams x265hip_tme_frame keeps per caller stream; call before destroying that stream */

/* MotionEstimate::diamondSearch (motion.cpp:631-773) for n PUs of one size: the full-pel predictor search of ThreadedME's first stage
 * (Search::puMotionEstimation with isMVP, search.cpp:355-363 -- the CTU and its four sub-CUs at search range 32; the results seed m_areaBestMV for
 * the PU searches, analysis.cpp:248-306).  Uses of x265hip_me_task: curOff, refOff, mvmin / mvmax (full pel), qmvp (the MVD origin setMVP was
 * given; (0,0) in the reference's use); the other fields are ignored.  results[i]: mv = FULL-pel outMV, cost = the return value, mvcost = the
 * lambda-scaled MVD cost of mv << 2.  The reference's COST_MV_X4 offsets the second loop's points twice (kern_diamond.hip header): positions up to
 * twice the window's half-width from the PU are read, inside the plane's padding as in the reference. */
int x265hip_diamond_batch(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                          const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange, x265hip_me_result* results);

/* ---- several references, two lists: the per-PU choice after the per-reference searches --------------------------------------------------
 * x265hip_inter_merge_batch replaces the tail of Search::puMotionEstimation / predInterSearch for 2Nx2N PUs (search.cpp:258-556): bits and cost of every
 * (list, reference) search -- listSelBits + MVP_IDX_BITS + getTUBits(ref) + BitCost::bitcost(mv - mvp), (satd - mvcost) + RDCost::getCost(bits) --, the best
 * reference of each list, for B slices the bidirectional candidate (predInterLumaPixel of both bests -> pixelavg_pp -> SATD, and the same with both MVs zero),
 * and the final choice, as a record like MEData (encoder/threadedme.h:122-130).  There is no AMVP list here: the MVP of a search is its predictor (the task's
 * qmvp, or mvpSource[list][ref][task.mvpFrom].mv) and checkBestMVP / updateMVP have nothing to choose from.
 * results[list][ref]: the x265hip_me_batch outputs of the same task list searched in that reference.  subpelPlanes[list][ref]: the 16-slot phase-plane buffer
 * of that reference (needed when bidir != 0 and list 1 is not empty); all references share refStride / planeElems and the tasks' refOff. */
typedef struct x265hip_inter_choice {
    int16_t  mv[2][2];           /* per list: quarter-pel MV (0 when the list is unused)                */
    int16_t  mvp[2][2];          /* per list: the predictor the bits were counted against               */
    uint32_t mvCost[2];          /* per list: lambda-scaled MVD cost of the chosen search (MEData.mvCost) */
    int8_t   ref[2];             /* per list: reference index, -1 = list unused (REF_NOT_VALID)          */
    int16_t  reserved;
    int32_t  bits;               /* MEData.bits                                                          */
    uint32_t cost;               /* MEData.cost                                                          */
} x265hip_inter_choice;          /* 36 bytes */
typedef struct x265hip_merge_params {
    int numRef[2];                                   /* references searched per list: 1..16 and 0..16 (0: P slice)          */
    const x265hip_me_result* results[2][X265HIP_MAX_REF];
    const x265hip_me_result* mvpSource[2][X265HIP_MAX_REF];        /* per (list, ref): the array task.mvpFrom indexes, or NULL             */
    const void* subpelPlanes[2][X265HIP_MAX_REF]; int64_t planeElems;
    const float* bitsRow; int bitsHalfRange;         /* device copy of x265hip_mvbits_row: entry [bitsHalfRange + d] = s_bitsizes[|d|] */
    uint64_t lambda;                                 /* x265hip_rd_lambda(qp) = RDCost::m_lambda                            */
    int bidir;                                       /* != 0: evaluate the bidirectional candidate (B slices)               */
    int sourceMaxDim;                                /* max(param->sourceWidth, sourceHeight): range of the zero-MV try     */
} x265hip_merge_params;
int x265hip_mvbits_row(int halfRange, float* out /* 2 * halfRange + 1 */);      /* host: BitCost::CalculateLogs (bitcost.cpp:72-86)        */
uint64_t x265hip_rd_lambda(int qp);                                             /* host: RDCost::setLambda on x265_lambda_tab[qp]          */
int x265hip_inter_merge_batch(void* stream, int w, int h, const void* curPlane, intptr_t curStride, intptr_t refStride,
                              const x265hip_me_task* tasks, int n, const x265hip_merge_params* params, x265hip_inter_choice* out);

/* Pre-interpolate a padded reference plane (or a stack of planes: `rows` counts every row of the allocation) into
 * its 15 quarter-pel phase planes: outPlanes + f*planeElems for f = yFrac*4 + xFrac = 1..15 holds, at the same
 * (stride, row) addressing as refPlane, exactly luma_hpp / luma_vpp / luma_hvpp of the pixel (ipfilter.cpp:79-118,
 * 164-203, 362-369).  Slot f = 0 receives a copy of refPlane, so that x265hip_me_batch can address every candidate --
 * integer or sub-pel -- as "buffer base + 32-bit offset".  With the planes, x265hip_me_batch costs every sub-pel
 * candidate as a plain SAD/SATD at an integer offset -- the device-memory-rich version of the reference lookahead's
 * half-pel planes (lowres.h:104-124).  Values within 4 pixels of the allocation border are
 * computed from clamped coordinates and must not be used (they lie in the picture margins).
 * stride and planeElems must be multiples of 4 pixels, planeElems >= stride*rows, 16 planes of planeElems allocated. */
int x265hip_subpel_planes(void* stream, const void* refPlane, intptr_t stride, int rows, void* outPlanes, int64_t planeElems);
/* The rows rowFirst .. rowEnd - 1 of the same 16 planes: a reference picture that is still being reconstructed becomes valid CTU row by CTU row (frame threads:
 * Frame::m_reconRowFlag, encoder/frameencoder.cpp:1029-1036, framefilter.cpp:676).  A row's vertical taps read the source rows y - 3 .. y + 4 of the whole plane (clamped at 0
 * and rows - 1 exactly as the whole-plane call clamps them), so the last 4 rows of a range whose successor rows are not filled yet mean nothing until a later call covers them
 * again: the caller re-submits them with the next range (x265hip_tme_picture does).  The planes of rowFirst = 0, rowEnd = rows are those of x265hip_subpel_planes. */
int x265hip_subpel_planes_rows(void* stream, const void* refPlane, intptr_t stride, int rows, int rowFirst, int rowEnd, void* outPlanes, int64_t planeElems);

/* one transform unit of the inter residual path; all TUs of one call share log2 size.
 * replaces the chain Predict::predInterLumaPixel (predict.cpp:279-300: copy_pp | luma_hpp | luma_vpp |
 * luma_hvpp by MV fraction) -> cu[].sub_ps -> Quant::transformNxN (quant.cpp:397-480: cu[].dct -> quant)
 * and, when recon is requested, Quant::invtransformNxN (quant.cpp:543-605: dequant_normal -> cu[].idct,
 * DC shortcut :588-597) -> cu[].add_ps -> cu[].sse_pp (search.cpp:5563-5575). */
typedef struct x265hip_tu_task {
    int32_t curOff;              /* TU top-left in the source plane                                   */
    int32_t refOff;              /* co-located pixel in the reference plane                           */
    int16_t mv[2];               /* quarter-pel MV used for motion compensation                       */
    int32_t reconOff;            /* TU top-left in the recon plane (ignored without recon)            */
    int32_t mvFrom;              /* >= 0: mv = mvSource[mvFrom].mv (device-side hand-over from x265hip_me_batch) */
} x265hip_tu_task;               /* 20 bytes */

struct x265hip_inter_choice;
typedef struct x265hip_tq_params {
    int qp;                      /* 0..51: per = qp/6, rem = qp%6 (quant.cpp:465-469,555-568)         */
    int add;                     /* quant rounding numerator: 171 (intra) or 85 (inter), << (qBits-9) */
    const int32_t* quantCoeff;   /* numCoeff-long table or NULL = flat s_quantScales[rem] (scalinglist.cpp:129) */
    int32_t* deltaU;             /* optional: n * N*N int32 (quant_c's deltaU) or NULL                */
    const void* subpelPlanes;    /* optional: the 16-slot buffer x265hip_subpel_planes made from refPlane -> motion     */
    int64_t planeElems;          /*           compensation is a copy out of slot 4*yFrac + xFrac instead of a filter    */
    const struct x265hip_inter_choice* choice;   /* optional (several references): the TU's MV and reference come from choice[task.mvFrom] (x265hip_inter_merge_batch): */
    int choiceList, choiceRef;   /*           only TUs whose PU chose reference choiceRef of list choiceList are processed by this call (one call per reference
                                              plane; the outputs of the other TUs are left alone); uni-directional choices only */
    int chroma;                  /* != 0: the planes are a Cb or Cr plane of a 4:2:0 picture and the TUs chroma TUs (2^log2TrSize chroma samples): motion
                                    compensation is Predict::predInterChromaPixel (predict.cpp:340-380: the quarter-pel luma MV read as an eighth-pel chroma MV,
                                    4-tap filters); qp must be the CHROMA qp of the plane (the caller maps it, as Quant::setChromaQP does); subpelPlanes unused */
    const void* refPlane1;       /* != NULL: a BI-DIRECTIONAL launch (needs choice): refPlane is reference choiceRef of list 0, refPlane1 reference choiceRef1 of
                                    list 1 (same stride, the tasks' refOff); only TUs whose PU chose exactly that pair are processed; motion compensation is
                                    the B-slice branch of Predict::motionCompensation (predict.cpp:186-211): predInterLumaShort of both (14-bit) -> addAvg */
    int choiceRef1;
    int dst4;                    /* != 0 with log2TrSize 2: the 4x4 transform pair is DST-VII (dst4x4 / idst4x4, dct.cpp:43-81, 443-456, 528-541) and the DC-only
                                    shortcut of the inverse is off -- Quant's choice for intra luma 4x4 TUs (quant.cpp:429-432, 585-603).  With refPlane = a plane
                                    holding the caller's intra predictions, MVs zero and add = 171 the chain is the intra TU's */
} x265hip_tq_params;

int x265hip_tq_batch(void* stream, int log2TrSize,
                     const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                     const x265hip_tu_task* tasks, int n, const x265hip_tq_params* params,
                     int16_t* coeff /* n x N*N dense */, uint32_t* numSig /* n */,
                     void* reconPlane /* NULL = forward path only */, intptr_t reconStride, uint64_t* sse /* n, with recon */,
                     const x265hip_me_result* mvSource /* may be NULL */);

/* ---- SEA search (motion.cpp:1438-1591) ---------------------------------------------------------------------------------
 * x265hip_sea_integral_planes builds the 12 integral planes FrameFilter::processPostRow builds per reconstructed picture
 * (encoder/framefilter.cpp:38-139, 740-833; order 32x32, 32x24, 32x8, 24x32, 16x16, 16x12, 16x4, 12x16, 8x32, 8x8, 4x16, 4x4):
 * plane k, at every position whose W x H box lies inside the padded picture, holds the sum of the box whose top-left pixel is that
 * position.  picPadded = FIRST element of the padded picture (row -marginY, column -marginX), rows = all its rows; plane k starts at
 * planes + k * planeElems with the same layout, so the planes take the tasks' refOff unchanged.  workspace: device scratch of
 * x265hip_sea_integral_workspace(stride, rows) bytes.
 * x265hip_me_batch_sea = x265hip_me_batch with method SEA.  The PU shapes 8x4, 4x8, 8x32 and 32x8 are refused (X265HIP_EARG): for them
 * the reference's result depends on stale contents of MotionEstimate::fencPUYuv (its DC terms read outside the PU, :1467-1468). */
size_t x265hip_sea_integral_workspace(intptr_t stride, int rows);
int x265hip_sea_integral_planes(void* stream, const void* picPadded, intptr_t stride, int rows, uint32_t* planes, int64_t planeElems,
                                void* workspace, size_t workspaceBytes);
int x265hip_me_batch_sea(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                         const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
                         int merange, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource,
                         const void* subpelPlanes, int64_t planeElems, const uint32_t* integralPlanes, int64_t integralPlaneElems);

/* ---- lookahead frame costs on half-resolution pictures ------------------------------------------------
 * The lowres buffer holds F pictures x 4 planes (full-pel, H, V, HV half-pel: what x265hip_frame_init_lowres +
 * x265hip_extend_pic_border produce, the layout of Lowres::lowresPlane[0..3], lowres.cpp:381-391), each plane planeElems
 * long with pixel (0,0) at element `origin`; plane k of picture f starts at ((f * 4 + k) * planeElems).  Blocks are the
 * lookahead's 8x8 (X265_LOWRES_CU_SIZE); widthInCU x heightInCU of them per picture; margins >= 8 + 8 + 1 pixels.
 *
 * x265hip_lookahead_intra_batch  replaces LookaheadTLD::lowresIntraEstimate (slicetype.cpp:755-864) for nFrames pictures:
 *   intraCost / intraMode / lowresCosts are nFrames x ncu, rowSatds nFrames x heightInCU, sums nFrames x { costEst, costEstAq }.
 * x265hip_lookahead_cost_batch   replaces CostEstimateGroup::estimateFrameCost (slicetype.cpp:4365-4463, serial branch; no
 *   HME, no slices; weighted list-0 reference through x265hip_la_task.weighted0) with estimateCUCost (:4467-4640) for nTasks (p0, b, p1) choices at once.
 *   rowsPerSlice: 0 = the serial sweep of estimateFrameCost; > 0 = Lookahead::m_numRowsPerSlice of the cooperative --lookahead-slices branch
 *   (slicetype.cpp:1173-1176, 4347-4357, 4394-4426): each slice of that many block rows (the last one with the remainder) is swept on its own.
 *   invQscale: nFrames x ncu 8.8 fixed-point AQ factors (Lowres::invQscaleFactor / invQscaleFactor8x8) or NULL.
 *   costRow: the row of x265hip_mvcost_row(x265hip_lookahead_qp(), ...), costHalfRange >= 4 * (8 * max(wcu, hcu) + 32).
 *   The call is fully asynchronous (no copy back, no synchronisation: it may be captured into a hipGraph); tasks with p0 == b
 *   or pictures not among the nFrames pictures of the buffer are skipped on the device (their outputs keep their previous contents).
 *   mvs (int16 x, y per block) and mvCosts are arrays of ncu-long SLOTS, the device form of Lowres::lowresMvs[list][dist] /
 *   lowresMvCosts[list][dist]: a task searches into its slot when doSearch[list] != 0 and reads it otherwise (the
 *   reference's bDoSearch caching, :4376-4377).  Two tasks of one call must not search the same slot.
 *   Outputs per outSlot: lowresCosts (ncu, cost | listused << 14), rowSatds (heightInCU), sums { costEst before the
 *   B-frame normalisation (:4456-4457), costEstAq, intraMbs }. */
typedef struct x265hip_la_task {
    int32_t b, p0, p1;           /* picture indices in the lowres buffer (places, not display order); p1 == b: P estimate */
    int32_t doSearch[2];         /* per list: run the motion search (else the slot already holds its result)          */
    int32_t mvSlot[2];           /* per list: slot of mvs / mvCosts (mvSlot[1] unused for a P estimate)               */
    int32_t outSlot;             /* slot of lowresCosts / rowSatds / sums                                             */
    int32_t weighted0;           /* 0, or 1 + index of the picture list 0 is SEARCHED in: the weighted copy of p0 that
                                    LookaheadTLD::weightsAnalyse made (slicetype.cpp:919-1020, wfref0 :4474; x265hip blockop
                                    weight_pp on the four planes); the bidirectional average always uses p0 itself      */
} x265hip_la_task;               /* 36 bytes */

int x265hip_lookahead_qp(void);  /* X265_LOOKAHEAD_QP of this library's bit depth (common.h:223) */
int x265hip_lookahead_intra_batch(void* stream, const void* lowres, int64_t planeElems, intptr_t stride, int64_t origin, int widthInCU, int heightInCU,
                                  int nFrames, const int32_t* invQscale, int32_t* intraCost, uint8_t* intraMode, uint16_t* lowresCosts,
                                  int32_t* rowSatds, int64_t* sums);
int x265hip_lookahead_cost_batch(void* stream, const void* lowres, int64_t planeElems, intptr_t stride, int64_t origin, int widthInCU, int heightInCU,
                                 const x265hip_la_task* tasks, int nTasks, int nFrames /* pictures in the lowres buffer */, const int32_t* intraCost, const int32_t* invQscale,
                                 const uint16_t* costRow, int costHalfRange, int rowsPerSlice, int16_t* mvs, int32_t* mvCosts,
                                 uint16_t* lowresCosts, int32_t* rowSatds, int64_t* sums);

/* --hme (param->bEnableHME, slicetype.cpp:4430-4437, 4483-4575): the sweep runs first on the QUARTER-resolution pictures (Lowres::lowerResPlane[0..3], four planes per
 * picture at the same picture indices as the lowres buffer; blocks of 8x8 on the Lookahead::m_4x4Width x m_4x4Height grid) with hmeRange[0] / hmeSearchMethod[0] into its own
 * MV / cost slots (mvs / mvCosts below: the device form of Lowres::lowerResMvs / lowerResMvCosts, same slot numbers as the call's mvSlot), then on the half-resolution pictures
 * with hmeRange[1] / hmeSearchMethod[1], where twice the quarter-resolution MV of the block above a block joins its predictor candidates.  Methods: X265HIP_ME_DIA, _HEX, _UMH, _STAR
 * or _FULL (the reference's default is hex, umh; an exhaustive level is cut to +-range around the zero vector, motion.cpp:1598-1605; a star level's raster covers the picture
 * and costs one placement in four at the doubled vector, :1392 -- costHalfRange >= 12 * the larger picture side + 192 then; a sea level does not exist: the reference's
 * lookahead has no integral planes and dies there).  The serial sweep only (rowsPerSlice 0): the reference's cooperative slices read each other's quarter-resolution results
 * while they are being written (slicetype.cpp:4332-4358, 4532). */
typedef struct x265hip_la_hme {
    const void* lowerRes; int64_t planeElems; intptr_t stride; int64_t origin; int widthInCU, heightInCU;
    int method[2], range[2];
    int16_t* mvs; int32_t* mvCosts;            /* slots of widthInCU * heightInCU entries */
} x265hip_la_hme;
int x265hip_lookahead_cost_batch_hme(void* stream, const void* lowres, int64_t planeElems, intptr_t stride, int64_t origin, int widthInCU, int heightInCU,
                                     const x265hip_la_task* tasks, int nTasks, int nFrames, const int32_t* intraCost, const int32_t* invQscale,
                                     const uint16_t* costRow, int costHalfRange, int rowsPerSlice, int16_t* mvs, int32_t* mvCosts,
                                     uint16_t* lowresCosts, int32_t* rowSatds, int64_t* sums, const x265hip_la_hme* hme /* NULL: no HME */);

/* cuTree: one propagation step (Lookahead::estimateCUPropagate, slicetype.cpp:3850-3953, with primitives.propagateCost, pixel.cpp:906-931)
 * on the arrays the calls above left in HBM: intraCost / invQscale of picture b, the lowresCosts and the two MV slots of its estimate,
 * and the three pictures' propagateCost arrays (uint16, ncu each; prop1 may be NULL for a P picture).  distP0 = b - p0, distP1 = p1 - b;
 * fpsFactor = CLIP_DURATION(frame duration) / CLIP_DURATION(average duration) (:3863).  Double arithmetic as in the reference; the
 * saturating adds are order-independent (non-negative addends).  workspace: 16 * ncu bytes of device scratch. */
int x265hip_cutree_propagate(void* stream, int widthInCU, int heightInCU, int distP0, int distP1, int weightedBiPred, double fpsFactor, int referenced,
                             const int32_t* intraCost, const uint16_t* lowresCosts, const int32_t* invQscale, const int16_t* mvs0, const int16_t* mvs1,
                             uint16_t* propB, uint16_t* prop0, uint16_t* prop1, void* workspace, size_t workspaceBytes);

/* cuTree: the last step for a picture (Lookahead::cuTreeFinish, slicetype.cpp:4098-4150; default configuration: no --hevc-aq, qg-size above 8): the qp offset of every
 * block of the half-resolution picture from its accumulated propagateCost: qpCuTreeOffset = qpAqOffset - strength * (log2(intra + propagate) - log2(intra) + weightdelta),
 * intra = (intraCost * invQscale + 128) >> 8, propagate = (propagateCost * fpsFactor + 128) >> 8, fpsFactor = (int)(CLIP_DURATION(averageDuration) / CLIP_DURATION(frame
 * duration) * 256), weightdelta = 1 - weightedCostDelta when ref0Distance != 0 and weightedCostDelta > 0, strength = Lookahead::m_cuTreeStrength = 5 * (1 - qcomp).
 * Blocks without intra cost keep their qpCuTreeOffset.  Doubles; the two log2 come from the device math library (tolerance 1e-12 against the host's). */
int x265hip_cutree_finish(void* stream, int ncu, const int32_t* intraCost, const int32_t* invQscale, const uint16_t* propagateCost, const double* qpAqOffset,
                          int fpsFactor, double weightedCostDelta, int ref0Distance, double cuTreeStrength, double* qpCuTreeOffset);

/* SAO statistics of a whole deblocked picture: SAO::calcSaoStatsCTU (encoder/sao.cpp:729-905) for one plane of every CTU, one slice,
 * bLimitSAO off.  fenc / recon point at pixel (0,0) of the source and the reconstructed plane (same stride).  Luma: planeOffset 0; a 4:2:0 chroma
 * plane: its own width / height / CTU size (picture and CTU sizes halved, sao.cpp:748-756) and planeOffset 2 (:773).  out: per CTU (raster order)
 * [2][5][32] int32 = m_offsetOrg then m_count, types in the order SAO_EO_0..3, SAO_BO (sao.h:43-50).  nonDeblocked = param bSaoNonDeblocked (0 / 1);
 * nonDeblocked = 2: the statistics SAO::calcSaoStatsCu_BeforeDblk collects instead (sao.cpp:908-1207) -- the bottom / right border of every CTU that the deblocked
 * statistics leave out under sao-non-deblock, with `recon` = the picture BEFORE deblocking (m_offsetOrgPreDblk / m_countPreDblk of each CTU, same layout). */
int x265hip_sao_stats_frame(void* stream, const void* fenc, const void* recon, intptr_t stride, int picWidth, int picHeight, int ctuSize, int nonDeblocked,
                            int planeOffset, int32_t* out);
/* the same with --slices: sliceFirstRow (device, ceil(picHeight / ctuSize) + 1 bytes, the last one 0) is non-zero for every CTU row that begins a slice -- its CTUs have no
 * row above (m_bFirstRowInSlice, sao.cpp:744-746), and the CTUs of the row before it count down to their bottom line like the picture's last row (m_bLastRowInSlice,
 * :763-766).  NULL = x265hip_sao_stats_frame. */
int x265hip_sao_stats_frame_slices(void* stream, const void* fenc, const void* recon, intptr_t stride, int picWidth, int picHeight, int ctuSize, int nonDeblocked,
                                   int planeOffset, int32_t* out, const uint8_t* sliceFirstRow);
/* the CTUs of the rows [ctuRow0, ctuRow1) only -- their entries of `out`, the others are not touched.  A band of FrameFilter::processRow's pipeline (framefilter.cpp:490-500:
 * rdoSaoUnitCu of row r runs before row r + 1 is deblocked): the rows below need not be deblocked yet, a CTU's statistics leave out the lines the next row's deblocking changes
 * (skipB) and read one line beyond them, which it does not change. */
int x265hip_sao_stats_rows(void* stream, const void* fenc, const void* recon, intptr_t stride, int picWidth, int picHeight, int ctuWidth, int ctuHeight, int nonDeblocked,
                           int planeOffset, int32_t* out, const uint8_t* sliceFirstRow, int ctuRow0, int ctuRow1);      /* ctuWidth x ctuHeight: the CTU in this plane (4:2:2 chroma: w = h / 2) */

/* SAO of a whole luma plane, OUT OF PLACE (in != out): SAO::generateLumaOffsets + applyPixelOffsets (encoder/sao.cpp:268-623) for every CTU.  The
 * reference filters in place and classifies against saved unmodified neighbours (m_tmpU, m_tmpL); reading the input plane is the same thing.
 * params (device): per CTU in raster order 6 int32 = typeIdx (-1 = off, 0..3 = SAO_EO_0..3, 4 = SAO_BO), bandPos, offset[4], with SAO_MERGE_LEFT / UP
 * already resolved to the merged CTU's values.  One slice (the picture's first / last rows are the slice's).
 * Chroma (SAO::generateChromaOffsets, sao.cpp:626-730): call it on the Cb / Cr plane with the plane's width, height and CTU size (ctuSize / 2 for 4:2:0, so 8..32);
 * the two chroma planes share one typeIdx per CTU (the reference applies Cr with Cb's type, sao.cpp:721) -- pass Cb's typeIdx in Cr's records. */
int x265hip_sao_apply_frame(void* stream, const void* in, void* out, intptr_t stride, int picWidth, int picHeight, int ctuSize, const int32_t* params);

/* PSNR numerator of one plane: Encoder::computeSSD (encoder/encoder.cpp:1203-1270), the exact 64-bit sum of squared differences of the source and
 * the reconstructed plane (width <= 16384).  *out is a device uint64. */
int x265hip_plane_ssd(void* stream, const void* fenc, const void* recon, intptr_t stride, int width, int height, uint64_t* out);

/* SSIM of the reconstructed picture against the source as the frame filter accumulates it (FrameFilter::processPostRow, encoder/framefilter.cpp:704-722 ->
 * calculateSSIM :839-865 -> ssim_4x4x2_core / ssim_end_4, common/pixel.cpp:623-693): rowSsim[r] / rowCnt[r] = what CTU row r adds to
 * FrameEncoder::m_ssim / m_ssimCnt (float sum of its windows in the reference's order of additions, number of windows); frame[0] = the double sum of the
 * rows in row order, frame[1] = the window count: the picture's SSIM is frame[0] / frame[1] (Encoder::finishFrameStats, encoder.cpp:3193-3198).  Float
 * arithmetic, but the order of operations is the reference's, so the result is IDENTICAL to the single-threaded reference (the tests use tolerance 0).
 * workspace: x265hip_ssim_workspace(width, height) bytes of device memory.  width, height >= 10, width <= 16384. */
size_t x265hip_ssim_workspace(int width, int height);
int x265hip_ssim_frame(void* stream, const void* recon, intptr_t stride1, const void* fenc, intptr_t stride2, int width, int height, int ctuSize,
                       void* workspace, float* rowSsim, uint32_t* rowCnt, double* frame);

/* ---------------------------------------------------------------------------------------------------------------------------------------------
 * Deblocking of a reconstructed 4:2:0 picture, in place in HBM: replaces the Deblock::deblockCTU(ctu, geom, EDGE_VER / EDGE_HOR) calls of the frame
 * filter (FrameFilter::ParallelFilter::processTasks, encoder/framefilter.cpp:383-443 -> common/deblock.cpp:37-497 -> primitives.pelFilterLumaStrong /
 * pelFilterChroma, common/loopfilter.cpp:136-232).  The picture is described by CUData's own per-partition arrays, copied to the device as they are:
 * CTU after CTU (raster order of CTUs), numPartitions = (ctuSize / 4)^2 entries per CTU in z-scan order -- m_log2CUSize, m_partSize, m_tuDepth,
 * m_predMode (0 = not coded / outside the picture), m_cbf[0], m_tqBypass (may be NULL when !tqBypassEnabled), m_qp, m_refIdx[0], m_refIdx[1] (B slices),
 * m_mv[0], m_mv[1] (int32 x, y pairs).  refPic[list][refIdx] = any integer identifying the picture behind slice->m_refFrameList[list][refIdx] (the reference
 * compares Frame pointers; POCs do).  betaOffsetDiv2 / tcOffsetDiv2 / cb / crQpOffset / tqBypassEnabled are the PPS fields.  width and height are
 * multiples of 8 (the minimum CU size); slices are whole CTU rows (sliceFirstRow), no tiles.
 * bsOut (optional, device, 2 * (height/4) * (width/4) bytes): the boundary strength of every examined edge segment, [dir][unitY][unitX].
 * --------------------------------------------------------------------------------------------------------------------------------------------- */
typedef struct x265hip_deblock_pic
{
    int width, height, ctuSize, sliceIsP, betaOffsetDiv2, tcOffsetDiv2, cbQpOffset, crQpOffset, tqBypassEnabled;
    const uint8_t *log2CUSize, *partSize, *tuDepth, *predMode, *cbfLuma, *tqBypass;
    const int8_t *qp, *refIdx0, *refIdx1;
    const int32_t *mv0, *mv1;
    int32_t refPic[2][16];
    const uint8_t* sliceFirstRow;   /* --slices: per CTU row, non-zero where the row begins a slice (CUData::m_bFirstRowInSlice: the CTU above is no neighbour, cudata.cpp:323 -- the
                                       row's top edge is not filtered); NULL = one slice.  ceil(height / ctuSize) + 1 entries, the last one 0.  Like the other arrays: device memory for
                                       x265hip_deblock_frame / _pictures, host memory inside x265hip_ff_picture_desc */
    int chromaFormat;               /* X265_CSP_*: 0 or 1 = 4:2:0, 2 = 4:2:2, 3 = 4:4:4 -- the subsampling of Cb / Cr (their size in samples, the grid their edges lie on, the QP rule:
                                       deblock.cpp:104-113, 417-497) */
} x265hip_deblock_pic;
int x265hip_deblock_frame(void* stream, const x265hip_deblock_pic* desc, void* Y, intptr_t strideY, void* Cb, void* Cr, intptr_t strideC, uint8_t* bsOut);
/* A band of CTU rows [ctuRow0, ctuRow1) (FrameFilter::processRow's order, encoder/framefilter.cpp:576-676): the edges of those rows' CTUs, the band's top edge included -- it changes
 * the last 3 luma / 1 chroma lines of the row above, which must hold that row's own deblocking.  Bands in increasing order give x265hip_deblock_frame's picture.  bsOut: the band's
 * unit rows only, not cleared. */
int x265hip_deblock_rows(void* stream, const x265hip_deblock_pic* desc, void* Y, intptr_t strideY, void* Cb, void* Cr, intptr_t strideC, uint8_t* bsOut, int ctuRow0, int ctuRow1);

/* ---- the same stages for a BATCH of pictures of one size in one launch per stage (a 1080p plane does not fill 256 CUs: per-picture launches of 7-20 us
 * on <= 17 workgroups are launch-latency bound; a batch is not).  Pictures are `pictureElems` elements apart in their buffers; per-picture outputs are
 * consecutive (statistics: nCtu x 320 int32 per picture; SAO parameters: nCtu x 6 int32; SSD: one uint64; SSIM: rowSsim / rowCnt numRows each, frame 2 doubles,
 * workspace x265hip_ssim_workspace bytes each).  Deblocking takes one description + plane pointers per picture: the job list both as a host copy (checked,
 * sizes the launch) and as its device copy (read by the kernels). */
typedef struct x265hip_deblock_job { x265hip_deblock_pic pic; void *Y, *Cb, *Cr; uint8_t* bsOut; } x265hip_deblock_job;
int x265hip_deblock_pictures(void* stream, const x265hip_deblock_job* jobsDevice, const x265hip_deblock_job* jobsHost, int nPictures, intptr_t strideY, intptr_t strideC);
int x265hip_sao_stats_pictures(void* stream, const void* fenc, const void* recon, intptr_t stride, int picWidth, int picHeight, int ctuSize, int nonDeblocked,
                               int planeOffset, int32_t* out, int nPictures, int64_t pictureElems);
int x265hip_sao_apply_pictures(void* stream, const void* in, void* out, intptr_t stride, int picWidth, int picHeight, int ctuSize, const int32_t* params,
                               int nPictures, int64_t pictureElems);
int x265hip_plane_ssd_pictures(void* stream, const void* fenc, const void* recon, intptr_t stride, int width, int height, uint64_t* out, int nPictures,
                               int64_t fencPictureElems, int64_t reconPictureElems);
int x265hip_ssim_pictures(void* stream, const void* recon, intptr_t stride1, const void* fenc, intptr_t stride2, int width, int height, int ctuSize,
                          void* workspace, float* rowSsim, uint32_t* rowCnt, double* frame, int nPictures, int64_t reconPictureElems, int64_t fencPictureElems);

#ifdef __cplusplus
}
#endif
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#endif
