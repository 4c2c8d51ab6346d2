/*
 * x265hip_ctx.h -- the host side of the frame-batched path for a C / C++ caller (an encoder): device context, resident planes and the
 * CTU-pyramid batch "motion search -> motion compensation / DCT / quant" of x265hip_frame.h, without any Python or torch in the process.
 *
 * This is the `x265hip_ctx_create / upload_plane / me_batch / tq_batch / sync` surface of SURVEY 8(b)(3).  What it replaces on the encoder
 * side: the per-CTU producer of the decoupled-ME seam (ThreadedME::findJob -> Analysis::deriveMVsForCTU, encoder/threadedme.cpp:207-261,
 * analysis.cpp:248-306) and the inter-residual chain of Search::estimateResidualQT (search.cpp:5515-5636) for whole frames:
 *
 *   x265hip_ctx_create(device)                    one device, one stream
 *   x265hip_batch_create(ctx, desc)               planes, phase planes, pyramid task lists, result / coefficient arrays for F frame pairs
 *   x265hip_batch_upload_plane(...)               source / reference picture -> padded device plane (borders replicated on the device:
 *                                                 extendPicBorder, common/pixel.cpp:1044-1058)
 *   x265hip_batch_step(batch)                     phase planes -> ME 64 / 32 / 16 / 8 (each level seeded by its parent CU's MV the way
 *                                                 computeMVForPUs seeds PUs from m_areaBestMV) -> TQ; asynchronous on the context's stream
 *   x265hip_batch_read_results / read_coeffs      MVs + costs per pyramid level (the flat MEData array of threadedme.h:122-130), coefficients
 *
 * All functions return X265HIP_OK or a negative X265HIP_E* code (x265hip_last_error() has the text).  Nothing here has a CPU fallback.
 */
#ifndef X265HIP_CTX_H
#define X265HIP_CTX_H
#include "x265hip_frame.h"

/* the library is built with -fvisibility=hidden: what these headers declare is its whole exported surface */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif
#ifdef __cplusplus
extern "C" {
#endif

typedef struct x265hip_ctx x265hip_ctx;
typedef struct x265hip_batch x265hip_batch;

int   x265hip_ctx_create(int device, x265hip_ctx** ctx);
void  x265hip_ctx_destroy(x265hip_ctx* ctx);
size_t x265hip_ctx_trim(x265hip_ctx* ctx);           /* a context keeps the large device blocks (>= 64 MiB, up to 96 GiB in all) of its destroyed batches for its next ones (a host that creates and
                                                         destroys 8K batches does not unmap and map multi-GB ranges each time); this gives them back to the driver now.  Returns the bytes released */
void* x265hip_ctx_stream(x265hip_ctx* ctx);          /* hipStream_t: every call on this context is ordered on it (a batch with desc.streams > 1: after x265hip_batch_join) */
int   x265hip_ctx_sync(x265hip_ctx* ctx);              /* waits for everything queued through the context, the sub-streams of its batches included */
int   x265hip_ctx_device(const x265hip_ctx* ctx);    /* the device the context was created on; every entry point that takes a context selects it for the calling thread */

#define X265HIP_MAX_PIC_DIM 8184                 /* (dim + 8 - 1) << 2 must fit the int16 quarter-pel limits of the task records */

typedef struct x265hip_batch_desc
{
    int width, height;      /* luma picture size, multiples of the CTU size 64 (pad the picture as the encoder does, picyuv.cpp:91-111) */
    int frames;             /* (source, reference) picture pairs processed per step                                                    */
    int margin;             /* padding on every side of a plane; >= 64 + 16 + 8 (PicYuv uses maxCUSize + 32 = 96)                         */
    int qp;                 /* lambda of the MVD cost row (BitCost::setQP) and quantiser of the TQ stage                               */
    int merange, method, subme;   /* param->searchRange, X265_*_SEARCH (x265hip_me_method), param->subpelRefine                        */
    int tuLog2;             /* transform size of the TQ stage: 2..5                                                                    */
    int recon;              /* != 0: also dequant -> IDCT -> reconstruction + SSE (S4)                                                 */
    int usePlanes;          /* != 0: quarter-pel phase planes of the reference stack (x265hip_subpel_planes)                           */
    int refs;               /* list-0 reference pictures searched per source picture: 1..X265HIP_MAX_REF (0 = 1; param->maxNumReferences: medium 3, slow 4,
                               slower 5 -- param.cpp:567-608).  Every reference is searched down the pyramid with its own predictor chain (m_areaBestMV[area][list][ref],
                               analysis.cpp:248-306), x265hip_inter_merge_batch chooses per PU, the TQ stage compensates each TU from the chosen reference.  > 1 needs usePlanes */
    int rect;               /* != 0: also the 2NxN and Nx2N PUs of every CU (param->bEnableRectInter: preset slow and up): 425 PUs per CTU instead of 85, each seeded
                               by its own CU's 2Nx2N result in the same reference                                                       */
    int streams;            /* 1..8 (0 = 1): the batch is cut into this many sub-batches of whole pictures, each stepped on its own stream (pictures are independent,
                               the levels of one picture are not); x265hip_batch_step stays ordered behind the context's stream.
                               2: the two streams ALTERNATE on the 64x64 level (its workgroups fill a CU's LDS: it then always runs beside the other stream's smaller
                               levels, never beside itself) and are not joined between steps -- a stream's next pass follows its own previous one; x265hip_ctx_sync and
                               the upload / read calls join them                                                                            */
    int bandRows;           /* must be 0 (kept for the record's layout): the band-major schedule of round 3 was a measured loss (profiles/r03_band_major_ab.txt) and left the
                               library in round 5; x265hip_batch_create refuses any other value                                                      */
    int amp;                /* != 0: also the asymmetric PUs (2NxnU, 2NxnD, nLx2N, nRx2N) of every CU of 64, 32 and 16 pixels (param->bEnableAMP: preset slower and up,
                               param.cpp:592-593; searched at analysis.cpp:2756-2860): 168 more PUs per CTU (593 with rect = the MEData entries of a CTU, threadedme.h:67-92),
                               each seeded by its own CU's 2Nx2N result in the same reference                                             */
    int refs1;              /* list-1 reference pictures: 0 = a P picture.  > 0 = a B picture: every list-1 reference is searched down the pyramid like list 0's, the per-PU
                               choice is among both lists and -- for the PUs of split CUs above 8x8, as in the reference (search.cpp:420-503; 2Nx2N bi-prediction belongs to
                               the mode decision, 8x8 CUs are bi-prediction restricted) -- the bidirectional candidate; the TQ stage compensates every TU from the list and
                               reference its 2Nx2N PU chose.  Needs usePlanes                                                              */
} x265hip_batch_desc;

/* Pure host code (no GPU needed): the task lists x265hip_batch_create uploads.  level = 64, 32, 16 or 8.  Task k of a level is PU
 * (frame, row, column) in raster order; mvpFrom of a level below 64 = index of the parent CU's task in the level above; limits are
 * CUData::clipMv's (cudata.cpp:2094-2107) with the search window derived on the device (X265HIP_ME_WINDOW). */
int   x265hip_batch_task_count(const x265hip_batch_desc* desc, int level);
int   x265hip_batch_build_me_tasks(const x265hip_batch_desc* desc, int level, x265hip_me_task* out);
/* the rectangular partitions w x h (2NxN: w = 2h = CU size; Nx2N: h = 2w): task k = PU (frame, row, column) in raster order of the w x h grid, mvpFrom = its CU's 2Nx2N task */
int   x265hip_batch_rect_task_count(const x265hip_batch_desc* desc, int w, int h);
int   x265hip_batch_build_rect_tasks(const x265hip_batch_desc* desc, int w, int h, x265hip_me_task* out);
/* the asymmetric partitions (desc.amp) w x h with max(w, h) = the CU size (64, 32, 16) and min(w, h) a quarter or three quarters of it.  A shape occurs twice in every CU (as the
 * first PU of one mode and as the second of its sibling); task k = PU (frame, CU row, occurrence 0 / 1, CU column), mvpFrom = its CU's 2Nx2N task */
int   x265hip_batch_amp_task_count(const x265hip_batch_desc* desc, int w, int h);
int   x265hip_batch_build_amp_tasks(const x265hip_batch_desc* desc, int w, int h, x265hip_me_task* out);
int   x265hip_batch_tu_count(const x265hip_batch_desc* desc);
int   x265hip_batch_build_tu_tasks(const x265hip_batch_desc* desc, x265hip_tu_task* out);

int   x265hip_batch_create(x265hip_ctx* ctx, const x265hip_batch_desc* desc, x265hip_batch** batch);
void  x265hip_batch_destroy(x265hip_batch* batch);
/* which: 0 = source, 1 + r = list-0 reference r, 1 + refs + r = list-1 reference r (desc.refs1).  pixels: the unpadded width x height picture in HOST memory (pixel = uint8_t / uint16_t by library),
 * strideElems its row pitch.  The padded plane is assembled on the device. */
int   x265hip_batch_upload_plane(x265hip_batch* batch, int which, int frame, const void* pixels, intptr_t strideElems);
int   x265hip_batch_step(x265hip_batch* batch);
int   x265hip_batch_step_one_stream(x265hip_batch* batch);      /* the same pass as ONE sub-batch on the context's stream, whatever desc.streams says (same results): its stages
                                                                 * run one after the other, which per-stage timing needs */
int   x265hip_batch_read_results(x265hip_batch* batch, int level, x265hip_me_result* out /* x265hip_batch_task_count entries */);       /* reference 0, 2Nx2N */
int   x265hip_batch_read_results_ref(x265hip_batch* batch, int w, int h, int ref, x265hip_me_result* out);      /* any searched shape (square or, with rect, 2NxN / Nx2N), any reference */
int   x265hip_batch_read_results_list(x265hip_batch* batch, int w, int h, int list, int ref, x265hip_me_result* out);      /* any searched shape (with amp the asymmetric ones too), either list */
int   x265hip_batch_read_choices(x265hip_batch* batch, int w, int h, struct x265hip_inter_choice* out);         /* refs > 1 or refs1 > 0: the per-PU choice among the references / lists */
/* per-stage timing of sub-batch 0 (HIP events on the stream the stage runs on): while set_timing is 1 every step records its own event set (the last 64 are kept);
 * read_timing synchronises the batch's streams, returns the number of steps averaged (> 0) and the mean milliseconds of every stage over them, in
 * x265hip_batch_stage_name order ("planes", "me64", ["rect64",] ["amp64",] ..., "tq"), and starts a new record */
int   x265hip_batch_set_timing(x265hip_batch* batch, int on);
/* How a step launches its kernels -- every combination gives the same bytes (tests/test_host_batch_gpu.py compares them); the default, 0, is the fastest measured form.
 *   X265HIP_BATCH_START64_LAUNCH: the 64x64 level of a STAR search WITH its start-stage launch (by default the batch's own top-level tasks -- zero predictor, no candidates --
 *     are started inside the full-pel kernel, csrc/star64_body.inc: two launches instead of three).
 *   (Flags 1, 2 and 8 -- the 16x16 / 8x8 [/ 32x32] levels fused into one launch, tiled phase planes -- were measured losses, profiles/r03_fused_ab.txt and r03_tiled_ab.txt,
 *   and are refused since round 5.) */
#define X265HIP_BATCH_START64_LAUNCH 4
#define X265HIP_BATCH_PLANE_GROUPS_OF_2 16      /* the phase planes in groups of two pictures (what a batch does by itself when its 16-slot plane buffer outgrows the kernels' 32-bit
                                                   byte offsets -- 8K 10 bit beyond two pictures: each group keeps its 16 slots back to back, the kernels get a pointer biased by the
                                                   group's first picture); same bytes: for tests of the grouping at small sizes.  x265hip_batch_device_ptr(.., 2 / 200 + r / 400 + r) hands
                                                   out the allocation, whose layout is then per group */
int   x265hip_batch_set_mode(x265hip_batch* batch, int flags);
int   x265hip_batch_set_fused(x265hip_batch* batch, int flags);      /* the same call under its first name */
int   x265hip_batch_stage_count(const x265hip_batch* batch);
const char* x265hip_batch_stage_name(const x265hip_batch* batch, int i);
int   x265hip_batch_read_timing(x265hip_batch* batch, float* msPerStage);
/* the longest single kernel of a pass on its own -- star64_kernel of the first reference (STAR search with the phase planes; batches stepped on ONE stream) -- bracketed by its
 * own events while set_timing is 1: mean milliseconds over the timed steps since the last call; returns the number of steps averaged, 0 when no such launch was timed */
int   x265hip_batch_read_kernel_timing(x265hip_batch* batch, float* msStar64Kernel);
int   x265hip_batch_read_plane(x265hip_batch* batch, int which, int frame, void* out /* the padded plane as it sits on the device: (width + 2 margin) x (height + 2 margin) pixels */);
int   x265hip_batch_read_coeffs(x265hip_batch* batch, int16_t* coeff /* tu_count << (2 * tuLog2) */, uint32_t* numSig /* tu_count */);
/* device pointers for consumers that stay on the GPU: what = 0 source planes, 1 reference planes, 2 phase planes, 3 coefficients, 4 numSig,
 * 5 reconstruction, 10 + log2(level) - 3 = results of a pyramid level (10: 8x8 ... 13: 64x64; reference 0), 100 + r = plane stack of list-0 reference r, 200 + r = its phase planes,
 * 300 + r / 400 + r = the same for list-1 reference r */
void* x265hip_batch_device_ptr(x265hip_batch* batch, int what);
/* desc.streams = 2 leaves the second sub-batch on its own stream when x265hip_batch_step returns (the read_*, upload and x265hip_ctx_sync calls join it).  A consumer that
 * stays on the GPU and queues its own work on x265hip_ctx_stream() against the pointers above calls this first: everything every sub-batch queued so far is then
 * ordered in front of what follows on the context's stream (no host wait). */
int   x265hip_batch_join(x265hip_batch* batch);

/* ---- ThreadedME producer for a C++ encoder: one picture's MEData table from HOST data ------------------------------------------------------------------
 * What ThreadedME::findJob -> Analysis::deriveMVsForCTU does for every CTU of a picture (threadedme.cpp:207-261, analysis.cpp:248-306): the diamond searches of the
 * CTU and its four sub-CUs per reference (m_areaBestMV), the collocated-median override, then every PU of the schedule (x265hip_tme_frame).  The caller hands over
 * host pointers -- the padded planes as PicYuv holds them (first element of the allocation; all planes of one geometry), the picture's MEData table
 * (slice->m_ctuMV, as x265hip_inter_choice records) as it is before the picture, per reference that reference picture's table and the lookahead's MVs -- and what only
 * the encoder's own state yields: per CTU and reference the median of the collocated MVs (CUData::getMedianColMV, cudata.cpp:1744-1790), per (CTU, entry, partition)
 * the temporal neighbour (CUData::getNeighbourMV's collocated part), and the qp of every CU (Analysis::calculateQpforCuSize). */
typedef struct x265hip_tme x265hip_tme;
typedef struct x265hip_tme_host_ref {
    const void* mePlane;                       /* slice->m_mref[l][r].fpelPlane[0] allocation (weighted or not)                                  */
    const void* reconPlane;                    /* the reconstructed reference picture's plane (== mePlane without weighting)                     */
    const struct x265hip_inter_choice* refTable;   /* that picture's own table or NULL (intra picture)                                           */
    const int16_t* lowresMv;                   /* Lowres::lowresMvs[l][dist] as x, y per 16x16 block, or NULL (not estimated / distance out of range) */
    uint64_t reconKey;                         /* identity of the reconstructed picture (e.g. Frame::m_encodeOrder + 1): the producer keeps the planes of the last
                                                  pictures it saw on the device and uploads / phase-interpolates a picture once; 0 = no identity, upload every time.  A keyed picture is
                                                  either COMPLETE when first seen (reconRowsValid = 0) or grows: see reconRowsValid */
    uint64_t meKey;                            /* identity of mePlane when it is not the reconstruction (the weighted plane MotionReference::applyWeight fills for the CURRENT picture, e.g.
                                                  (picture's encode order + 1) << 8 | list << 5 | ref ... any value unique among live planes); 0 = uploaded with every call                */
    int reconRowsValid, meRowsValid;           /* frame threads: rows of the plane allocation (counted from its first row, the top margin included) that are final NOW -- the reference is
                                                  still being reconstructed (Frame::m_reconRowFlag, frameencoder.cpp:1029-1036) / weighted (MotionReference::numSliceWeightedRows).  0 = the
                                                  whole plane.  A keyed plane is uploaded and phase-interpolated incrementally: each call adds the rows the producer has not seen yet (a call
                                                  may declare FEWER rows than an earlier one -- each band declares what its own searches need: the producer keeps what it has).  The caller hands over what the searches of desc.ctuRowFirst .. + ctuRowCount may read (the encoder's own
                                                  row-lag rule: every reference row up to CTU row + FrameEncoder::m_refLagRows, the window clamped by Search::m_refLagPixels)                */
} x265hip_tme_host_ref;
typedef struct x265hip_tme_picture_desc {
    int isP, numRef[2], curPOC, temporalMvp, refPOC[2][16];
    int searchRange, searchMethod, subpelRefine;
    int width, height, lowresBlocksX;
    const void* curPlane; intptr_t stride; int64_t origin, planeElems;
    x265hip_tme_host_ref refs[2][X265HIP_MAX_REF];      /* numRef[l] <= X265HIP_MAX_REF = 16 (MAX_NUM_REF); more is X265HIP_EARG */
    struct x265hip_inter_choice* table;        /* [numCtu][593], in / out                                                                        */
    const int16_t* median;                     /* [numCtu][2][X265HIP_MAX_REF][3]: valid, x, y of getMedianColMV; NULL = none                   */
    const x265hip_tme_temporal* temporal;      /* [numCtu][entries][2]                                                                           */
    int nQp, qps[64];                           /* the distinct qps of the picture's CUs                                                          */
    const uint8_t* qpIndex;                    /* [numCtu][entries]: index into qps of the entry's CU                                            */
    const uint8_t* areaQpIndex;                /* [numCtu][5]: index into qps of the CTU (area 0) and its four sub-CUs (the diamond searches)    */
    int sourceHeight;                          /* param->sourceHeight (the unpadded picture); 0 = height                                         */
    int frameThreads;                          /* param->frameNumThreads; 0 or 1: reference pictures are complete.  > 1 models the encoder's window and selectMVP restrictions
                                                  (m_refLagPixels = searchRange, m_bFrameParallel); the caller must still hand over finished reference rows only            */
    int flags;                                 /* X265HIP_TME_LAUNCH_PER_STAGE / X265HIP_TME_PACKED_GROUPS (x265hip_frame.h); 0 for production   */
    int16_t* areaBestOut;                      /* optional [numCtu][5][2][X265HIP_MAX_REF][2]: m_areaBestMV as computed                         */
    int ctuRowFirst, ctuRowCount;              /* a band of CTU rows of the picture (ThreadedME under frame threads queues a picture's rows as their reference rows become
                                                  final, threadedme.cpp:121-150): only these rows' CTUs are searched, only their entries of table / median / temporal / qpIndex /
                                                  areaQpIndex / areaBestOut are read and written (the arrays keep the picture's CTU addressing).  0, 0 = the whole picture.
                                                  CTUs of a picture do not depend on each other (findJob takes them in any order), so bands in any order give the picture's table */
    int pirStartCol, pirSafeX;                 /* --intra-refresh (Search::setSearchRange, search.cpp:4987-4996): in a P picture whose first reference has not finished its refresh sweep
                                                  (its pirEndCol < numCuInWidth) the CUs of CTU columns left of the picture's own pirStartCol search no further right than
                                                  4 * (pirSafeX - cuX) quarter-pels, pirSafeX = the reference's pirEndCol * ctuSize - 3.  pirStartCol = 0: no restriction
                                                  (B pictures, intra refresh off, sweep finished)                                                     */
} x265hip_tme_picture_desc;
int  x265hip_tme_create(x265hip_ctx* ctx, int width, int height, int ctuSize, int minCuSize, int rect, int amp, x265hip_tme** tme);
void x265hip_tme_destroy(x265hip_tme* tme);
int  x265hip_tme_entries(const x265hip_tme* tme, const x265hip_tme_step** steps);     /* the schedule (x265hip_tme_schedule) this producer steps through */
/* optional: page-lock a long-lived host buffer of the caller (a PicYuv plane allocation, a FrameData table) once; copies from / to it are then DMA transfers */
int  x265hip_host_register(void* p, size_t bytes);
int  x265hip_host_unregister(void* p);
int  x265hip_tme_picture(x265hip_tme* tme, const x265hip_tme_picture_desc* desc);     /* synchronous: desc->table holds the picture's (band's) records on return; one call at a time per producer */

/* ---- lookahead producer for a C++ encoder: lowres intra costs and frame-cost estimates from HOST data ------------------------------------------------------------
 * What the lookahead's workers do per picture and per (p0, b, p1) choice: LookaheadTLD::lowresIntraEstimate (slicetype.cpp:755-864) and CostEstimateGroup::estimateFrameCost
 * (:4366-4463; estimateCUCost :4467-4640).  The half-resolution pictures (Lowres::buffer[0]: four planes of planeElems pixels back to back, lowres.cpp:84-140) stay on
 * the device under the caller's key (e.g. Lowres::frameNum + 1; the least recently used of maxPictures makes room, a missing picture is uploaded again from `planes`).
 * Arrays are the reference's own: MVs as int16 x, y per 8x8 block (Lowres::lowresMvs[list][dist], MV = two int32 there), costs int32 (lowresMvCosts), lowresCosts uint16
 * (cost | listused << 14), rowSatds int32 per block row; sums = { costEst before the B-frame normalisation (:4456-4457), costEstAq, intraMbs }.  Calls are synchronous
 * (the lookahead's workers may call concurrently).  --hme is offered for the serial sweep (x265hip_la_enable_hme + desc.hme). */
typedef struct x265hip_la x265hip_la;
int  x265hip_la_create(x265hip_ctx* ctx, int widthInCU, int heightInCU, intptr_t stride /* Lowres::lumaStride */, int64_t planeElems /* buffer[1] - buffer[0] */,
                       int64_t origin /* lowresPlane[0] - buffer[0] */, int maxPictures, x265hip_la** la);
void x265hip_la_destroy(x265hip_la* la);
/* --hme: the quarter-resolution pictures (Lowres::lowerResBuffer[0], four planes of planeElems4 = lowerResBuffer[1] - lowerResBuffer[0] pixels; rows lumaStride / 2 apart;
 * origin4 = lowerResPlane[0] - lowerResBuffer[0]) on the Lookahead::m_4x4Width x m_4x4Height grid of 8x8 blocks.  Once, before the first estimate with desc.hme. */
int  x265hip_la_enable_hme(x265hip_la* la, int widthInCU4, int heightInCU4, intptr_t stride4, int64_t planeElems4, int64_t origin4);
/* make a picture resident (no-op when it is): planes4 = Lowres::buffer[0]; invQscale = Lowres::invQscaleFactor (or ...8x8 with qgSize 8) or NULL; intraCost: optional,
 * for callers that ran the intra estimate themselves */
int  x265hip_la_picture(x265hip_la* la, uint64_t key, const void* planes4, const int32_t* invQscale, const int32_t* intraCost);
int  x265hip_la_forget(x265hip_la* la, uint64_t key);         /* the picture left the lookahead: its slot is free */
/* lowresIntraEstimate of one picture (made resident on the way): intraCost / intraMode / lowresCosts[0][0] per block, rowSatds[0][0], sums2 = { costEst[0][0], costEstAq[0][0] } */
int  x265hip_la_intra(x265hip_la* la, uint64_t key, const void* planes4, const int32_t* invQscale, int32_t* intraCost, uint8_t* intraMode, uint16_t* lowresCosts,
                      int32_t* rowSatds, int64_t* sums2);
typedef struct x265hip_la_estimate_desc {
    uint64_t key[3];                 /* p0, b, p1; key[2] == key[1]: P estimate                                                                       */
    const void* planes[3];           /* Lowres::buffer[0] of each, used when the picture is not resident (may be NULL when it certainly is)           */
    const int32_t* invQscale;        /* of b, used with planes[1]                                                                                     */
    const int32_t* intraCost;        /* of b, used with planes[1] (a resident b holds the costs x265hip_la_intra / x265hip_la_picture left)          */
    const void* weightedPlanes;      /* NULL, or the four planes of the weighted copy of p0 (LookaheadTLD::wbuffer[0], slicetype.cpp:919-1020): list 0 is searched there */
    int doSearch[2];                 /* per list: search (mvs / mvCosts are outputs) or reuse the list's earlier result (inputs) -- bDoSearch, :4376-4377 */
    int rowsPerSlice;                /* 0 = the serial sweep; > 0 = Lookahead::m_numRowsPerSlice of the cooperative --lookahead-slices sweep          */
    int16_t* mvs[2]; int32_t* mvCosts[2];       /* per list, ncu entries (list 1 unused for a P estimate)                                              */
    uint16_t* lowresCosts; int32_t* rowSatds; int64_t* sums;      /* ncu, heightInCU, 3                                                                 */
    /* --hme (after x265hip_la_enable_hme): the quarter-resolution sweep runs first and seeds the half-resolution one (x265hip_lookahead_cost_batch_hme) */
    int hme;                         /* != 0: param->bEnableHME                                                                                       */
    const void* lowerPlanes[3];      /* Lowres::lowerResBuffer[0] of p0, b, p1 (uploaded once per picture)                                            */
    int hmeMethod[2], hmeRange[2];   /* param->hmeSearchMethod[0..1] (X265HIP_ME_DIA / _HEX / _UMH / _STAR / _FULL), param->hmeRange[0..1]                  */
    int16_t* lowerMvs[2]; int32_t* lowerMvCosts[2];      /* optional outputs per searched list: Lowres::lowerResMvs / lowerResMvCosts (m_4x4Width * m_4x4Height entries) */
} x265hip_la_estimate_desc;
int  x265hip_la_estimate(x265hip_la* la, const x265hip_la_estimate_desc* desc);
/* n estimates at once -- what CostEstimateGroup::finishBatch holds (slicetype.cpp:4236-4278: up to 512 queued (p0, b, p1) triples): ceil(n / X265HIP_LA_MAX_BATCH) launches, in the
 * caller's order, synchronous.  Estimates of one call must not depend on each other's searches (an estimate that reuses a list search another one of the SAME call makes belongs in
 * the next call -- integration/lookahead_adapter.cpp splits a batch into such waves); two estimates that both search the same (b, list, distance) simply both do. */
#define X265HIP_LA_MAX_BATCH 32
int  x265hip_la_estimate_batch(x265hip_la* la, const x265hip_la_estimate_desc* descs, int n);
/* x265hip_la_estimate is synchronous for its caller, but estimates that arrive from other threads while a launch is in flight go up TOGETHER as the next launch (up to X265HIP_LA_MAX_BATCH):
 * the lookahead's batched frame costs (b-adapt 2 with a thread pool) become batches on the device.  launches / estimates so far: */
int  x265hip_la_batch_stats(const x265hip_la* la, int64_t* launches, int64_t* estimates);
/* cuTree, one propagation step for a host caller: Lookahead::estimateCUPropagate (slicetype.cpp:3850-3953; primitives.propagateCost, pixel.cpp:906-931) on the caller's
 * arrays of picture b and its two references -- x265hip_cutree_propagate (include/x265hip_frame.h) with the staging around it.  All arrays have ncu = widthInCU * heightInCU
 * entries (the producer's geometry); mvs are int16 x, y pairs (4-byte aligned); propB is read (the reference memsets its first row itself when !referenced, :3870-3871) and
 * prop0 / prop1 are updated in place (prop1 / mvs1 may be NULL for a P picture: distP1 == 0).  Integer results, the reference's double arithmetic: identical.  Synchronous. */
typedef struct x265hip_la_cutree_desc {
    int distP0, distP1, weightedBiPred, referenced; double fpsFactor;
    const int32_t* intraCost; const uint16_t* lowresCosts; const int32_t* invQscale; const int16_t* mvs0; const int16_t* mvs1;
    const uint16_t* propB; uint16_t* prop0; uint16_t* prop1;
} x265hip_la_cutree_desc;
int  x265hip_la_cutree_propagate(x265hip_la* la, const x265hip_la_cutree_desc* desc);

/* ---------------------------------------------------------------------------------------------------------------------------------------------
 * In-loop filter producer for a host caller: what FrameFilter does to one reconstructed 4:2:0 picture between its reconstruction and the SAO decision
 * (encoder/framefilter.cpp:451-573 ParallelFilter::processTasks): the deblocking filter of the whole picture (Deblock::deblockCTU on every CTU, both edge directions)
 * and the SAO statistics of every CTU on the deblocked picture (SAO::calcSaoStatsCTU for the three planes) -- one call, host arrays in and out, on top of
 * x265hip_deblock_frame / x265hip_sao_stats_frame (include/x265hip_frame.h).  The SAO decision (rate-distortion search with the encoder's entropy coder) and what
 * follows stay with the caller.  What integration/filter_adapter.cpp binds inside the reference encoder.
 * The picture is described as for x265hip_deblock_frame (CUData's per-partition arrays, CTU after CTU -- here HOST arrays); planes are host pointers to pixel (0,0),
 * rows strideY / strideC (given at creation) apart; the source planes have the same strides (PicYuv of one encoder).  4:2:0; --slices through pic.sliceFirstRow (a HOST array
 * here); every class of every CTU is returned (--limit-sao is the caller's choice of which to use, integration/filter_adapter.cpp).
 * --------------------------------------------------------------------------------------------------------------------------------------------- */
typedef struct x265hip_ff x265hip_ff;
int  x265hip_ff_create(x265hip_ctx* ctx, int width, int height, int ctuSize, intptr_t strideY, intptr_t strideC, x265hip_ff** out);
void x265hip_ff_destroy(x265hip_ff* ff);
typedef struct x265hip_ff_picture_desc
{
    x265hip_deblock_pic pic;                   /* width / height / ctuSize as at creation; the array pointers are HOST pointers (numCtu * numPartitions entries each)         */
    void *reconY, *reconCb, *reconCr;          /* in: the reconstructed picture; out (deblock != 0): the deblocked picture (the picture area only, the borders are untouched) */
    const void *fencY, *fencCb, *fencCr;       /* the source picture (saoStats != 0)                                                                                            */
    int deblock;                               /* param->bEnableLoopFilter                                                                                                      */
    int saoStats;                              /* bit 0: luma statistics, bit 1: the two chroma planes' (SAOParam::bSaoFlag[0] / [1])                                           */
    int saoNonDeblocked;                       /* param->bSaoNonDeblocked (the skipped border widths of calcSaoStatsCTU)                                                         */
    int32_t* stats[3];                         /* out, per plane: per CTU in raster order [2][5][32] int32 = m_offsetOrg then m_count of that plane (x265hip_sao_stats_frame) */
    int ctuRowFirst, ctuRowCount;              /* a band of CTU rows (FrameFilter::processRow runs a picture's filters row by row behind the row encoders, framefilter.cpp:576-676, and
                                                  under frame threads the next pictures wait for the rows it finishes): only these rows' CTUs are deblocked -- the band's top edge
                                                  included, which changes the last 3 luma / 1 chroma lines of the row above -- and only their statistics are written.  The arrays and
                                                  planes keep the picture's addressing; read: the CU arrays of the band's rows and of the row above, the reconstruction from 8 luma lines
                                                  above the band (the row above as the previous band left it: deblocked, SAO not yet applied) to the band's last line; written back: from
                                                  4 luma lines above the band.  A band that begins the picture or a slice (pic.sliceFirstRow) reads and writes nothing above itself.  Bands of a picture must come in increasing order; bands of different pictures may interleave.  0, 0 = the whole picture */
} x265hip_ff_picture_desc;
int  x265hip_ff_picture(x265hip_ff* ff, const x265hip_ff_picture_desc* desc);

#ifdef __cplusplus
}
#endif
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#endif
