/*
 * x265hip.h -- C ABI of the MI355X (gfx950) back end for x265's encoder-primitives hot path.
 *
 * One shared library per bit depth, like the reference's multilib builds:
 *   libx265hip_8.so   (pixel = uint8_t,  sse_t = uint32_t)   <-> X265_DEPTH 8
 *   libx265hip_10.so  (pixel = uint16_t, sse_t = uint64_t)   <-> X265_DEPTH 10, HIGH_BIT_DEPTH
 * (reference: source/common/common.h:127-149).
 *
 * Two layers:
 *  (1) DROP-IN TABLE.  x265hip_setup_primitives() overwrites the hot-path slots of an
 *      x265 `EncoderPrimitives` (reference: source/common/primitives.h:239-432) with functions
 *      that have exactly the reference's per-slot C signatures (primitives.h:133-236, strides in
 *      ELEMENTS, caller-owned host buffers) and run the arithmetic on the GPU.  It is the
 *      "one more overwrite pass" of x265_setup_primitives() (primitives.cpp:336-376): call it
 *      after setupAssemblyPrimitives() and before setupAliasPrimitives().
 *  (2) BATCHED DEVICE API.  The same arithmetic over thousands of blocks / whole frames per
 *      launch, on device-resident planes (plain device pointers + a hipStream_t passed as
 *      void*).  This is where the throughput is; (1) exists for parity and drop-in semantics.
 *
 * All functions return 0 on success or a negative X265HIP_E* code; slot functions cannot return
 * errors (reference convention: primitives.h typedefs return only results) and abort() with a
 * message on stderr if the GPU is unusable -- there is NO CPU fallback anywhere in this library.
 */
#ifndef X265HIP_H
#define X265HIP_H
#include <stddef.h>
#include <stdint.h>

/* the library is built with -fvisibility=hidden: what these headers declare is its whole exported surface */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif
#ifdef __cplusplus
extern "C" {
#endif

#define X265HIP_OK 0
#define X265HIP_EABI (-1)     /* table size / bit depth mismatch */
#define X265HIP_EDEVICE (-2)  /* no usable gfx950 device / HIP runtime error */
#define X265HIP_EARG (-3)     /* invalid argument */

/* ---- EncoderPrimitives layout (x86-64, both depths; probed from the reference headers and
 *      re-checked by tests/test_abi_layout.py against offsetof() of the real struct) ---- */
#define X265HIP_SIZEOF_TABLE 18240
#define X265HIP_NUM_PU 25          /* enum LumaPU, primitives.h:41-55 */
#define X265HIP_NUM_CU 5           /* enum LumaCU, primitives.h:57-65 */
#define X265HIP_OFF_PU 0
#define X265HIP_PU_PTRS 19         /* sizeof(PU) = 152 */
#define X265HIP_OFF_CU 3800
#define X265HIP_CU_PTRS 73         /* sizeof(CU) = 584 */
#define X265HIP_OFF_SCALARS 6720   /* dst4x4 */
#define X265HIP_OFF_CHROMA 7200
#define X265HIP_CHROMA_BYTES 2760  /* pu[25] x 12 ptrs + cu[5] x 9 ptrs */
#define X265HIP_CHROMA_PU_PTRS 12
#define X265HIP_CHROMA_CU_PTRS 9

/* pointer index inside one PU entry (primitives.h:247-267) */
enum x265hip_pu_slot {
    X265HIP_PU_SAD, X265HIP_PU_SAD_X3, X265HIP_PU_SAD_X4, X265HIP_PU_ADS, X265HIP_PU_SATD,
    X265HIP_PU_LUMA_HPP, X265HIP_PU_LUMA_HPS, X265HIP_PU_LUMA_VPP, X265HIP_PU_LUMA_VPS,
    X265HIP_PU_LUMA_VSP, X265HIP_PU_LUMA_VSS, X265HIP_PU_LUMA_HVPP,
    X265HIP_PU_PIXELAVG_PP, X265HIP_PU_PIXELAVG_PP_ALIGNED,
    X265HIP_PU_ADDAVG, X265HIP_PU_ADDAVG_ALIGNED, X265HIP_PU_COPY_PP,
    X265HIP_PU_CONVERT_P2S, X265HIP_PU_CONVERT_P2S_ALIGNED
};
/* pointer index inside one CU entry (primitives.h:275-316) */
enum x265hip_cu_slot {
    X265HIP_CU_DCT, X265HIP_CU_IDCT, X265HIP_CU_STANDARD_DCT, X265HIP_CU_LOWPASS_DCT,
    X265HIP_CU_CALCRESIDUAL, X265HIP_CU_CALCRESIDUAL_ALIGNED, X265HIP_CU_SUB_PS,
    X265HIP_CU_ADD_PS, X265HIP_CU_ADD_PS_ALIGNED, X265HIP_CU_BLOCKFILL_S, X265HIP_CU_BLOCKFILL_S_ALIGNED,
    X265HIP_CU_COPY_CNT, X265HIP_CU_COUNT_NONZERO, X265HIP_CU_CPY2DTO1D_SHL, X265HIP_CU_CPY2DTO1D_SHR,
    X265HIP_CU_CPY1DTO2D_SHL, X265HIP_CU_CPY1DTO2D_SHL_ALIGNED, X265HIP_CU_CPY1DTO2D_SHR,
    X265HIP_CU_COPY_SP, X265HIP_CU_COPY_PS, X265HIP_CU_COPY_SS, X265HIP_CU_COPY_PP, X265HIP_CU_VAR,
    X265HIP_CU_SSE_PP, X265HIP_CU_SSE_SS, X265HIP_CU_PSY_COST_PP, X265HIP_CU_SSD_S, X265HIP_CU_SSD_S_ALIGNED,
    X265HIP_CU_SA8D, X265HIP_CU_TRANSPOSE, X265HIP_CU_INTRA_PRED_ALLANGS, X265HIP_CU_INTRA_FILTER,
    X265HIP_CU_INTRA_PRED /* [35] */, X265HIP_CU_NONPSYRDOQUANT = X265HIP_CU_INTRA_PRED + 35
};
/* byte offsets of the scalar slots we fill (primitives.h:320-371) */
#define X265HIP_OFF_DST4X4 6720
#define X265HIP_OFF_IDST4X4 6728
#define X265HIP_OFF_QUANT 6736
#define X265HIP_OFF_NQUANT 6744
#define X265HIP_OFF_DEQUANT_SCALING 6752
#define X265HIP_OFF_DEQUANT_NORMAL 6760
#define X265HIP_OFF_DENOISEDCT 6768
#define X265HIP_OFF_SCALE1D_128TO64 6776   /* [2] */
#define X265HIP_OFF_SCALE2D_64TO32 6792
#define X265HIP_OFF_SAOCUSTATSBO 6888          /* then saoCuStatsE0..E3 at +8, +16, +24, +32 */
#define X265HIP_OFF_FRAMEINITLOWRES 6928
#define X265HIP_OFF_FRAMEINITLOWERRES 6936
#define X265HIP_OFF_PROPAGATECOST 6944
#define X265HIP_OFF_FIX8UNPACK 6952
#define X265HIP_OFF_FIX8PACK 6960
#define X265HIP_OFF_EXTENDROWBORDER 6968
#define X265HIP_OFF_INTEGRAL_INITV 7104     /* [6]: 4, 8, 12, 16, 24, 32 rows (primitives.h:122-131) */
#define X265HIP_OFF_INTEGRAL_INITH 7152     /* [6]: 4, 8, 12, 16, 24, 32 columns */
#define X265HIP_OFF_WEIGHT_SP 7016
#define X265HIP_OFF_WEIGHT_PP 7024
/* pointer index inside Chroma::PUChroma / CUChroma (primitives.h:399-428) */
enum x265hip_chroma_pu_slot {
    X265HIP_CPU_SATD, X265HIP_CPU_FILTER_VPP, X265HIP_CPU_FILTER_VPS, X265HIP_CPU_FILTER_VSP, X265HIP_CPU_FILTER_VSS,
    X265HIP_CPU_FILTER_HPP, X265HIP_CPU_FILTER_HPS, X265HIP_CPU_ADDAVG, X265HIP_CPU_ADDAVG_ALIGNED,
    X265HIP_CPU_COPY_PP, X265HIP_CPU_P2S, X265HIP_CPU_P2S_ALIGNED
};
enum x265hip_chroma_cu_slot {
    X265HIP_CCU_SA8D, X265HIP_CCU_SSE_PP, X265HIP_CCU_SUB_PS, X265HIP_CCU_ADD_PS, X265HIP_CCU_ADD_PS_ALIGNED,
    X265HIP_CCU_COPY_PS, X265HIP_CCU_COPY_SP, X265HIP_CCU_COPY_SS, X265HIP_CCU_COPY_PP
};

/* flags for x265hip_setup_primitives */
#define X265HIP_FILL_ALL 0u
#define X265HIP_FILL_NO_ALLANGS 1u   /* leave cu[].intra_pred_allangs untouched (callers test it for NULL, search.cpp:1723) */

/* ---------------------------------------------------------------------------------- */
/* (0) library / device                                                               */
/* ---------------------------------------------------------------------------------- */
int x265hip_bit_depth(void);                         /* 8 or 10: the X265_DEPTH this .so was built for */
const char* x265hip_last_error(void);                /* thread-local message of the last failing call */
int x265hip_device_init(int device);                 /* select the HIP device for the calling thread */

/* replaces the sizeof/depth checks a compiled-in back end gets for free (primitives.cpp:336-376) */
int x265hip_abi_check(size_t sizeof_table, int bit_depth);
/* the overwrite pass: void setup*Primitives(EncoderPrimitives&, int cpuMask) (primitives.h:471-474) */
int x265hip_setup_primitives(void* encoder_primitives, int bit_depth, uint32_t flags);

/* ---------------------------------------------------------------------------------- */
/* (2) batched device API.  `stream` is a hipStream_t; every pointer is DEVICE memory.  */
/*     Item i works at base + off[i] (element offsets, int32) -- one launch per call.   */
/* ---------------------------------------------------------------------------------- */

/* pixel-compare ops; result per item: int32 (SAD/SATD/SA8D/PSY) or uint64 (SSE*) */
enum x265hip_cmp_op { X265HIP_CMP_SAD, X265HIP_CMP_SATD, X265HIP_CMP_SA8D, X265HIP_CMP_SSE_PP,
                      X265HIP_CMP_PSY_COST, X265HIP_CMP_SSE_SS, X265HIP_CMP_SSD_S };
/* replaces pu[].sad / pu[].satd / cu[].sa8d / cu[].sse_pp / cu[].psy_cost_pp / cu[].sse_ss / cu[].ssd_s
 * (pixel.cpp:40-383,718-749) over n block pairs.  a/b are pixel planes (int16 planes for SSE_SS/SSD_S). */
int x265hip_pixelcmp_batch(void* stream, int op, int w, int h,
                           const void* a, intptr_t strideA, const int32_t* offA,
                           const void* b, intptr_t strideB, const int32_t* offB,
                           int n, void* out);

/* element-wise block ops (pixel.cpp:385-483,485-594,751-854) over n blocks */
enum x265hip_blk_op { X265HIP_BLK_CALCRESIDUAL, X265HIP_BLK_SUB_PS, X265HIP_BLK_ADD_PS, X265HIP_BLK_COPY_PP,
                      X265HIP_BLK_COPY_SS, X265HIP_BLK_COPY_SP, X265HIP_BLK_COPY_PS, X265HIP_BLK_FILL_S,
                      X265HIP_BLK_2DTO1D_SHL, X265HIP_BLK_2DTO1D_SHR, X265HIP_BLK_1DTO2D_SHL, X265HIP_BLK_1DTO2D_SHR,
                      X265HIP_BLK_TRANSPOSE, X265HIP_BLK_ADDAVG, X265HIP_BLK_PIXELAVG, X265HIP_BLK_WEIGHT_SP,
                      X265HIP_BLK_WEIGHT_PP, X265HIP_BLK_SCALE1D, X265HIP_BLK_SCALE2D };
typedef struct x265hip_blk_args {
    void* dst; intptr_t dstStride; const int32_t* dstOff;
    const void* src0; intptr_t src0Stride; const int32_t* src0Off;
    const void* src1; intptr_t src1Stride; const int32_t* src1Off;
    int p0, p1, p2, p3;   /* op parameters: shift / fill value / (w0, round, shift, offset) */
} x265hip_blk_args;
int x265hip_blockop_batch(void* stream, int op, int w, int h, const x265hip_blk_args* args, int n);

/* transforms (dct.cpp:443-611): n TUs of size N; src item i at src + srcOff[i] with srcStride,
 * dst dense N*N at dst + i*N*N (forward) / the mirror image for inverse. dst4 = DST-VII 4x4. */
enum x265hip_tr_op { X265HIP_TR_DCT, X265HIP_TR_IDCT, X265HIP_TR_DST4, X265HIP_TR_IDST4,
                     X265HIP_TR_LOWPASS /* lowPassDct8/16/32_c, lowpassdct.cpp:34-116; N = full block size */ };
int x265hip_transform_batch(void* stream, int op, int N,
                            const int16_t* src, intptr_t srcStride, const int32_t* srcOff,
                            int16_t* dst, intptr_t dstStride, const int32_t* dstOff, int n);

/* Lookahead plane preparation: frame_init_lowres_core (pixel.cpp:596-622) -- the half-resolution plane and its
 * three half-pel companions of a width x height LOWRES picture; reads source rows 0 .. 2*height and columns 0 .. 2*width.
 * Device pointers. */
int x265hip_frame_init_lowres(void* stream, const void* src, intptr_t srcStride, void* dst0, void* dsth, void* dstv, void* dstc,
                              intptr_t dstStride, int width, int height);

/* Reference-plane preparation: extendPicBorder (pixel.cpp:1044-1058) on nPictures padded pictures of one
 * device allocation -- picOrg = pixel (0,0) of picture 0, picture i at picOrg + i*pictureElems.  Rows are widened first
 * (p.extendRowBorder, ipfilter.cpp:59-77), then the widened top / bottom rows are replicated into the vertical margins,
 * so a reconstructed frame becomes a searchable reference without leaving HBM. */
int x265hip_extend_pic_border(void* stream, void* picOrg, intptr_t stride, int width, int height, int marginX, int marginY,
                              int nPictures, int64_t pictureElems);

/* Row-granular helpers behind the slots propagateCost, fix8Pack / fix8Unpack (cuTree; pixel.cpp:906-948) and integral_init{4..32}{h,v}
 * (SEA; encoder/framefilter.cpp:38-139).  Device pointers.  The frame-level forms are x265hip_cutree_propagate and
 * x265hip_sea_integral_planes (x265hip_frame.h).
 *   integral_init_h: sum[x] = (pix[x] + ... + pix[x + boxWidth - 1]) + above[x] for x in [0, positions); above = the row before sum
 *   integral_init_v: top[x] = below[x] - top[x]                                  for x in [0, positions); below = top + N rows */
int x265hip_propagate_cost_row(void* stream, int32_t* dst, const uint16_t* propagateIn, const int32_t* intraCosts, const uint16_t* interCosts,
                               const int32_t* invQscales, double fpsFactor, int len);
int x265hip_fix8_convert(void* stream, int pack, void* dst, const void* src, int count);     /* pack: double -> Q8.8 uint16; else the reverse */
int x265hip_integral_init_h(void* stream, uint32_t* sum, const uint32_t* above, const void* pix, int boxWidth, int positions);
int x265hip_integral_init_v(void* stream, uint32_t* top, const uint32_t* below, int positions);

/* SAO statistics of one block (encoder/sao.cpp:1774-1937): type 0..3 = saoCuStatsE0..E3, 4 = saoCuStatsBO.  diff has stride 64;
 * rec points at the block's first pixel (the edge classes read one pixel / row around it).  out[0..31] = per-class sums of diff, out[32..63] =
 * counts (edge classes in SAO::s_eoTable order) -- deltas, the caller adds them.  upIn = the reference's upBuff1 on entry (types 1-3); upOutA /
 * upOutB receive what the reference leaves in upBuff1 / upBufft (type 1: [0, endX); type 2: A = [0, endX] of the last odd row, B = of the last even
 * row; type 3: upOutA[0] stands for upBuff1[-1], [1 .. endX] for [0 .. endX - 1]).  Device pointers. */
int x265hip_sao_stats(void* stream, int type, const int16_t* diff, const void* rec, intptr_t stride, const int8_t* upIn, int endX, int endY,
                      int32_t* out, int8_t* upOutA, int8_t* upOutB);

/* SEA pre-filter pu[].ads (pixel.cpp:121-165; which PU uses x1 / x2 / x4: pixel.cpp:1122-1146): for i in [0, width)
 * ads = sum |encDC[k] - sums[i + off_k]| + costMvX[i]; positions with ads < thresh are appended, in order, to mvs.
 * parts = 1, 2 or 4; lx = PU width.  All pointers are device pointers; *nmv receives the count. */
int x265hip_ads(void* stream, int parts, int lx, const int32_t* encDC, const uint32_t* sums, int delta,
                const uint16_t* costMvX, int16_t* mvs, int width, int thresh, int32_t* nmv);

/* quant family (dct.cpp:614-715): n dense blocks of numCoeff coefficients each.
 * quantCoeff is ONE numCoeff-long table shared by all blocks (per-TU-size scaling list). */
int x265hip_quant_batch(void* stream, const int16_t* coef, const int32_t* quantCoeff, int32_t* deltaU /*may be NULL*/,
                        int16_t* qCoef, int qBits, int add, int numCoeff, int n, uint32_t* numSig);
int x265hip_nquant_batch(void* stream, const int16_t* coef, const int32_t* quantCoeff, int16_t* qCoef,
                         int qBits, int add, int numCoeff, int n, uint32_t* numSig);
int x265hip_dequant_normal_batch(void* stream, const int16_t* q, int16_t* coef, int num, int scale, int shift);
int x265hip_dequant_scaling_batch(void* stream, const int16_t* q, const int32_t* deq, int16_t* coef,
                                  int numCoeff, int n, int per, int shift);

/* interpolation (ipfilter.cpp:40-369): taps 8 = luma, 4 = chroma */
enum x265hip_ip_op { X265HIP_IP_HPP, X265HIP_IP_HPS, X265HIP_IP_VPP, X265HIP_IP_VPS, X265HIP_IP_VSP, X265HIP_IP_VSS,
                     X265HIP_IP_HVPP, X265HIP_IP_P2S };
int x265hip_interp_batch(void* stream, int op, int taps, int w, int h,
                         const void* src, intptr_t srcStride, const int32_t* srcOff,
                         void* dst, intptr_t dstStride, const int32_t* dstOff,
                         const int32_t* coeffIdx /* per item: idxX | idxY<<8 | isRowExt<<16 */, int n);

/* intra (intrapred.cpp:31-234): neighbour arrays of 4N+1 pixels at nb + nbOff[i] */
int x265hip_intra_filter_batch(void* stream, int N, const void* nb, const int32_t* nbOff, void* filt, const int32_t* filtOff, int n);
int x265hip_intra_pred_batch(void* stream, int N, const void* nb, const int32_t* nbOff,
                             void* dst, intptr_t dstStride, const int32_t* dstOff,
                             const int32_t* modeFilter /* mode | bFilter<<8 */, int n);
int x265hip_intra_allangs_batch(void* stream, int N, const void* ref, const int32_t* refOff,
                                const void* filt, const int32_t* filtOff, void* dst /* n x 33*N*N dense */, int bLuma, int n);

/* Intra mode scan of n square CUs -- the data-parallel half of Search::estIntraPredQT (encoder/search.cpp:1655-1745):
 * costs[i*35 + mode] = sa8d(source block, prediction of `mode`) for mode 0 (planar), 1 (DC) and the 33 angular modes, with the
 * reference's choices of neighbour array (g_intraFilterFlags), edge filter (size <= 16) and 64x64 handling (source and neighbours
 * scaled to 32x32 by scale2D_64to32 / scale1D_128to64, no filtered neighbours, costs << 2).  nbRef / nbFilt: per CU the 4*size+1
 * pixels of Predict::intraNeighbourBuf[0] / [1] after initAdiPattern (predict.h:75), CU i at + i*nbPitch.  Mode bits, MPMs and
 * the RD comparison stay on the host.  64x64 needs x265hip_intra_cost_workspace() bytes of device workspace. */
size_t x265hip_intra_cost_workspace(int log2Size, int n);
int x265hip_intra_cost_batch(void* stream, int log2Size, const void* srcPlane, intptr_t srcStride, const int32_t* srcOff,
                             const void* nbRef, const void* nbFilt, int nbPitch, int n, int32_t* costs,
                             void* workspace, size_t workspaceBytes);

#ifdef __cplusplus
}
#endif
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#endif /* X265HIP_H */
