/*
 * x265_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's C encoder primitives for the
 * hot path named in BASELINE.json (SAD/SATD/sa8d/SSE, interpolation, DCT/IDCT,
 * quant/dequant, intra prediction, residual/recon helpers).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (libx265hip_*.so) never links or calls it.
 *
 * Pinned against the real reference: oracle/_ref/x265ref_{8,10} (built from
 * /root/reference sources by oracle/Makefile) and tests/golden/*.npz.
 *
 * Built twice: -DX265_DEPTH=8 (pixel=u8, sse=u32) and -DX265_DEPTH=10
 * (pixel=u16, sse=u64), like the reference's multilib (common.h:127-149).
 * Strides are in ELEMENTS, as in primitives.h:133-236.
 */
#ifndef X265_ORACLE_H
#define X265_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifndef X265_DEPTH
#define X265_DEPTH 8
#endif
#if X265_DEPTH > 8
typedef uint16_t xo_pixel;
#else
typedef uint8_t xo_pixel;
#endif

#ifdef __cplusplus
extern "C" {
#endif

int xo_bit_depth(void);

/* ---- pixel compare family (pixel.cpp:40-383,718-749) ---- */
int      xo_sad(int w, int h, const xo_pixel* a, intptr_t sa, const xo_pixel* b, intptr_t sb);
void     xo_sad_x3(int w, int h, const xo_pixel* fenc, const xo_pixel* r0, const xo_pixel* r1, const xo_pixel* r2, intptr_t rs, int32_t* res);
void     xo_sad_x4(int w, int h, const xo_pixel* fenc, const xo_pixel* r0, const xo_pixel* r1, const xo_pixel* r2, const xo_pixel* r3, intptr_t rs, int32_t* res);
int      xo_satd(int w, int h, const xo_pixel* a, intptr_t sa, const xo_pixel* b, intptr_t sb);
int      xo_sa8d(int size, const xo_pixel* a, intptr_t sa, const xo_pixel* b, intptr_t sb);
uint64_t xo_sse_pp(int w, int h, const xo_pixel* a, intptr_t sa, const xo_pixel* b, intptr_t sb);
uint64_t xo_sse_ss(int w, int h, const int16_t* a, intptr_t sa, const int16_t* b, intptr_t sb);
uint64_t xo_ssd_s(int size, const int16_t* a, intptr_t sa);
int      xo_psy_cost_pp(int size, const xo_pixel* src, intptr_t ss, const xo_pixel* rec, intptr_t rs);

/* ---- block ops (pixel.cpp:385-483,485-594,751-854) ---- */
void xo_calcresidual(int size, const xo_pixel* fenc, const xo_pixel* pred, int16_t* resi, intptr_t stride);
void xo_sub_ps(int w, int h, int16_t* dst, intptr_t ds, const xo_pixel* s0, const xo_pixel* s1, intptr_t ss0, intptr_t ss1);
void xo_add_ps(int w, int h, xo_pixel* dst, intptr_t ds, const xo_pixel* s0, const int16_t* s1, intptr_t ss0, intptr_t ss1);
void xo_copy_pp(int w, int h, xo_pixel* dst, intptr_t ds, const xo_pixel* src, intptr_t ss);
void xo_copy_ss(int w, int h, int16_t* dst, intptr_t ds, const int16_t* src, intptr_t ss);
void xo_copy_sp(int w, int h, xo_pixel* dst, intptr_t ds, const int16_t* src, intptr_t ss);
void xo_copy_ps(int w, int h, int16_t* dst, intptr_t ds, const xo_pixel* src, intptr_t ss);
void xo_blockfill_s(int size, int16_t* dst, intptr_t ds, int16_t val);
void xo_cpy2Dto1D_shl(int size, int16_t* dst, const int16_t* src, intptr_t ss, int shift);
void xo_cpy2Dto1D_shr(int size, int16_t* dst, const int16_t* src, intptr_t ss, int shift);
void xo_cpy1Dto2D_shl(int size, int16_t* dst, const int16_t* src, intptr_t ds, int shift);
void xo_cpy1Dto2D_shr(int size, int16_t* dst, const int16_t* src, intptr_t ds, int shift);
void xo_transpose(int size, xo_pixel* dst, const xo_pixel* src, intptr_t ss);
void xo_addAvg(int w, int h, const int16_t* s0, const int16_t* s1, xo_pixel* dst, intptr_t ss0, intptr_t ss1, intptr_t ds);
void xo_pixelavg_pp(int w, int h, xo_pixel* dst, intptr_t ds, const xo_pixel* s0, intptr_t ss0, const xo_pixel* s1, intptr_t ss1);
void xo_weight_sp(const int16_t* src, xo_pixel* dst, intptr_t ss, intptr_t ds, int w, int h, int w0, int round, int shift, int offset);
void xo_weight_pp(const xo_pixel* src, xo_pixel* dst, intptr_t stride, int w, int h, int w0, int round, int shift, int offset);
void xo_scale1D_128to64(xo_pixel* dst, const xo_pixel* src);
void xo_scale2D_64to32(xo_pixel* dst, const xo_pixel* src, intptr_t stride);

/* ---- transform / quant family (dct.cpp:43-757) ---- */
void     xo_dct(int n, const int16_t* src, int16_t* dst, intptr_t srcStride);   /* n = 4,8,16,32 */
void     xo_intra_costs(int size, const xo_pixel* fenc, intptr_t stride, const xo_pixel* nbRef, const xo_pixel* nbFilt, int32_t* costs);   /* search.cpp:1655-1745 */
void     xo_frame_init_lowres(const xo_pixel* src0, xo_pixel* dst0, xo_pixel* dsth, xo_pixel* dstv, xo_pixel* dstc,
                              intptr_t srcStride, intptr_t dstStride, int width, int height);                      /* pixel.cpp:596-622 */
void     xo_extend_row_border(xo_pixel* txt, intptr_t stride, int width, int height, int marginX);                  /* ipfilter.cpp:59-77 */
void     xo_extend_pic_border(xo_pixel* pic, intptr_t stride, int width, int height, int marginX, int marginY);    /* pixel.cpp:1044-1058 */
void     xo_lowpass_dct(int n, const int16_t* src, int16_t* dst, intptr_t srcStride);   /* n = 8,16,32 (lowpassdct.cpp:34-116) */
int      xo_ads(int parts, int lx, const int* encDC, const uint32_t* sums, int delta, const uint16_t* costMvX, int16_t* mvs, int width, int thresh); /* pixel.cpp:121-165 */
void     xo_idct(int n, const int16_t* src, int16_t* dst, intptr_t dstStride);
void     xo_dst4(const int16_t* src, int16_t* dst, intptr_t srcStride);
void     xo_idst4(const int16_t* src, int16_t* dst, intptr_t dstStride);
uint32_t xo_quant(const int16_t* coef, const int32_t* quantCoeff, int32_t* deltaU, int16_t* qCoef, int qBits, int add, int numCoeff);
uint32_t xo_nquant(const int16_t* coef, const int32_t* quantCoeff, int16_t* qCoef, int qBits, int add, int numCoeff);
void     xo_dequant_normal(const int16_t* q, int16_t* coef, int num, int scale, int shift);
void     xo_dequant_scaling(const int16_t* q, const int32_t* deq, int16_t* coef, int num, int per, int shift);
int      xo_count_nonzero(int n, const int16_t* q);
uint32_t xo_copy_count(int n, int16_t* coef, const int16_t* resi, intptr_t rs);
void     xo_denoise_dct(int16_t* coef, uint32_t* resSum, const uint16_t* offset, int num);
const int16_t* xo_dct_matrix(int n);   /* n x n, row-major (constants.cpp:270-344) */

/* ---- SAO statistics (encoder/sao.cpp:1774-1937): type 0..3 = saoCuStatsE0..E3, 4 = saoCuStatsBO; diff stride 64 ---- */
void xo_sao_stats(int type, const int16_t* diff, const xo_pixel* rec, intptr_t stride, int8_t* upBuff1, int8_t* upBufft,
                  int endX, int endY, int32_t* stats, int32_t* count);

/* SAO::calcSaoStatsCTU, every CTU of one plane of a picture (sao.cpp:729-905; chroma: the plane's own sizes + planeOffset 2); out: per CTU [2][5][32] int32 (offsetOrg, count; EO_0..3, BO) */
void xo_sao_stats_frame(const xo_pixel* fenc, const xo_pixel* recon, intptr_t stride, int picWidth, int picHeight, int ctuSize, int nonDeblocked, int planeOffset, int32_t* out);
void xo_sao_stats_frame_slices(const xo_pixel* fenc, const xo_pixel* recon, intptr_t stride, int picWidth, int picHeight, int ctuSize, int nonDeblocked, int planeOffset, int32_t* out,
                               const uint8_t* sliceFirstRow);
void xo_sao_stats_rows(const xo_pixel* fenc, const xo_pixel* recon, intptr_t stride, int picWidth, int picHeight, int ctuSize, int nonDeblocked, int planeOffset, int32_t* out,
                       const uint8_t* sliceFirstRow, int ctuRow0, int ctuRow1);   /* the CTUs of those rows only */
void xo_sao_stats_rows_wh(const xo_pixel* fenc, const xo_pixel* recon, intptr_t stride, int picWidth, int picHeight, int ctuW, int ctuH, int nonDeblocked, int planeOffset, int32_t* out,
                          const uint8_t* sliceFirstRow, int ctuRow0, int ctuRow1);   /* CTU width and height apart (4:2:2 chroma planes) */
void xo_sao_stats_frame_predeblock(const xo_pixel* fenc, const xo_pixel* recon, intptr_t stride, int picWidth, int picHeight, int ctuSize, int planeOffset, int32_t* out);

/* SAO of a luma plane, out of place (sao.cpp:268-623); params: per CTU { typeIdx, bandPos, offset[4] } */
void xo_sao_apply_frame(const xo_pixel* in, xo_pixel* out, intptr_t stride, int picWidth, int picHeight, int ctuSize, const int32_t* params);
uint64_t xo_plane_ssd(const xo_pixel* fenc, const xo_pixel* rec, intptr_t stride, int width, int height);   /* encoder.cpp:1203-1270 computeSSD */
/* deblocking of a 4:2:0 picture, common/deblock.cpp:37-497 + common/loopfilter.cpp:136-232.  The per-partition arrays are CUData's (CTU after CTU,
 * z-scan order inside a CTU): m_log2CUSize, m_partSize, m_tuDepth, m_predMode, m_cbf[0], m_tqBypass, m_qp, m_refIdx[0..1], m_mv[0..1] (int32 x, y);
 * refPic[list][refIdx] identifies the picture (what the reference compares as Frame pointers). */
typedef struct xo_deblock_pic
{
    int width, height, ctuSize, sliceIsP, betaOffsetDiv2, tcOffsetDiv2, cbQpOffset, crQpOffset, tqBypassEnabled;
    const uint8_t *log2CUSize, *partSize, *tuDepth, *predMode, *cbfLuma, *tqBypass;
    const int8_t *qp, *refIdx0, *refIdx1;
    const int32_t *mv0, *mv1;
    int32_t refPic[2][16];
    const uint8_t* sliceFirstRow;     /* --slices: per CTU row (+ one 0 entry), non-zero where a slice begins; NULL = one slice */
    int chromaFormat;                 /* X265_CSP_*: 0 or 1 = 4:2:0, 2 = 4:2:2, 3 = 4:4:4 (the chroma planes' subsampling; deblock.cpp:104-113, 417-497) */
} xo_deblock_pic;
int xo_deblock_bs(const xo_deblock_pic* d, int ux, int uy, int dir);
void xo_deblock_frame(const xo_deblock_pic* d, xo_pixel* Y, intptr_t strideY, xo_pixel* Cb, xo_pixel* Cr, intptr_t strideC, uint8_t* bsOut);
void xo_deblock_rows(const xo_deblock_pic* d, xo_pixel* Y, intptr_t strideY, xo_pixel* Cb, xo_pixel* Cr, intptr_t strideC, uint8_t* bsOut, int ctuRow0, int ctuRow1);   /* a band of CTU rows (the rows above already deblocked) */
/* framefilter.cpp:704-722, 839-865 + pixel.cpp:623-693: per CTU row float sums and window counts, frame total in double */
void xo_ssim_frame(const xo_pixel* rec, intptr_t stride1, const xo_pixel* fenc, intptr_t stride2, int width, int height, int ctuSize,
                   float* rowSsim, uint32_t* rowCnt, double* total, uint32_t* cnt);

/* ---- interpolation family (ipfilter.cpp:40-369); taps = 8 (luma) or 4 (chroma) ---- */
void xo_interp_hpp(int taps, int w, int h, const xo_pixel* src, intptr_t ss, xo_pixel* dst, intptr_t ds, int coeffIdx);
void xo_interp_hps(int taps, int w, int h, const xo_pixel* src, intptr_t ss, int16_t* dst, intptr_t ds, int coeffIdx, int isRowExt);
void xo_interp_vpp(int taps, int w, int h, const xo_pixel* src, intptr_t ss, xo_pixel* dst, intptr_t ds, int coeffIdx);
void xo_interp_vps(int taps, int w, int h, const xo_pixel* src, intptr_t ss, int16_t* dst, intptr_t ds, int coeffIdx);
void xo_interp_vsp(int taps, int w, int h, const int16_t* src, intptr_t ss, xo_pixel* dst, intptr_t ds, int coeffIdx);
void xo_interp_vss(int taps, int w, int h, const int16_t* src, intptr_t ss, int16_t* dst, intptr_t ds, int coeffIdx);
void xo_interp_hvpp(int taps, int w, int h, const xo_pixel* src, intptr_t ss, xo_pixel* dst, intptr_t ds, int idxX, int idxY);
void xo_p2s(int w, int h, const xo_pixel* src, intptr_t ss, int16_t* dst, intptr_t ds);

/* ---- intra family (intrapred.cpp:31-234) ---- */
void xo_intra_filter(int size, const xo_pixel* samples, xo_pixel* filtered);
void xo_intra_pred(int size, xo_pixel* dst, intptr_t ds, const xo_pixel* srcPix, int dirMode, int bFilter);
void xo_intra_allangs(int size, xo_pixel* dst, const xo_pixel* refPix, const xo_pixel* filtPix, int bLuma);

#ifdef __cplusplus
}
#endif
#endif
