/*
 * ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Our own driver (no reference code in here) that is compiled TOGETHER WITH the
 * reference's primitive sources where they lie under /root/reference
 * (common/pixel.cpp, dct.cpp, ipfilter.cpp, intrapred.cpp, constants.cpp, lowpassdct.cpp,
 * primitives.cpp, common.cpp, yuv.cpp, encoder/bitcost.cpp, encoder/motion.cpp) into
 * oracle/_ref/x265ref_{8,10}.  It fills a real
 * EncoderPrimitives table through the reference's own setup*Primitives_c()
 * functions (primitives.cpp:56-75 lists them) and executes slot calls requested
 * over stdin/stdout, so the Python tests can compare oracle/x265_oracle.c (and
 * generate tests/golden) against the REAL reference arithmetic.
 *
 * Wire format (little endian), request:
 *   u32 oplen, op bytes, u32 n_ints, i64 ints[], u32 n_bufs, { u64 len, bytes }[]
 * response:
 *   u32 n_bufs, { u64 len, bytes }[]          (n_bufs = 0xFFFFFFFF on unknown op)
 */
#include "common.h"
#include "primitives.h"
#include "motion.h"
#include "lowres.h"
#include "rdcost.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>

namespace X265_NS {
void setupPixelPrimitives_c(EncoderPrimitives& p);
void setupDCTPrimitives_c(EncoderPrimitives& p);
void setupFilterPrimitives_c(EncoderPrimitives& p);
void setupIntraPrimitives_c(EncoderPrimitives& p);
void setupLowPassPrimitives_c(EncoderPrimitives& p);
void setupSeaIntegralPrimitives_c(EncoderPrimitives& p);   /* framefilter.cpp:142-157 */
void setupSaoPrimitives_c(EncoderPrimitives& p);           /* sao.cpp:1939-1950 */
extern const int16_t g_t4[4][4];
extern const int16_t g_t8[8][8];
extern const int16_t g_t16[16][16];
extern const int16_t g_t32[32][32];
}
using namespace X265_NS;

static EncoderPrimitives& T = X265_NS::primitives;   /* the reference's own global table (primitives.cpp:54) */

/* exposes the protected lambda-scaled MVD cost row of BitCost (bitcost.h:54-67) */
struct MEx : public MotionEstimate
{
    const uint16_t* costRow() const { return m_cost; }
};
static MEx* g_me;

typedef std::vector<uint8_t> Buf;
struct Req { std::string op; std::vector<int64_t> I; std::vector<Buf> B; };

static bool rd(void* p, size_t n) { return fread(p, 1, n, stdin) == n; }
static void wr(const void* p, size_t n) { fwrite(p, 1, n, stdout); }

static bool readReq(Req& r)
{
    uint32_t n;
    if (!rd(&n, 4)) return false;
    r.op.resize(n); if (n && !rd(&r.op[0], n)) return false;
    if (!rd(&n, 4)) return false;
    r.I.resize(n); if (n && !rd(r.I.data(), 8 * (size_t)n)) return false;
    if (!rd(&n, 4)) return false;
    r.B.resize(n);
    for (uint32_t i = 0; i < n; i++)
    {
        uint64_t l; if (!rd(&l, 8)) return false;
        r.B[i].resize(l); if (l && !rd(r.B[i].data(), l)) return false;
    }
    return true;
}
static void writeResp(const std::vector<Buf>& out)
{
    uint32_t n = (uint32_t)out.size(); wr(&n, 4);
    for (auto& b : out) { uint64_t l = b.size(); wr(&l, 8); if (l) wr(b.data(), l); }
    fflush(stdout);
}
template<class V> static Buf scalar(V v) { Buf b(sizeof(V)); memcpy(b.data(), &v, sizeof(V)); return b; }

/* map (w,h) to LumaPU without primitives.cpp's table */
static int puIndex(int w, int h)
{
    static const int dims[25][2] = { {4,4},{8,8},{16,16},{32,32},{64,64},{8,4},{4,8},{16,8},{8,16},{32,16},{16,32},{64,32},{32,64},
        {16,12},{12,16},{16,4},{4,16},{32,24},{24,32},{32,8},{8,32},{64,48},{48,64},{64,16},{16,64} };
    for (int i = 0; i < 25; i++) if (dims[i][0] == w && dims[i][1] == h) return i;
    return -1;
}
static int cuIndex(int n) { int i = 0; while ((4 << i) < n) i++; return i; }

#define PX(b, off) (reinterpret_cast<pixel*>((b).data()) + (off))
#define S16(b, off) (reinterpret_cast<int16_t*>((b).data()) + (off))
#define I32(b) (reinterpret_cast<int32_t*>((b).data()))

static Buf g_benchPlanes[2];
static bool dispatch(Req& r, std::vector<Buf>& out)
{
    const std::string& op = r.op;
    auto& I = r.I; auto& B = r.B;
    if (op == "info")
    {
        out.push_back(scalar<int32_t>(X265_DEPTH));
        out.push_back(scalar<int32_t>((int32_t)sizeof(EncoderPrimitives)));
        out.push_back(scalar<int32_t>((int32_t)sizeof(sse_t)));
        return true;
    }
    if (op == "layout")
    {   /* byte offsets of the slots the HIP back end fills; checked against include/x265hip.h */
        EncoderPrimitives* z = 0; (void)z;
        std::vector<int32_t> v;
#define OFF(m) v.push_back((int32_t)offsetof(EncoderPrimitives, m))
        OFF(pu); OFF(pu[1]); OFF(pu[0].sad); OFF(pu[0].sad_x3); OFF(pu[0].sad_x4); OFF(pu[0].ads); OFF(pu[0].satd);
        OFF(pu[0].luma_hpp); OFF(pu[0].luma_hps); OFF(pu[0].luma_vpp); OFF(pu[0].luma_vps); OFF(pu[0].luma_vsp); OFF(pu[0].luma_vss);
        OFF(pu[0].luma_hvpp); OFF(pu[0].pixelavg_pp); OFF(pu[0].addAvg); OFF(pu[0].copy_pp); OFF(pu[0].convert_p2s);
        OFF(cu); OFF(cu[1]); OFF(cu[0].dct); OFF(cu[0].idct); OFF(cu[0].standard_dct); OFF(cu[0].lowpass_dct); OFF(cu[0].calcresidual);
        OFF(cu[0].sub_ps); OFF(cu[0].add_ps); OFF(cu[0].blockfill_s); OFF(cu[0].copy_cnt); OFF(cu[0].count_nonzero);
        OFF(cu[0].cpy2Dto1D_shl); OFF(cu[0].cpy2Dto1D_shr); OFF(cu[0].cpy1Dto2D_shl); OFF(cu[0].cpy1Dto2D_shr);
        OFF(cu[0].copy_sp); OFF(cu[0].copy_ps); OFF(cu[0].copy_ss); OFF(cu[0].copy_pp); OFF(cu[0].var);
        OFF(cu[0].sse_pp); OFF(cu[0].sse_ss); OFF(cu[0].psy_cost_pp); OFF(cu[0].ssd_s); OFF(cu[0].sa8d); OFF(cu[0].transpose);
        OFF(cu[0].intra_pred_allangs); OFF(cu[0].intra_filter); OFF(cu[0].intra_pred);
        OFF(dst4x4); OFF(idst4x4); OFF(quant); OFF(nquant); OFF(dequant_scaling); OFF(dequant_normal); OFF(denoiseDct);
        OFF(scale1D_128to64); OFF(scale2D_64to32); OFF(weight_sp); OFF(weight_pp);
        OFF(chroma); OFF(chroma[1]); OFF(chroma[0].pu[1]); OFF(chroma[0].pu[0].satd); OFF(chroma[0].pu[0].filter_vpp);
        OFF(chroma[0].pu[0].filter_vps); OFF(chroma[0].pu[0].filter_vsp); OFF(chroma[0].pu[0].filter_vss); OFF(chroma[0].pu[0].filter_hpp);
        OFF(chroma[0].pu[0].filter_hps); OFF(chroma[0].pu[0].addAvg); OFF(chroma[0].pu[0].copy_pp); OFF(chroma[0].pu[0].p2s);
        OFF(chroma[0].cu); OFF(chroma[0].cu[1]); OFF(chroma[0].cu[0].sa8d); OFF(chroma[0].cu[0].sse_pp); OFF(chroma[0].cu[0].sub_ps);
        OFF(chroma[0].cu[0].add_ps); OFF(chroma[0].cu[0].copy_ps); OFF(chroma[0].cu[0].copy_sp); OFF(chroma[0].cu[0].copy_ss); OFF(chroma[0].cu[0].copy_pp);
        OFF(extendRowBorder); OFF(frameInitLowres); OFF(frameInitLowerRes);
        OFF(propagateCost); OFF(fix8Unpack); OFF(fix8Pack); OFF(integral_initv); OFF(integral_inith);
        OFF(saoCuStatsBO); OFF(saoCuStatsE0); OFF(saoCuStatsE1); OFF(saoCuStatsE2); OFF(saoCuStatsE3);
#undef OFF
        v.push_back((int32_t)sizeof(EncoderPrimitives));
        Buf b(v.size() * 4); memcpy(b.data(), v.data(), b.size()); out.push_back(b);
        return true;
    }
    if (op == "dct_matrix")
    {
        int n = (int)I[0];
        const int16_t* m = n == 4 ? &g_t4[0][0] : n == 8 ? &g_t8[0][0] : n == 16 ? &g_t16[0][0] : &g_t32[0][0];
        Buf b(n * n * 2); memcpy(b.data(), m, b.size()); out.push_back(b);
        return true;
    }
    /* ---- pixel compare: ints = w,h,strideA,strideB,offA,offB ; bufs = A,B ---- */
    if (op == "sad" || op == "satd")
    {
        int pu = puIndex((int)I[0], (int)I[1]);
        pixelcmp_t f = op == "sad" ? T.pu[pu].sad : T.pu[pu].satd;
        out.push_back(scalar<int32_t>(f(PX(B[0], I[4]), I[2], PX(B[1], I[5]), I[3])));
        return true;
    }
    if (op == "sa8d" || op == "psy_cost_pp")
    {
        int cu = cuIndex((int)I[0]);
        pixelcmp_t f = op == "sa8d" ? T.cu[cu].sa8d : T.cu[cu].psy_cost_pp;
        out.push_back(scalar<int32_t>(f(PX(B[0], I[4]), I[2], PX(B[1], I[5]), I[3])));
        return true;
    }
    if (op == "sse_pp")
    {
        out.push_back(scalar<uint64_t>((uint64_t)T.cu[cuIndex((int)I[0])].sse_pp(PX(B[0], I[4]), I[2], PX(B[1], I[5]), I[3])));
        return true;
    }
    if (op == "sse_ss")
    {
        out.push_back(scalar<uint64_t>((uint64_t)T.cu[cuIndex((int)I[0])].sse_ss(S16(B[0], I[4]), I[2], S16(B[1], I[5]), I[3])));
        return true;
    }
    if (op == "ssd_s")
    {   /* ints = size, stride, off */
        out.push_back(scalar<uint64_t>((uint64_t)T.cu[cuIndex((int)I[0])].ssd_s[0](S16(B[0], I[2]), I[1])));
        return true;
    }
    if (op == "sad_x3" || op == "sad_x4")
    {   /* ints = w,h,refStride,offFenc,off0,off1,off2[,off3]; bufs = fenc(stride 64), ref */
        int pu = puIndex((int)I[0], (int)I[1]);
        int32_t res[4] = { 0, 0, 0, 0 };
        if (op == "sad_x3") T.pu[pu].sad_x3(PX(B[0], I[3]), PX(B[1], I[4]), PX(B[1], I[5]), PX(B[1], I[6]), I[2], res);
        else T.pu[pu].sad_x4(PX(B[0], I[3]), PX(B[1], I[4]), PX(B[1], I[5]), PX(B[1], I[6]), PX(B[1], I[7]), I[2], res);
        Buf b(16); memcpy(b.data(), res, 16); out.push_back(b);
        return true;
    }
    /* ---- block ops; output buffers are passed in pre-filled (so untouched bytes are checked too) ---- */
    if (op == "calcresidual")
    {   /* ints = size, stride ; bufs = fenc, pred, resi(out) */
        T.cu[cuIndex((int)I[0])].calcresidual[0](PX(B[0], 0), PX(B[1], 0), S16(B[2], 0), I[1]);
        out.push_back(B[2]); return true;
    }
    if (op == "sub_ps")
    {   /* ints = size, ds, ss0, ss1 ; bufs = dst(out), s0, s1 */
        T.cu[cuIndex((int)I[0])].sub_ps(S16(B[0], 0), I[1], PX(B[1], 0), PX(B[2], 0), I[2], I[3]);
        out.push_back(B[0]); return true;
    }
    if (op == "add_ps")
    {   /* ints = size, ds, ss0, ss1 ; bufs = dst(out), s0(pixel), s1(int16) */
        T.cu[cuIndex((int)I[0])].add_ps[0](PX(B[0], 0), I[1], PX(B[1], 0), S16(B[2], 0), I[2], I[3]);
        out.push_back(B[0]); return true;
    }
    if (op == "copy_pp")
    {   /* ints = w,h,ds,ss */
        T.pu[puIndex((int)I[0], (int)I[1])].copy_pp(PX(B[0], 0), I[2], PX(B[1], 0), I[3]);
        out.push_back(B[0]); return true;
    }
    if (op == "copy_ss") { T.cu[cuIndex((int)I[0])].copy_ss(S16(B[0], 0), I[2], S16(B[1], 0), I[3]); out.push_back(B[0]); return true; }
    if (op == "copy_sp") { T.cu[cuIndex((int)I[0])].copy_sp(PX(B[0], 0), I[2], S16(B[1], 0), I[3]); out.push_back(B[0]); return true; }
    if (op == "copy_ps") { T.cu[cuIndex((int)I[0])].copy_ps(S16(B[0], 0), I[2], PX(B[1], 0), I[3]); out.push_back(B[0]); return true; }
    if (op == "blockfill_s") { T.cu[cuIndex((int)I[0])].blockfill_s[0](S16(B[0], 0), I[1], (int16_t)I[2]); out.push_back(B[0]); return true; }
    if (op == "cpy2Dto1D_shl") { T.cu[cuIndex((int)I[0])].cpy2Dto1D_shl(S16(B[0], 0), S16(B[1], 0), I[1], (int)I[2]); out.push_back(B[0]); return true; }
    if (op == "cpy2Dto1D_shr") { T.cu[cuIndex((int)I[0])].cpy2Dto1D_shr(S16(B[0], 0), S16(B[1], 0), I[1], (int)I[2]); out.push_back(B[0]); return true; }
    if (op == "cpy1Dto2D_shl") { T.cu[cuIndex((int)I[0])].cpy1Dto2D_shl[0](S16(B[0], 0), S16(B[1], 0), I[1], (int)I[2]); out.push_back(B[0]); return true; }
    if (op == "cpy1Dto2D_shr") { T.cu[cuIndex((int)I[0])].cpy1Dto2D_shr(S16(B[0], 0), S16(B[1], 0), I[1], (int)I[2]); out.push_back(B[0]); return true; }
    if (op == "transpose") { T.cu[cuIndex((int)I[0])].transpose(PX(B[0], 0), PX(B[1], 0), I[1]); out.push_back(B[0]); return true; }
    if (op == "addAvg")
    {   /* ints = w,h,ss0,ss1,ds ; bufs = s0,s1,dst(out) */
        T.pu[puIndex((int)I[0], (int)I[1])].addAvg[0](S16(B[0], 0), S16(B[1], 0), PX(B[2], 0), I[2], I[3], I[4]);
        out.push_back(B[2]); return true;
    }
    if (op == "pixelavg_pp")
    {   /* ints = w,h,ds,ss0,ss1 ; bufs = dst(out), s0, s1 */
        T.pu[puIndex((int)I[0], (int)I[1])].pixelavg_pp[0](PX(B[0], 0), I[2], PX(B[1], 0), I[3], PX(B[2], 0), I[4], 32);
        out.push_back(B[0]); return true;
    }
    if (op == "weight_sp")
    {   /* ints = ss, ds, w, h, w0, round, shift, offset ; bufs = src, dst(out) */
        T.weight_sp(S16(B[0], 0), PX(B[1], 0), I[0], I[1], (int)I[2], (int)I[3], (int)I[4], (int)I[5], (int)I[6], (int)I[7]);
        out.push_back(B[1]); return true;
    }
    if (op == "weight_pp")
    {   /* ints = stride, w, h, w0, round, shift, offset ; bufs = src, dst(out) */
        T.weight_pp(PX(B[0], 0), PX(B[1], 0), I[0], (int)I[1], (int)I[2], (int)I[3], (int)I[4], (int)I[5], (int)I[6]);
        out.push_back(B[1]); return true;
    }
    if (op == "scale1D_128to64") { T.scale1D_128to64[0](PX(B[0], 0), PX(B[1], 0)); out.push_back(B[0]); return true; }
    if (op == "scale2D_64to32") { T.scale2D_64to32(PX(B[0], 0), PX(B[1], 0), I[0]); out.push_back(B[0]); return true; }
    /* ---- transforms ---- */
    if (op == "dct")
    {   /* ints = n, srcStride ; bufs = src ; dst dense */
        int n = (int)I[0]; Buf d(n * n * 2 + 64); int16_t* dp = (int16_t*)(((uintptr_t)d.data() + 31) & ~(uintptr_t)31);
        T.cu[cuIndex(n)].dct(S16(B[0], 0), dp, I[1]);
        Buf o(n * n * 2); memcpy(o.data(), dp, o.size()); out.push_back(o); return true;
    }
    if (op == "intra_costs")
    {   /* ints = size, stride, off ; bufs = src plane, nbRef (4*size+1), nbFilt -> int32 costs[35].
         * The reference's own primitives in the order of Search::estIntraPredQT (search.cpp:1655-1745, allangs branch). */
        int tuSize = (int)I[0]; intptr_t stride = I[1];
        const pixel* fenc = PX(B[0], 0) + I[2];
        static pixel fencScaled[32 * 32], fencT[32 * 32], predBuf[33 * 32 * 32], nb[2][258];
        memcpy(nb[0], B[1].data(), (4 * tuSize + 1) * sizeof(pixel)); memcpy(nb[1], B[2].data(), (4 * tuSize + 1) * sizeof(pixel));
        int scaleTuSize = tuSize, scaleStride = (int)stride, costShift = 0, sizeIdx = cuIndex(tuSize);
        if (tuSize > 32)
        {
            T.scale2D_64to32(fencScaled, fenc, stride); fenc = fencScaled;
            pixel nScale[129];
            nb[1][0] = nb[0][0];
            T.scale1D_128to64[0](nScale + 1, nb[0] + 1);
            memcpy(&nb[0][1], &nScale[1], 2 * 64 * sizeof(pixel)); memcpy(&nb[1][1], &nScale[1], 2 * 64 * sizeof(pixel));
            scaleTuSize = 32; scaleStride = 32; costShift = 2; sizeIdx = 3;
        }
        pixelcmp_t sa8d = T.cu[sizeIdx].sa8d;
        int predsize = scaleTuSize * scaleTuSize;
        std::vector<int32_t> c(35);
        /* (in search.cpp fenc is a compact CU buffer, so its stride doubles as the prediction stride; here fenc lives in a plane) */
        T.cu[sizeIdx].intra_pred[DC_IDX](predBuf, scaleTuSize, nb[0], 0, (scaleTuSize <= 16));
        c[DC_IDX] = sa8d(fenc, scaleStride, predBuf, scaleTuSize) << costShift;
        pixel* planar = nb[0];
        if (tuSize & (8 | 16 | 32)) planar = nb[1];
        T.cu[sizeIdx].intra_pred[PLANAR_IDX](predBuf, scaleTuSize, planar, 0, 0);
        c[PLANAR_IDX] = sa8d(fenc, scaleStride, predBuf, scaleTuSize) << costShift;
        T.cu[sizeIdx].transpose(fencT, fenc, scaleStride);
        /* the C table leaves intra_pred_allangs NULL (primitives.cpp:348); all_angs_pred_c is what the asm slot computes */
        for (int mode = 2; mode < 35; mode++)
        {
            int filter = !!(g_intraFilterFlags[mode] & scaleTuSize);
            T.cu[sizeIdx].intra_pred[mode](predBuf, scaleTuSize, nb[filter], mode, scaleTuSize <= 16);
            c[mode] = sa8d(fenc, scaleStride, predBuf, scaleTuSize) << costShift;     /* the !allangs arm of TRY_ANGLE */
        }
        (void)fencT; (void)predsize;
        Buf o(35 * 4); memcpy(o.data(), c.data(), o.size()); out.push_back(o); return true;
    }
    if (op == "frame_init_lowres")
    {   /* ints = srcStride, dstStride, width, height (lowres size) ; bufs = src, dst0, dsth, dstv, dstc (outs pre-filled) */
        T.frameInitLowres(PX(B[0], 0), PX(B[1], 0), PX(B[2], 0), PX(B[3], 0), PX(B[4], 0), I[0], I[1], (int)I[2], (int)I[3]);
        out.push_back(B[1]); out.push_back(B[2]); out.push_back(B[3]); out.push_back(B[4]); return true;
    }
    if (op == "extend_pic_border")
    {   /* ints = stride, width, height, marginX, marginY ; bufs = padded plane (in/out), picture origin at (marginX, marginY) */
        pixel* pic = PX(B[0], 0) + I[4] * I[0] + I[3];
        extendPicBorder(pic, I[0], (int)I[1], (int)I[2], (int)I[3], (int)I[4]);          /* pixel.cpp:1044-1058 */
        out.push_back(B[0]); return true;
    }
    if (op == "extend_row_border")
    {   /* the table slot alone (ipfilter.cpp:59-77): ints = stride, width, height, marginX ; bufs = rows (in/out), origin at x = marginX */
        T.extendRowBorder(PX(B[0], 0) + I[3], I[0], (int)I[1], (int)I[2], (int)I[3]);
        out.push_back(B[0]); return true;
    }
    if (op == "lowpass_dct")
    {   /* ints = n, srcStride ; bufs = src ; dst dense (cu[].lowpass_dct, lowpassdct.cpp) */
        int n = (int)I[0]; Buf d(n * n * 2 + 64); int16_t* dp = (int16_t*)(((uintptr_t)d.data() + 31) & ~(uintptr_t)31);
        T.cu[cuIndex(n)].lowpass_dct(S16(B[0], 0), dp, I[1]);
        Buf o(n * n * 2); memcpy(o.data(), dp, o.size()); out.push_back(o); return true;
    }
    if (op == "ads")
    {   /* ints = w, h, delta, width, thresh ; bufs = encDC (int32[4]), sums (uint32[]), costMvX (uint16[]) -> nmv, mvs */
        int width = (int)I[3];
        Buf mv((size_t)(width > 0 ? width : 1) * 2);
        int n = T.pu[puIndex((int)I[0], (int)I[1])].ads((int*)B[0].data(), (uint32_t*)B[1].data(), (int)I[2], (uint16_t*)B[2].data(),
                                                         (int16_t*)mv.data(), width, (int)I[4]);
        out.push_back(scalar<int32_t>(n)); out.push_back(mv); return true;
    }
    if (op == "dst4")
    {
        Buf d(32 + 64); int16_t* dp = (int16_t*)(((uintptr_t)d.data() + 31) & ~(uintptr_t)31);
        T.dst4x4(S16(B[0], 0), dp, I[0]);
        Buf o(32); memcpy(o.data(), dp, 32); out.push_back(o); return true;
    }
    if (op == "idct")
    {   /* ints = n, dstStride ; bufs = src dense, dst(out, pre-filled) */
        T.cu[cuIndex((int)I[0])].idct(S16(B[0], 0), S16(B[1], 0), I[1]);
        out.push_back(B[1]); return true;
    }
    if (op == "idst4") { T.idst4x4(S16(B[0], 0), S16(B[1], 0), I[0]); out.push_back(B[1]); return true; }
    if (op == "quant")
    {   /* ints = qBits, add, numCoeff ; bufs = coef, quantCoeff */
        int n = (int)I[2]; Buf du(n * 4), q(n * 2);
        uint32_t ns = T.quant(S16(B[0], 0), I32(B[1]), (int32_t*)du.data(), (int16_t*)q.data(), (int)I[0], (int)I[1], n);
        out.push_back(scalar<uint32_t>(ns)); out.push_back(q); out.push_back(du); return true;
    }
    if (op == "nquant")
    {
        int n = (int)I[2]; Buf q(n * 2);
        uint32_t ns = T.nquant(S16(B[0], 0), I32(B[1]), (int16_t*)q.data(), (int)I[0], (int)I[1], n);
        out.push_back(scalar<uint32_t>(ns)); out.push_back(q); return true;
    }
    if (op == "dequant_normal")
    {   /* ints = num, scale, shift ; bufs = q */
        int n = (int)I[0]; Buf c(n * 2);
        T.dequant_normal(S16(B[0], 0), (int16_t*)c.data(), n, (int)I[1], (int)I[2]);
        out.push_back(c); return true;
    }
    if (op == "dequant_scaling")
    {   /* ints = num, per, shift ; bufs = q, deq */
        int n = (int)I[0]; Buf c(n * 2);
        T.dequant_scaling(S16(B[0], 0), I32(B[1]), (int16_t*)c.data(), n, (int)I[1], (int)I[2]);
        out.push_back(c); return true;
    }
    if (op == "count_nonzero") { out.push_back(scalar<int32_t>(T.cu[cuIndex((int)I[0])].count_nonzero(S16(B[0], 0)))); return true; }
    if (op == "copy_cnt")
    {   /* ints = n, resiStride ; bufs = resi */
        int n = (int)I[0]; Buf c(n * n * 2);
        uint32_t ns = T.cu[cuIndex(n)].copy_cnt((int16_t*)c.data(), S16(B[0], 0), I[1]);
        out.push_back(scalar<uint32_t>(ns)); out.push_back(c); return true;
    }
    if (op == "denoise_dct")
    {   /* ints = num ; bufs = coef(inout), resSum(inout u32), offset(u16) */
        T.denoiseDct(S16(B[0], 0), (uint32_t*)B[1].data(), (const uint16_t*)B[2].data(), (int)I[0]);
        out.push_back(B[0]); out.push_back(B[1]); return true;
    }
    /* ---- interpolation: ints = taps,w,h,ss,ds,srcOff,idx[,idx2/isRowExt] ; bufs = src, dst(out, pre-filled) ---- */
    if (op.compare(0, 7, "interp_") == 0 || op == "p2s")
    {
        int taps = (int)I[0], w = (int)I[1], h = (int)I[2];
        intptr_t ss = I[3], ds = I[4], so = I[5]; int idx = (int)I[6]; int idx2 = I.size() > 7 ? (int)I[7] : 0;
        bool luma = taps == 8;
        /* chroma tables are indexed by the LUMA partition enum; 4:2:0 chroma block is (w,h) = luma/2 */
        int pu = luma ? puIndex(w, h) : puIndex(w * 2, h * 2);
        EncoderPrimitives::PU& L = T.pu[pu];
        EncoderPrimitives::Chroma::PUChroma& C = T.chroma[X265_CSP_I420].pu[pu];
        if (op == "interp_hpp") (luma ? L.luma_hpp : C.filter_hpp)(PX(B[0], so), ss, PX(B[1], 0), ds, idx);
        else if (op == "interp_hps") (luma ? L.luma_hps : C.filter_hps)(PX(B[0], so), ss, S16(B[1], 0), ds, idx, idx2);
        else if (op == "interp_vpp") (luma ? L.luma_vpp : C.filter_vpp)(PX(B[0], so), ss, PX(B[1], 0), ds, idx);
        else if (op == "interp_vps") (luma ? L.luma_vps : C.filter_vps)(PX(B[0], so), ss, S16(B[1], 0), ds, idx);
        else if (op == "interp_vsp") (luma ? L.luma_vsp : C.filter_vsp)(S16(B[0], so), ss, PX(B[1], 0), ds, idx);
        else if (op == "interp_vss") (luma ? L.luma_vss : C.filter_vss)(S16(B[0], so), ss, S16(B[1], 0), ds, idx);
        else if (op == "interp_hvpp") L.luma_hvpp(PX(B[0], so), ss, PX(B[1], 0), ds, idx, idx2);
        else if (op == "p2s") (luma ? L.convert_p2s[0] : C.p2s[0])(PX(B[0], so), ss, S16(B[1], 0), ds);
        else return false;
        out.push_back(B[1]); return true;
    }
    /* ---- intra ---- */
    if (op == "intra_filter")
    {   /* ints = size ; bufs = samples, filtered(out, pre-filled) */
        T.cu[cuIndex((int)I[0])].intra_filter(PX(B[0], 0), PX(B[1], 0)); out.push_back(B[1]); return true;
    }
    if (op == "intra_pred")
    {   /* ints = size, ds, mode, bFilter ; bufs = srcPix, dst(out) */
        T.cu[cuIndex((int)I[0])].intra_pred[I[2]](PX(B[1], 0), I[1], PX(B[0], 0), (int)I[2], (int)I[3]); out.push_back(B[1]); return true;
    }
    if (op == "intra_allangs")
    {   /* ints = size, bLuma ; bufs = ref, filt */
        int n = (int)I[0]; Buf d(33 * n * n * sizeof(pixel));
        T.cu[cuIndex(n)].intra_pred_allangs((pixel*)d.data(), PX(B[0], 0), PX(B[1], 0), (int)I[1]); out.push_back(d); return true;
    }
    /* ---- timing helper for bench.py cpu_baseline(kind="reference"): ints = family, reps, w/h... ---- */
    if (op == "lambda_tab")
    {
        Buf b((QP_MAX_MAX + 1) * 8); memcpy(b.data(), x265_lambda_tab, b.size()); out.push_back(b); return true;
    }
    if (op == "sao_stats")
    {   /* ints = type (0..3 = E0..E3, 4 = BO), stride, recOff, endX, endY, upOff ; bufs = diff (int16, stride 64), rec plane, upBuff1, upBufft,
           stats (int32), count (int32).  returns stats, count, upBuff1, upBufft after the reference's primitive (sao.cpp:1774-1937). */
        const int type = (int)I[0]; const intptr_t stride = I[1]; const int endX = (int)I[3], endY = (int)I[4];
        Buf st = B[4], ct = B[5], u1 = B[2], ut = B[3];
        const int16_t* diff = (const int16_t*)B[0].data(); const pixel* rec = PX(B[1], I[2]);
        int8_t* up1 = (int8_t*)u1.data() + I[5]; int8_t* upt = (int8_t*)ut.data() + I[5];
        int32_t* s = (int32_t*)st.data(); int32_t* c = (int32_t*)ct.data();
        if (type == 4) T.saoCuStatsBO(diff, rec, stride, endX, endY, s, c);
        else if (type == 0) T.saoCuStatsE0(diff, rec, stride, endX, endY, s, c);
        else if (type == 1) T.saoCuStatsE1(diff, rec, stride, up1, endX, endY, s, c);
        else if (type == 2) T.saoCuStatsE2(diff, rec, stride, up1, upt, endX, endY, s, c);
        else T.saoCuStatsE3(diff, rec, stride, up1, endX, endY, s, c);
        out.push_back(st); out.push_back(ct); out.push_back(u1); out.push_back(ut); return true;
    }
    if (op == "mvcost_row")
    {   /* ints = qp, halfRange ; returns u16 cost[-halfRange..halfRange] (bitcost.cpp:30-56) */
        g_me->setQP((unsigned)I[0]);
        int hr = (int)I[1]; Buf b((2 * hr + 1) * 2);
        memcpy(b.data(), g_me->costRow() - hr, b.size()); out.push_back(b); return true;
    }
    if (op == "bits_cost")
    {   /* ints = qp, n, then n x (mv.x, mv.y, mvp.x, mvp.y, bits): returns u32[n] BitCost::bitcost(mv, mvp) (bitcost.h:66-70) and u32[n] RDCost::getCost(bits)
           for the slice-independent lambda of that qp (rdcost.h:88-92,164-169) */
        g_me->setQP((unsigned)I[0]);
        RDCost rd; rd.setLambda(x265_lambda2_tab[I[0]], x265_lambda_tab[I[0]]);
        int n = (int)I[1]; Buf a(4 * n), b(4 * n);
        for (int i = 0; i < n; i++)
        {
            ((uint32_t*)a.data())[i] = g_me->bitcost(MV((int)I[2 + 5 * i], (int)I[3 + 5 * i]), MV((int)I[4 + 5 * i], (int)I[5 + 5 * i]));
            ((uint32_t*)b.data())[i] = rd.getCost((uint32_t)I[6 + 5 * i]);
        }
        out.push_back(a); out.push_back(b); return true;
    }
    if (op == "me")
    {   /* ints = w,h, fencStride,fencOff, refStride,refOff, mvmin.x,mvmin.y,mvmax.x,mvmax.y, qmvp.x,qmvp.y,
                  merange, searchMethod, subpelRefine, qp, numCand, (mvc.x,mvc.y)* ; bufs = fenc plane, ref plane.
           Runs the reference's MotionEstimate::motionEstimate (motion.cpp:923-1773) with the luma-only
           setSourcePU (motion.cpp:203-231). */
        int w = (int)I[0], h = (int)I[1];
        ReferencePlanes rp;
        rp.fpelPlane[0] = PX(B[1], I[5]);
        rp.lumaStride = I[4];
        rp.isLowres = false; rp.isHMELowres = false;
        g_me->setQP((unsigned)I[15]);
        g_me->setSourcePU(PX(B[0], 0), I[2], I[3], w, h, (int)I[13], (int)I[14]);
        MV mvmin((int)I[6], (int)I[7]), mvmax((int)I[8], (int)I[9]), qmvp((int)I[10], (int)I[11]), outmv;
        int nc = (int)I[16]; MV mvc[16];
        for (int i = 0; i < nc && i < 16; i++) mvc[i] = MV((int)I[17 + 2 * i], (int)I[18 + 2 * i]);
        /* blockOffset is the PU offset inside the fenc plane; the reference plane pointer is pre-offset so the
           same blockOffset addresses the co-located block (lookahead calling convention, slicetype.cpp:4484) */
        rp.fpelPlane[0] = PX(B[1], I[5]) - I[3];
        std::vector<std::vector<uint32_t> > integ;
        if ((int)I[13] == X265_SEA)
        {   /* ints after the candidates: element index of pixel (0,0) in the reference buffer, CTU-aligned picture height, padX, padY.
               The 12 integral planes are built by the reference's integral_init primitives in the order FrameFilter::processPostRow
               calls them (framefilter.cpp:757-833), then handed to the search at the PU's co-located offset (search.cpp:2153-2157). */
            const int64_t org = I[17 + 2 * nc]; const int maxHeight = (int)I[18 + 2 * nc], padX = (int)I[19 + 2 * nc], padY = (int)I[20 + 2 * nc];
            const intptr_t stride = I[4];
            const size_t elems = B[1].size() / sizeof(pixel);
            static const int Wk[12] = { INTEGRAL_32, INTEGRAL_32, INTEGRAL_32, INTEGRAL_24, INTEGRAL_16, INTEGRAL_16, INTEGRAL_16, INTEGRAL_12, INTEGRAL_8, INTEGRAL_8, INTEGRAL_4, INTEGRAL_4 };
            static const int Hk[12] = { INTEGRAL_32, INTEGRAL_24, INTEGRAL_8, INTEGRAL_32, INTEGRAL_16, INTEGRAL_12, INTEGRAL_4, INTEGRAL_16, INTEGRAL_32, INTEGRAL_8, INTEGRAL_16, INTEGRAL_4 };
            static const int Hn[12] = { 32, 24, 8, 32, 16, 12, 4, 16, 32, 8, 16, 4 };
            integ.assign(12, std::vector<uint32_t>(elems + 64, 0xdeadbeefu));
            for (int k = 0; k < 12; k++)
            {
                uint32_t* Iorg = integ[k].data() + org;
                memset(Iorg - padY * stride - padX, 0, stride * sizeof(uint32_t));
                for (int y = -padY; y < maxHeight + padY - 1; y++)
                {
                    pixel* pix = PX(B[1], org) + y * stride - padX;
                    uint32_t* sum = Iorg + (y + 1) * stride - padX;
                    T.integral_inith[Wk[k]](sum, pix, stride);
                    if (y >= Hn[k] - padY) T.integral_initv[Hk[k]](sum - Hn[k] * stride, stride);
                }
                g_me->integral[k] = Iorg + (I[5] - org);             /* the plane at the PU's co-located position (search.cpp:355) */
            }
        }
        int cost = g_me->motionEstimate(&rp, mvmin, mvmax, qmvp, nc, mvc, (int)I[12], outmv, 1, false);
        int32_t r[3] = { outmv.x, outmv.y, cost };
        Buf b(12); memcpy(b.data(), r, 12); out.push_back(b); return true;
    }
    if (op == "bench_planes")
    {   /* the two planes of the bench_me / bench_tq requests that follow, read once from a file (bufs = path; ints = bytes of the cur plane, bytes of the ref plane, back to
           back in the file): bench.py starts one process per host core, and the planes do not travel through every process's pipe */
        std::string path((const char*)B[0].data(), B[0].size());
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) return false;
        g_benchPlanes[0].resize((size_t)I[0]); g_benchPlanes[1].resize((size_t)I[1]);
        const bool ok = fread(g_benchPlanes[0].data(), 1, g_benchPlanes[0].size(), f) == g_benchPlanes[0].size() && fread(g_benchPlanes[1].data(), 1, g_benchPlanes[1].size(), f) == g_benchPlanes[1].size();
        fclose(f);
        out.push_back(scalar<int32_t>(ok ? 1 : 0)); return true;
    }
    if ((op == "bench_me" || op == "bench_tq") && B.size() < 2) { B.resize(2); B[0].swap(g_benchPlanes[0]); B[1].swap(g_benchPlanes[1]); const bool r2 = dispatch(r, out); B[0].swap(g_benchPlanes[0]); B[1].swap(g_benchPlanes[1]); return r2; }
    if (op == "bench_me")
    {   /* many PUs of one size through the reference's motionEstimate, timed inside the process (no IPC in the
           timed region).  ints = w,h,stride,merange,method,subme,qp,n,reps, then n x (off, mvmin.x,mvmin.y,mvmax.x,mvmax.y,
           qmvp.x,qmvp.y); bufs = cur plane, ref plane.  returns elapsed ns, n x (mvx, mvy, cost). */
        int w = (int)I[0], h = (int)I[1]; intptr_t stride = I[2];
        int merange = (int)I[3], method = (int)I[4], subme = (int)I[5], qp = (int)I[6], n = (int)I[7], reps = (int)I[8];
        std::vector<int32_t> res(3 * (size_t)n);
        g_me->setQP((unsigned)qp);
        ReferencePlanes rp; rp.lumaStride = stride; rp.isLowres = false; rp.isHMELowres = false;
        rp.fpelPlane[0] = PX(B[1], 0);
        auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < reps; rep++)
        for (int i = 0; i < n; i++)
        {
            const int64_t* a = &I[9 + 7 * (size_t)i];
            g_me->setSourcePU(PX(B[0], 0), stride, a[0], w, h, method, subme);
            MV mvmin((int)a[1], (int)a[2]), mvmax((int)a[3], (int)a[4]), qmvp((int)a[5], (int)a[6]), outmv;
            int cost = g_me->motionEstimate(&rp, mvmin, mvmax, qmvp, 0, NULL, merange, outmv, 1, false);
            res[3 * i] = outmv.x; res[3 * i + 1] = outmv.y; res[3 * i + 2] = cost;
        }
        auto t1 = std::chrono::steady_clock::now();
        out.push_back(scalar<int64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count()));
        Buf b(res.size() * 4); memcpy(b.data(), res.data(), b.size()); out.push_back(b); return true;
    }
    if (op == "bench_tq")
    {   /* many inter TUs through the reference's primitives in the order of predict.cpp:279-300 and
           quant.cpp:397-480 (rdoq off, flat lists): copy_pp|luma_hpp|luma_vpp|luma_hvpp -> sub_ps -> dct -> quant.
           ints = log2n, stride, qp, addNumerator, n, reps, then n x (off, mvx, mvy); bufs = cur, ref.
           returns elapsed ns, numSig[n], coefficient checksum (sum of coef * (index+1) mod 2^32). */
        static const int qs[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };
        int lg = (int)I[0], N = 1 << lg; intptr_t stride = I[1]; int qp = (int)I[2], addNum = (int)I[3], n = (int)I[4], reps = (int)I[5];
        int pu = puIndex(N, N), cu = cuIndex(N);
        pixel* pred = (pixel*)aligned_alloc(64, 32 * 32 * sizeof(pixel));
        int16_t* resi = (int16_t*)aligned_alloc(64, 2048); int16_t* dctc = (int16_t*)aligned_alloc(64, 2048);
        int16_t* q = (int16_t*)aligned_alloc(64, 2048); int32_t* du = (int32_t*)aligned_alloc(64, 4096);
        int32_t* qc = (int32_t*)aligned_alloc(64, 4096);
        for (int i = 0; i < N * N; i++) qc[i] = qs[qp % 6];
        const int tshift = 15 - X265_DEPTH - lg, qbits = 14 + qp / 6 + tshift, add = addNum << (qbits - 9);
        std::vector<uint32_t> ns(n); uint32_t csum = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < reps; rep++)
        for (int i = 0; i < n; i++)
        {
            const int64_t* a = &I[6 + 3 * (size_t)i];
            const pixel* cur = PX(B[0], a[0]);
            int mvx = (int)a[1], mvy = (int)a[2];
            const pixel* src = PX(B[1], a[0]) + (mvx >> 2) + (mvy >> 2) * stride;
            int xf = mvx & 3, yf = mvy & 3;
            if (!(xf | yf)) T.pu[pu].copy_pp(pred, N, src, stride);
            else if (!yf) T.pu[pu].luma_hpp(src, stride, pred, N, xf);
            else if (!xf) T.pu[pu].luma_vpp(src, stride, pred, N, yf);
            else T.pu[pu].luma_hvpp(src, stride, pred, N, xf, yf);
            T.cu[cu].sub_ps(resi, N, cur, pred, stride, N);
            T.cu[cu].dct(resi, dctc, N);
            ns[i] = T.quant(dctc, qc, du, q, qbits, add, N * N);
            if (rep == 0) for (int k = 0; k < N * N; k++) csum += (uint32_t)(int32_t)q[k] * (uint32_t)(k + 1);
        }
        auto t1 = std::chrono::steady_clock::now();
        free(pred); free(resi); free(dctc); free(q); free(du); free(qc);
        out.push_back(scalar<int64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count()));
        Buf b(ns.size() * 4); memcpy(b.data(), ns.data(), b.size()); out.push_back(b);
        out.push_back(scalar<uint32_t>(csum)); return true;
    }
    if (op == "time_tu")
    {   /* residual -> dct32 -> quant over `I[0]` TUs of B[0] (int16 residual, dense 32x32 each); returns ns */
        int ntu = (int)I[0], qBits = (int)I[1], add = (int)I[2];
        int16_t* coef = (int16_t*)aligned_alloc(64, 2048); int16_t* q = (int16_t*)aligned_alloc(64, 2048);
        int32_t* du = (int32_t*)aligned_alloc(64, 4096); uint32_t acc = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < ntu; i++)
        {
            T.cu[3].dct(S16(B[0], (size_t)i * 1024), coef, 32);
            acc += T.quant(coef, I32(B[1]), du, q, qBits, add, 1024);
        }
        auto t1 = std::chrono::steady_clock::now();
        free(coef); free(q); free(du);
        out.push_back(scalar<int64_t>(std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count()));
        out.push_back(scalar<uint32_t>(acc)); return true;
    }
    return false;
}

int main()
{
    memset(&T, 0, sizeof(T));
    setupPixelPrimitives_c(T);
    setupDCTPrimitives_c(T);
    for (int i = 0; i < 4; i++) T.cu[i].standard_dct = T.cu[i].dct;      /* enableLowpassDCTPrimitives, primitives.cpp:77-83 (without its dct <- lowpass_dct switch) */
    setupLowPassPrimitives_c(T);
    setupFilterPrimitives_c(T);
    setupIntraPrimitives_c(T);
    setupSeaIntegralPrimitives_c(T);
    setupSaoPrimitives_c(T);
    setupAliasPrimitives(T);           /* primitives.cpp:178-284 */
    MotionEstimate::initScales();
    g_me = new MEx();
    g_me->init(X265_CSP_I400);
    Req r;
    while (readReq(r))
    {
        std::vector<Buf> out;
        if (!dispatch(r, out)) { uint32_t bad = 0xFFFFFFFFu; wr(&bad, 4); fflush(stdout); continue; }
        writeResp(out);
    }
    return 0;
}
