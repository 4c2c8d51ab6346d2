// kern_tq.hip -- fused inter-TU pipeline, one wavefront per transform unit:
//   motion compensation at the chosen quarter-pel MV   predict.cpp:279-300 (copy_pp | luma_hpp | luma_vpp | luma_hvpp)
//   residual = source - prediction                     pixel.cpp:806-818   (cu[].sub_ps)
//   forward DCT                                        dct.cpp:443-526     (32x32: matrix cores, xh_dct32.h)
//   quantisation, numSig, optional deltaU              dct.cpp:666-688     (Quant::transformNxN, quant.cpp:458-469)
//   [recon] dequant -> inverse DCT (or DC shortcut) -> pred + resid -> SSE   quant.cpp:543-605, search.cpp:5563-5575
// Source and reference pixels are read once from HBM, everything in between lives in LDS / registers; the only
// outputs are the quantised coefficients (2 B/pixel), numSig and -- on request -- the reconstruction and its SSE.
#include "xh_mc.h"
#include "xh_dct32.h"
#include "../../include/x265hip_frame.h"
using namespace xh;

namespace {

__device__ const int k_quantScales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };   // scalinglist.cpp:129
__device__ const int k_invQuantScales[6] = { 40, 45, 51, 57, 64, 72 };                  // scalinglist.cpp:130
__device__ const int8_t k_dst4m[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };   // dct.cpp:43-81 (fastForwardDst / inversedst as a matrix)

struct TqArgs
{
    const pixel* cur; intptr_t cs; const pixel* ref; intptr_t rs;
    const x265hip_tu_task* tasks; int n;
    int qp, add; const int32_t* quantCoeff; int32_t* deltaU;
    int16_t* coeff; uint32_t* numSig;
    pixel* recon; intptr_t reconStride; uint64_t* sse;
    const x265hip_me_result* mvSource;
    const pixel* planes; int64_t planeElems;
    const x265hip_inter_choice* choice; int choiceList, choiceRef;
    int chroma;
    const pixel* ref1; int choiceRef1;          // bi-directional launch: list-1 reference plane and index
    int dst4;                                   // 4x4 TUs: DST-VII instead of the DCT (intra luma, quant.cpp:429-432, 585-603)
    int tiled;                                  // 16 bit: slots 1..15 of `planes` are tiled (xh_mc.h tile_off; xh_tq_batch_tiled)
};

template<int N> struct Lg { static const int v = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5; };

// dct.cpp:666-688 for one coefficient; returns the signed clipped level, sets nz / delta
__device__ __forceinline__ int quant_one(int coef, int q, int qBits, int add, int& nz, int& delta)
{
    const int sign = coef < 0 ? -1 : 1;
    const int32_t tmplevel = (int32_t)((uint32_t)abs(coef) * (uint32_t)q);
    int level = (int32_t)((uint32_t)tmplevel + (uint32_t)add) >> qBits;
    delta = (int32_t)((uint32_t)tmplevel - ((uint32_t)level << qBits)) >> (qBits - 8);
    nz = level != 0;
    return clip3(-32768, 32767, level * sign);
}

// BI: the launch has bi-directional TUs (a second 14-bit prediction per wavefront in LDS).  Without it a workgroup of the 32x32 form holds 26 KB instead of 34.5 KB of LDS:
// six wavefronts per SIMD instead of four, for a kernel that spends three quarters of its time waiting on memory
template<int N, bool BI>
__global__ __launch_bounds__(256) void tq_kernel(TqArgs a)
{
    constexpr int NN = N * N, LG = Lg<N>::v;
    __shared__ __attribute__((aligned(16))) pixel s_pred[4][NN];
    __shared__ __attribute__((aligned(16))) int16_t s_a[4][NN];
    __shared__ __attribute__((aligned(16))) int16_t s_b[4][(N + 7) * N];
    __shared__ int8_t s_m[NN];
    __shared__ __attribute__((aligned(16))) int16_t s_c[BI ? 4 : 1][BI ? NN : 8];      // second 14-bit prediction of a bi-directional TU

    if (N < 32)
    {
        for (int i = threadIdx.x; i < NN; i += 256) s_m[i] = (N == 4 && a.dst4) ? k_dst4m[i & 15] : (int8_t)dct_coef((i / N) * (32 / N), i % N);
    }
    else if (a.recon)
    {
        for (int i = threadIdx.x; i < NN; i += 256) s_m[i] = (int8_t)dct_coef(i / N, i % N);
    }
    __syncthreads();                   // the only block-level barrier; waves are independent from here on

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = xcd_contiguous_block(blockIdx.x, gridDim.x) * 4 + wave;
    if (item >= a.n) return;
    x265hip_tu_task tk = a.tasks[item];
    if (tk.mvFrom >= 0 && a.mvSource) { tk.mv[0] = a.mvSource[tk.mvFrom].mv[0]; tk.mv[1] = a.mvSource[tk.mvFrom].mv[1]; }
    int mv1x = 0, mv1y = 0;
    if (tk.mvFrom >= 0 && a.choice)
    {   // several references: this launch compensates from ONE of them (or one PAIR, a.ref1) -- only the TUs whose PU chose it
        const x265hip_inter_choice ch = a.choice[tk.mvFrom];
        if (a.ref1)
        {
            if (ch.ref[0] != a.choiceRef || ch.ref[1] != a.choiceRef1) return;
            tk.mv[0] = ch.mv[0][0]; tk.mv[1] = ch.mv[0][1]; mv1x = ch.mv[1][0]; mv1y = ch.mv[1][1];
        }
        else
        {
            if (ch.ref[a.choiceList] != a.choiceRef || ch.ref[a.choiceList ^ 1] >= 0) return;
            tk.mv[0] = ch.mv[a.choiceList][0]; tk.mv[1] = ch.mv[a.choiceList][1];
        }
    }

    McCtx c;
    c.setGeometry(N, N, lane);
    c.fref = a.ref + tk.refOff; c.rs = a.rs;
    c.pred = (lpixel*)s_pred[wave]; c.immed = (lshort*)s_b[wave];
    lshort* sa = (lshort*)s_a[wave];
    lshort* sb = (lshort*)s_b[wave];
    const lpixel* pred = c.pred;

    // ---- motion compensation into LDS (ends with a wave_sync): the reference's copy_pp | hpp | vpp | hvpp dispatch, or -- with
    //      the phase planes of this reference -- a copy of the block at the integer part of the MV out of plane 4*yFrac + xFrac ----
    if (BI && a.ref1)
    {   // bi-directional TU: Predict::motionCompensation's B-slice branch (predict.cpp:186-196, 211): two 14-bit predictions, addAvg
        build_pred_short(c, plane_view(c), tk.mv[0], tk.mv[1], sa);
        McCtx c1 = c; c1.fref = a.ref1 + tk.refOff;
        build_pred_short(c1, plane_view(c1), mv1x, mv1y, (lshort*)s_c[wave]);
        add_avg(c, sa, (const lshort*)s_c[wave]);
    }
    else if (a.chroma)
    {   // chroma TU of a 4:2:0 picture: predInterChromaPixel (4-tap filters at the eighth-pel chroma MV = the quarter-pel luma MV)
        CCtx cc; cc.ref[0] = a.ref + tk.refOff; cc.ref[1] = cc.ref[0]; cc.rs = a.rs; cc.fenc[0] = cc.fenc[1] = nullptr; cc.pred = c.pred; cc.immed = c.immed;      // (N + 3) intermediate rows fit the luma path's (N + 7)-row buffer
        cc.w = N; cc.h = N; cc.qpr = N >> 2; cc.nquads = (N >> 2) * N; cc.lane = lane; cc.on = true;
        chroma_pred<64>(cc, 0, tk.mv[0], tk.mv[1]);
    }
    else if (a.planes)
    {
        const int f = (tk.mv[1] & 3) * 4 + (tk.mv[0] & 3);
        if (false)
        {   // the block out of the tiles of slot f: a quad that straddles a tile column pixel by pixel
            const pixel* slot = a.planes + (int64_t)f * a.planeElems;
            const uint32_t rs32 = (uint32_t)a.rs, py = (uint32_t)tk.refOff / rs32, px = (uint32_t)tk.refOff - py * rs32;
            const uint32_t X0 = px + (uint32_t)(tk.mv[0] >> 2), Y0 = py + (uint32_t)(tk.mv[1] >> 2);
            QUAD_LOOP(c, q, y, x4)
                int v[4];
                const uint32_t X = X0 + (uint32_t)x4, Y = Y0 + (uint32_t)y;
                if ((X & 15u) <= 12u) load4u(slot + tile_off(X, Y, rs32), v);
                else
                {
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = slot[tile_off(X + (uint32_t)e, Y, rs32)];
                }
                store4(c.pred + y * N + x4, v);
            QUAD_END
        }
        else
        {
        const pixel* src = (f ? a.planes + (int64_t)f * a.planeElems : a.ref) + tk.refOff + (intptr_t)(tk.mv[1] >> 2) * a.rs + (tk.mv[0] >> 2);
        QUAD_LOOP(c, q, y, x4)
            int v[4]; load4u(src + (intptr_t)y * a.rs + x4, v); store4(c.pred + y * N + x4, v);
        QUAD_END
        }
        wave_sync();
    }
    else
        build_pred(c, tk.mv[0], tk.mv[1]);

    // ---- residual (kept in LDS; the source quad is re-read from L2 for the SSE only when recon is requested) ----
    const pixel* cur = a.cur + tk.curOff;
    QUAD_LOOP(c, q, y, x4)
        int s[4], p[4]; load4u(cur + (intptr_t)y * a.cs + x4, s); load4(pred + y * N + x4, p);
        u32x2 r; r.x = (uint32_t)(uint16_t)(s[0] - p[0]) | ((uint32_t)(uint16_t)(s[1] - p[1]) << 16);
        r.y = (uint32_t)(uint16_t)(s[2] - p[2]) | ((uint32_t)(uint16_t)(s[3] - p[3]) << 16);
        *(lu2*)(sa + y * N + x4) = r;
    QUAD_END
    wave_sync();

    // ---- forward transform + quantisation ----
    const int per = a.qp / 6, rem = a.qp % 6;
    const int transformShift = 15 - X265_DEPTH - LG;
    const int qBits = 14 + per + transformShift;
    const int add = a.add << (qBits - 9);
    const int flatQ = k_quantScales[rem];
    int16_t* outCoef = a.coeff + (intptr_t)item * NN;
    int32_t* outDelta = a.deltaU ? a.deltaU + (intptr_t)item * NN : nullptr;
    int nzCount = 0;

    if (N == 32)
    {
        const int r = lane & 31, g = lane >> 5;
        v4i tB1, tA2;
        dct32_operands(r, g, tB1, tA2);
        int d[8];
#pragma unroll
        for (int q = 0; q < 4; q++)
        {
            u32x2 v = *(const lu2*)(sa + r * 32 + 16 * g + 4 * q);
            d[2 * q] = (int)v.x; d[2 * q + 1] = (int)v.y;
        }
        v16i acc;
        dct32_forward(d, tB1, tA2, acc);
        wave_sync();                   // all lanes have consumed the residual before sb is reused below
#pragma unroll
        for (int i = 0; i < 16; i++)
        {
            const int k = (i & 3) + 8 * (i >> 2) + 4 * g, idx = k * 32 + r;
            const int coef = (int)(int16_t)((acc[i] + (1 << 10)) >> 11);
            int nz, delta;
            const int level = quant_one(coef, a.quantCoeff ? a.quantCoeff[idx] : flatQ, qBits, add, nz, delta);
            outCoef[idx] = (int16_t)level;
            if (outDelta) outDelta[idx] = delta;
            nzCount += nz;
            if (a.recon) sb[idx] = (int16_t)level;
        }
    }
    else
    {
        const int shift1 = LG - 1 + X265_DEPTH - 8, shift2 = LG + 6;
        // stage 1: sb[k*N + j] = (sum_m M[k][m] * sa[j*N + m] + add) >> shift1   (dct.cpp:83-240 as a matrix product)
        for (int i = lane; i < NN; i += 64)
        {
            const int k = i >> LG, j = i & (N - 1);
            int s = 0;
#pragma unroll
            for (int m = 0; m < N; m++) s += (int)s_m[k * N + m] * (int)sa[j * N + m];
            sb[i] = (int16_t)((s + (1 << (shift1 - 1))) >> shift1);
        }
        wave_sync();
        int coefReg[(NN + 63) / 64];
#pragma unroll
        for (int e = 0; e < (NN + 63) / 64; e++)
        {
            const int i = lane + 64 * e;
            int s = 0;
            if (i < NN)
            {
                const int k = i >> LG, j = i & (N - 1);
#pragma unroll
                for (int m = 0; m < N; m++) s += (int)s_m[k * N + m] * (int)sb[j * N + m];
            }
            coefReg[e] = (int)(int16_t)((s + (1 << (shift2 - 1))) >> shift2);
        }
        wave_sync();
#pragma unroll
        for (int e = 0; e < (NN + 63) / 64; e++)
        {
            const int i = lane + 64 * e;
            if (i < NN)
            {
                int nz, delta;
                const int level = quant_one(coefReg[e], a.quantCoeff ? a.quantCoeff[i] : flatQ, qBits, add, nz, delta);
                outCoef[i] = (int16_t)level;
                if (outDelta) outDelta[i] = delta;
                nzCount += nz;
                if (a.recon) sb[i] = (int16_t)level;
            }
        }
    }
    const int numSig = wsum_u(nzCount);
    if (lane == 0) a.numSig[item] = (uint32_t)numSig;
    if (!a.recon) return;

    // ---- reconstruction (quant.cpp:543-605 with flat lists; search.cpp:5563-5575,5630) ----
    wave_sync();
    const int q0 = uni((int)sb[0]);
    int dcVal = 0;
    bool fill = false;
    if (numSig == 0) fill = true;
    else
    {
        const int shift = 20 - 14 - transformShift, scale = k_invQuantScales[rem] << per, dqAdd = 1 << (shift - 1);
        if (numSig == 1 && q0 != 0 && !(N == 4 && a.dst4))
        {   // DC shortcut (quant.cpp:588-597; not with the DST)
            const int deq0 = clip3(-32768, 32767, (int32_t)((uint32_t)(q0 * scale) + (uint32_t)dqAdd) >> shift);
            const int shift_2nd = 12 - (X265_DEPTH - 8) - 3;
            dcVal = (int)(int16_t)(((((deq0 + 1) >> 1) * 8) + (1 << (shift_2nd - 1))) >> shift_2nd);
            fill = true;
        }
        else
        {
            for (int i = lane; i < NN; i += 64)
                sa[i] = (int16_t)clip3(-32768, 32767, (int32_t)((uint32_t)((int)sb[i] * scale) + (uint32_t)dqAdd) >> shift);
            wave_sync();
            if constexpr (N == 32)
            {   // both inverse stages on the matrix cores (xh_dct32.h): the lane's column of the dequantised block in, row r of the residual out
                const int r = lane & 31, g = lane >> 5;
                v4i tB1, tA2;
                idct32_operands(r, g, tB1, tA2);
                int d[8];
#pragma unroll
                for (int q = 0; q < 8; q++)
                    d[q] = (int)((unsigned)(uint16_t)sa[(16 * g + 2 * q) * 32 + r] | ((unsigned)(uint16_t)sa[(16 * g + 2 * q + 1) * 32 + r] << 16));
                v16i acc;
                idct32_inverse(d, tB1, tA2, acc);
                wave_sync();                                   // every lane has its column: sa becomes the residual
                const int shift2 = 12 - (X265_DEPTH - 8);
#pragma unroll
                for (int i = 0; i < 16; i++)
                    sa[r * 32 + (i & 3) + 8 * (i >> 2) + 4 * g] = (int16_t)clip3(-32768, 32767, (acc[i] + (1 << (shift2 - 1))) >> shift2);
                wave_sync();
            }
            else
            {
            // inverse stage 1: sb[j*N + k] = clip16((sum_m M[m][k] * sa[m*N + j] + 64) >> 7)   (dct.cpp:242-416)
            for (int i = lane; i < NN; i += 64)
            {
                const int j = i >> LG, k = i & (N - 1);
                int s = 0;
#pragma unroll
                for (int m = 0; m < N; m++) s += (int)s_m[m * N + k] * (int)sa[m * N + j];
                sb[i] = (int16_t)clip3(-32768, 32767, (s + 64) >> 7);
            }
            wave_sync();
            const int shift2 = 12 - (X265_DEPTH - 8);
            int resReg[(NN + 63) / 64];
#pragma unroll
            for (int e = 0; e < (NN + 63) / 64; e++)
            {
                const int i = lane + 64 * e;
                int s = 0;
                if (i < NN)
                {
                    const int j = i >> LG, k = i & (N - 1);
#pragma unroll
                    for (int m = 0; m < N; m++) s += (int)s_m[m * N + k] * (int)sb[m * N + j];
                }
                resReg[e] = clip3(-32768, 32767, (s + (1 << (shift2 - 1))) >> shift2);
            }
            wave_sync();
#pragma unroll
            for (int e = 0; e < (NN + 63) / 64; e++)
            {
                const int i = lane + 64 * e;
                if (i < NN) sa[i] = (int16_t)resReg[e];
            }
            wave_sync();
            }
        }
    }
    // recon = clip(pred + resid), SSE against the source (pixel.cpp:820-832, 167-186)
    pixel* rec = a.recon + tk.reconOff;
    unsigned long long sse = 0;
    for (int i = lane; i < NN; i += 64)
    {
        const int y = i >> LG, x = i & (N - 1);
        const int res = fill ? dcVal : (int)sa[i];
        const int v = clip3(0, XH_PIXEL_MAX, (int)pred[i] + res);
        rec[(intptr_t)y * a.reconStride + x] = (pixel)v;
        const int t = (int)cur[(intptr_t)y * a.cs + x] - v;
        sse += (unsigned long long)(t * t);
    }
    sse = wave_sum64(sse);
    if (lane == 0 && a.sse) a.sse[item] = (uint64_t)(sse_t)sse;
}

template<int N> int launch_tq(hipStream_t st, const TqArgs& a)
{
    if (a.ref1) XH_KLAUNCH((tq_kernel<N, true>), dim3((a.n + 3) / 4), dim3(256), 0, st, a);
    else XH_KLAUNCH((tq_kernel<N, false>), dim3((a.n + 3) / 4), dim3(256), 0, st, a);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

} // namespace

static int tq_batch(void* stream, int log2TrSize, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                    const x265hip_tu_task* tasks, int n, const x265hip_tq_params* params,
                    int16_t* coeff, uint32_t* numSig, void* reconPlane, intptr_t reconStride, uint64_t* sse,
                    const x265hip_me_result* mvSource, int tiled)
{
    if (n <= 0) return X265HIP_OK;
    if (!tasks || !params || !coeff || !numSig || log2TrSize < 2 || log2TrSize > 5 || params->qp < 0 || params->qp > 51 ||
        (params->add != 171 && params->add != 85))
    { set_error("tq_batch: bad arguments"); return X265HIP_EARG; }
    TqArgs a = { (const pixel*)curPlane, curStride, (const pixel*)refPlane, refStride, tasks, n,
                 params->qp, params->add, params->quantCoeff, params->deltaU, coeff, numSig,
                 (pixel*)reconPlane, reconStride, sse, mvSource, (const pixel*)params->subpelPlanes, params->planeElems,
                 params->choice, params->choiceList, params->choiceRef, params->chroma, (const pixel*)params->refPlane1, params->choiceRef1, params->dst4, tiled };
    if (params->refPlane1 && (!params->choice || params->chroma || params->choiceRef1 < 0 || params->choiceRef1 >= X265HIP_MAX_REF)) { set_error("tq_batch: a bi-directional launch needs choice records, luma planes and choiceRef1 in 0..15"); return X265HIP_EARG; }
    if (params->choice && (params->choiceList < 0 || params->choiceList > 1 || params->choiceRef < 0 || params->choiceRef >= X265HIP_MAX_REF)) { set_error("tq_batch: bad choiceList / choiceRef"); return X265HIP_EARG; }
    hipStream_t st = (hipStream_t)stream;
    switch (log2TrSize)
    {
    case 2: return launch_tq<4>(st, a);
    case 3: return launch_tq<8>(st, a);
    case 4: return launch_tq<16>(st, a);
    default: return launch_tq<32>(st, a);
    }
}

extern "C" int x265hip_tq_batch(void* stream, int log2TrSize, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                                const x265hip_tu_task* tasks, int n, const x265hip_tq_params* params,
                                int16_t* coeff, uint32_t* numSig, void* reconPlane, intptr_t reconStride, uint64_t* sse,
                                const x265hip_me_result* mvSource)
{
    return tq_batch(stream, log2TrSize, curPlane, curStride, refPlane, refStride, tasks, n, params, coeff, numSig, reconPlane, reconStride, sse, mvSource, 0);
}
