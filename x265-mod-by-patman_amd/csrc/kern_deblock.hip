// kern_deblock.hip -- the deblocking filter of a whole 4:2:0 picture in two launches (all vertical edges, then all horizontal edges), in place on the
// reconstructed planes in HBM.  Reference: Deblock::deblockCTU and everything under it (common/deblock.cpp:37-497: deblockCU, setEdgefilterTU / PU /
// Multiple, bsCuEdge, getBoundaryStrength, edgeFilterLuma, edgeFilterChroma) + pelFilterLumaStrong_c / pelFilterChroma_{V,H}_c (common/loopfilter.cpp:136-232).
//
// The reference walks every CTU's CU quadtree recursively and marks edges in a per-CTU strength array; here ONE THREAD owns one 4-sample edge segment of
// the 8x8 deblocking grid: it derives the segment's boundary strength from the per-partition arrays of the two units it separates (CU size, partition
// shape, transform depth, modes, cbf, references, motion vectors -- CUData's own arrays, CTU after CTU in z-scan order, uploaded as they are), makes the
// filter decisions on its 4 x 8 samples held in registers, and writes at most 3 samples each side.  Segments of one direction never touch each other's
// samples (8 apart, 4 read / 3 written each side), so a pass needs no synchronisation; the horizontal pass reads what the vertical pass wrote (second launch).
// The two chroma planes are filtered by the thread of the luma segment they belong to (edges on the 16-sample luma grid, strength 2 only).
// Lanes run along picture rows: a wavefront reads 64 neighbouring segments = contiguous row pieces.
#include "xh_common.h"
#include "../../include/x265hip_frame.h"
using namespace xh;

namespace {

__constant__ uint8_t c_tc[54] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 7, 8, 9,
                                  10, 11, 13, 14, 16, 18, 20, 22, 24 };                         // deblock.cpp:499-503
__constant__ uint8_t c_beta[52] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36,
                                    38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };    // deblock.cpp:505-509
__constant__ uint8_t c_chromaScale[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 33,
                                           33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };   // constants.cpp:346-350

__device__ __forceinline__ uint32_t spread4(uint32_t v) { v = (v | (v << 2)) & 0x33u; return (v | (v << 1)) & 0x55u; }     // abcd -> 0a0b0c0d

// index of unit (ux, uy) in the CTU-major, z-scan-inside-a-CTU arrays
__device__ __forceinline__ uint32_t part_index(int ux, int uy, int lgUpc, int nx)
{
    const int m = (1 << lgUpc) - 1;
    return (uint32_t)((uy >> lgUpc) * nx + (ux >> lgUpc)) << (2 * lgUpc) | spread4(ux & m) | spread4(uy & m) << 1;
}

__device__ __forceinline__ bool mv_far(int2 a, int2 b) { return abs(a.x - b.x) >= 4 || abs(a.y - b.y) >= 4; }

// everything the strength derivation reads about one unit, fetched in one go (all loads independent of each other)
struct UnitInfo { int log2CU, part, tu, mode, cbf, ref0, ref1, qp, bypass; int2 mv0, mv1; };

__device__ __forceinline__ UnitInfo load_unit(const x265hip_deblock_pic& d, uint32_t i, bool needL1)
{
    UnitInfo u;
    u.log2CU = d.log2CUSize[i]; u.part = d.partSize[i]; u.tu = d.tuDepth[i]; u.mode = d.predMode[i]; u.cbf = d.cbfLuma[i]; u.qp = d.qp[i];
    u.ref0 = d.refIdx0[i]; u.mv0 = ((const int2*)d.mv0)[i];
    u.ref1 = needL1 ? (int)d.refIdx1[i] : -1; u.mv1 = needL1 ? ((const int2*)d.mv1)[i] : make_int2(0, 0);
    u.bypass = d.tqBypassEnabled ? (int)d.tqBypass[i] : 0;
    return u;
}

template<int DIR>
__device__ __forceinline__ int boundary_strength(const x265hip_deblock_pic& d, int ux, int uy, const UnitInfo& Q, const UnitInfo& P)
{
    const int pos = (DIR ? uy : ux) * 4, cuSize = 1 << Q.log2CU, rel = pos & (cuSize - 1);
    int bs;
    if (!rel) bs = pos > 0 ? 2 : 0;                                                    // CU edge (bsCuEdge): the neighbour exists except at the picture border
    else if (!(rel & ((cuSize >> Q.tu) - 1))) bs = 2;                                   // transform edge
    else
    {
        const int ps = Q.part;
        const int at = DIR ? (ps == 1 || ps == 3 ? cuSize >> 1 : ps == 4 ? cuSize >> 2 : ps == 5 ? cuSize - (cuSize >> 2) : -1)
                           : (ps == 2 || ps == 3 ? cuSize >> 1 : ps == 6 ? cuSize >> 2 : ps == 7 ? cuSize - (cuSize >> 2) : -1);
        bs = rel == at ? 1 : 0;                                                        // prediction edge inside the CU
    }
    if (!bs) return 0;
    if (P.mode == 2 || Q.mode == 2) return 2;
    if (bs > 1 && (((Q.cbf >> Q.tu) & 1) || ((P.cbf >> P.tu) & 1))) return 1;
    const int rp0 = P.ref0 >= 0 ? d.refPic[0][P.ref0 & 15] : -1, rq0 = Q.ref0 >= 0 ? d.refPic[0][Q.ref0 & 15] : -1;
    const int2 zero = make_int2(0, 0);
    const int2 mp0 = rp0 >= 0 ? P.mv0 : zero, mq0 = rq0 >= 0 ? Q.mv0 : zero;
    if (d.sliceIsP) return rp0 != rq0 || mv_far(mq0, mp0);
    const int rp1 = P.ref1 >= 0 ? d.refPic[1][P.ref1 & 15] : -1, rq1 = Q.ref1 >= 0 ? d.refPic[1][Q.ref1 & 15] : -1;
    const int2 mp1 = rp1 >= 0 ? P.mv1 : zero, mq1 = rq1 >= 0 ? Q.mv1 : zero;
    if ((rp0 == rq0 && rp1 == rq1) || (rp0 == rq1 && rp1 == rq0))
    {
        if (rp0 != rp1)
            return rp0 == rq0 ? (mv_far(mq0, mp0) || mv_far(mq1, mp1)) : (mv_far(mq1, mp0) || mv_far(mq0, mp1));
        return (mv_far(mq0, mp0) || mv_far(mq1, mp1)) && (mv_far(mq1, mp0) || mv_far(mq0, mp1));
    }
    return 1;
}

template<int DIR>
__device__ __forceinline__ void deblock_body(const x265hip_deblock_pic& d, pixel* __restrict__ Y, intptr_t strideY, pixel* __restrict__ Cb, pixel* __restrict__ Cr,
                                             intptr_t strideC, int lgUpc, int nx, uint8_t* __restrict__ bsOut, int uy0, int uy1)
{   // unit rows [uy0, uy1): the whole picture, or a band of CTU rows (x265hip_deblock_rows; uy0 is a multiple of the CTU's units, so even)
    const int uw = d.width >> 2, uh = d.height >> 2;
    const int a = blockIdx.x * 64 + (threadIdx.x & 63), b = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int ux = DIR ? a : 2 * a, uy = uy0 + (DIR ? 2 * b : b);                       // the 8x8 grid: even units across the edge
    if (ux >= uw || uy >= uy1 || !(DIR ? uy : ux)) return;
    // --slices: the CTU above the first row of a slice is not a neighbour (CUData::initCTU: m_cuAbove = NULL, cudata.cpp:323), its top edge is not filtered
    if (DIR && d.sliceFirstRow && !(uy & ((1 << lgUpc) - 1)) && d.sliceFirstRow[uy >> lgUpc]) return;
    // one round of independent loads: the two units' records and the segment's 4 lines x 8 luma samples (wasted where the strength turns out 0, but a
    // segment that is filtered no longer waits for the records, then the strength inputs, then the samples one round trip after the other)
    const uint32_t q = part_index(ux, uy, lgUpc, nx);
    const uint32_t p = DIR ? part_index(ux, uy - 1, lgUpc, nx) : part_index(ux - 1, uy, lgUpc, nx);
    const UnitInfo Q = load_unit(d, q, !d.sliceIsP), P = load_unit(d, p, !d.sliceIsP);
    pixel* src = Y + (intptr_t)uy * 4 * strideY + ux * 4;
    const intptr_t step = DIR ? 1 : strideY, off = DIR ? strideY : 1;
    int m[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int k = 0; k < 8; k++) m[i][k] = src[i * step + (k - 4) * off];
    if (!Q.mode) return;
    const int bs = boundary_strength<DIR>(d, ux, uy, Q, P);
    if (bsOut) bsOut[((size_t)DIR * uh + uy) * uw + ux] = (uint8_t)bs;
    if (!bs) return;
    int maskP = -1, maskQ = -1;
    if (d.tqBypassEnabled)
    {
        maskP = P.bypass ? 0 : -1; maskQ = Q.bypass ? 0 : -1;
        if (!(maskP | maskQ)) return;
    }
    const int qp = (P.qp + Q.qp + 1) >> 1;
    constexpr int sh = X265_DEPTH - 8;
    const int tcOffset = 2 * d.tcOffsetDiv2;

    // chroma first (independent planes): edges on the 8-sample CHROMA grid across the edge, one 4-sample chroma segment per (1 << the chroma shift along the edge) luma units,
    // with the strength and QPs of the first of them (deblock.cpp:104-113, 457-459).  4:2:0: the 16-sample luma grid, a segment per two units; 4:2:2: 16 across vertical /
    // 8 across horizontal edges; 4:4:4: the luma grid.  The QP goes through the table for 4:2:0 only (:483-484)
    const int hs = d.chromaFormat == 3 ? 0 : 1, vs = (d.chromaFormat == 2 || d.chromaFormat == 3) ? 0 : 1;
    const int across = DIR ? vs : hs, along = DIR ? hs : vs;
    if (bs == 2 && !((DIR ? uy : ux) & ((2 << across) - 1)) && !((DIR ? ux : uy) & ((1 << along) - 1)))
    {
        const intptr_t step = DIR ? 1 : strideC, off = DIR ? strideC : 1;
#pragma unroll
        for (int c = 0; c < 2; c++)
        {
            int cqp = qp + (c ? d.crQpOffset : d.cbQpOffset);
            if (cqp >= 30) cqp = d.chromaFormat <= 1 ? (int)c_chromaScale[min(cqp, 57)] : min(cqp, 51);
            const int tc = c_tc[clip3(0, 53, cqp + 2 + tcOffset)] << sh;
            pixel* s = (c ? Cr : Cb) + (intptr_t)((uy * 4) >> vs) * strideC + ((ux * 4) >> hs);
#pragma unroll
            for (int i = 0; i < 4; i++, s += step)
            {
                const int m2 = s[-2 * off], m3 = s[-off], m4 = s[0], m5 = s[off];
                const int delta = clip3(-tc, tc, ((m4 - m3) * 4 + m2 - m5 + 4) >> 3);
                s[-off] = clip_pixel(m3 + (delta & maskP)); s[0] = clip_pixel(m4 - (delta & maskQ));
            }
        }
    }

    // luma: the segment's 4 lines x 8 samples are in registers already
    const int beta = c_beta[clip3(0, 51, qp + 2 * d.betaOffsetDiv2)] << sh;
    const int dp0 = abs(m[0][1] - 2 * m[0][2] + m[0][3]), dq0 = abs(m[0][4] - 2 * m[0][5] + m[0][6]);
    const int dp3 = abs(m[3][1] - 2 * m[3][2] + m[3][3]), dq3 = abs(m[3][4] - 2 * m[3][5] + m[3][6]);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 >= beta) return;
    const int tc = c_tc[clip3(0, 53, qp + 2 * (bs - 1) + tcOffset)] << sh;
    const int thr = (tc * 5 + 1) >> 1;
    const bool strong = 2 * d0 < (beta >> 2) && 2 * d3 < (beta >> 2) &&
                        abs(m[0][0] - m[0][3]) + abs(m[0][7] - m[0][4]) < (beta >> 3) && abs(m[0][3] - m[0][4]) < thr &&
                        abs(m[3][0] - m[3][3]) + abs(m[3][7] - m[3][4]) < (beta >> 3) && abs(m[3][3] - m[3][4]) < thr;
    if (strong)
    {   // pelFilterLumaStrong_c (results are cast, not clipped to the sample range)
        const int tcP = (2 * tc) & maskP, tcQ = (2 * tc) & maskQ;
#pragma unroll
        for (int i = 0; i < 4; i++)
        {
            const int m0 = m[i][0], m1 = m[i][1], m2 = m[i][2], m3 = m[i][3], m4 = m[i][4], m5 = m[i][5], m6 = m[i][6], m7 = m[i][7];
            pixel* s = src + i * step;
            s[-3 * off] = (pixel)(clip3(-tcP, tcP, ((2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3) - m1) + m1);
            s[-2 * off] = (pixel)(clip3(-tcP, tcP, ((m1 + m2 + m3 + m4 + 2) >> 2) - m2) + m2);
            s[-off] = (pixel)(clip3(-tcP, tcP, ((m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3) - m3) + m3);
            s[0] = (pixel)(clip3(-tcQ, tcQ, ((m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3) - m4) + m4);
            s[off] = (pixel)(clip3(-tcQ, tcQ, ((m3 + m4 + m5 + m6 + 2) >> 2) - m5) + m5);
            s[2 * off] = (pixel)(clip3(-tcQ, tcQ, ((m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3) - m6) + m6);
        }
        return;
    }
    const int side = (beta + (beta >> 1)) >> 3, tc2 = tc >> 1;
    const bool sideP = (dp0 + dp3 < side) && maskP, sideQ = (dq0 + dq3 < side) && maskQ;
#pragma unroll
    for (int i = 0; i < 4; i++)
    {
        const int m1 = m[i][1], m2 = m[i][2], m3 = m[i][3], m4 = m[i][4], m5 = m[i][5], m6 = m[i][6];
        int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
        if (abs(delta) >= tc * 10) continue;
        delta = clip3(-tc, tc, delta);
        pixel* s = src + i * step;
        s[-off] = clip_pixel(m3 + (delta & maskP));
        s[0] = clip_pixel(m4 - (delta & maskQ));
        if (sideP) s[-2 * off] = clip_pixel(m2 + clip3(-tc2, tc2, (((m1 + m3 + 1) >> 1) - m2 + delta) >> 1));
        if (sideQ) s[off] = clip_pixel(m5 + clip3(-tc2, tc2, (((m6 + m4 + 1) >> 1) - m5 - delta) >> 1));
    }
}

template<int DIR>
__global__ __launch_bounds__(256) void deblock_kernel(x265hip_deblock_pic d, pixel* __restrict__ Y, intptr_t strideY, pixel* __restrict__ Cb, pixel* __restrict__ Cr,
                                                      intptr_t strideC, int lgUpc, int nx, uint8_t* __restrict__ bsOut, int uy0, int uy1)
{
    deblock_body<DIR>(d, Y, strideY, Cb, Cr, strideC, lgUpc, nx, bsOut, uy0, uy1);
}
// a batch of pictures of one size: picture = grid z, its description and planes from the device copy of the job list
template<int DIR>
__global__ __launch_bounds__(256) void deblock_pictures_kernel(const x265hip_deblock_job* __restrict__ jobs, intptr_t strideY, intptr_t strideC, int lgUpc, int nx)
{
    const x265hip_deblock_job& j = jobs[blockIdx.z];              // left in memory (uniform address: scalar loads); a local copy would put refPic[][] in scratch
    deblock_body<DIR>(j.pic, (pixel*)j.Y, strideY, (pixel*)j.Cb, (pixel*)j.Cr, strideC, lgUpc, nx, j.bsOut, 0, j.pic.height >> 2);
}

} // namespace

static bool deblock_desc_ok(const x265hip_deblock_pic& d, intptr_t strideY, intptr_t strideC)
{
    return !(d.width < 8 || d.height < 8 || (d.width & 7) || (d.height & 7) || (d.ctuSize != 16 && d.ctuSize != 32 && d.ctuSize != 64) || strideY < d.width || strideC < (d.chromaFormat == 3 ? d.width : d.width / 2) || d.chromaFormat < 0 || d.chromaFormat > 3 ||
             !d.log2CUSize || !d.partSize || !d.tuDepth || !d.predMode || !d.cbfLuma || !d.qp || !d.refIdx0 || !d.mv0 || (!d.sliceIsP && (!d.refIdx1 || !d.mv1)) ||
             (d.tqBypassEnabled && !d.tqBypass));
}

extern "C" int x265hip_deblock_pictures(void* stream, const x265hip_deblock_job* jobsDevice, const x265hip_deblock_job* jobsHost, int nPictures, intptr_t strideY, intptr_t strideC)
{
    if (!jobsDevice || !jobsHost || nPictures < 1 || nPictures > 65535) { set_error("deblock_pictures: bad arguments"); return X265HIP_EARG; }
    const x265hip_deblock_pic& d0 = jobsHost[0].pic;
    for (int i = 0; i < nPictures; i++)
    {
        const x265hip_deblock_job& j = jobsHost[i];
        if (!j.Y || !j.Cb || !j.Cr || !deblock_desc_ok(j.pic, strideY, strideC) || j.pic.width != d0.width || j.pic.height != d0.height || j.pic.ctuSize != d0.ctuSize)
        { set_error("deblock_pictures: bad description of picture %d (all pictures of a batch share width, height and CTU size)", i); return X265HIP_EARG; }
    }
    hipStream_t st = (hipStream_t)stream;
    const int lgUpc = d0.ctuSize == 64 ? 4 : d0.ctuSize == 32 ? 3 : 2, nx = (d0.width + d0.ctuSize - 1) / d0.ctuSize, uw = d0.width >> 2, uh = d0.height >> 2;
    for (int i = 0; i < nPictures; i++)
        if (jobsHost[i].bsOut) XH_HIP(hipMemsetAsync(jobsHost[i].bsOut, 0, (size_t)2 * uw * uh, st));
    XH_KLAUNCH(deblock_pictures_kernel<0>, dim3((uw / 2 + 63) / 64, (uh + 3) / 4, nPictures), dim3(256), 0, st, jobsDevice, strideY, strideC, lgUpc, nx);
    XH_KLAUNCH(deblock_pictures_kernel<1>, dim3((uw + 63) / 64, (uh / 2 + 3) / 4, nPictures), dim3(256), 0, st, jobsDevice, strideY, strideC, lgUpc, nx);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

extern "C" int x265hip_deblock_frame(void* stream, const x265hip_deblock_pic* desc, void* Y, intptr_t strideY, void* Cb, void* Cr, intptr_t strideC, uint8_t* bsOut)
{
    if (!desc) { set_error("deblock_frame: null argument"); return X265HIP_EARG; }
    return x265hip_deblock_rows(stream, desc, Y, strideY, Cb, Cr, strideC, bsOut, 0, (desc->height + (desc->ctuSize > 0 ? desc->ctuSize : 64) - 1) / (desc->ctuSize > 0 ? desc->ctuSize : 64));
}

// A band of CTU rows [ctuRow0, ctuRow1) of the picture: the edges of those rows' CTUs, the band's top edge included (it changes the last 3 luma / 1 chroma lines of the row above,
// which must hold that row's own deblocking already) -- FrameFilter::processRow's order (framefilter.cpp:576-676).  Bands in increasing order over a picture give the whole
// picture's result (oracle: xo_deblock_rows, tests/test_filters_bands.py).  bsOut (optional) is written for the band's unit rows only and is NOT cleared here.
extern "C" int x265hip_deblock_rows(void* stream, const x265hip_deblock_pic* desc, void* Y, intptr_t strideY, void* Cb, void* Cr, intptr_t strideC, uint8_t* bsOut, int ctuRow0, int ctuRow1)
{
    if (!desc || !Y || !Cb || !Cr) { set_error("deblock_frame: null argument"); return X265HIP_EARG; }
    const x265hip_deblock_pic& d = *desc;
    if (d.width < 8 || d.height < 8 || (d.width & 7) || (d.height & 7) || (d.ctuSize != 16 && d.ctuSize != 32 && d.ctuSize != 64) || strideY < d.width || strideC < (d.chromaFormat == 3 ? d.width : d.width / 2) || d.chromaFormat < 0 || d.chromaFormat > 3 ||
        !d.log2CUSize || !d.partSize || !d.tuDepth || !d.predMode || !d.cbfLuma || !d.qp || !d.refIdx0 || !d.mv0 || (!d.sliceIsP && (!d.refIdx1 || !d.mv1)) ||
        (d.tqBypassEnabled && !d.tqBypass))
    { set_error("deblock_frame: bad picture description (dimensions are multiples of 8, CTU 16/32/64, 4:2:0)"); return X265HIP_EARG; }
    hipStream_t st = (hipStream_t)stream;
    const int lgUpc = d.ctuSize == 64 ? 4 : d.ctuSize == 32 ? 3 : 2, nx = (d.width + d.ctuSize - 1) / d.ctuSize, uw = d.width >> 2, uh = d.height >> 2;
    const int nRows = (d.height + d.ctuSize - 1) / d.ctuSize;
    if (ctuRow0 < 0 || ctuRow1 <= ctuRow0 || ctuRow1 > nRows) { set_error("deblock_rows: CTU rows %d..%d of %d", ctuRow0, ctuRow1, nRows); return X265HIP_EARG; }
    const int uy0 = ctuRow0 << lgUpc, uy1 = min(ctuRow1 << lgUpc, uh), nUy = uy1 - uy0;
    if (bsOut && ctuRow0 == 0 && ctuRow1 == nRows) XH_HIP(hipMemsetAsync(bsOut, 0, (size_t)2 * uw * uh, st));
    XH_KLAUNCH(deblock_kernel<0>, dim3((uw / 2 + 63) / 64, (nUy + 3) / 4), dim3(256), 0, st, d, (pixel*)Y, strideY, (pixel*)Cb, (pixel*)Cr, strideC, lgUpc, nx, bsOut, uy0, uy1);
    XH_KLAUNCH(deblock_kernel<1>, dim3((uw + 63) / 64, ((nUy + 1) / 2 + 3) / 4), dim3(256), 0, st, d, (pixel*)Y, strideY, (pixel*)Cb, (pixel*)Cr, strideC, lgUpc, nx, bsOut, uy0, uy1);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
