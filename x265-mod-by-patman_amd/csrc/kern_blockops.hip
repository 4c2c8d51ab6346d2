// kern_blockops.hip -- batched element-wise block primitives: residual / reconstruction / copies /
// shifts / transpose / bi-pred averaging / weighted prediction / down-scaling
// (reference pixel.cpp:385-483, 485-594, 751-854).  One 256-thread workgroup per block; lanes walk
// the block row-major so global accesses are coalesced along rows.
#include "xh_common.h"
#include "../../include/x265hip_frame.h"
using namespace xh;

namespace {

struct BlkArgs
{
    void* dst; intptr_t ds; const int32_t* dOff;
    const void* s0; intptr_t ss0; const int32_t* s0Off;
    const void* s1; intptr_t ss1; const int32_t* s1Off;
    int p0, p1, p2, p3;
};

template<int OP>
__global__ __launch_bounds__(256) void blockop_kernel(int w, int h, BlkArgs a, int n)
{
    const int item = blockIdx.x;
    const intptr_t dO = a.dOff ? a.dOff[item] : 0, o0 = a.s0Off ? a.s0Off[item] : 0, o1 = a.s1Off ? a.s1Off[item] : 0;
    const int ow = (OP == X265HIP_BLK_SCALE2D) ? 32 : (OP == X265HIP_BLK_SCALE1D ? 64 : w);
    const int oh = (OP == X265HIP_BLK_SCALE2D) ? 32 : (OP == X265HIP_BLK_SCALE1D ? 2 : h);
    // blockIdx.y spreads big blocks (weight_pp / weight_sp run over whole reference planes, reference.cpp:161-163)
    for (int i = blockIdx.y * 256 + threadIdx.x; i < ow * oh; i += 256 * gridDim.y)
    {
        const int y = i / ow, x = i - y * ow;
        if (OP == X265HIP_BLK_CALCRESIDUAL || OP == X265HIP_BLK_SUB_PS)
        {   // residual = fenc - pred (pixel.cpp:463-475, 806-818)
            const pixel* f = (const pixel*)a.s0 + o0; const pixel* p = (const pixel*)a.s1 + o1;
            ((int16_t*)a.dst + dO)[y * a.ds + x] = (int16_t)((int)f[y * a.ss0 + x] - (int)p[y * a.ss1 + x]);
        }
        else if (OP == X265HIP_BLK_ADD_PS)
        {   // recon = clip(pred + resid) (pixel.cpp:820-832)
            const pixel* p = (const pixel*)a.s0 + o0; const int16_t* r = (const int16_t*)a.s1 + o1;
            ((pixel*)a.dst + dO)[y * a.ds + x] = clip_pixel((int)p[y * a.ss0 + x] + (int)r[y * a.ss1 + x]);
        }
        else if (OP == X265HIP_BLK_COPY_PP)
            ((pixel*)a.dst + dO)[y * a.ds + x] = ((const pixel*)a.s0 + o0)[y * a.ss0 + x];
        else if (OP == X265HIP_BLK_COPY_SS)
            ((int16_t*)a.dst + dO)[y * a.ds + x] = ((const int16_t*)a.s0 + o0)[y * a.ss0 + x];
        else if (OP == X265HIP_BLK_COPY_SP)
            ((pixel*)a.dst + dO)[y * a.ds + x] = (pixel)((const int16_t*)a.s0 + o0)[y * a.ss0 + x];
        else if (OP == X265HIP_BLK_COPY_PS)
            ((int16_t*)a.dst + dO)[y * a.ds + x] = (int16_t)((const pixel*)a.s0 + o0)[y * a.ss0 + x];
        else if (OP == X265HIP_BLK_FILL_S)
            ((int16_t*)a.dst + dO)[y * a.ds + x] = (int16_t)a.p0;
        else if (OP == X265HIP_BLK_2DTO1D_SHL || OP == X265HIP_BLK_1DTO2D_SHL)
        {   // pixel.cpp:393-409, 427-443
            int v = ((const int16_t*)a.s0 + o0)[y * a.ss0 + x];
            ((int16_t*)a.dst + dO)[y * a.ds + x] = (int16_t)((uint32_t)v << a.p0);
        }
        else if (OP == X265HIP_BLK_2DTO1D_SHR || OP == X265HIP_BLK_1DTO2D_SHR)
        {   // pixel.cpp:411-425, 445-461: round is an int16 (1 << (shift-1))
            int v = ((const int16_t*)a.s0 + o0)[y * a.ss0 + x];
            int16_t round = (int16_t)(1 << (a.p0 - 1));
            ((int16_t*)a.dst + dO)[y * a.ds + x] = (int16_t)((v + round) >> a.p0);
        }
        else if (OP == X265HIP_BLK_TRANSPOSE)
            ((pixel*)a.dst + dO)[y * a.ds + x] = ((const pixel*)a.s0 + o0)[x * a.ss0 + y];
        else if (OP == X265HIP_BLK_ADDAVG)
        {   // pixel.cpp:834-854
            const int shift = XH_IF_INTERNAL_PREC + 1 - X265_DEPTH, offset = (1 << (shift - 1)) + 2 * XH_IF_INTERNAL_OFFS;
            int v = (int)((const int16_t*)a.s0 + o0)[y * a.ss0 + x] + (int)((const int16_t*)a.s1 + o1)[y * a.ss1 + x];
            ((pixel*)a.dst + dO)[y * a.ds + x] = clip_pixel((v + offset) >> shift);
        }
        else if (OP == X265HIP_BLK_PIXELAVG)
        {   // pixel.cpp:537-549
            int v = (int)((const pixel*)a.s0 + o0)[y * a.ss0 + x] + (int)((const pixel*)a.s1 + o1)[y * a.ss1 + x];
            ((pixel*)a.dst + dO)[y * a.ds + x] = (pixel)((v + 1) >> 1);
        }
        else if (OP == X265HIP_BLK_WEIGHT_SP)
        {   // pixel.cpp:485-508: p0=w0 p1=round p2=shift p3=offset
            int v = ((const int16_t*)a.s0 + o0)[y * a.ss0 + x];
            ((pixel*)a.dst + dO)[y * a.ds + x] = clip_pixel(((a.p0 * (v + XH_IF_INTERNAL_OFFS) + a.p1) >> a.p2) + a.p3);
        }
        else if (OP == X265HIP_BLK_WEIGHT_PP)
        {   // pixel.cpp:510-535
            int16_t v = (int16_t)((int)((const pixel*)a.s0 + o0)[y * a.ss0 + x] << (XH_IF_INTERNAL_PREC - X265_DEPTH));
            ((pixel*)a.dst + dO)[y * a.ds + x] = clip_pixel(((a.p0 * (int)v + a.p1) >> a.p2) + a.p3);
        }
        else if (OP == X265HIP_BLK_SCALE1D)
        {   // pixel.cpp:551-577: rows at src and src+128 -> rows at dst and dst+64
            const pixel* s = (const pixel*)a.s0 + o0 + 128 * y;
            ((pixel*)a.dst + dO)[64 * y + x] = (pixel)(((int)s[2 * x] + (int)s[2 * x + 1] + 1) >> 1);
        }
        else if (OP == X265HIP_BLK_SCALE2D)
        {   // pixel.cpp:579-594
            const pixel* s = (const pixel*)a.s0 + o0 + 2 * y * a.ss0 + 2 * x;
            ((pixel*)a.dst + dO)[32 * y + x] = (pixel)(((int)s[0] + (int)s[1] + (int)s[a.ss0] + (int)s[a.ss0 + 1] + 2) >> 2);
        }
    }
}

template<int OP> int launch(hipStream_t st, int w, int h, const BlkArgs& a, int n)
{
    const long long elems = (long long)w * h;
    const unsigned gy = (unsigned)(elems <= 4096 ? 1 : (elems + 4095) / 4096 > 2048 ? 2048 : (elems + 4095) / 4096);
    XH_KLAUNCH(blockop_kernel<OP>, dim3(n, gy), dim3(256), 0, st, w, h, a, n);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}


// ---- SAO statistics (encoder/sao.cpp:1774-1937): saoCuStatsE0..E3 (type 0..3) and saoCuStatsBO (type 4) of one block ----
// Per pixel the class is sign(c - a) + sign(c - b) + 2 for the two neighbours of the edge direction (band: c >> (depth - 5)); the
// reference's rolling sign buffers are the same signs carried from row to row, so every pixel is independent here.  Row 0 takes
// its "upper neighbour" signs from upIn, the signs the reference would leave behind for the next block go to upOutA / upOutB.
// out[0..31] = sums of diff per class, out[32..63] = counts (edge classes already mapped through SAO::s_eoTable).
__global__ __launch_bounds__(256) void sao_stats_kernel(int type, const int16_t* __restrict__ diff, const pixel* __restrict__ rec, intptr_t stride,
                                                        const int8_t* __restrict__ upIn, int endX, int endY, int32_t* __restrict__ out,
                                                        int8_t* __restrict__ upOutA, int8_t* __restrict__ upOutB)
{
    __shared__ int s_sum[32], s_cnt[32];
    const int t = threadIdx.x;
    if (t < 32) { s_sum[t] = 0; s_cnt[t] = 0; }
    __syncthreads();
    auto sgn = [](int v) { return (v > 0) - (v < 0); };
    // neighbour offsets (a = the one the sign buffer stands for, b = the opposite one)
    const int ax = type == 0 ? -1 : type == 2 ? -1 : type == 3 ? 1 : 0, ay = type == 0 ? 0 : -1;
    const int bx = -ax, by = -ay;
    for (int i = t; i < endX * endY; i += 256)
    {
        const int y = i / endX, x = i - y * endX;
        const int c = rec[(intptr_t)y * stride + x];
        int cls;
        if (type == 4) cls = c >> (X265_DEPTH - 5);
        else
        {
            const int sa = (ay == -1 && y == 0) ? (int)upIn[x] : sgn(c - (int)rec[(intptr_t)(y + ay) * stride + x + ax]);
            const int sb = sgn(c - (int)rec[(intptr_t)(y + by) * stride + x + bx]);
            const int e = sa + sb + 2;
            cls = (int)((0x43021u >> (4 * e)) & 15);                            // s_eoTable = { 1, 2, 0, 3, 4 }
        }
        atomicAdd(&s_sum[cls], (int)diff[y * 64 + x]);
        atomicAdd(&s_cnt[cls], 1);
    }
    // the sign buffers after the last row (only the entries the reference writes)
    const int L = endY - 1;
    if (type == 1)
        for (int x = t; x < endX; x += 256) upOutA[x] = (int8_t)(-sgn((int)rec[(intptr_t)L * stride + x] - (int)rec[(intptr_t)(L + 1) * stride + x]));
    else if (type == 2)
    {   // row y writes buffer (y even ? upBufft : upBuff1): [0] = sign(rec[y+1][0] - rec[y][-1]), [x + 1] = -sign(rec[y][x] - rec[y+1][x+1])
        for (int k = 0; k < 2; k++)
        {
            const int y = L - k;                                                  // last row and the one before it
            if (y < 0) break;
            int8_t* dst = (y & 1) ? upOutA : upOutB;
            for (int x = t; x <= endX; x += 256)
                dst[x] = x == 0 ? (int8_t)sgn((int)rec[(intptr_t)(y + 1) * stride] - (int)rec[(intptr_t)y * stride - 1])
                                : (int8_t)(-sgn((int)rec[(intptr_t)y * stride + x - 1] - (int)rec[(intptr_t)(y + 1) * stride + x]));
        }
    }
    else if (type == 3)
    {   // upOutA[0] stands for upBuff1[-1]
        for (int x = t; x < endX; x += 256) upOutA[x] = (int8_t)(-sgn((int)rec[(intptr_t)L * stride + x] - (int)rec[(intptr_t)(L + 1) * stride + x - 1]));
        if (t == 0) upOutA[endX] = (int8_t)sgn((int)rec[(intptr_t)(L + 1) * stride + endX - 1] - (int)rec[(intptr_t)L * stride + endX]);
    }
    __syncthreads();
    if (t < 32) { out[t] = s_sum[t]; out[32 + t] = s_cnt[t]; }
}


// ---- SAO statistics of a whole picture (SAO::calcSaoStatsCTU, encoder/sao.cpp:729-905, luma, one slice): one workgroup per CTU ----
// Every pixel is classified against its real neighbours (the reference's sign buffers hold exactly those signs); which pixels of a CTU
// each class counts is a rectangle per type (skipB / skipR keep away from rows / columns the neighbours' deblocking has not finalised).
// Edge classes accumulate in registers (5 classes x 4 types, compile-time indexed), the 32 bands through LDS atomics (one copy per wavefront).
__global__ __launch_bounds__(1024) void sao_frame_kernel(const pixel* __restrict__ fenc, const pixel* __restrict__ recon, intptr_t stride, int picWidth, int picHeight,
                                                        int ctuSize, int nonDeblocked, int po, int32_t* __restrict__ out, int64_t picElems, int64_t outPicInts,
                                                        const uint8_t* __restrict__ sliceFirstRow, int ctuFirst, int ctuH)
{
    fenc += blockIdx.z * picElems; recon += blockIdx.z * picElems; out += blockIdx.z * outPicInts;      // picture of a batch (grid z)
    constexpr int NW = 16;                                         // wavefronts of the workgroup: a 64x64 CTU is 4 pixels per thread
    __shared__ int s_bo[NW][2][32];
    __shared__ int s_eo[NW][40];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    for (int i = t; i < NW * 2 * 32; i += 1024) (&s_bo[0][0][0])[i] = 0;
    __syncthreads();
    // (ctuSize = the CTU's width in this plane, ctuH its height: they differ in the chroma planes of 4:2:2 -- ctuWidth >>= m_hChromaShift, ctuHeight >>= m_vChromaShift, sao.cpp:748-756)
    const int nx = (picWidth + ctuSize - 1) / ctuSize, ny = (picHeight + ctuH - 1) / ctuH;
    const int addr = ctuFirst + blockIdx.x, lpelx = (addr % nx) * ctuSize, tpely = (addr / nx) * ctuH;      // (ctuFirst: a band of CTU rows, x265hip_sao_stats_rows)
    // --slices: CTU rows that begin a slice (CUData::m_bFirstRowInSlice / m_bLastRowInSlice, sao.cpp:744-746, 763-766): no neighbours above the slice's first row,
    // the slice's last row counts down to its bottom line like the picture's last row
    const int row = addr / nx;
    const bool lastRow = row == ny - 1 || (sliceFirstRow && sliceFirstRow[row + 1]);
    const int above = (!tpely) | (sliceFirstRow ? (int)(sliceFirstRow[row] != 0) : 0);
    const int rpelx = min(lpelx + ctuSize, picWidth), bpely = min(tpely + ctuH, picHeight);
    const int cw = rpelx - lpelx, ch = bpely - tpely;
    const int picH = lastRow ? bpely : picHeight;
    const bool atRight = rpelx == picWidth, atBottom = bpely == picH;
    const int startX = !lpelx;
    // regions (:800-901); deblocked statistics: skipB 4 / skipR 5 throughout, non-deblocked: per class
    // po = plane_offset (:773): 0 for luma, 2 for the chroma planes (whose sizes the caller passes already shifted)
    const int boX = atRight ? cw : cw - (nonDeblocked ? 4 : 5) + po, boY = atBottom ? ch : ch - (nonDeblocked ? 3 : 4) + po;
    const int e0X = atRight ? cw - 1 : cw - 5 + po, e0Y = ch - (nonDeblocked ? 3 : 4) + po;
    const int e1X = atRight ? cw : cw - (nonDeblocked ? 4 : 5) + po, e1Y = atBottom ? ch - 1 : ch - 4 + po;
    const int e2X = atRight ? cw - 1 : cw - 5 + po, e2Y = atBottom ? ch - 1 : ch - 4 + po;
    // nonDeblocked == 2: the border statistics before deblocking (skipB / skipR per class: BO 3 / 4, EO_0 3 / 5, EO_1 4 / 4, EO_2 and EO_3 4 / 5)
    const bool pre = nonDeblocked == 2;
    const int pBoX = atRight ? cw : cw - 4 + po, pBoY = atBottom ? ch : ch - 3 + po;            // BO startX / startY; EO_0 shares startY, EO_1 startX
    const int pE2X = atRight ? cw - 1 : cw - 5 + po, pE2Y = atBottom ? ch - 1 : ch - 4 + po;    // EO_0 / EO_2 / EO_3 startX; EO_1 / EO_2 / EO_3 startY
    int sum[4][5], cnt[4][5];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int c = 0; c < 5; c++) { sum[k][c] = 0; cnt[k][c] = 0; }
    auto sgn = [](int v) { return (v > 0) - (v < 0); };
    for (int i = t; i < cw * ch; i += 1024)
    {
        const int y = i / cw, x = i - y * cw;
        // the pixel, the source pixel and the eight neighbours are loaded unconditionally and together (coordinates clamped into the picture: wherever an
        // edge class counts the pixel its neighbours are inside, so clamping never changes a value that is used) -- a load per class inside its region test
        // made every pixel wait for up to five load round trips one after the other
        const int gx = lpelx + x, gy = tpely + y;
        const int xm = max(gx - 1, 0), xp = min(gx + 1, picWidth - 1);
        const pixel* rc = recon + (intptr_t)gy * stride;
        const pixel* rm = recon + (intptr_t)max(gy - 1, 0) * stride;
        const pixel* rp = recon + (intptr_t)min(gy + 1, picHeight - 1) * stride;
        const int c = rc[gx], d = (int)fenc[(intptr_t)gy * stride + gx] - c;
        const int nA[4] = { (int)rc[xm], (int)rm[gx], (int)rm[xm], (int)rm[xp] };      // (-1,0) (0,-1) (-1,-1) (1,-1)
        const int nB[4] = { (int)rc[xp], (int)rp[gx], (int)rp[xp], (int)rp[xm] };      // (1,0)  (0,1)  (1,1)   (-1,1)
        // pre: SAO::calcSaoStatsCu_BeforeDblk (:908-1207) -- the complement, the CTU's bottom / right border on the picture before deblocking: inside the
        // class's window [firstX, cw - 1) x [above, ch - 1) and right of its startX or below its startY
        const bool boHit = pre ? (x >= pBoX || y >= pBoY) : (x < boX && y < boY);
        if (boHit) { atomicAdd(&s_bo[wave][0][c >> (X265_DEPTH - 5)], d); atomicAdd(&s_bo[wave][1][c >> (X265_DEPTH - 5)], 1); }
        const bool inD[4] = { x >= startX && x < e0X && y < e0Y, x < e1X && y >= above && y < e1Y,
                              x >= startX && x < e2X && y >= above && y < e2Y, x >= startX && x < e2X && y >= above && y < e2Y };
        const bool diagP = y >= above && y < ch - 1 && x >= startX && x < cw - 1 && (x >= pE2X || y >= pE2Y);
        const bool inP[4] = { x < cw - 1 && x >= (y < pBoY ? pE2X : startX), y >= above && y < ch - 1 && (x >= pBoX || y >= pE2Y), diagP, diagP };
        const bool in[4] = { pre ? inP[0] : inD[0], pre ? inP[1] : inD[1], pre ? inP[2] : inD[2], pre ? inP[3] : inD[3] };
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int e = in[k] ? sgn(c - nA[k]) + sgn(c - nB[k]) + 2 : -1;
#pragma unroll
            for (int q = 0; q < 5; q++) { const bool hit = e == q; sum[k][q] += hit ? d : 0; cnt[k][q] += hit ? 1 : 0; }
        }
    }
    // per-wavefront totals of the edge classes, mapped through s_eoTable = { 1, 2, 0, 3, 4 }
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int q = 0; q < 5; q++)
        {
            const int cls = (int)((0x43021u >> (4 * q)) & 15);
            const int a = wave_sum(sum[k][q]), b = wave_sum(cnt[k][q]);
            if (lane == 0) { s_eo[wave][k * 10 + cls] = a; s_eo[wave][k * 10 + 5 + cls] = b; }
        }
    __syncthreads();
    int32_t* o = out + (int64_t)addr * 320;                        // [2][5][32]: offsetOrg, count; types EO_0..3, BO
    for (int i = t; i < 320; i += 1024)
    {
        const int which = i / 160, type = (i % 160) / 32, cls = i % 32;
        int v = 0;
        if (type == 4) { for (int w = 0; w < NW; w++) v += s_bo[w][which][cls]; }
        else if (cls < 5) { for (int w = 0; w < NW; w++) v += s_eo[w][type * 10 + which * 5 + cls]; }
        o[i] = v;
    }
}


// Encoder::computeSSD (encoder/encoder.cpp:1203-1270): sum of squared differences of two planes (PSNR numerator), exact in 64 bits.
// A workgroup takes a 256 x 8 tile: a thread's 16 loads are independent and issued together (a per-thread loop along the row waited for one load
// round trip per step: 10.7 us per 1080p plane).
__global__ __launch_bounds__(256) void plane_ssd_kernel(const pixel* __restrict__ a, const pixel* __restrict__ b, intptr_t stride, int width, int height, unsigned long long* out,
                                                        int64_t aPicElems, int64_t bPicElems, int outPicStride)
{
    a += blockIdx.z * aPicElems; b += blockIdx.z * bPicElems; out += blockIdx.z * outPicStride;         // picture of a batch (grid z)
    __shared__ unsigned long long s_part[4];
    const int x = blockIdx.x * 256 + threadIdx.x, y0 = blockIdx.y * 8;
    int va[8], vb[8];
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
        const bool ok = x < width && y0 + r < height;
        const intptr_t o = ok ? (intptr_t)(y0 + r) * stride + x : 0;
        va[r] = ok ? (int)a[o] : 0; vb[r] = ok ? (int)b[o] : 0;
    }
    unsigned acc32 = 0;                                             // 8 squares of < 2^20 each
#pragma unroll
    for (int r = 0; r < 8; r++) { const int d = va[r] - vb[r]; acc32 += (unsigned)(d * d); }
    const unsigned long long acc = wave_sum64((unsigned long long)acc32);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}


// SSIM of a picture as the frame filter accumulates it (FrameFilter::processPostRow, encoder/framefilter.cpp:704-722; calculateSSIM :839-865; ssim_4x4x2_core /
// ssim_end_1 / ssim_end_4, common/pixel.cpp:623-693).  4x4 blocks on a grid shifted by (2,2), a window = 2x2 blocks.  The window values are computed in
// parallel; what makes the float result identical to the reference's is the ORDER of the additions, kept literally: four windows -> a partial, partials
// added to the CTU row's running float sum left to right, top to bottom; CTU rows added in double.  No FMA contraction anywhere in these functions.
#pragma clang fp contract(off)
__device__ __forceinline__ float ssim_window(int s1, int s2, int ss, int s12)
{
#if X265_DEPTH > 8
    constexpr float c1 = (float)(.01 * .01 * XH_PIXEL_MAX * XH_PIXEL_MAX * 64), c2 = (float)(.03 * .03 * XH_PIXEL_MAX * XH_PIXEL_MAX * 64 * 63);
    const float f1 = (float)s1, f2 = (float)s2, fss = (float)ss, f12 = (float)s12;
    const float vars = fss * 64 - f1 * f1 - f2 * f2, covar = f12 * 64 - f1 * f2;
#else
    constexpr int c1 = (int)(.01 * .01 * XH_PIXEL_MAX * XH_PIXEL_MAX * 64 + .5), c2 = (int)(.03 * .03 * XH_PIXEL_MAX * XH_PIXEL_MAX * 64 * 63 + .5);
    const int f1 = s1, f2 = s2;
    const int vars = ss * 64 - f1 * f1 - f2 * f2, covar = s12 * 64 - f1 * f2;
#endif
    return (float)(2 * f1 * f2 + c1) * (float)(2 * covar + c2) / ((float)(f1 * f1 + f2 * f2 + c1) * (float)(vars + c2));
}

// one workgroup: a tile of 32 x 8 windows = 33 x 9 block sums in LDS; E[wy * nwx + wx] = the window's ssim_end_1 value
__global__ __launch_bounds__(256) void ssim_window_kernel(const pixel* __restrict__ rec, intptr_t stride1, const pixel* __restrict__ fenc, intptr_t stride2,
                                                          int nwx, int nwy, float* __restrict__ E, int64_t recPicElems, int64_t fencPicElems, int64_t ePicFloats)
{
    rec += blockIdx.z * recPicElems; fenc += blockIdx.z * fencPicElems; E += blockIdx.z * ePicFloats;   // picture of a batch (grid z)
    __shared__ uint4 s_blk[9][33];
    const int bx0 = blockIdx.x * 32, by0 = blockIdx.y * 8;
    for (int t = threadIdx.x; t < 9 * 33; t += 256)
    {
        const int ly = t / 33, lx = t - ly * 33, bx = bx0 + lx, by = by0 + ly;
        uint4 s = make_uint4(0, 0, 0, 0);
        if (bx <= nwx && by <= nwy)
        {
            const pixel* a = rec + (intptr_t)(2 + 4 * by) * stride1 + 2 + 4 * bx;
            const pixel* b = fenc + (intptr_t)(2 + 4 * by) * stride2 + 2 + 4 * bx;
            for (int y = 0; y < 4; y++, a += stride1, b += stride2)
                for (int x = 0; x < 4; x++) { const uint32_t p = a[x], q = b[x]; s.x += p; s.y += q; s.z += p * p + q * q; s.w += p * q; }
        }
        s_blk[ly][lx] = s;
    }
    __syncthreads();
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5, wx = bx0 + lx, wy = by0 + ly;
    if (wx >= nwx || wy >= nwy) return;
    const uint4 a = s_blk[ly][lx], b = s_blk[ly][lx + 1], c = s_blk[ly + 1][lx], d = s_blk[ly + 1][lx + 1];
    E[(size_t)wy * nwx + wx] = ssim_window((int)(a.x + b.x + c.x + d.x), (int)(a.y + b.y + c.y + d.y), (int)(a.z + b.z + c.z + d.z), (int)(a.w + b.w + c.w + d.w));
}

// one workgroup per CTU row: 1024 threads form the four-window partials (ssim_end_4) of as many window rows as fit in 32 KB of LDS (all 16 of a 64-row
// CTU row up to 2K-wide pictures), then thread 0 adds them to the running sum in order, 16 partials per step fetched as four 16-byte LDS reads.
// What bounds it is the latency of the global loads of the window values (written by the previous launch, i.e. not in this XCD's L2): with 64 threads
// every thread walked 32 groups one load-wait after the other (40 us per 1080p picture, however the additions were fed -- LDS scalar / vector reads or
// v_readlane, 60 us); 1024 threads take two groups each.
__global__ __launch_bounds__(1024) void ssim_rows_kernel(const float* __restrict__ E, int nwx, int width, int height, int ctuSize, int numRows,
                                                       float* __restrict__ rowSsim, uint32_t* __restrict__ rowCnt, int64_t ePicFloats)
{
    E += blockIdx.y * ePicFloats; rowSsim += blockIdx.y * numRows; rowCnt += blockIdx.y * numRows;      // picture of a batch (grid y)
    constexpr int CAP = 8192;
    __shared__ __attribute__((aligned(16))) float s_part[CAP];
    const int r = blockIdx.x, start = r == 0, end = r == numRows - 1;
    uint32_t minY = r * ctuSize - 4 * !start, maxY = min((uint32_t)((r + 1) * ctuSize - 4 * !end), (uint32_t)height);
    minY += start ? 2 : -6;
    const uint32_t hb = (maxY - minY) >> 2, wy0 = (minY - 2) >> 2;
    const int groups = (nwx + 3) >> 2, padded = (groups + 15) & ~15;      // <= 1024
    const int chunk = max(1, CAP / padded);
    float ssim = 0.0f;
    for (uint32_t y = 1; y < hb; y += chunk)
    {
        const int rows = min((uint32_t)chunk, hb - y);
        for (int t = threadIdx.x; t < rows * padded; t += 1024)
        {
            const int k = t / padded, g = t - k * padded;
            const float* e = E + (size_t)(wy0 + y + k - 1) * nwx;
            float part = 0.0f;
            for (int i = 4 * g; i < 4 * g + 4 && i < nwx; i++) part += e[i];
            s_part[t] = part;
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int k = 0; k < rows; k++)
            {
                const float* sp = s_part + k * padded;
                int g = 0;
                for (; g + 16 <= groups; g += 16)
                {
                    const float4 a = *(const float4*)&sp[g], b = *(const float4*)&sp[g + 4], c = *(const float4*)&sp[g + 8], d = *(const float4*)&sp[g + 12];
                    ssim += a.x; ssim += a.y; ssim += a.z; ssim += a.w; ssim += b.x; ssim += b.y; ssim += b.z; ssim += b.w;
                    ssim += c.x; ssim += c.y; ssim += c.z; ssim += c.w; ssim += d.x; ssim += d.y; ssim += d.z; ssim += d.w;
                }
                for (; g < groups; g++) ssim += sp[g];
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) { rowSsim[r] = ssim; rowCnt[r] = (hb - 1) * (uint32_t)nwx; }
}

__global__ void ssim_total_kernel(const float* __restrict__ rowSsim, const uint32_t* __restrict__ rowCnt, int numRows, double* __restrict__ frame)
{
    rowSsim += blockIdx.x * numRows; rowCnt += blockIdx.x * numRows; frame += 2 * blockIdx.x;           // picture of a batch (grid x)
    double t = 0; uint32_t c = 0;
    for (int r = 0; r < numRows; r++) { t += rowSsim[r]; c += rowCnt[r]; }
    frame[0] = t; frame[1] = (double)c;
}
#pragma clang fp contract(fast)

// SAO of a whole luma plane, out of place (SAO::generateLumaOffsets + applyPixelOffsets, encoder/sao.cpp:268-623): the reference filters in place
// CTU by CTU and classifies against saved copies of the unmodified neighbours (m_tmpU / m_tmpL) -- i.e. against the picture before SAO, which is
// simply the input here.  params: per CTU { typeIdx (-1 off, 0..3 EO, 4 BO), bandPos, offset[4] }.  Each thread filters four neighbouring pixels.
__global__ __launch_bounds__(256) void sao_apply_kernel(const pixel* __restrict__ in, pixel* __restrict__ out, intptr_t stride, int picWidth, int picHeight,
                                                        int ctuSize, int lgCtu, const int32_t* __restrict__ params, int64_t inPicElems, int64_t outPicElems, int64_t paramPicInts)
{
    in += blockIdx.z * inPicElems; out += blockIdx.z * outPicElems; params += blockIdx.z * paramPicInts;   // picture of a batch (grid z)
    const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x0 >= picWidth || y >= picHeight) return;
    const int nx = (picWidth + ctuSize - 1) >> lgCtu, ny = (picHeight + ctuSize - 1) >> lgCtu;
    const int cy = y >> lgCtu;
    auto sgn = [](int v) { return (v > 0) - (v < 0); };
    const pixel* r = in + (intptr_t)y * stride;
    pixel* o = out + (intptr_t)y * stride;
    const bool bottomRowOfPic = y == picHeight - 1;               // (bpely == picHeight) and the last row of that CTU
    for (int i = 0; i < 4; i++)
    {
        const int x = x0 + i;
        if (x >= picWidth) break;
        const int32_t* p = params + 6 * (cy * nx + (x >> lgCtu));
        const int type = p[0], c = r[x];
        int v = c;
        if (type == 4)
        {
            const int band = c >> (X265_DEPTH - 5), k = (band - p[1]) & 31;
            if (k < 4) v = c + (int8_t)p[2 + k];
        }
        else if (type >= 0)
        {
            const bool horiz = type != 1, vert = type != 0;        // which picture edges make the pixel ineligible (:318-320, 372-374, 408-412, 474-478)
            const bool ok = !(horiz && (x == 0 || x == picWidth - 1)) && !(vert && (y == 0 || bottomRowOfPic));
            if (ok)
            {
                const int ax = type == 0 ? -1 : type == 1 ? 0 : type == 2 ? -1 : 1, ay = type == 0 ? 0 : -1;
                const int e = sgn(c - (int)r[(intptr_t)ay * stride + x + ax]) + sgn(c - (int)r[-(intptr_t)ay * stride + x - ax]) + 2;
                const int cls = (int)((0x43021u >> (4 * e)) & 15);  // s_eoTable; class 0 carries no offset
                if (cls) v = c + (int8_t)p[1 + cls];
            }
        }
        o[x] = (pixel)min(max(v, 0), XH_PIXEL_MAX);
    }
    (void)ny;
}

} // namespace

// ---- frame_init_lowres_core (pixel.cpp:596-622): one thread per lowres pixel, 3x3 source neighbourhood -> 4 outputs ----
namespace {
__global__ __launch_bounds__(256) void lowres_kernel(const pixel* __restrict__ src, intptr_t ss, pixel* __restrict__ d0, pixel* __restrict__ dh,
                                                     pixel* __restrict__ dv, pixel* __restrict__ dc, intptr_t ds, int width, int height)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= width || y >= height) return;
    const pixel* s0 = src + (intptr_t)(2 * y) * ss + 2 * x; const pixel* s1 = s0 + ss; const pixel* s2 = s1 + ss;
    const int a0 = s0[0], a1 = s0[1], a2 = s0[2], b0 = s1[0], b1 = s1[1], b2 = s1[2], c0 = s2[0], c1 = s2[1], c2 = s2[2];
    const int v00 = (a0 + b0 + 1) >> 1, v01 = (a1 + b1 + 1) >> 1, v02 = (a2 + b2 + 1) >> 1;      // rows 0/1, columns 0..2
    const int v10 = (b0 + c0 + 1) >> 1, v11 = (b1 + c1 + 1) >> 1, v12 = (b2 + c2 + 1) >> 1;      // rows 1/2
    const intptr_t o = (intptr_t)y * ds + x;
    d0[o] = (pixel)((v00 + v01 + 1) >> 1); dh[o] = (pixel)((v01 + v02 + 1) >> 1);
    dv[o] = (pixel)((v10 + v11 + 1) >> 1); dc[o] = (pixel)((v11 + v12 + 1) >> 1);
}
} // namespace
extern "C" int x265hip_frame_init_lowres(void* stream, const void* src, intptr_t srcStride, void* dst0, void* dsth, void* dstv, void* dstc,
                                         intptr_t dstStride, int width, int height)
{
    if (width <= 0 || height <= 0) return X265HIP_OK;
    if (!src || !dst0 || !dsth || !dstv || !dstc) { set_error("frame_init_lowres: NULL plane"); return X265HIP_EARG; }
    XH_KLAUNCH(lowres_kernel, dim3((unsigned)((width + 63) / 64), (unsigned)((height + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const pixel*)src, srcStride, (pixel*)dst0, (pixel*)dsth, (pixel*)dstv, (pixel*)dstc, dstStride, width, height);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

// ---- extendPicBorder (pixel.cpp:1044-1058) / p.extendRowBorder (ipfilter.cpp:59-77) on pictures resident in HBM ----
// grid.x walks the rows of the padded picture (height + 2*marginY), grid.y the pictures.  A row of the picture proper gets
// its two margins; a margin row is a copy of the widened first / last picture row, built from that row's end pixels so
// that the launch has no ordering between rows.
namespace {
__global__ __launch_bounds__(256) void extend_border_kernel(pixel* __restrict__ picOrg, intptr_t stride, int width, int height, int marginX, int marginY,
                                                            int64_t pictureElems)
{
    pixel* pic = picOrg + (int64_t)blockIdx.y * pictureElems;
    const int row = (int)blockIdx.x - marginY;                          // -marginY .. height + marginY - 1
    const int srcRow = min(max(row, 0), height - 1);
    const pixel* src = pic + (intptr_t)srcRow * stride;
    pixel* dst = pic + (intptr_t)row * stride;
    const pixel left = src[0], right = src[width - 1];
    if (row == srcRow)
    {   // picture row: only the margins are written
        for (int x = threadIdx.x; x < marginX; x += 256) { dst[-marginX + x] = left; dst[width + x] = right; }
    }
    else
    {   // margin row: the whole widened row -- `stride` elements from the row start like the reference's memcpy, i.e.
        // including whatever lies between the right margin and the next row (pixel.cpp:1050-1057)
        for (int x = threadIdx.x - marginX; x < (int)stride - marginX; x += 256)
            dst[x] = x < 0 ? left : (x < width ? src[x] : (x < width + marginX ? right : src[x]));
    }
}
} // namespace
extern "C" int x265hip_extend_pic_border(void* stream, void* picOrg, intptr_t stride, int width, int height, int marginX, int marginY,
                                         int nPictures, int64_t pictureElems)
{
    if (nPictures <= 0) return X265HIP_OK;
    if (!picOrg || width <= 0 || height <= 0 || marginX < 0 || marginY < 0 || stride < width + 2 * marginX)
    { set_error("extend_pic_border: bad arguments"); return X265HIP_EARG; }
    XH_KLAUNCH(extend_border_kernel, dim3((unsigned)(height + 2 * marginY), (unsigned)nPictures), dim3(256), 0, (hipStream_t)stream,
                       (pixel*)picOrg, stride, width, height, marginX, marginY, pictureElems);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

extern "C" int x265hip_blockop_batch(void* stream, int op, int w, int h, const x265hip_blk_args* args, int n)
{
    if (n <= 0) return X265HIP_OK;
    if (!args || w < 1 || h < 1 || w > 16384 || h > 16384) { set_error("blockop_batch: bad arguments (op %d, %dx%d)", op, w, h); return X265HIP_EARG; }
    BlkArgs a = { args->dst, args->dstStride, args->dstOff, args->src0, args->src0Stride, args->src0Off,
                  args->src1, args->src1Stride, args->src1Off, args->p0, args->p1, args->p2, args->p3 };
    hipStream_t st = (hipStream_t)stream;
    switch (op)
    {
#define C(O) case O: return launch<O>(st, w, h, a, n);
    C(X265HIP_BLK_CALCRESIDUAL) C(X265HIP_BLK_SUB_PS) C(X265HIP_BLK_ADD_PS) C(X265HIP_BLK_COPY_PP) C(X265HIP_BLK_COPY_SS)
    C(X265HIP_BLK_COPY_SP) C(X265HIP_BLK_COPY_PS) C(X265HIP_BLK_FILL_S) C(X265HIP_BLK_2DTO1D_SHL) C(X265HIP_BLK_2DTO1D_SHR)
    C(X265HIP_BLK_1DTO2D_SHL) C(X265HIP_BLK_1DTO2D_SHR) C(X265HIP_BLK_TRANSPOSE) C(X265HIP_BLK_ADDAVG) C(X265HIP_BLK_PIXELAVG)
    C(X265HIP_BLK_WEIGHT_SP) C(X265HIP_BLK_WEIGHT_PP) C(X265HIP_BLK_SCALE1D) C(X265HIP_BLK_SCALE2D)
#undef C
    }
    set_error("blockop_batch: unknown op %d", op);
    return X265HIP_EARG;
}

extern "C" int x265hip_sao_stats(void* stream, int type, const int16_t* diff, const void* rec, intptr_t stride, const int8_t* upIn, int endX, int endY,
                                 int32_t* out, int8_t* upOutA, int8_t* upOutB)
{
    if (endX <= 0 || endY <= 0) return X265HIP_OK;
    if (type < 0 || type > 4 || endX > 64 || endY > 64 || !diff || !rec || !out || (type >= 1 && type <= 3 && (!upIn || !upOutA)) || (type == 2 && !upOutB))
    { set_error("sao_stats: bad arguments"); return X265HIP_EARG; }
    XH_KLAUNCH(sao_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, type, diff, (const pixel*)rec, stride, upIn, endX, endY, out, upOutA, upOutB);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

static int sao_stats_launch(void* stream, const void* fenc, const void* recon, intptr_t stride, int picWidth, int picHeight, int ctuSize, int nonDeblocked,
                            int planeOffset, int32_t* out, int nPictures, int64_t pictureElems, const uint8_t* sliceFirstRow, int ctuRow0 = 0, int ctuRow1 = -1, int ctuH = 0)
{
    if (ctuH == 0) ctuH = ctuSize;
    if (!fenc || !recon || !out || picWidth < 1 || picHeight < 1 || (ctuSize != 8 && ctuSize != 16 && ctuSize != 32 && ctuSize != 64) || (ctuH != ctuSize && ctuH != 2 * ctuSize) || ctuH > 64 || stride < picWidth ||
        (planeOffset != 0 && planeOffset != 2) || nPictures < 1 || nPictures > 65535 || (nPictures > 1 && pictureElems < stride * (intptr_t)picHeight))
    { set_error("sao_stats: bad arguments"); return X265HIP_EARG; }
    const int nx = (picWidth + ctuSize - 1) / ctuSize, ny = (picHeight + ctuH - 1) / ctuH, n = nx * ny;
    if (ctuRow1 < 0) ctuRow1 = ny;
    if (ctuRow0 < 0 || ctuRow1 <= ctuRow0 || ctuRow1 > ny) { set_error("sao_stats: CTU rows %d..%d of %d", ctuRow0, ctuRow1, ny); return X265HIP_EARG; }
    XH_KLAUNCH(sao_frame_kernel, dim3((ctuRow1 - ctuRow0) * nx, 1, nPictures), dim3(1024), 0, (hipStream_t)stream, (const pixel*)fenc, (const pixel*)recon, stride, picWidth, picHeight, ctuSize,
                       nonDeblocked, planeOffset, out, pictureElems, (int64_t)n * 320, sliceFirstRow, ctuRow0 * nx, ctuH);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
extern "C" int x265hip_sao_stats_pictures(void* stream, const void* fenc, const void* recon, intptr_t stride, int picWidth, int picHeight, int ctuSize, int nonDeblocked,
                                          int planeOffset, int32_t* out, int nPictures, int64_t pictureElems)
{
    return sao_stats_launch(stream, fenc, recon, stride, picWidth, picHeight, ctuSize, nonDeblocked, planeOffset, out, nPictures, pictureElems, nullptr);
}
extern "C" int x265hip_sao_stats_frame_slices(void* stream, const void* fenc, const void* recon, intptr_t stride, int picWidth, int picHeight, int ctuSize, int nonDeblocked,
                                              int planeOffset, int32_t* out, const uint8_t* sliceFirstRow)
{
    return sao_stats_launch(stream, fenc, recon, stride, picWidth, picHeight, ctuSize, nonDeblocked, planeOffset, out, 1, 0, sliceFirstRow);
}
// the CTUs of the rows [ctuRow0, ctuRow1) only (their entries of `out`): a band of FrameFilter::processRow's pipeline -- the rows below need not be deblocked yet (a CTU's
// statistics leave out what the next row's deblocking changes: skipB, sao.cpp:729-905; oracle: xo_sao_stats_rows)
extern "C" int x265hip_sao_stats_rows(void* stream, const void* fenc, const void* recon, intptr_t stride, int picWidth, int picHeight, int ctuWidth, int ctuHeight, int nonDeblocked,
                                      int planeOffset, int32_t* out, const uint8_t* sliceFirstRow, int ctuRow0, int ctuRow1)
{   // ctuWidth x ctuHeight: the CTU in THIS plane (square but for the chroma planes of 4:2:2, where it is half as wide as high)
    return sao_stats_launch(stream, fenc, recon, stride, picWidth, picHeight, ctuWidth, nonDeblocked, planeOffset, out, 1, 0, sliceFirstRow, ctuRow0, ctuRow1, ctuHeight);
}
extern "C" int x265hip_sao_stats_frame(void* stream, const void* fenc, const void* recon, intptr_t stride, int picWidth, int picHeight, int ctuSize, int nonDeblocked,
                                       int planeOffset, int32_t* out)
{
    return x265hip_sao_stats_pictures(stream, fenc, recon, stride, picWidth, picHeight, ctuSize, nonDeblocked, planeOffset, out, 1, 0);
}

extern "C" int x265hip_plane_ssd_pictures(void* stream, const void* fenc, const void* recon, intptr_t stride, int width, int height, uint64_t* out, int nPictures,
                                          int64_t fencPictureElems, int64_t reconPictureElems)
{
    if (!fenc || !recon || !out || width < 1 || height < 1 || stride < width || width > 16384 || nPictures < 1 || nPictures > 65535) { set_error("plane_ssd: bad arguments"); return X265HIP_EARG; }
    hipStream_t st = (hipStream_t)stream;
    XH_HIP(hipMemsetAsync(out, 0, sizeof(uint64_t) * nPictures, st));
    XH_KLAUNCH(plane_ssd_kernel, dim3((width + 255) / 256, (height + 7) / 8, nPictures), dim3(256), 0, st, (const pixel*)fenc, (const pixel*)recon, stride, width, height,
                       (unsigned long long*)out, fencPictureElems, reconPictureElems, 1);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
extern "C" int x265hip_plane_ssd(void* stream, const void* fenc, const void* recon, intptr_t stride, int width, int height, uint64_t* out)
{
    return x265hip_plane_ssd_pictures(stream, fenc, recon, stride, width, height, out, 1, 0, 0);
}

extern "C" size_t x265hip_ssim_workspace(int width, int height)
{
    if (width < 10 || height < 10) return 0;
    return (size_t)((width - 2) >> 2) * ((height - 2) >> 2) * sizeof(float);
}

extern "C" int x265hip_ssim_pictures(void* stream, const void* recon, intptr_t stride1, const void* fenc, intptr_t stride2, int width, int height, int ctuSize,
                                     void* workspace, float* rowSsim, uint32_t* rowCnt, double* frame, int nPictures, int64_t reconPictureElems, int64_t fencPictureElems)
{
    if (!recon || !fenc || !workspace || !rowSsim || !rowCnt || !frame || width < 10 || height < 10 || width > 16384 || (ctuSize != 16 && ctuSize != 32 && ctuSize != 64) ||
        stride1 < width || stride2 < width || nPictures < 1 || nPictures > 65535)
    { set_error("ssim_frame: bad arguments"); return X265HIP_EARG; }
    hipStream_t st = (hipStream_t)stream;
    const int numRows = (height + ctuSize - 1) / ctuSize;
    const int nwx = ((width - 2) >> 2) - 1;                              // windows per row; block columns 0..nwx
    /* block rows the CTU rows cover: the last CTU row ends at block row B0 + hb - 1 (the arithmetic of ssim_rows_kernel) */
    const int lastMin = numRows == 1 ? 2 : (numRows - 1) * ctuSize - 10;
    const int nby = ((lastMin - 2) >> 2) + ((height - lastMin) >> 2);
    const int nwy = nby - 1;
    if (nwx < 1 || nwy < 1)
    {   /* no complete window: the reference reports zero windows (single CTU row pictures lower than 10 rows cannot get here) */
        XH_HIP(hipMemsetAsync(rowSsim, 0, (size_t)nPictures * numRows * sizeof(float), st)); XH_HIP(hipMemsetAsync(rowCnt, 0, (size_t)nPictures * numRows * sizeof(uint32_t), st));
        XH_HIP(hipMemsetAsync(frame, 0, (size_t)nPictures * 2 * sizeof(double), st));
        return X265HIP_OK;
    }
    const int64_t ePic = (int64_t)(x265hip_ssim_workspace(width, height) / sizeof(float));
    XH_KLAUNCH(ssim_window_kernel, dim3((nwx + 31) / 32, (nwy + 7) / 8, nPictures), dim3(256), 0, st, (const pixel*)recon, stride1, (const pixel*)fenc, stride2, nwx, nwy,
                       (float*)workspace, reconPictureElems, fencPictureElems, ePic);
    XH_KLAUNCH(ssim_rows_kernel, dim3(numRows, nPictures), dim3(1024), 0, st, (const float*)workspace, nwx, width, height, ctuSize, numRows, rowSsim, rowCnt, ePic);
    XH_KLAUNCH(ssim_total_kernel, dim3(nPictures), dim3(1), 0, st, (const float*)rowSsim, (const uint32_t*)rowCnt, numRows, frame);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
extern "C" int x265hip_ssim_frame(void* stream, const void* recon, intptr_t stride1, const void* fenc, intptr_t stride2, int width, int height, int ctuSize,
                                  void* workspace, float* rowSsim, uint32_t* rowCnt, double* frame)
{
    return x265hip_ssim_pictures(stream, recon, stride1, fenc, stride2, width, height, ctuSize, workspace, rowSsim, rowCnt, frame, 1, 0, 0);
}

extern "C" int x265hip_sao_apply_pictures(void* stream, const void* in, void* out, intptr_t stride, int picWidth, int picHeight, int ctuSize, const int32_t* params,
                                          int nPictures, int64_t pictureElems)
{
    if (!in || !out || in == out || !params || picWidth < 1 || picHeight < 1 || (ctuSize != 8 && ctuSize != 16 && ctuSize != 32 && ctuSize != 64) || stride < picWidth ||
        nPictures < 1 || nPictures > 65535)
    { set_error("sao_apply: bad arguments (out of place only)"); return X265HIP_EARG; }
    const int lg = ctuSize == 64 ? 6 : ctuSize == 32 ? 5 : ctuSize == 16 ? 4 : 3;        // 8: the chroma plane of a 4:2:0 picture with 16x16 CTUs
    const int n = ((picWidth + ctuSize - 1) / ctuSize) * ((picHeight + ctuSize - 1) / ctuSize);
    XH_KLAUNCH(sao_apply_kernel, dim3((picWidth + 255) / 256, (picHeight + 3) / 4, nPictures), dim3(256), 0, (hipStream_t)stream, (const pixel*)in, (pixel*)out, stride,
                       picWidth, picHeight, ctuSize, lg, params, pictureElems, pictureElems, (int64_t)n * 6);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
extern "C" int x265hip_sao_apply_frame(void* stream, const void* in, void* out, intptr_t stride, int picWidth, int picHeight, int ctuSize, const int32_t* params)
{
    return x265hip_sao_apply_pictures(stream, in, out, stride, picWidth, picHeight, ctuSize, params, 1, 0);
}
