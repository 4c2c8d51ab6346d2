// kern_lookahead.hip -- the lookahead's frame-cost path on half-resolution ("lowres") pictures resident in HBM:
//   la_intra_kernel   LookaheadTLD::lowresIntraEstimate           (reference encoder/slicetype.cpp:755-864)
//   la_search_kernel  the motion-search half of estimateCUCost     (slicetype.cpp:4467-4572; motion.cpp:923-1140, 1644-1773, lowres branch)
//   la_finish_kernel  the decision half of estimateCUCost          (slicetype.cpp:4574-4640)
//
// estimateFrameCost walks the 8x8 blocks of a picture in REVERSE raster order because every block takes its MV
// predictor from the blocks to its right and below (right, below, below-left, below-right).  That is a wavefront
// dependency, not a serial one: block (x, y) can start once step (W-1-x) + 2*(H-1-y) - 1 is done.  One workgroup owns
// one (estimate, list) pair and sweeps the wavefront with one barrier per step; an 8-lane group (lane = block row)
// runs the complete lowres motionEstimate of a block.  Independent estimates -- the lookahead asks for O(bframes^2)
// per mini-GOP, and list 0 / list 1 of one estimate are independent too -- fill the machine.
#include "xh_common.h"
#include "../../include/x265hip_frame.h"
#include <algorithm>
#include <cmath>
#include <vector>
using namespace xh;

namespace {

constexpr int CU = 8;                      // X265_LOWRES_CU_SIZE
constexpr int COST_MAX = 1 << 28;          // MotionEstimate::COST_MAX (motion.h:65)
constexpr int LOWRES_COST_MASK = (1 << 14) - 1, LOWRES_COST_SHIFT = 14;   // slicetype.h:42-43
constexpr int LA_MERANGE = 16;             // CostEstimateGroup::s_merange (slicetype.h:337)

#define LA_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)
__device__ __forceinline__ int quad_sum(int v) { v += LA_DPP(v, 0xB1); v += LA_DPP(v, 0x4E); return v; }          // quad_perm [1,0,3,2], [2,3,0,1]
__device__ __forceinline__ int xor4(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x101F); }                      // lane ^ 4 (bit mode: and 0x1f, xor 4)
__device__ __forceinline__ int group8_sum(int v) { v = quad_sum(v); return v + xor4(v); }

// one row of eight pixels
#if X265_DEPTH == 8
struct Row { uint32_t w[2]; };
__device__ __forceinline__ uint32_t avg_word(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }   // (a + b + 1) >> 1 per byte
__device__ __forceinline__ int sad_row(const Row& a, const Row& b)
{
    return (int)__builtin_amdgcn_sad_u8(a.w[1], b.w[1], __builtin_amdgcn_sad_u8(a.w[0], b.w[0], 0));
}
__device__ __forceinline__ void diff_row(const Row& a, const Row& b, int (&d)[8])
{
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = (int)((a.w[i >> 2] >> (8 * (i & 3))) & 255) - (int)((b.w[i >> 2] >> (8 * (i & 3))) & 255);
}
constexpr int ROW_WORDS = 2;
#else
struct Row { uint32_t w[4]; };
__device__ __forceinline__ uint32_t avg_word(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) >> 1) & 0x7fff7fffu); }
__device__ __forceinline__ int sad_row(const Row& a, const Row& b)
{
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) s = __builtin_amdgcn_sad_u16(a.w[i], b.w[i], s);
    return (int)s;
}
__device__ __forceinline__ void diff_row(const Row& a, const Row& b, int (&d)[8])
{
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = (int)((a.w[i >> 1] >> (16 * (i & 1))) & 0xffff) - (int)((b.w[i >> 1] >> (16 * (i & 1))) & 0xffff);
}
constexpr int ROW_WORDS = 4;
#endif
// A row at any pixel alignment, fetched as DWORD-ALIGNED loads + funnel shifts: one lane per cache line is already the
// slowest pattern for the L1/TA front end (profiles/micro/RESULTS.md: ~37 clk per wave load), and a sub-dword-misaligned
// address would split every lane's load again (2-4x).  Reads up to 3 bytes past the row (inside the plane margins).
// The address is "uniform 4-byte-aligned buffer base + 32-bit element offset": SGPR-base addressing, no 64-bit VALU math.
__device__ __forceinline__ Row ld_row(const pixel* base, uint32_t e)
{
    const uint32_t boff = e * (uint32_t)sizeof(pixel);
    const char* a = (const char*)base + (boff & ~3u);
    const unsigned m = boff & 3u;
    Row r;
#if X265_DEPTH == 8
    struct W3 { uint32_t x, y, z; } w;
    __builtin_memcpy(&w, __builtin_assume_aligned(a, 4), 12);
    r.w[0] = __builtin_amdgcn_alignbyte(w.y, w.x, m); r.w[1] = __builtin_amdgcn_alignbyte(w.z, w.y, m);
#else
    struct W4 { uint32_t x, y, z, t; } w; uint32_t w4;
    __builtin_memcpy(&w, __builtin_assume_aligned(a, 4), 16);
    __builtin_memcpy(&w4, __builtin_assume_aligned(a + 16, 4), 4);
    r.w[0] = __builtin_amdgcn_alignbyte(w.y, w.x, m); r.w[1] = __builtin_amdgcn_alignbyte(w.z, w.y, m);
    r.w[2] = __builtin_amdgcn_alignbyte(w.t, w.z, m); r.w[3] = __builtin_amdgcn_alignbyte(w4, w.t, m);
#endif
    return r;
}
__device__ __forceinline__ Row avg_rows(const Row& a, const Row& b)
{   // pixelavg_pp (pixel.cpp:537-549) with the 32/32 weights the lookahead passes
    Row r;
#pragma unroll
    for (int i = 0; i < ROW_WORDS; i++) r.w[i] = avg_word(a.w[i], b.w[i]);
    return r;
}

// SATD of the 8x8 block whose row `lane & 7` differences are d[]: two satd_8x4 (pixel.cpp:237-265), each >> 1.
// Horizontal 4-point Hadamards in registers, vertical ones across the four lanes of a quad (two DPP butterflies).
__device__ __forceinline__ int satd_rows(const int (&d)[8], int lane)
{
    int h[8];
#pragma unroll
    for (int k = 0; k < 8; k += 4)
    {
        const int a0 = d[k] + d[k + 1], a1 = d[k] - d[k + 1], a2 = d[k + 2] + d[k + 3], a3 = d[k + 2] - d[k + 3];
        h[k] = a0 + a2; h[k + 1] = a1 + a3; h[k + 2] = a0 - a2; h[k + 3] = a1 - a3;
    }
    const int s1 = (lane & 1) ? -1 : 1, s2 = (lane & 2) ? -1 : 1;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
    {
        int v = h[k];
        v = LA_DPP(v, 0xB1) + __mul24(v, s1);         // full-rate 24-bit multiply: |v| < 2^15
        v = LA_DPP(v, 0x4E) + __mul24(v, s2);
        s += abs(v);
    }
    const int half = quad_sum(s) >> 1;           // one 8x4
    return half + xor4(half);
}

// The same SATD on PACKED 16-bit lanes (|coefficient| <= 16 * 1023 fits int16): the left and the right 4x4 block of the row travel in
// the low and the high half of one register -- the pairing of the reference's SWAR satd_8x4 (pixel.cpp:237-265) -- so the horizontal
// butterflies are four v_pk_add/sub pairs between registers and the two vertical (cross-lane) stages one DPP move + one packed
// multiply-add per register instead of two scalar pairs.  About half the vector instructions of satd_rows.
typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2 as_s2(uint32_t v) { return __builtin_bit_cast(s2, v); }
__device__ __forceinline__ uint32_t as_u32(s2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ int satd_rows_pk(const Row& a, const Row& b, int lane)
{
#if X265_DEPTH > 10
    {   // 12 bit: a coefficient reaches 16 * 4095, beyond int16 -- the 32-bit form
        int d[8];
#pragma unroll
        for (int k = 0; k < 8; k++) d[k] = (int)((a.w[k >> 1] >> (16 * (k & 1))) & 0xffffu) - (int)((b.w[k >> 1] >> (16 * (k & 1))) & 0xffffu);
        return satd_rows(d, lane);
    }
#endif
    s2 r[4];                                          // r[k] = (d[k], d[k + 4]): column k of the left block, column k of the right block
#if X265_DEPTH == 8
#pragma unroll
    for (int k = 0; k < 4; k++)
    {   // v_perm_b32: bytes (w0.b[k], 0, w1.b[k], 0); selector 0-3 = second operand, 4-7 = first, 0x0C = zero
        const uint32_t sel = 0x0C040C00u + 0x00010001u * (uint32_t)k;
        r[k] = as_s2(__builtin_amdgcn_perm(a.w[1], a.w[0], sel)) - as_s2(__builtin_amdgcn_perm(b.w[1], b.w[0], sel));
    }
#else
    {
        const s2 d01 = as_s2(a.w[0]) - as_s2(b.w[0]), d23 = as_s2(a.w[1]) - as_s2(b.w[1]), d45 = as_s2(a.w[2]) - as_s2(b.w[2]), d67 = as_s2(a.w[3]) - as_s2(b.w[3]);
        r[0] = as_s2(__builtin_amdgcn_perm(as_u32(d45), as_u32(d01), 0x05040100u)); r[1] = as_s2(__builtin_amdgcn_perm(as_u32(d45), as_u32(d01), 0x07060302u));
        r[2] = as_s2(__builtin_amdgcn_perm(as_u32(d67), as_u32(d23), 0x05040100u)); r[3] = as_s2(__builtin_amdgcn_perm(as_u32(d67), as_u32(d23), 0x07060302u));
    }
#endif
    const s2 a0 = r[0] + r[1], a1 = r[0] - r[1], a2 = r[2] + r[3], a3 = r[2] - r[3];
    s2 h[4] = { a0 + a2, a1 + a3, a0 - a2, a1 - a3 };
    const short m1 = (lane & 1) ? -1 : 1, m2 = (lane & 2) ? -1 : 1;
    const s2 s1 = { m1, m1 }, sg2 = { m2, m2 };
    u2 acc = { 0, 0 };
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
        s2 v = h[k];
        v = as_s2((uint32_t)LA_DPP((int)as_u32(v), 0xB1)) + v * s1;
        v = as_s2((uint32_t)LA_DPP((int)as_u32(v), 0x4E)) + v * sg2;
        const s2 n = -v;
        acc += __builtin_bit_cast(u2, __builtin_elementwise_max(v, n));      // |v|; four of them stay below 2^16 as UNSIGNED 16-bit
    }
    const uint32_t u = __builtin_bit_cast(uint32_t, acc);
    const int s = (int)(u & 0xffffu) + (int)(u >> 16);
    const int half = quad_sum(s) >> 1;                // one 8x4
    return half + xor4(half);
}

// ---- per-block context of the search / finish kernels ----
struct Blk
{
    Row fenc;
    const pixel* base; uint32_t ref0, pe;   // lowres buffer; element offset of the lane's row at MV 0 in the reference's full-pel plane; the H / V / HV half-pel planes follow pe elements apart (lowres.h:75-124)
    int stride;
    const uint16_t* lcost; int mvpx, mvpy;   // MVD cost row in LDS (centred); it covers every |mv - mvp| a picture of this size can produce
    int lane;
    // the final zero-MV check of motionEstimate (motion.cpp:1763-1768) goes through subpelCompare, which always reads ReferencePlanes::fpelPlane[0] with lumaStride: on the
    // quarter-resolution level of --hme that is the HALF-resolution plane, at the quarter-resolution block offset.  zbase != NULL: the lane's row of that block (element offset zref).
    const pixel* zbase; uint32_t zref;
    const uint16_t* gcost;                   // centre of the whole MVD cost row in memory: the star search's raster takes one cost at twice the quarter-pel vector (motion.cpp:1392), beyond the LDS copy
};
__device__ __forceinline__ int mvcost(const Blk& c, int qx, int qy) { return (uint16_t)((int)c.lcost[qx - c.mvpx] + (int)c.lcost[qy - c.mvpy]); }   // bitcost.h:57
// ReferencePlanes::lowresMC: half-pel positions are planes, quarter-pel positions the rounded average of two of them
__device__ __forceinline__ Row mc_row(const Blk& c, int qx, int qy)
{
    const int hpelA = (qy & 2) | ((qx & 2) >> 1);
    const Row a = ld_row(c.base, c.ref0 + __umul24(hpelA, c.pe) + (uint32_t)((qx >> 2) + __mul24(qy >> 2, c.stride)));
    if (!((qx | qy) & 1)) return a;
    const int qx2 = qx + (qx & 1), qy2 = qy + (qy & 1);
    const int hpelB = (qy2 & 2) | ((qx2 & 2) >> 1);
    const Row b = ld_row(c.base, c.ref0 + __umul24(hpelB, c.pe) + (uint32_t)((qx2 >> 2) + __mul24(qy2 >> 2, c.stride)));
    return avg_rows(a, b);
}
// K candidates (quarter-pel coordinates) costed in one pass: loads first, then the reductions
template<int K, bool SATD>
__device__ __forceinline__ void eval(const Blk& c, const int (&qx)[K], const int (&qy)[K], int (&out)[K])
{
    Row r[K];
#pragma unroll
    for (int k = 0; k < K; k++) r[k] = mc_row(c, qx[k], qy[k]);
#pragma unroll
    for (int k = 0; k < K; k++)
    {
        if (SATD) out[k] = satd_rows_pk(c.fenc, r[k], c.lane);
        else out[k] = group8_sum(sad_row(c.fenc, r[k]));
    }
}

__device__ __forceinline__ int k_hex2(int i, int col)
{   // motion.cpp:64: {-1,-2} {-2,0} {-1,2} {1,2} {2,0} {1,-2} {-1,-2} {-2,0}, packed (value + 2) in 3 bits
    const unsigned v = col ? 0x402910u /* y */ : 0x05c641u /* x */;
    return (int)((v >> (3 * i)) & 7) - 2;
}
__device__ __forceinline__ int k_mod6m1(int i) { return (int)((0x05432105u >> (4 * i)) & 15); }   // motion.cpp:65
__device__ __forceinline__ int k_sq(int i, int col)
{   // square1 (motion.cpp:66): {0,0} {0,-1} {0,1} {-1,0} {1,0} {-1,-1} {-1,1} {1,-1} {1,1}, packed (value + 1) in 2 bits
    const unsigned v = col ? 0x22161u /* y */ : 0x28215u /* x */;
    return (int)((v >> (2 * i)) & 3) - 1;
}

// hex4 of the uneven multi-hexagon search (motion.cpp:67-73), packed (value + 4) per nibble
__device__ __forceinline__ int k_hex4(int j, int col)
{
    const unsigned long long v = col ? 0x7766554433221180ull : 0x6280808080806244ull;
    return (int)((v >> (4 * j)) & 15) - 4;
}
// X265_UMH_SEARCH up to its `goto me_hex2` (motion.cpp:1142-1326) for an 8x8 lowres block (no motion candidates: the range is never adapted) -- what --hme runs on the
// half-resolution level by default (hmeSearchMethod[1]).  The same sequence as umh_stage of me_body.inc, on this kernel's evaluator; candidates the reference skips are
// measured at the origin and ignored.  Returns whether the hexagon refinement follows.
__device__ __forceinline__ bool lowres_umh(Blk& c, int px, int py, int merange, int mnx, int mny, int mxx, int mxy, int& bx, int& by, int& bcost)
{
    int ox = bx, oy = by;
    constexpr int scale = (CU * CU) >> 4;                                  // sizeScale of the 8x8 PU (:60-61, 123-153)
    auto sadThresh = [&](int v) { return bcost < (v >> 4) * scale; };
    auto x4 = [&](int ax, int ay, int bx_, int by_, int cx, int cy, int dx, int dy) {      // COST_MV_X4 (:296-317): only the vertical range is tested
        const int X[4] = { ox + ax, ox + bx_, ox + cx, ox + dx }, Y[4] = { oy + ay, oy + by_, oy + cy, oy + dy };
        const int QX[4] = { X[0] * 4, X[1] * 4, X[2] * 4, X[3] * 4 }, QY[4] = { Y[0] * 4, Y[1] * 4, Y[2] * 4, Y[3] * 4 };
        int C[4];
        eval<4, false>(c, QX, QY, C);
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            const int cost = C[k] + mvcost(c, QX[k], QY[k]);
            if ((Y[k] >= mny) & (Y[k] <= mxy) && cost < bcost) { bcost = cost; bx = X[k]; by = Y[k]; }
        }
    };
    auto pair = [&](int xa, int ya, bool oka, int xb, int yb, bool okb) {                  // two COST_MV of one CROSS step
        const int QX[2] = { (oka ? xa : ox) * 4, (okb ? xb : ox) * 4 }, QY[2] = { (oka ? ya : oy) * 4, (okb ? yb : oy) * 4 };
        int C[2];
        eval<2, false>(c, QX, QY, C);
        const int c0 = C[0] + mvcost(c, QX[0], QY[0]), c1 = C[1] + mvcost(c, QX[1], QY[1]);
        if (oka && c0 < bcost) { bcost = c0; bx = xa; by = ya; }
        if (okb && c1 < bcost) { bcost = c1; bx = xb; by = yb; }
    };
    auto cross = [&](int start, int xMax, int yMax) {                                        // CROSS (:361-385)
        int i = start;
        if (xMax <= min(mxx - ox, ox - mnx))
            for (; i < xMax - 2; i += 4) x4(i, 0, -i, 0, i + 2, 0, -i - 2, 0);
        for (; i < xMax; i += 2) pair(ox + i, oy, ox + i <= mxx, ox - i, oy, ox - i >= mnx);
        i = start;
        if (yMax <= min(mxy - oy, oy - mny))
            for (; i < yMax - 2; i += 4) x4(0, i, 0, -i, 0, i + 2, 0, -i - 2);
        for (; i < yMax; i += 2) pair(ox, oy + i, oy + i <= mxy, ox, oy - i, oy - i >= mny);
    };
    const int ucost1 = bcost;                                              // refine predictors (:1147-1159)
    int crossStart = 1;
    ox = px; oy = py; x4(0, -1, 0, 1, -1, 0, 1, 0);
    if (px | py) { ox = 0; oy = 0; x4(0, -1, 0, 1, -1, 0, 1, 0); }
    const int ucost2 = bcost;
    if ((bx | by) && !(bx == px && by == py)) { ox = bx; oy = by; x4(0, -1, 0, 1, -1, 0, 1, 0); }
    if (bcost == ucost2) crossStart = 3;
    ox = bx; oy = by;
    if (bcost == ucost2 && sadThresh(2000))
    {   // early termination (:1161-1180)
        x4(0, -2, -1, -1, 1, -1, -2, 0);
        x4(2, 0, -1, 1, 1, 1, 0, 2);
        if (bcost == ucost1 && sadThresh(500)) return false;
        if (bcost == ucost2)
        {
            const int range = (int16_t)(merange >> 1) | 1;
            cross(3, range, range);
            x4(-1, -2, 1, -2, -2, -1, 2, -1);
            x4(-2, 1, 2, 1, -1, 2, 1, 2);
            if (bcost == ucost2) return false;
            crossStart = range + 2;
        }
    }
    cross(crossStart, merange, merange >> 1);
    x4(-2, -2, -2, 2, 2, -2, 2, 2);
    ox = bx; oy = by;                                                      // hexagon grid (:1243-1320)
    int i = 1;
    do
    {
        for (int j0 = 0; j0 < 16; j0 += 4)
        {
            int X[4], Y[4], QX[4], QY[4], C[4]; bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const int x = ox + k_hex4(j0 + k, 0) * i, y = oy + k_hex4(j0 + k, 1) * i;
                ok[k] = x >= mnx && x <= mxx && y >= mny && y <= mxy;
                X[k] = ok[k] ? x : ox; Y[k] = ok[k] ? y : oy; QX[k] = X[k] * 4; QY[k] = Y[k] * 4;
            }
            eval<4, false>(c, QX, QY, C);
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const int cost = C[k] + mvcost(c, QX[k], QY[k]);
                if (ok[k] && cost < bcost) { bcost = cost; bx = X[k]; by = Y[k]; }
            }
        }
    }
    while (++i <= merange >> 2);
    return bx >= mnx && bx <= mxx && by >= mny && by <= mxy;               // `goto me_hex2` (:1323-1324)
}

// X265_STAR_SEARCH (motion.cpp:1328-1436) with StarPatternSearch (:387-629) for an 8x8 lowres block.  A round's points are costed together (4, 8 or 2 x 8 at a time) and
// compared in the reference's order; a point the reference's window tests leave out is measured at the round's centre and ignored.  The x4 form of a round visits the
// points of the guarded form in the same order, and `all inside` implies every guard, so the guards alone decide.
struct StarSt { int bx, by, bcost, pointNr, dist; };
template<int K>
__device__ __forceinline__ void star_batch(const Blk& c, StarSt& s, int ox, int oy, const int (&X)[K], const int (&Y)[K], const bool (&ok)[K], const int (&nr)[K], const int (&ds)[K])
{
    int QX[K], QY[K], C[K];
#pragma unroll
    for (int k = 0; k < K; k++) { QX[k] = (ok[k] ? X[k] : ox) * 4; QY[k] = (ok[k] ? Y[k] : oy) * 4; }
    eval<K, false>(c, QX, QY, C);
#pragma unroll
    for (int k = 0; k < K; k++)
    {
        const int cost = C[k] + mvcost(c, QX[k], QY[k]);
        if (ok[k] && cost < s.bcost) { s.bcost = cost; s.bx = X[k]; s.by = Y[k]; s.pointNr = nr[k]; s.dist = ds[k]; }
    }
}
__device__ __forceinline__ void lowres_star_pattern(const Blk& c, int mnx, int mny, int mxx, int mxy, StarSt& s, int earlyExitIters, int merange)
{
    const int ox = s.bx, oy = s.by;
    int rounds = 0;
    {   // distance 1 (:406-448): 2, 4, 5, 7
        const int saved = s.bcost;
        const int X[4] = { ox, ox - 1, ox + 1, ox }, Y[4] = { oy - 1, oy, oy, oy + 1 };
        const bool ok[4] = { oy - 1 >= mny, ox - 1 >= mnx, ox + 1 <= mxx, oy + 1 <= mxy };
        const int nr[4] = { 2, 4, 5, 7 }, ds[4] = { 1, 1, 1, 1 };
        star_batch<4>(c, s, ox, oy, X, Y, ok, nr, ds);
        if (s.bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
    for (int dist = 2; dist <= 8; dist <<= 1)
    {   // :450-527: 2, 1, 3, 4, 5, 6, 8, 7 -- the diagonal points at half the distance
        const int saved = s.bcost, h = dist >> 1;
        const int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist, top2 = oy - h, bottom2 = oy + h, left2 = ox - h, right2 = ox + h;
        const int X[8] = { ox, left2, right2, left, right, left2, right2, ox }, Y[8] = { top, top2, top2, oy, oy, bottom2, bottom2, bottom };
        const bool ok[8] = { top >= mny, top2 >= mny && left2 >= mnx, top2 >= mny && right2 <= mxx, left >= mnx, right <= mxx, bottom2 <= mxy && left2 >= mnx, bottom2 <= mxy && right2 <= mxx, bottom <= mxy };
        const int nr[8] = { 2, 1, 3, 4, 5, 6, 8, 7 }, ds[8] = { dist, h, h, dist, dist, h, h, dist };
        star_batch<8>(c, s, ox, oy, X, Y, ok, nr, ds);
        if (s.bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
    for (int dist = 16; dist <= (int16_t)merange; dist <<= 1)
    {   // :529-628: top, left, right, bottom, then three rings of four points between them; point number 0
        const int saved = s.bcost, q = dist >> 2;
        const int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        {
            const int yt = top + q, yb = bottom - q, xl = ox - q, xr = ox + q;
            const int X[8] = { ox, left, right, ox, xl, xr, xl, xr }, Y[8] = { top, oy, oy, bottom, yt, yt, yb, yb };
            const bool ok[8] = { top >= mny, left >= mnx, right <= mxx, bottom <= mxy, yt >= mny && xl >= mnx, yt >= mny && xr <= mxx, yb <= mxy && xl >= mnx, yb <= mxy && xr <= mxx };
            const int nr[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, ds[8] = { dist, dist, dist, dist, dist, dist, dist, dist };
            star_batch<8>(c, s, ox, oy, X, Y, ok, nr, ds);
        }
        {
            const int yt2 = top + 2 * q, yb2 = bottom - 2 * q, xl2 = ox - 2 * q, xr2 = ox + 2 * q, yt3 = top + 3 * q, yb3 = bottom - 3 * q, xl3 = ox - 3 * q, xr3 = ox + 3 * q;
            const int X[8] = { xl2, xr2, xl2, xr2, xl3, xr3, xl3, xr3 }, Y[8] = { yt2, yt2, yb2, yb2, yt3, yt3, yb3, yb3 };
            const bool ok[8] = { yt2 >= mny && xl2 >= mnx, yt2 >= mny && xr2 <= mxx, yb2 <= mxy && xl2 >= mnx, yb2 <= mxy && xr2 <= mxx,
                                 yt3 >= mny && xl3 >= mnx, yt3 >= mny && xr3 <= mxx, yb3 <= mxy && xl3 >= mnx, yb3 <= mxy && xr3 <= mxx };
            const int nr[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, ds[8] = { dist, dist, dist, dist, dist, dist, dist, dist };
            star_batch<8>(c, s, ox, oy, X, Y, ok, nr, ds);
        }
        if (s.bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
}
__device__ __forceinline__ void lowres_star(const Blk& c, int mnx, int mny, int mxx, int mxy, int merange, int& bx, int& by, int& bcost)
{
    auto twoPoints = [&](int nr) {   // offsets[] (motion.cpp:75-85): the two outer neighbours of direction nr = 1..8, packed (value + 1) in 2 bits, x then y
        const unsigned px = 0x684a0884u, py = 0x9a982501u;
        const int i = (nr - 1) * 2;
        const int ax = bx + (int)((px >> (2 * i)) & 3) - 1, ay = by + (int)((py >> (2 * i)) & 3) - 1, bx2 = bx + (int)((px >> (2 * i + 2)) & 3) - 1, by2 = by + (int)((py >> (2 * i + 2)) & 3) - 1;
        const bool oka = ax >= mnx && ax <= mxx && ay >= mny && ay <= mxy, okb = bx2 >= mnx && bx2 <= mxx && by2 >= mny && by2 <= mxy;
        const int QX[2] = { (oka ? ax : bx) * 4, (okb ? bx2 : bx) * 4 }, QY[2] = { (oka ? ay : by) * 4, (okb ? by2 : by) * 4 };
        int C[2];
        eval<2, false>(c, QX, QY, C);
        const int c0 = C[0] + mvcost(c, QX[0], QY[0]), c1 = C[1] + mvcost(c, QX[1], QY[1]);
        if (oka && c0 < bcost) { bcost = c0; bx = ax; by = ay; }
        if (okb && c1 < bcost) { bcost = c1; bx = bx2; by = by2; }
    };
    StarSt s = { bx, by, bcost, 0, 0 };
    lowres_star_pattern(c, mnx, mny, mxx, mxy, s, 3, merange);
    bx = s.bx; by = s.by; bcost = s.bcost;
    if (s.dist == 1)
    {   // :1335-1364: the two missing points; no new best ends the search
        if (!s.pointNr) return;
        const int saved = bcost;
        twoPoints(s.pointNr);
        if (bcost == saved) return;
    }
    if (s.dist > 5)
    {   // raster refinement over the WHOLE window, every fifth position (:1366-1401): four placements of a row at a time while four fit (the fourth one's mv cost at
        // tmv << 3, :1392), then the rest of the row one by one
        for (int ty = mny; ty <= mxy; ty += 5)
            for (int tx = mnx; tx <= mxx; tx += 20)
            {
                const bool four = tx + 15 <= mxx;
                int X[4], Y[4], C[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { X[k] = min(tx + 5 * k, mxx) * 4; Y[k] = ty * 4; }
                eval<4, false>(c, X, Y, C);
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const int mvc = (k == 3 && four) ? (int)(uint16_t)((int)c.gcost[2 * X[k] - c.mvpx] + (int)c.gcost[2 * Y[k] - c.mvpy]) : mvcost(c, X[k], Y[k]);
                    const int cost = C[k] + mvc;
                    if (tx + 5 * k <= mxx && cost < bcost) { bcost = cost; bx = tx + 5 * k; by = ty; }
                }
            }
    }
    int dist = s.dist;
    while (dist > 0)
    {   // :1403-1434: a new search centred on the best so far, until the best distance is 0; distance 1 ends it with the two missing points
        s.bx = bx; s.by = by; s.bcost = bcost; s.pointNr = 0; s.dist = 0;
        lowres_star_pattern(c, mnx, mny, mxx, mxy, s, 32, merange);
        bx = s.bx; by = s.by; bcost = s.bcost; dist = s.dist;
        if (dist == 1)
        {
            if (s.pointNr) twoPoints(s.pointNr);
            break;
        }
    }
}

// MotionEstimate::motionEstimate for a lowres reference (no candidates, subme 1; the hexagon search, or -- with --hme -- the level's method: hexagon or uneven
// multi-hexagon, motion.cpp:1013): returns the cost, MV in (ox, oy)
__device__ __forceinline__ int lowres_me(Blk& c, int mnx, int mny, int mxx, int mxy, int mvpx, int mvpy, int& ox, int& oy, int merange = LA_MERANGE, int method = X265HIP_ME_HEX)
{
    const bool umh = method == X265HIP_ME_UMH;
    c.mvpx = mvpx; c.mvpy = mvpy;
    const int qmnx = mnx * 4, qmny = mny * 4, qmxx = mxx * 4, qmxy = mxy * 4;
    const int pmx = min(max(mvpx, qmnx), qmxx), pmy = min(max(mvpy, qmny), qmxy);
    int bx = (pmx + 2) >> 2, by = (pmy + 2) >> 2, bcost, bprecost;
    {   // motion.cpp:966-988: clipped MVP (sub-pel SAD, no mv cost), rounded MVP, zero MV
        const int X[3] = { pmx, bx * 4, 0 }, Y[3] = { pmy, by * 4, 0 };
        int C[3];
        eval<3, false>(c, X, Y, C);
        bprecost = bcost = C[0];
        if ((pmx | pmy) & 3) bcost = C[1] + mvcost(c, bx * 4, by * 4);
        if (pmx | pmy)
        {
            const int cost = C[2] + mvcost(c, 0, 0);
            if (cost < bcost) { bcost = cost; bx = 0; by = max(min(0, mxy), mny); }
        }
    }
    if (bcost == 0) { ox = bx * 4; oy = by * 4; return mvcost(c, ox, oy); }
    auto inY = [&](int y) { return (y >= mny) & (y <= mxy); };
    bool hexToo = true;
    if (method == X265HIP_ME_DIA)
    {   // diamond search, radius 1 (motion.cpp:1016-1039): the four neighbours costed together; the row test decides which may win; the step sits in the low bits of the cost
        int i = merange;
        bcost <<= 4;
        do
        {
            const int X[4] = { bx * 4, bx * 4, (bx - 1) * 4, (bx + 1) * 4 }, Y[4] = { (by - 1) * 4, (by + 1) * 4, by * 4, by * 4 };
            int C[4];
            eval<4, false>(c, X, Y, C);
#pragma unroll
            for (int k = 0; k < 4; k++) C[k] += mvcost(c, X[k], Y[k]);
            if (inY(by - 1)) bcost = min(bcost, (C[0] << 4) + 1);
            if (inY(by + 1)) bcost = min(bcost, (C[1] << 4) + 3);
            bcost = min(bcost, (C[2] << 4) + 4);
            bcost = min(bcost, (C[3] << 4) + 12);
            if (!(bcost & 15)) break;
            bx -= (int32_t)((uint32_t)bcost << 28) >> 30;
            by -= (int32_t)((uint32_t)bcost << 30) >> 30;
            bcost &= ~15;
        }
        while (--i && bx >= mnx && bx <= mxx && by >= mny && by <= mxy);
        bcost >>= 4;
        hexToo = false;
    }
    else if (method == X265HIP_ME_FULL)
    {   // exhaustive search of an --hme reference (motion.cpp:1593-1632): the window cut to +-merange around the ZERO vector (:1598-1605), raster order, strict `<`
        const int r = merange < 0 ? -merange : merange;
        const int y0 = max(mny, -r), x0 = max(mnx, -r), y1 = min(mxy, r), x1 = min(mxx, r);
        for (int ty = y0; ty <= y1; ty++)
            for (int tx = x0; tx <= x1; tx += 4)
            {
                int X[4], Y[4], C[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { X[k] = min(tx + k, x1) * 4; Y[k] = ty * 4; }
                eval<4, false>(c, X, Y, C);
#pragma unroll
                for (int k = 0; k < 4; k++)
                {
                    const int cost = C[k] + mvcost(c, X[k], Y[k]);
                    if (tx + k <= x1 && cost < bcost) { bcost = cost; bx = tx + k; by = ty; }
                }
            }
        hexToo = false;
    }
    else if (method == X265HIP_ME_STAR) { lowres_star(c, mnx, mny, mxx, mxy, merange, bx, by, bcost); hexToo = false; }
    if (umh) hexToo = lowres_umh(c, (pmx + 2) >> 2, (pmy + 2) >> 2, merange, mnx, mny, mxx, mxy, bx, by, bcost);
    if (hexToo)
    {
    {   // hexagon search (motion.cpp:1041-1140)
        int X[6], Y[6], C[6];
#pragma unroll
        for (int k = 0; k < 6; k++) { X[k] = (bx + k_hex2(k + 1, 0)) * 4; Y[k] = (by + k_hex2(k + 1, 1)) * 4; }     // (-2,0) (-1,2) (1,2) (2,0) (1,-2) (-1,-2)
        eval<6, false>(c, X, Y, C);
#pragma unroll
        for (int k = 0; k < 6; k++) C[k] += mvcost(c, X[k], Y[k]);
        bcost <<= 3;
        if (inY(by)) bcost = min(bcost, (C[0] << 3) + 2);
        if (inY(by + 2)) { bcost = min(bcost, (C[1] << 3) + 3); bcost = min(bcost, (C[2] << 3) + 4); }
        if (inY(by)) bcost = min(bcost, (C[3] << 3) + 5);
        if (inY(by - 2)) { bcost = min(bcost, (C[4] << 3) + 6); bcost = min(bcost, (C[5] << 3) + 7); }
    }
    if (bcost & 7)
    {
        int dir = (bcost & 7) - 2;
        if (inY(by + k_hex2(dir + 1, 1)))
        {
            bx += k_hex2(dir + 1, 0); by += k_hex2(dir + 1, 1);
            for (int i = (merange >> 1) - 1; i > 0 && bx >= mnx && bx <= mxx && inY(by); i--)
            {
                int X[3], Y[3], C[3];
#pragma unroll
                for (int k = 0; k < 3; k++) { X[k] = (bx + k_hex2(dir + k, 0)) * 4; Y[k] = (by + k_hex2(dir + k, 1)) * 4; }
                eval<3, false>(c, X, Y, C);
                bcost &= ~7;
#pragma unroll
                for (int k = 0; k < 3; k++)
                    if (inY(by + k_hex2(dir + k, 1))) bcost = min(bcost, ((C[k] + mvcost(c, X[k], Y[k])) << 3) + k + 1);
                if (!(bcost & 7)) break;
                dir += (bcost & 7) - 2;
                dir = k_mod6m1(dir + 1);
                bx += k_hex2(dir + 1, 0); by += k_hex2(dir + 1, 1);
            }
        }
    }
    bcost >>= 3;
    {   // square refine (:1116-1138)
        int X[8], Y[8], C[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { X[k] = (bx + k_sq(k + 1, 0)) * 4; Y[k] = (by + k_sq(k + 1, 1)) * 4; }
        eval<8, false>(c, X, Y, C);
        int dir = 0;
#pragma unroll
        for (int k = 0; k < 8; k++)
        {
            const int cost = C[k] + mvcost(c, X[k], Y[k]);
            if ((k_sq(k + 1, 1) == 0 || inY(by + k_sq(k + 1, 1))) && cost < bcost) { bcost = cost; dir = k + 1; }
        }
        bx += k_sq(dir, 0); by += k_sq(dir, 1);
    }
    }
    // motion.cpp:1644-1699
    int qx, qy;
    if (bprecost < bcost) { qx = pmx; qy = pmy; bcost = bprecost; }
    else { qx = bx * 4; qy = by * 4; }
    int zeroSatd = -1;
    if (!bcost)
        bcost = mvcost(c, qx, qy);
    else
    {   // lowres branch: 4 half-pel directions at SAD, re-measure at SATD, 4 quarter-pel directions at SATD
        int X[4], Y[4], C[4], bdir = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { X[k] = qx + k_sq(k + 1, 0) * 2; Y[k] = qy + k_sq(k + 1, 1) * 2; }
        eval<4, false>(c, X, Y, C);
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            if ((Y[k] < qmny) | (Y[k] > qmxy)) continue;
            const int cost = C[k] + mvcost(c, X[k], Y[k]);
            if (cost < bcost) { bcost = cost; bdir = k + 1; }
        }
        qx += k_sq(bdir, 0) * 2; qy += k_sq(bdir, 1) * 2;
        if (c.zbase)
        {
            const int X1[1] = { qx }, Y1[1] = { qy };
            int C1[1];
            eval<1, true>(c, X1, Y1, C1);
            bcost = C1[0] + mvcost(c, qx, qy);
        }
        else
        {   // the zero-MV SATD of the final check rides along
            const int X2[2] = { qx, 0 }, Y2[2] = { qy, 0 };
            int C2[2];
            eval<2, true>(c, X2, Y2, C2);
            bcost = C2[0] + mvcost(c, qx, qy); zeroSatd = C2[1];
        }
        bdir = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { X[k] = qx + k_sq(k + 1, 0); Y[k] = qy + k_sq(k + 1, 1); }
        eval<4, true>(c, X, Y, C);
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
            if ((Y[k] < qmny) | (Y[k] > qmxy)) continue;
            const int cost = C[k] + mvcost(c, X[k], Y[k]);
            if (cost < bcost) { bcost = cost; bdir = k + 1; }
        }
        qx += k_sq(bdir, 0); qy += k_sq(bdir, 1);
    }
    if (qx | qy)
    {   // motion.cpp:1763-1768 (the cost is NOT replaced when the zero MV wins)
        if (c.zbase) zeroSatd = satd_rows_pk(c.fenc, ld_row(c.zbase, c.zref), c.lane);
        else if (zeroSatd < 0)
        {
            const int X1[1] = { 0 }, Y1[1] = { 0 };
            int C1[1];
            eval<1, true>(c, X1, Y1, C1);
            zeroSatd = C1[0];
        }
        if (zeroSatd + mvcost(c, 0, 0) <= bcost) { qx = 0; qy = 0; }
    }
    ox = qx; oy = qy;
    return bcost;
}

struct LaGeom { const pixel* lowres; int64_t planeElems; intptr_t stride; int64_t origin; int wcu, hcu; };

__device__ __forceinline__ const pixel* plane_of(const LaGeom& g, int frame, int k) { return g.lowres + ((int64_t)frame * 4 + k) * g.planeElems + g.origin; }
// element offset of pixel (0,0) of plane 0 of a picture inside the lowres buffer (the whole buffer is < 2^31 elements, checked on the host)
__device__ __forceinline__ uint32_t plane0_off(const LaGeom& g, int frame) { return (uint32_t)((int64_t)frame * 4 * g.planeElems + g.origin); }

// packed MV (x low, y high 16 bits), exchanged between the wavefronts of a workgroup through global memory
__device__ __forceinline__ uint32_t ld_mv(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_mv(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// a task the host could not look at (the list lives in device memory): pictures inside the buffer and b not its own list-0 reference, else the estimate is skipped.
// The indices are places in the lowres buffer, not display order (a caller that keeps pictures in reusable slots has none): p1 == b marks a P estimate.
__device__ __forceinline__ bool la_task_ok(const x265hip_la_task* tp, int nFrames)
{
    const int w0 = tp->weighted0;
    return tp->p0 >= 0 && tp->p0 < nFrames && tp->b >= 0 && tp->b < nFrames && tp->p1 >= 0 && tp->p1 < nFrames && tp->p0 != tp->b && w0 >= 0 && w0 <= nFrames;
}

// --hme (slicetype.cpp:4430-4437, 4483-4575): the same sweep runs first on the quarter-resolution pictures (level 0: its own geometry g, range hmeRange[0], method
// hmeSearchMethod[0], never the weighted copy) into its own MV / cost slots; the half-resolution sweep (level 1: hmeRange[1], hmeSearchMethod[1]) then takes twice the
// level-0 MV of the block above it as a fifth predictor candidate.  hme5Mvs == NULL: no such candidate (no HME, or level 0 itself).  gz.lowres != NULL on level 0 only: the
// half-resolution pictures, for the zero-MV check (Blk::zbase).
__global__ __launch_bounds__(1024) void la_search_kernel(LaGeom g, const x265hip_la_task* __restrict__ tasks, int nFrames, const uint16_t* __restrict__ costCentre, int costR, int rowsPerSlice,
                                                         uint32_t* mvs, int32_t* mvCosts, int merange, int umh /* the level's search method (X265HIP_ME_*) */, int useWeighted,
                                                         const uint32_t* __restrict__ hme5Mvs, const int32_t* __restrict__ hme5Costs, int hme5N, LaGeom gz)
{
    const x265hip_la_task* tp = tasks + (blockIdx.x >> 1);
    if (!la_task_ok(tp, nFrames)) return;
    const int list = blockIdx.x & 1;
    const int tb = tp->b, tp0 = tp->p0, tp1 = tp->p1;
    const bool bidir = tp1 != tb;
    if ((list && !bidir) || !tp->doSearch[list]) return;
    const int W = g.wcu, H = g.hcu, ncu = W * H;
    const int grp = threadIdx.x >> 3, ngrp = blockDim.x >> 3, lane = threadIdx.x & 7;
    const int slot = tp->mvSlot[list];
    uint32_t* mv = mvs + (int64_t)slot * ncu;
    int32_t* mvCost = mvCosts + (int64_t)slot * ncu;
    const int w0 = tp->weighted0;
    const int refFrame = list ? tp1 : ((w0 > 0 && useWeighted) ? w0 - 1 : tp0);
    HIP_DYNAMIC_SHARED(uint16_t, s_cost)                       // 2 * costR + 1 entries of the cost row
    for (int i = threadIdx.x; i <= 2 * costR; i += blockDim.x) s_cost[i] = costCentre[i - costR];
    __syncthreads();
    const uint32_t fencPlane = plane0_off(g, tb), rp = plane0_off(g, refFrame);
    const int stride = (int)g.stride;

    // cooperative slices (--lookahead-slices, slicetype.cpp:1173-1176, 4347-4357): blockIdx.y = slice; a slice is its own sweep, its bottom row
    // takes no predictors from below (lastRow), the last slice also takes the remainder rows
    const int nslices = H / rowsPerSlice, sl = blockIdx.y;
    const int firstY = sl * rowsPerSlice, lastY = sl == nslices - 1 ? H - 1 : firstY + rowsPerSlice - 1, HS = lastY - firstY + 1;
    const int steps = W + 2 * (HS - 1);
    for (int s = 0; s < steps; s++)
    {
        const int jmin = max(0, (s - W + 2) >> 1), jmax = min(HS - 1, s >> 1);
        for (int j = jmin + grp; j <= jmax; j += ngrp)
        {
            const int cuY = lastY - j, cuX = W - 1 - (s - 2 * j), cuXY = cuX + cuY * W;
            const bool lastRow = j == 0;
            const uint32_t pel = (uint32_t)(CU * cuX + __mul24(CU * cuY + lane, stride));
            Blk c;
            c.lane = lane; c.stride = stride; c.base = g.lowres; c.lcost = s_cost + costR; c.mvpx = 0; c.mvpy = 0; c.gcost = costCentre;
            c.fenc = ld_row(g.lowres, fencPlane + pel);
            c.ref0 = rp + pel; c.pe = (uint32_t)g.planeElems;
            c.zbase = gz.lowres; c.zref = gz.lowres ? plane0_off(gz, list ? tp1 : tp0) + (uint32_t)(CU * cuX + __mul24(CU * cuY, stride) + __mul24(lane, (int)gz.stride)) : 0u;
            const int mnx = -cuX * CU - 8, mny = -cuY * CU - 8, mxx = (W - cuX - 1) * CU + 8, mxy = (H - cuY - 1) * CU + 8;
            // reverse-order MV prediction (slicetype.cpp:4520-4536): right, below, below-left, below-right; with --hme twice the quarter-resolution MV (:4537-4540)
            constexpr int NC = 5;
            bool valid[NC] = { cuX < W - 1, !lastRow, !lastRow && cuX > 0, !lastRow && cuX < W - 1, false };
            const int where[4] = { 1, W, W - 1, W + 1 };
            int X[NC], Y[NC];
#pragma unroll
            for (int k = 0; k < 4; k++)
            {
                const uint32_t v = valid[k] ? ld_mv(mv + cuXY + where[k]) : 0u;
                X[k] = (int16_t)(v & 0xffff); Y[k] = (int16_t)(v >> 16);
            }
            X[4] = Y[4] = 0;
            if (hme5Mvs)
            {   // cuXY_4x4 = (cuX / 2) + (cuY / 2) * widthInCU / 2 -- with the HALF-resolution width, as the reference writes it
                const int i4 = (cuX / 2) + ((cuY / 2) * W) / 2;
                if (i4 < hme5N && hme5Costs[(int64_t)slot * hme5N + i4] > 0)
                {
                    const uint32_t v = hme5Mvs[(int64_t)slot * hme5N + i4];
                    valid[4] = true; X[4] = 2 * (int)(int16_t)(v & 0xffff); Y[4] = 2 * (int)(int16_t)(v >> 16);
                }
            }
            int mvpx = 0, mvpy = 0, skipCost = 0x7fffffff;
            if (valid[0] | valid[1] | valid[4])
            {   // the candidate with the lowest SATD becomes the predictor (:4541-4556).  Neighbours mostly agree: a candidate
                // equal to an earlier one reuses its cost, and a slot nobody in the wavefront needs is skipped (uniform branches)
                int C[NC] = { 0, 0, 0, 0, 0 };
                bool need[NC]; Row r[NC];
#pragma unroll
                for (int k = 0; k < NC; k++)
                {
                    bool dup = false;
#pragma unroll
                    for (int j = 0; j < k; j++) dup |= valid[j] && X[j] == X[k] && Y[j] == Y[k];
                    need[k] = __builtin_amdgcn_ballot_w64(valid[k] && !dup) != 0;
                    if (need[k]) r[k] = mc_row(c, X[k], Y[k]);
                }
#pragma unroll
                for (int k = 0; k < NC; k++)
                {
                    if (need[k]) C[k] = satd_rows_pk(c.fenc, r[k], lane);
#pragma unroll
                    for (int j = k - 1; j >= 0; j--) if (valid[j] && X[j] == X[k] && Y[j] == Y[k]) C[k] = C[j];
                }
                int mvpcost = COST_MAX;
#pragma unroll
                for (int k = 0; k < NC; k++)
                {
                    if (!valid[k]) continue;
                    if (C[k] < mvpcost) { mvpcost = C[k]; mvpx = X[k]; mvpy = Y[k]; }
                    if (!(mvpx | mvpy) && bidir) skipCost = C[k];
                }
            }
            int ox, oy;
            int fencCost = lowres_me(c, mnx, mny, mxx, mxy, mvpx, mvpy, ox, oy, merange, umh);
            if (skipCost < 64 && skipCost < fencCost && bidir) { fencCost = skipCost; ox = 0; oy = 0; }
            if (lane == 0)
            {
                st_mv(mv + cuXY, (uint32_t)(uint16_t)ox | ((uint32_t)(uint16_t)oy << 16));
                mvCost[cuXY] = fencCost;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void la_zero_kernel(const x265hip_la_task* __restrict__ tasks, int nTasks, unsigned long long* sums)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nTasks * 3) sums[(int64_t)tasks[i / 3].outSlot * 3 + i % 3] = 0;
}

// workgroup-wide sum of three per-thread values -> out[0..2] valid in thread 0 (the totals of one block row)
__device__ __forceinline__ void row_totals(int a, int b, int c, int (&out)[3])
{
    __shared__ int s_red[4][3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    if (lane == 0) { s_red[wave][0] = a; s_red[wave][1] = b; s_red[wave][2] = c; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 0; k < 3; k++) out[k] = s_red[0][k] + s_red[1][k] + s_red[2][k] + s_red[3][k];
}

// the decision half of estimateCUCost (:4574-4640): one workgroup per (block row, estimate), 8 lanes per block; the row's
// totals are reduced inside the workgroup, so the frame sums see one atomic per row instead of one per block
__global__ __launch_bounds__(256) void la_finish_kernel(LaGeom g, const x265hip_la_task* __restrict__ tasks, int nFrames, const uint32_t* __restrict__ mvs,
                                                        const int32_t* __restrict__ mvCosts, const int32_t* __restrict__ intraCost,
                                                        const int32_t* __restrict__ invQscale, uint16_t* __restrict__ lowresCosts,
                                                        int32_t* __restrict__ rowSatds, unsigned long long* sums)
{
    const x265hip_la_task* tp = tasks + blockIdx.y;
    if (!la_task_ok(tp, nFrames)) return;
    const int tb = tp->b, tp0 = tp->p0, tp1 = tp->p1, slot0 = tp->mvSlot[0], slot1 = tp->mvSlot[1], outSlot = tp->outSlot;
    const int W = g.wcu, H = g.hcu, ncu = W * H;
    const int cuY = blockIdx.x, lane = threadIdx.x & 7;
    const bool bidir = tp1 != tb;
    int accCost = 0, accAq = 0, accRow = 0, accIntra = 0;
    for (int cuX = threadIdx.x >> 3; cuX < W; cuX += 32)
    {
        const int cuXY = cuX + cuY * W;
        const uint32_t pel = (uint32_t)(CU * cuX + __mul24(CU * cuY + lane, (int)g.stride));
        int bcost = COST_MAX, listused = 0;
        {
            const int c0 = mvCosts[(int64_t)slot0 * ncu + cuXY];
            if (c0 < bcost) { bcost = c0; listused = 1; }
            if (bidir)
            {
                const int c1 = mvCosts[(int64_t)slot1 * ncu + cuXY];
                if (c1 < bcost) { bcost = c1; listused = 2; }
            }
        }
        if (bidir)
        {
            Blk c0, c1;
            c0.lane = c1.lane = lane; c0.stride = c1.stride = (int)g.stride; c0.base = c1.base = g.lowres;
            c0.fenc = ld_row(g.lowres, plane0_off(g, tb) + pel);
            c0.ref0 = plane0_off(g, tp0) + pel; c1.ref0 = plane0_off(g, tp1) + pel; c0.pe = c1.pe = (uint32_t)g.planeElems;
            const uint32_t m0 = mvs[(int64_t)slot0 * ncu + cuXY], m1 = mvs[(int64_t)slot1 * ncu + cuXY];
            const Row a = avg_rows(mc_row(c0, (int16_t)(m0 & 0xffff), (int16_t)(m0 >> 16)), mc_row(c1, (int16_t)(m1 & 0xffff), (int16_t)(m1 >> 16)));   // avg(l0-mv, l1-mv)
            const Row z = avg_rows(ld_row(c0.base, c0.ref0), ld_row(c1.base, c1.ref0));                                                                              // co-located
            int bicost = satd_rows_pk(c0.fenc, a, lane);
            if (bicost < bcost) { bcost = bicost; listused = 3; }
            bicost = satd_rows_pk(c0.fenc, z, lane);
            if (bicost < bcost) { bcost = bicost; listused = 3; }
            bcost += 4;                                        // lowresPenalty
        }
        else
        {
            bcost += 4;
            const int ic = intraCost[(int64_t)tb * ncu + cuXY];
            if (ic < bcost) { bcost = ic; listused = 0; }
        }
        if (lane == 0)
        {
            const bool score = (cuX > 0 && cuX < W - 1 && cuY > 0 && cuY < H - 1) || W <= 2 || H <= 2;
            const int bcostAq = (score && invQscale) ? ((bcost * invQscale[(int64_t)tb * ncu + cuXY] + 128) >> 8) : bcost;
            if (score) { accCost += bcost; accAq += bcostAq; accIntra += (!listused && !bidir) ? 1 : 0; }
            accRow += bcostAq;
            lowresCosts[(int64_t)outSlot * ncu + cuXY] = (uint16_t)(min(bcost, LOWRES_COST_MASK) | (listused << LOWRES_COST_SHIFT));
        }
    }
    // a row holds at most a few thousand blocks of cost < 2^18: 32-bit partial sums are safe
    int t3[3];
    row_totals(accCost, accAq, accRow, t3);
    __syncthreads();
    int t1[3];
    row_totals(accIntra, 0, 0, t1);
    if (threadIdx.x == 0)
    {
        rowSatds[(int64_t)outSlot * H + cuY] = t3[2];
        unsigned long long* sm = sums + (int64_t)outSlot * 3;
        if (t3[0]) atomicAdd(sm, (unsigned long long)t3[0]);
        if (t3[1]) atomicAdd(sm + 1, (unsigned long long)t3[1]);
        if (t1[0]) atomicAdd(sm + 2, (unsigned long long)t1[0]);
    }
}

// ---- intra cost of every 8x8 block (slicetype.cpp:755-864): one wavefront per block, lane = pixel ----
__device__ const int8_t k_angleTable[17] = { -32, -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
__device__ const int16_t k_invAngleTable[8] = { 4096, 1638, 910, 630, 482, 390, 315, 256 };
__device__ const uint8_t k_intraFilterFlags[35] = {   // constants.cpp:561
    0x38, 0x00, 0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30,
    0x38, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x20, 0x00, 0x20, 0x30, 0x30, 0x30, 0x30, 0x30, 0x30, 0x38 };

// satd of an 8x8 whose pixel (lane & 7, lane >> 3) difference is d: 4x4 Hadamards across lane bits 0,1 (x) and 3,4 (y)
__device__ __forceinline__ int satd8x8_px(int d, int lane)
{
    int v = d;
    v = LA_DPP(v, 0xB1) + ((lane & 1) ? -v : v);
    v = LA_DPP(v, 0x4E) + ((lane & 2) ? -v : v);
    v = LA_DPP(v, 0x128) + ((lane & 8) ? -v : v);                                   // row_ror:8 = lane ^ 8 inside a row of 16
    v = __builtin_amdgcn_ds_swizzle(v, 0x401F) + ((lane & 16) ? -v : v);            // lane ^ 16
    const int a = abs(v);
    const int top = wave_sum(lane < 32 ? a : 0), bot = wave_sum(lane < 32 ? 0 : a); // the two 8x4 halves
    return (top >> 1) + (bot >> 1);
}

__global__ __launch_bounds__(256) void la_intra_kernel(LaGeom g, const int32_t* __restrict__ invQscale, int penalty,
                                                       int32_t* __restrict__ intraCost, uint8_t* __restrict__ intraMode, uint16_t* __restrict__ lowresCosts,
                                                       int32_t* rowSatds, unsigned long long* sums)
{
    constexpr int N = CU, n2 = 2 * N, NB = 4 * N + 1;
    __shared__ pixel s_nbAll[4][2][NB + 3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int W = g.wcu, H = g.hcu, ncu = W * H;
    const int frame = blockIdx.y, cuY = blockIdx.x;            // one workgroup per block row: its four wavefronts take every fourth block
    pixel (*s_nb)[NB + 3] = s_nbAll[wave];
    int accCost = 0, accAq = 0, accRow = 0;
    for (int cuX = wave; cuX < W; cuX += 4)
    {
    const int cuXY = cuX + cuY * W, item = frame * ncu + cuXY;
    const pixel* cur = plane_of(g, frame, 0) + (intptr_t)CU * cuX + (intptr_t)CU * cuY * g.stride;
    const int x = lane & 7, y = lane >> 3;
    const int f = cur[(intptr_t)y * g.stride + x];
    {   // reference samples straight from the source plane (:795-799), then the [1 2 1] smoothing (intrapred.cpp:31-51)
        const pixel* tl = cur - g.stride - 1;
        if (lane <= 4 * N) s_nb[0][lane] = lane <= n2 ? tl[lane] : tl[(intptr_t)(lane - n2) * g.stride];
        wave_sync();
        if (lane <= 4 * N)
        {
            const pixel* s = s_nb[0];
            const int i = lane, t = s[0];
            int v;
            if (i == 0) v = (2 * t + s[1] + s[n2 + 1] + 2) >> 2;
            else if (i == n2 || i == 2 * n2) v = s[i];
            else if (i == n2 + 1) v = (2 * s[i] + t + s[i + 1] + 2) >> 2;
            else v = (2 * s[i] + s[i - 1] + s[i + 1] + 2) >> 2;
            s_nb[1][i] = (pixel)v;
        }
        wave_sync();
    }
    auto nbAt = [&](const pixel* a, int i, bool flip) -> int { return a[flip && i >= 1 ? (i <= n2 ? i + n2 : i - n2) : i]; };
    auto modeCost = [&](int mode) -> int {
        // the prediction of pixel (x, y); horizontal modes (< 18) are the transpose of the vertical construction on the
        // flipped neighbour array (intrapred.cpp:111-120, 192-203): evaluate that construction at (y, x)
        const bool hor = mode >= 2 && mode < 18;
        const int px = hor ? y : x, py = hor ? x : y;
        int p;
        if (mode == 0)
        {   // planar on the filtered samples (intrapred.cpp:87-100; :812)
            const pixel* a = s_nb[1];
            p = ((N - 1 - x) * a[n2 + 1 + y] + (N - 1 - y) * a[1 + x] + (x + 1) * a[1 + N] + (y + 1) * a[n2 + 1 + N] + N) >> 4;
        }
        else if (mode == 1)
        {   // DC with the edge filter (intrapred.cpp:53-85; :808)
            const pixel* a = s_nb[0];
            int v = lane < n2 ? (lane < N ? a[1 + lane] : a[n2 + 1 + lane - N]) : 0;
            const int dc = (wave_sum(v) + N) / n2;
            p = dc;
            if (x == 0 && y == 0) p = (a[1] + a[n2 + 1] + 2 * dc + 2) >> 2;
            else if (y == 0) p = (a[1 + x] + 3 * dc + 2) >> 2;
            else if (x == 0) p = (a[n2 + 1 + y] + 3 * dc + 2) >> 2;
        }
        else
        {
            const pixel* a = s_nb[(k_intraFilterFlags[mode] & N) ? 1 : 0];
            const int angleOffset = hor ? 10 - mode : mode - 26;
            const int angle = k_angleTable[8 + angleOffset];
            if (!angle)
            {   // pure vertical / horizontal with the edge filter (intrapred.cpp:152-166)
                p = nbAt(a, 1 + px, hor);
                if (px == 0) p = clip_pixel((int16_t)(nbAt(a, 1, hor) + ((nbAt(a, n2 + 1 + py, hor) - (int)a[0]) >> 1)));
            }
            else
            {
                const int invAngle = angle < 0 ? k_invAngleTable[-angleOffset - 1] : 0;
                const int angleSum = (py + 1) * angle, off = angleSum >> 5, frac = angleSum & 31;
                auto ref = [&](int j) -> int { return j >= -1 ? nbAt(a, 1 + j, hor) : nbAt(a, n2 + ((128 + (-1 - j) * invAngle) >> 8), hor); };
                p = frac ? ((32 - frac) * ref(off + px) + frac * ref(off + px + 1) + 16) >> 5 : ref(off + px);
            }
        }
        return satd8x8_px(f - p, lane);
    };
    int icost = COST_MAX, ilow = 0, cost;
    cost = modeCost(1); if (cost < icost) { icost = cost; ilow = 1; }
    cost = modeCost(0); if (cost < icost) { icost = cost; ilow = 0; }
    int acost = COST_MAX, alow = 4;
    for (int mode = 5; mode < 35; mode += 5) { cost = modeCost(mode); if (cost < acost) { acost = cost; alow = mode; } }
    for (int dist = 2; dist >= 1; dist--)
    {
        const int minus = alow - dist, plus = alow + dist;
        cost = modeCost(minus); if (cost < acost) { acost = cost; alow = minus; }
        cost = modeCost(plus); if (cost < acost) { acost = cost; alow = plus; }
    }
    if (acost < icost) { icost = acost; ilow = alow; }
    icost += penalty;
    if (lane == 0)
    {
        intraCost[item] = icost; intraMode[item] = (uint8_t)ilow;
        lowresCosts[item] = (uint16_t)min(icost, LOWRES_COST_MASK);
        const bool score = (cuX > 0 && cuX < W - 1 && cuY > 0 && cuY < H - 1) || W <= 2 || H <= 2;
        const int icostAq = (score && invQscale) ? ((icost * invQscale[item] + 128) >> 8) : icost;
        if (score) { accCost += icost; accAq += icostAq; }
        accRow += icostAq;
    }
    wave_sync();                                               // the next block overwrites this wavefront's neighbour arrays
    }
    int t3[3];
    row_totals(accCost, accAq, accRow, t3);
    if (threadIdx.x == 0)
    {
        rowSatds[(int64_t)frame * H + cuY] = t3[2];
        if (t3[0]) atomicAdd(sums + 2 * frame, (unsigned long long)t3[0]);
        if (t3[1]) atomicAdd(sums + 2 * frame + 1, (unsigned long long)t3[1]);
    }
}

// ---- cuTree: propagate the cost of picture b into its references (slicetype.cpp:3850-3953; pixel.cpp:906-931) ----
// The reference walks the blocks in raster order and adds into the references' uint16 arrays with saturation.  All addends are
// non-negative, so "saturating add after saturating add" equals min(start + sum, 65535): the sums are gathered with 64-bit atomics
// in any order and folded in afterwards.  The per-block amount is the reference's DOUBLE arithmetic operation by operation (no
// contraction into fused multiply-adds, IEEE division), so the truncated integer is the same.
__global__ __launch_bounds__(256) void cutree_scatter_kernel(int W, int H, int bipredWeight0, double fps, int referenced,
                                                             const int32_t* __restrict__ intraCost, const uint16_t* __restrict__ lowresCosts,
                                                             const int32_t* __restrict__ invQscale, const uint32_t* __restrict__ mvs0,
                                                             const uint32_t* __restrict__ mvs1, uint16_t* propB, unsigned long long* acc)
{
#pragma clang fp contract(off)
    const int cuIndex = blockIdx.x * 256 + threadIdx.x, ncu = W * H;
    if (cuIndex >= ncu) return;
    const int blockx = cuIndex % W, blocky = cuIndex / W;
    const int propagateIn = referenced ? (int)propB[cuIndex] : 0;             // unreferenced pictures use (and leave) a zeroed first row (:3866-3867)
    if (!referenced && blocky == 0) propB[cuIndex] = 0;
    const int intra = intraCost[cuIndex];
    const int lc = lowresCosts[cuIndex];
    const int inter = min(intra, lc & LOWRES_COST_MASK);
    const double propagateIntra = (double)(intra * invQscale[cuIndex]);
    const double propagateAmount = (double)propagateIn + propagateIntra * fps;
    const double propagateNum = (double)(intra - inter);
    const double propagateDenom = (double)intra;
    const int amount = (int)(propagateAmount * propagateNum / propagateDenom + 0.5);
    if (amount <= 0) return;                                                  // intra blocks do not propagate
    const int listsUsed = lc >> LOWRES_COST_SHIFT;
    for (int list = 0; list < 2; list++)
    {
        if (!((listsUsed >> list) & 1)) continue;
        int listamount = amount;
        if (listsUsed == 3) listamount = (listamount * (list ? 64 - bipredWeight0 : bipredWeight0) + 32) >> 6;
        const uint32_t mv = (list ? mvs1 : mvs0)[cuIndex];
        unsigned long long* rc = acc + (int64_t)list * ncu;
        if (!mv) { atomicAdd(rc + cuIndex, (unsigned long long)listamount); continue; }
        int x = (int16_t)(mv & 0xffff), y = (int16_t)(mv >> 16);
        const int cux = (x >> 5) + blockx, cuy = (y >> 5) + blocky;
        const int idx0 = cux + cuy * W;
        x &= 31; y &= 31;
        const int wgt[4] = { (32 - y) * (32 - x), (32 - y) * x, y * (32 - x), y * x };
#pragma unroll
        for (int k = 0; k < 4; k++)
        {   // blocks outside the picture receive nothing (:3930-3947)
            const int tx = cux + (k & 1), ty = cuy + (k >> 1);
            if (tx >= 0 && tx < W && ty >= 0 && ty < H)
                atomicAdd(rc + idx0 + (k & 1) + (k >> 1) * W, (unsigned long long)((listamount * wgt[k] + 512) >> 10));
        }
    }
}
__global__ __launch_bounds__(256) void cutree_fold_kernel(int ncu, const unsigned long long* __restrict__ acc, uint16_t* prop0, uint16_t* prop1)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ncu) return;
    const unsigned long long a0 = acc[i], a1 = acc[ncu + i];
    if (a0) prop0[i] = (uint16_t)min((unsigned long long)prop0[i] + a0, 65535ull);
    if (a1) prop1[i] = (uint16_t)min((unsigned long long)prop1[i] + a1, 65535ull);
}

// primitives.propagateCost on one row of blocks (pixel.cpp:906-931) and the Q8.8 converters cuTree stores its offsets with (:935-948)
__global__ __launch_bounds__(256) void propagate_row_kernel(int32_t* __restrict__ dst, const uint16_t* __restrict__ propagateIn, const int32_t* __restrict__ intraCosts,
                                                            const uint16_t* __restrict__ interCosts, const int32_t* __restrict__ invQscales, double fps, int len)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= len) return;
    const int intra = intraCosts[i];
    const int inter = min(intra, (int)interCosts[i] & LOWRES_COST_MASK);
    const double propagateIntra = (double)(intra * invQscales[i]);
    const double propagateAmount = (double)propagateIn[i] + propagateIntra * fps;
    const double propagateNum = (double)(intra - inter);
    const double propagateDenom = (double)intra;
    dst[i] = (int)(propagateAmount * propagateNum / propagateDenom + 0.5);
}
__global__ __launch_bounds__(256) void fix8_pack_kernel(uint16_t* __restrict__ dst, const double* __restrict__ src, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (uint16_t)(int16_t)(src[i] * 256.0);
}
__global__ __launch_bounds__(256) void fix8_unpack_kernel(double* __restrict__ dst, const uint16_t* __restrict__ src, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (double)(int16_t)src[i] / 256.0;
}

bool bad_geom(const void* lowres, int64_t planeElems, intptr_t stride, int64_t origin, int wcu, int hcu)
{
    return !lowres || ((uintptr_t)lowres & 3) || planeElems <= 0 || planeElems >= (1 << 24) || stride < wcu * CU || stride >= (1 << 23) || origin < 0 || wcu < 1 || hcu < 1 || origin >= planeElems;
}

} // namespace

extern "C" int x265hip_lookahead_qp(void) { return 12 + 6 * (X265_DEPTH - 8); }      // X265_LOOKAHEAD_QP (common.h:223)

extern "C" int x265hip_lookahead_intra_batch(void* stream, const void* lowres, int64_t planeElems, intptr_t stride, int64_t origin, int widthInCU, int heightInCU,
                                             int nFrames, const int32_t* invQscale, int32_t* intraCost, uint8_t* intraMode, uint16_t* lowresCosts,
                                             int32_t* rowSatds, int64_t* sums)
{
    if (nFrames <= 0) return X265HIP_OK;
    if (bad_geom(lowres, planeElems, stride, origin, widthInCU, heightInCU) || !intraCost || !intraMode || !lowresCosts || !rowSatds || !sums)
    { set_error("lookahead_intra_batch: bad arguments"); return X265HIP_EARG; }
    hipStream_t st = (hipStream_t)stream;
    const int qp = x265hip_lookahead_qp();
    const double lambda = std::floor(std::pow(2.0, (double)qp / 6.0 - 2.0) * (double)(1 << (X265_DEPTH - 8)) * 10000.0 + 0.5) / 10000.0;   // x265_lambda_tab[qp]
    const int penalty = 5 * (int)lambda + 4;                                           // intraPenalty + lowresPenalty (:762-764)
    XH_HIP(hipMemsetAsync(sums, 0, sizeof(int64_t) * 2 * (size_t)nFrames, st));
    const LaGeom g = { (const pixel*)lowres, planeElems, stride, origin, widthInCU, heightInCU };
    XH_KLAUNCH(la_intra_kernel, dim3(heightInCU, nFrames), dim3(256), 0, st, g, invQscale, penalty, intraCost, intraMode, lowresCosts, rowSatds,
                       (unsigned long long*)sums);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

extern "C" int x265hip_lookahead_cost_batch(void* stream, const void* lowres, int64_t planeElems, intptr_t stride, int64_t origin, int widthInCU, int heightInCU,
                                            const x265hip_la_task* tasks, int nTasks, int nFrames, const int32_t* intraCost, const int32_t* invQscale,
                                            const uint16_t* costRow, int costHalfRange, int rowsPerSlice, int16_t* mvs, int32_t* mvCosts,
                                            uint16_t* lowresCosts, int32_t* rowSatds, int64_t* sums)
{
    return x265hip_lookahead_cost_batch_hme(stream, lowres, planeElems, stride, origin, widthInCU, heightInCU, tasks, nTasks, nFrames, intraCost, invQscale, costRow, costHalfRange,
                                            rowsPerSlice, mvs, mvCosts, lowresCosts, rowSatds, sums, nullptr);
}

extern "C" int x265hip_lookahead_cost_batch_hme(void* stream, const void* lowres, int64_t planeElems, intptr_t stride, int64_t origin, int widthInCU, int heightInCU,
                                                const x265hip_la_task* tasks, int nTasks, int nFrames, const int32_t* intraCost, const int32_t* invQscale,
                                                const uint16_t* costRow, int costHalfRange, int rowsPerSlice, int16_t* mvs, int32_t* mvCosts,
                                                uint16_t* lowresCosts, int32_t* rowSatds, int64_t* sums, const x265hip_la_hme* hme)
{
    if (nTasks <= 0) return X265HIP_OK;
    if (hme)
    {
        if (bad_geom(hme->lowerRes, hme->planeElems, hme->stride, hme->origin, hme->widthInCU, hme->heightInCU) || !hme->mvs || !hme->mvCosts || ((uintptr_t)hme->mvs & 3))
        { set_error("lookahead_cost_batch_hme: bad quarter-resolution arguments"); return X265HIP_EARG; }
        for (int l = 0; l < 2; l++)
            if ((hme->method[l] != X265HIP_ME_DIA && hme->method[l] != X265HIP_ME_HEX && hme->method[l] != X265HIP_ME_UMH && hme->method[l] != X265HIP_ME_STAR && hme->method[l] != X265HIP_ME_FULL) || hme->range[l] < 1 || hme->range[l] > 64)
            { set_error("lookahead_cost_batch_hme: level %d: diamond, hexagon, uneven multi-hexagon, star or exhaustive search, range 1..64", l); return X265HIP_EARG; }
        // the raster of a star level costs one placement in four at twice its quarter-pel vector (motion.cpp:1392): |8 * mv - mvp| < 8 * (size + 8) + 4 * (size + 32)
        if ((hme->method[0] == X265HIP_ME_STAR || hme->method[1] == X265HIP_ME_STAR) && costHalfRange < 12 * max(widthInCU, heightInCU) * CU + 192)
        { set_error("lookahead_cost_batch_hme: cost row too short for a star level at this picture size (need >= %d)", 12 * max(widthInCU, heightInCU) * CU + 192); return X265HIP_EARG; }
        if (rowsPerSlice > 0 && rowsPerSlice < heightInCU) { set_error("lookahead_cost_batch_hme: the cooperative sweep is not offered with HME"); return X265HIP_EARG; }
        if (((int64_t)nFrames) * 4 * hme->planeElems >= ((int64_t)1 << 31)) { set_error("lookahead_cost_batch_hme: quarter-resolution buffer beyond 2^31 elements"); return X265HIP_EARG; }
        if (costHalfRange < 4 * (2 * max(widthInCU, heightInCU) * CU + 64))
        { set_error("lookahead_cost_batch_hme: cost row too short for --hme at this picture size (need >= %d)", 4 * (2 * max(widthInCU, heightInCU) * CU + 64)); return X265HIP_EARG; }
        if (sizeof(uint16_t) * (size_t)(8 * (2 * max(widthInCU, heightInCU) * CU + 64) + 2) > 160 * 1024) { set_error("lookahead_cost_batch_hme: picture too large for the cost row in LDS"); return X265HIP_EARG; }
    }
    if (nFrames <= 0) { set_error("lookahead_cost_batch: nFrames must be the number of pictures in the lowres buffer"); return X265HIP_EARG; }
    if (rowsPerSlice <= 0 || rowsPerSlice > heightInCU) rowsPerSlice = heightInCU;          // one slice
    if (bad_geom(lowres, planeElems, stride, origin, widthInCU, heightInCU) || !tasks || !intraCost || !costRow || !mvs || !mvCosts || !lowresCosts || !rowSatds || !sums)
    { set_error("lookahead_cost_batch: bad arguments"); return X265HIP_EARG; }
    // every MV and predictor lies within the picture + 8, + one block for the neighbour's own window, + 3 of pattern overshoot: |mvd| < 4 * (size + 32) quarter-pels
    if (costHalfRange < 4 * (max(widthInCU, heightInCU) * CU + 32))
    { set_error("lookahead_cost_batch: cost row too short for this picture size (need >= %d)", 4 * (max(widthInCU, heightInCU) * CU + 32)); return X265HIP_EARG; }
    if (((uintptr_t)mvs & 3)) { set_error("lookahead_cost_batch: mvs must be 4-byte aligned"); return X265HIP_EARG; }
    // Nothing is copied back or synchronised here (the call can be captured into a hipGraph): the task list stays on the device, the kernels
    // skip estimates whose pictures are out of order or outside the nFrames pictures of the buffer.  The kernels address the lowres buffer
    // with 32-bit element offsets: all of it must lie below 2^31 elements.
    const int maxFrame = nFrames - 1;
    if (((int64_t)maxFrame + 1) * 4 * planeElems >= ((int64_t)1 << 31)) { set_error("lookahead_cost_batch: lowres buffer beyond 2^31 elements"); return X265HIP_EARG; }
    hipStream_t st = (hipStream_t)stream;
    const LaGeom g = { (const pixel*)lowres, planeElems, stride, origin, widthInCU, heightInCU };
    XH_KLAUNCH(la_zero_kernel, dim3((unsigned)((nTasks * 3 + 255) / 256)), dim3(256), 0, st, tasks, nTasks, (unsigned long long*)sums);
    XH_LAUNCH_CHECK();
    // one 8-lane group per block of the widest wavefront step, in whole wavefronts, at most 1024 threads
    const int nslices = heightInCU / rowsPerSlice;
    const int widest = min(rowsPerSlice + heightInCU % rowsPerSlice, (widthInCU + 1) / 2);
    // A/B switches of the profiling scripts, clamped to what the kernel can run with (whole wavefronts, 64..1024 threads; LDS within the CU's 160 KB)
    static const int threadCap = [] { const char* e = xh_experiment("X265HIP_LA_THREADS"); int v = e ? atoi(e) : 1024; v = v / 64 * 64; return v < 64 ? 64 : v > 1024 ? 1024 : v; }();
    const int threads = min(min(1024, threadCap), max(64, (widest * 8 + 63) / 64 * 64));
    // the bound checked above; 2 bytes per entry of LDS.  With --hme the half-resolution sweep takes twice a quarter-resolution MV as a predictor candidate, and the block it
    // takes it from is indexed with the half-resolution width (slicetype.cpp:4482: for an odd width a block of another column, further down the picture, of any column):
    // that predictor lies anywhere within the doubled quarter-resolution window, |mv - mvp| < 4 * (2 * size + 64)
    const int costR = hme ? 4 * (2 * max(widthInCU, heightInCU) * CU + 64) : 4 * (max(widthInCU, heightInCU) * CU + 32);
    // Workgroup placement: a CU accepts four of these 8-wavefront workgroups, and the dispatcher fills CUs one after the other, so a
    // batch of ~2 workgroups per CU ends up four deep on some CUs and absent on others -- and four interleaved wavefront sweeps take
    // four times as long as one.  Asking for 56 KB of LDS (of 160 KB per CU) caps the depth at two.
    static const size_t ldsPad = [] { const char* e = xh_experiment("X265HIP_LA_LDS"); long v = e ? atol(e) : 56 * 1024; return (size_t)(v < 0 ? 0 : v > 160 * 1024 ? 160 * 1024 : v); }();
    const size_t lds = std::max(sizeof(uint16_t) * (size_t)(2 * costR + 2), ldsPad);
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)la_search_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    { set_error("lookahead_cost_batch: %zu bytes of LDS for the cost row refused", lds); return X265HIP_EDEVICE; }
    if (hme)
    {   // level 0: the quarter-resolution sweep into its own slots (same slot numbers), one slice, never the weighted copy
        const LaGeom g0 = { (const pixel*)hme->lowerRes, hme->planeElems, hme->stride, hme->origin, hme->widthInCU, hme->heightInCU };
        const int widest0 = min(hme->heightInCU, (hme->widthInCU + 1) / 2), threads0 = min(1024, max(64, (widest0 * 8 + 63) / 64 * 64));
        XH_KLAUNCH(la_search_kernel, dim3(2 * nTasks, 1), dim3(threads0), lds, st, g0, tasks, nFrames, costRow + costHalfRange, costR, hme->heightInCU, (uint32_t*)hme->mvs, hme->mvCosts,
                           hme->range[0], hme->method[0], 0, (const uint32_t*)nullptr, (const int32_t*)nullptr, 0, g);
        XH_LAUNCH_CHECK();
        XH_KLAUNCH(la_search_kernel, dim3(2 * nTasks, nslices), dim3(threads), lds, st, g, tasks, nFrames, costRow + costHalfRange, costR, rowsPerSlice, (uint32_t*)mvs, mvCosts,
                           hme->range[1], hme->method[1], 1, (const uint32_t*)hme->mvs, (const int32_t*)hme->mvCosts, hme->widthInCU * hme->heightInCU, LaGeom{});
    }
    else
        XH_KLAUNCH(la_search_kernel, dim3(2 * nTasks, nslices), dim3(threads), lds, st, g, tasks, nFrames, costRow + costHalfRange, costR, rowsPerSlice, (uint32_t*)mvs, mvCosts,
                           LA_MERANGE, X265HIP_ME_HEX, 1, (const uint32_t*)nullptr, (const int32_t*)nullptr, 0, LaGeom{});
    XH_LAUNCH_CHECK();
    XH_KLAUNCH(la_finish_kernel, dim3(heightInCU, nTasks), dim3(256), 0, st, g, tasks, nFrames, (const uint32_t*)mvs, mvCosts, intraCost, invQscale, lowresCosts, rowSatds,
                       (unsigned long long*)sums);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

extern "C" int x265hip_cutree_propagate(void* stream, int widthInCU, int heightInCU, int distP0, int distP1, int weightedBiPred, double fpsFactor, int referenced,
                                        const int32_t* intraCost, const uint16_t* lowresCosts, const int32_t* invQscale, const int16_t* mvs0, const int16_t* mvs1,
                                        uint16_t* propB, uint16_t* prop0, uint16_t* prop1, void* workspace, size_t workspaceBytes)
{
    const int ncu = widthInCU * heightInCU;
    if (widthInCU < 1 || heightInCU < 1 || distP0 < 1 || distP1 < 0 || !intraCost || !lowresCosts || !invQscale || !mvs0 || (distP1 > 0 && !mvs1) || !propB || !prop0 ||
        (distP1 > 0 && !prop1) || !workspace || workspaceBytes < sizeof(uint64_t) * 2 * (size_t)ncu || ((uintptr_t)mvs0 & 3) || ((uintptr_t)mvs1 & 3))
    { set_error("cutree_propagate: bad arguments"); return X265HIP_EARG; }
    hipStream_t st = (hipStream_t)stream;
    const int span = distP0 + distP1;
    const int distScaleFactor = ((distP0 << 8) + (span >> 1)) / span;                    // slicetype.cpp:3853-3855
    const int bipredWeight = weightedBiPred ? 64 - (distScaleFactor >> 2) : 32;
    XH_HIP(hipMemsetAsync(workspace, 0, sizeof(uint64_t) * 2 * (size_t)ncu, st));
    XH_KLAUNCH(cutree_scatter_kernel, dim3((ncu + 255) / 256), dim3(256), 0, st, widthInCU, heightInCU, bipredWeight, fpsFactor / 256, referenced,
                       intraCost, lowresCosts, invQscale, (const uint32_t*)mvs0, (const uint32_t*)(mvs1 ? mvs1 : mvs0), propB, (unsigned long long*)workspace);
    XH_LAUNCH_CHECK();
    XH_KLAUNCH(cutree_fold_kernel, dim3((ncu + 255) / 256), dim3(256), 0, st, ncu, (const unsigned long long*)workspace, prop0, prop1 ? prop1 : prop0);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

// Lookahead::cuTreeFinish (slicetype.cpp:4098-4150, default configuration: no hevc-aq, qgSize != 8): qp offset of every block from its propagated cost.
// Integer part exact; the two log2 of doubles are the device math library's (not guaranteed to round like the host's libm: tests state 1e-12).
static __global__ __launch_bounds__(256) void cutree_finish_kernel(int ncu, const int32_t* __restrict__ intraCost, const int32_t* __restrict__ invQscale, const uint16_t* __restrict__ prop,
                                                            const double* __restrict__ qpAq, int fpsFactor, double weightdelta, double strength, double* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ncu) return;
    const int intracost = (intraCost[i] * invQscale[i] + 128) >> 8;
    if (!intracost) return;                                       // the reference leaves qpCuTreeOffset of such blocks alone
    const int propagate = ((int)prop[i] * fpsFactor + 128) >> 8;
    const double log2_ratio = log2((double)(intracost + propagate)) - log2((double)intracost) + weightdelta;
    out[i] = qpAq[i] - strength * log2_ratio;
}
extern "C" int x265hip_cutree_finish(void* stream, int ncu, const int32_t* intraCost, const int32_t* invQscale, const uint16_t* propagateCost, const double* qpAqOffset,
                                     int fpsFactor, double weightedCostDelta, int ref0Distance, double cuTreeStrength, double* qpCuTreeOffset)
{
    if (ncu <= 0) return X265HIP_OK;
    if (!intraCost || !invQscale || !propagateCost || !qpAqOffset || !qpCuTreeOffset) { set_error("cutree_finish: bad arguments"); return X265HIP_EARG; }
    const double weightdelta = ref0Distance && weightedCostDelta > 0 ? 1.0 - weightedCostDelta : 0.0;          // slicetype.cpp:4107-4110
    XH_KLAUNCH(cutree_finish_kernel, dim3((ncu + 255) / 256), dim3(256), 0, (hipStream_t)stream, ncu, intraCost, invQscale, propagateCost, qpAqOffset, fpsFactor, weightdelta,
                       cuTreeStrength, qpCuTreeOffset);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

extern "C" int x265hip_propagate_cost_row(void* stream, int32_t* dst, const uint16_t* propagateIn, const int32_t* intraCosts, const uint16_t* interCosts,
                                          const int32_t* invQscales, double fpsFactor, int len)
{
    if (len <= 0) return X265HIP_OK;
    if (!dst || !propagateIn || !intraCosts || !interCosts || !invQscales) { set_error("propagate_cost_row: bad arguments"); return X265HIP_EARG; }
    XH_KLAUNCH(propagate_row_kernel, dim3((len + 255) / 256), dim3(256), 0, (hipStream_t)stream, dst, propagateIn, intraCosts, interCosts, invQscales, fpsFactor / 256, len);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
extern "C" int x265hip_fix8_convert(void* stream, int pack, void* dst, const void* src, int count)
{
    if (count <= 0) return X265HIP_OK;
    if (!dst || !src) { set_error("fix8_convert: bad arguments"); return X265HIP_EARG; }
    if (pack) XH_KLAUNCH(fix8_pack_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, (uint16_t*)dst, (const double*)src, count);
    else XH_KLAUNCH(fix8_unpack_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, (double*)dst, (const uint16_t*)src, count);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
