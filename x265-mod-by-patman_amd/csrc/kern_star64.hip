// kern_star64.hip -- the full-pel part of the STAR search (reference encoder/motion.cpp:1328-1436, StarPatternSearch :387-629) for 64x64 PUs
// with the whole search window of the PU in LDS: pattern passes, the two-point check, the raster refinement and the re-centred passes.
//
// A 64x64 PU at merange 57 searches a window of (114 + 64)^2 reference pixels.  The pattern kernels of me_body.inc cost four candidates per
// step straight from memory -- a trip through the L1 front end, a cross-lane reduction, a workgroup barrier and the decision logic per
// step, ~50 dependent steps per PU -- and the raster refinement costs 23 x 23 more placements (2.2 M absolute differences per PU) when the
// first pass lands far away.  Here the window is read from memory ONCE (66 KB at 16 bit, 34 KB at 8 bit), every cost is a pass over LDS, and
// all candidates of a pattern pass are costed together before the reference's decision sequence is replayed over their costs:
//
//   * band: the window as the dword image of its rows, stored column-dword-major (element [cdw][row], 179 rows per column) so that a lane's
//     walk down the rows of one placement is a sequence of immediate offsets off three base addresses;
//   * lane = (candidate, 8-byte unit u of a PU row); a wavefront permanently owns one half of the PU's rows (h = wave & 1) and keeps its
//     32 x 8 bytes of them in registers for the whole kernel;
//   * pattern passes: all points of a pass (60 at merange 57: distances 1, 2, 4, 8, 16, 32) are costed in rounds of 8 candidates x 2 halves,
//     one v_sad per 8 bytes + 1.5 LDS reads, summed over the unit lanes with four DPP steps; the first pass (early exit after 3 idle rounds)
//     costs distances 1-4 first and the rest only if the search goes on.  The decisions (strict `<` in the reference's point order, bPointNr /
//     bDistance bookkeeping, idle-round counting) are then replayed by every thread over the cost list;
//   * raster refinement (:1365-1399): lane = (candidate column i, unit u); the lane walks the reference rows R of the window once and
//     compares row R against every source row y with y + 5 j = R, i.e. feeds up to 7 vertical placements j at once (5.2 on average) -- one
//     v_sad per 8 bytes and 0.2 LDS reads per placement.  The walk is expanded at compile time: which (y, j) pairs a row serves is static,
//     the 23 accumulators and 32 source units never move.  The cost stage adds the MV cost (including the reference's `tmv << 3` on every
//     fourth placement of a row, :1392) and takes the minimum in raster order (COPY2_IF_LT).
//
// The launch sits between the two halves of the split 64x64 STAR kernel (me_body.inc, phase 1 / 2): phase 1 runs the start stage
// (clipped MVP, zero MV, candidates: sub-pel positions, costed from the phase planes) and parks the PU (full-pel position and cost in its
// x265hip_me_result, reserved = XH_PARKED); this kernel runs the integer search from there; phase 2 resumes with the sub-pel stage.
// Bound by the issue rate of v_sad_u16 / v_sad_u8 (one wave64 instruction per 4 clocks per SIMD, profiles/micro/valu_rate.hip).
#include "xh_mc.h"
#include "../../include/x265hip_frame.h"
#include <utility>
#include <cstdlib>
using namespace xh;

#define XH_PARKED 0x5041524B               // me_body.inc: x265hip_me_result.reserved of a parked PU

namespace {

constexpr int RD = 5;                                       // RasterDistance (motion.cpp:1364)
constexpr int PW = 64, PH = 64;                             // the PU
constexpr int MAXR = 57;                                    // largest merange whose window fits the band
constexpr int WPX = 2 * MAXR + PW;                          // window pixels each way: 178
constexpr int LPI = PW / XH_UNITPX;                         // unit lanes per candidate: 16 (16 bit) / 8 (8 bit)
constexpr int IPW = 64 / LPI;                               // candidates per wavefront: 4 / 8
constexpr int HR = 32;                                      // source rows per lane: half a PU (64 VGPRs)
constexpr int NJ = 23;                                      // vertical raster placements: (2 * 57) / 5 + 1
constexpr int NIC = 24;                                     // raster columns computed (a multiple of IPW; 23 used)
constexpr int RROWS = HR + RD * (NJ - 1);                   // reference rows one lane walks in the raster: 142
constexpr int BRP = (WPX + 1) | 1;                          // rows per band column (odd): 179
constexpr int MAXB = (RD * (NIC - 1) + PW - XH_UNITPX) * (int)sizeof(pixel) + 3;   // largest byte column a lane's 12-byte read starts at (24th raster column, last unit, misalignment 3)
constexpr int BCDW = (MAXB >> 2) + 3;                       // band columns in dwords: 91 / 46
static_assert(HR + RROWS - 1 < BRP && RROWS - 1 < 256, "row offsets of a raster walk are ds_read2 immediates inside one band column");
constexpr int LOADQ = (BCDW + 3) / 4;                       // 16-byte pieces per window row
constexpr int LOADROWS = 256 / LOADQ;
constexpr int LOADPASSES = (WPX + LOADROWS - 1) / LOADROWS;
constexpr int MAXCAND = 64;                                 // candidates of one costing batch (a whole pattern pass at merange <= 57: 60)
constexpr int COST_R = 512;                                 // MVD cost entries kept in LDS: d in [-512, 512] quarter-pels

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4a4 __attribute__((aligned(4)));
typedef XH_LDS uint16_t lu16;

#define XR_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)

struct Mv { const uint16_t* centre; const lu16* lcentre; int chr, mvpx, mvpy; };
__device__ __forceinline__ int cost1(const Mv& m, int d)
{
    return (unsigned)(d + COST_R) <= 2u * COST_R ? (int)m.lcentre[d] : (int)m.centre[min(max(d, -m.chr), m.chr)];
}
__device__ __forceinline__ int mvcost(const Mv& m, int qx, int qy) { return (uint16_t)(cost1(m, qx - m.mvpx) + cost1(m, qy - m.mvpy)); }   // bitcost.h:57

__device__ __forceinline__ int unit_sum(int v)
{   // sum over the LPI unit lanes of a candidate (they are an aligned lane group)
    v += XR_DPP(v, 0xB1); v += XR_DPP(v, 0x4E); v += XR_DPP(v, 0x141);             // quad, quad pair: 8 lanes
    if (LPI == 16) v += XR_DPP(v, 0x140);                                           // row_mirror: 16 lanes
    return v;
}

// ---- one placement of the wavefront's PU half against the band: 32 rows, one v_sad per 8 bytes ----------------------------------------
template<int... Y> __device__ __forceinline__ unsigned half_sad(const fquad (&f)[HR], const lu32* p0, unsigned mis, std::integer_sequence<int, Y...>)
{
    const lu32* p1 = p0 + BRP; const lu32* p2 = p1 + BRP;
    unsigned acc = 0;
    auto row = [&](int y, const fquad& fy) {
        const uint32_t w0 = p0[y], w1 = p1[y], w2 = p2[y];
        fquad r; r.x = __builtin_amdgcn_alignbyte(w1, w0, mis); r.y = __builtin_amdgcn_alignbyte(w2, w1, mis);
        acc = sadq(fy, r, acc);
    };
    (row(Y, f[Y]), ...);
    return acc;
}

// ---- raster: the walk over the reference rows, expanded at compile time (index sequences, not `#pragma unroll`: at 142 x 23 iterations the
// unroller gives up and the source units would be indexed dynamically, i.e. live in scratch memory) -------------------------------------
template<int R, int J> __device__ __forceinline__ void pair_step(const fquad (&f)[HR], unsigned (&acc)[NJ], const fquad& r)
{
    constexpr int y = R - RD * J;                                                   // source row that reference row R meets for placement J
    if constexpr (y >= 0 && y < HR) acc[J] = sadq(f[y], r, acc[J]);
}
template<int R, int... J> __device__ __forceinline__ void row_pairs(const fquad (&f)[HR], unsigned (&acc)[NJ], const fquad& r, std::integer_sequence<int, J...>)
{
    (pair_step<R, J>(f, acc, r), ...);
}
// Rows are taken in groups of GR: the LDS reads of group k + 1 are issued, then group k is computed.  The scheduling barriers keep the
// compiler from hoisting ALL 142 reads to the top (it did: 426 live registers, 1.4 KB of scratch per lane).
constexpr int GR = 6;
struct RowGroup { uint32_t w[GR][3]; };
template<int R0, int... K> __device__ __forceinline__ void load_group(RowGroup& g, const lu32* p0, std::integer_sequence<int, K...>)
{
    auto one = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (R0 + k < RROWS) { g.w[k][0] = p0[R0 + k]; g.w[k][1] = p0[BRP + R0 + k]; g.w[k][2] = p0[2 * BRP + R0 + k]; }
    };
    (one(std::integral_constant<int, K>{}), ...);
}
template<int R0, int... K> __device__ __forceinline__ void compute_group(const RowGroup& g, const fquad (&f)[HR], unsigned (&acc)[NJ], unsigned mis, std::integer_sequence<int, K...>)
{
    auto one = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (R0 + k < RROWS)
        {
            fquad r; r.x = __builtin_amdgcn_alignbyte(g.w[k][1], g.w[k][0], mis); r.y = __builtin_amdgcn_alignbyte(g.w[k][2], g.w[k][1], mis);
            row_pairs<R0 + k>(f, acc, r, std::make_integer_sequence<int, NJ>{});
        }
    };
    (one(std::integral_constant<int, K>{}), ...);
}
// The v_sad chains are pure arithmetic: neither the scheduling barrier nor program order keeps the compiler from gathering all 32 rows of ONE
// placement into a single dependent chain (it did, and spilled the aligned units of the other 22 placements it had to keep for later).  Passing
// the accumulators through an empty volatile asm at every group boundary ties each group's arithmetic to its place.
__device__ __forceinline__ void pin(unsigned (&acc)[NJ])
{
    static_assert(NJ == 23, "operand list below");
    asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]),
                      "+v"(acc[10]), "+v"(acc[11]), "+v"(acc[12]), "+v"(acc[13]), "+v"(acc[14]), "+v"(acc[15]), "+v"(acc[16]), "+v"(acc[17]), "+v"(acc[18]),
                      "+v"(acc[19]), "+v"(acc[20]), "+v"(acc[21]), "+v"(acc[22]));
}
template<int R0> __device__ __forceinline__ void row_pipeline(const RowGroup& cur, const fquad (&f)[HR], unsigned (&acc)[NJ], const lu32* p0, unsigned mis)
{
    if constexpr (R0 < RROWS)
    {
        RowGroup nxt;
        load_group<R0 + GR>(nxt, p0, std::make_integer_sequence<int, GR>{});
        __builtin_amdgcn_sched_barrier(0);
        compute_group<R0>(cur, f, acc, mis, std::make_integer_sequence<int, GR>{});
        pin(acc);
        __builtin_amdgcn_sched_barrier(0);
        row_pipeline<R0 + GR>(nxt, f, acc, p0, mis);
    }
}
template<int... J> __device__ __forceinline__ void fold_units(unsigned (&acc)[NJ], lu32* psum, int icol, bool first, std::integer_sequence<int, J...>)
{
    auto one = [&](int j, unsigned a) {
        const int v = unit_sum((int)a);
        if (first) __hip_atomic_fetch_add(psum + j * NIC + icol, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    (one(J, acc[J]), ...);
}
// one candidate-column group g of the raster for the wavefront's row half h: 64 lanes = IPW columns x LPI units
__device__ __forceinline__ void raster_task(const lu32* band, const fquad (&f)[HR], lu32* psum, int g, int h, int lane, int m0)
{
    const int il = lane / LPI, u = lane % LPI, icol = g * IPW + il;
    const int b = (RD * icol + XH_UNITPX * u) * (int)sizeof(pixel) + m0;           // byte column of the lane's unit in a window row
    const lu32* p0 = band + (b >> 2) * BRP + h * HR;
    unsigned acc[NJ] = {};
    RowGroup first;
    load_group<0>(first, p0, std::make_integer_sequence<int, GR>{});
    row_pipeline<0>(first, f, acc, p0, (unsigned)b & 3u);
    fold_units(acc, psum, icol, u == 0, std::make_integer_sequence<int, NJ>{});
}

// Point `k` of a StarPatternSearch round (motion.cpp:387-629) around (ox, oy): type 0 = distance 1 (4 points), type 1 = distances 2 / 4 / 8
// (8 points), type 2 = distances >= 16 (16 points).  Order and per-point range guards are the reference's; pd packs bPointNr | bDistance << 8.
__device__ __forceinline__ bool star_slot(int type, int dist, int ox, int oy, int k, int mnx, int mny, int mxx, int mxy, int& x, int& y, int& pd)
{
    const int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
    if (type == 0)
    {
        switch (k)
        {
        case 0: x = ox; y = top; pd = 2 | (1 << 8); return top >= mny;
        case 1: x = left; y = oy; pd = 4 | (1 << 8); return left >= mnx;
        case 2: x = right; y = oy; pd = 5 | (1 << 8); return right <= mxx;
        default: x = ox; y = bottom; pd = 7 | (1 << 8); return bottom <= mxy;
        }
    }
    if (type == 1)
    {
        const int half = dist >> 1, top2 = oy - half, bottom2 = oy + half, left2 = ox - half, right2 = ox + half;
        switch (k)
        {
        case 0: x = ox; y = top; pd = 2 | (dist << 8); return top >= mny;
        case 1: x = left2; y = top2; pd = 1 | (half << 8); return top2 >= mny && left2 >= mnx;
        case 2: x = right2; y = top2; pd = 3 | (half << 8); return top2 >= mny && right2 <= mxx;
        case 3: x = left; y = oy; pd = 4 | (dist << 8); return left >= mnx;
        case 4: x = right; y = oy; pd = 5 | (dist << 8); return right <= mxx;
        case 5: x = left2; y = bottom2; pd = 6 | (half << 8); return bottom2 <= mxy && left2 >= mnx;
        case 6: x = right2; y = bottom2; pd = 8 | (half << 8); return bottom2 <= mxy && right2 <= mxx;
        default: x = ox; y = bottom; pd = 7 | (dist << 8); return bottom <= mxy;
        }
    }
    pd = dist << 8;
    if (k < 4)
    {
        switch (k)
        {
        case 0: x = ox; y = top; return top >= mny;
        case 1: x = left; y = oy; return left >= mnx;
        case 2: x = right; y = oy; return right <= mxx;
        default: x = ox; y = bottom; return bottom <= mxy;
        }
    }
    const int index = ((k - 4) >> 2) + 1, j = (k - 4) & 3, q = (dist >> 2) * index;
    const int posYT = top + q, posYB = bottom - q, posXL = ox - q, posXR = ox + q;
    x = (j & 1) ? posXR : posXL; y = (j & 2) ? posYB : posYT;
    return ((j & 2) ? posYB <= mxy : posYT >= mny) && ((j & 1) ? posXR <= mxx : posXL >= mnx);
}
__device__ __forceinline__ int round_slots(int dist) { return dist == 1 ? 4 : (dist <= 8 ? 8 : 16); }
// the two "missing" neighbours of a distance-1 winner (motion.cpp:1345-1362, 1417-1430; table :421-431)
__device__ __forceinline__ int k_offsets(int i, int col)
{
    // motion.cpp:74-84 offsets[16] = {-1,0},{0,-1},{-1,-1},{1,-1},{-1,0},{1,0},{-1,1},{-1,-1},{1,-1},{1,1},{-1,0},{0,1},{-1,1},{1,1},{1,0},{0,1}: value + 1, two bits per entry
    constexpr uint32_t X = 0x684A0884u, Y = 0x9A982501u;
    return (int)(((col ? Y : X) >> (2 * i)) & 3) - 1;
}

struct Shared
{
    uint32_t band[BCDW * BRP];
    pixel fenc[PH * PW];
    uint32_t psum[NJ * NIC];
    unsigned long long best;
    uint32_t csad[MAXCAND];
    int cx[MAXCAND], cy[MAXCAND], cpd[MAXCAND];
    uint16_t mvc[2 * COST_R + 2];
};

// minimum of an unsigned value over the 64 lanes of the wavefront (every lane gets it)
__device__ __forceinline__ uint32_t wave_umin(uint32_t v)
{
#define XR_MIN(ctrl) v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFF, (int)v, (ctrl), 0xF, 0xF, false))
    XR_MIN(0xB1); XR_MIN(0x4E); XR_MIN(0x141); XR_MIN(0x140);
#undef XR_MIN
    return min(min((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
               min((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}

// Cost the candidates cx / cy [0, n) (full-pel MVs inside the window; n <= MAXCAND = the lanes of a wavefront).  Returns, in lane c of EVERY
// wavefront, the key (SAD + mv cost) << 8 | c of candidate c (all ones for an unvisited point, cpd < 0, and for lanes >= n): the
// reference's "first candidate, in its order, that beats the best so far" is the minimum key of a round.
__device__ __forceinline__ uint32_t cost_batch(Shared& s, const fquad (&f)[HR], const Mv& mv, int n, int mnx, int mny, int m0, int tid)
{
    const int lane = tid & 63, wave = tid >> 6, h = wave & 1, il = lane / LPI, u = lane % LPI;
    if (tid < n) s.csad[tid] = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 2 * IPW)
    {
        const int c = base + (wave >> 1) * IPW + il;
        if (base + (wave >> 1) * IPW >= n) break;                                   // wave-uniform
        const int cc = min(c, n - 1);
        const int px = s.cx[cc] - mnx, py = s.cy[cc] - mny;
        const int b = (px + XH_UNITPX * u) * (int)sizeof(pixel) + m0;
        const lu32* p0 = (const lu32*)s.band + (b >> 2) * BRP + py + h * HR;
        const int v = unit_sum((int)half_sad(f, p0, (unsigned)b & 3u, std::make_integer_sequence<int, HR>{}));
        if (u == 0 && c < n) __hip_atomic_fetch_add((lu32*)s.csad + c, (uint32_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    uint32_t key = 0xFFFFFFFFu;
    if (lane < n && s.cpd[lane] >= 0) key = ((uint32_t)((int)s.csad[lane] + mvcost(mv, s.cx[lane] * 4, s.cy[lane] * 4)) << 8) | (uint32_t)lane;
    return key;
}

__global__ __launch_bounds__(256, 2) void star64_kernel(const pixel* __restrict__ cur, intptr_t cs, const pixel* __restrict__ ref, intptr_t rs,
                                                        const x265hip_me_task* __restrict__ tasks, int n, const uint16_t* __restrict__ costCentre, int chr,
                                                        int merange, x265hip_me_result* __restrict__ results, const x265hip_me_result* __restrict__ mvpSource, int dbgAll)
{
    const int dbg = dbgAll & 15;
    __shared__ __attribute__((aligned(16))) Shared s;
    const int item = (dbgAll & 32) ? xcd_contiguous_block(blockIdx.x, gridDim.x) : (dbgAll & 64) ? xcd_chunked_block(blockIdx.x, gridDim.x, 60) : (int)blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (item >= n) return;
    const x265hip_me_result st = results[item];
    if (st.reserved != XH_PARKED) return;                                           // only PUs the first half parked (start stage done, search pending)
    const x265hip_me_task* __restrict__ tp = tasks + item;
    if (tp->flags & X265HIP_ME_ROWS) costCentre += (size_t)((tp->flags >> 8) & 0xFF) * (size_t)(2 * chr + 1);      // the task's own MVD cost row (one PU per workgroup: its LDS slice too)
    Mv mv; mv.centre = costCentre; mv.lcentre = (const lu16*)s.mvc + COST_R; mv.chr = chr;
    mv.mvpx = tp->qmvp[0]; mv.mvpy = tp->qmvp[1];
    { const int from = tp->mvpFrom; if (from >= 0 && mvpSource) { mv.mvpx = mvpSource[from].mv[0]; mv.mvpy = mvpSource[from].mv[1]; } }
    int mnx = tp->mvmin[0], mny = tp->mvmin[1], mxx = tp->mvmax[0], mxy = tp->mvmax[1];
    if (tp->flags & X265HIP_ME_WINDOW)
    {   // search.cpp:4969-5021 setSearchRange, as in me_kernel
        const int lx0 = mnx, ly0 = mny, lx1 = mxx, ly1 = mxy, d = merange << 2;
        mnx = min(lx1, max(lx0, mv.mvpx - d)) >> 2; mny = min(ly1, max(ly0, mv.mvpy - d)) >> 2;
        mxx = min(lx1, max(lx0, mv.mvpx + d)) >> 2; mxy = min(ly1, max(ly0, mv.mvpy + d)) >> 2;
        mxy = max(mxy, mny);
    }
    // window larger than the band (a caller's own bounds wider than 2 * 57): leave the PU parked, the second half of the split kernel
    // searches it the general way
    if (mxx - mnx > 2 * MAXR || mxy - mny > 2 * MAXR || mxx < mnx || mxy < mny) return;

    // ---- source PU, MVD cost slice, the window: every global load of the set-up is issued before the first LDS store (the loads sit behind
    //      no per-pass branch: rows past the window are clamped, not skipped -- a conditional load per pass made the 17 passes 17 exposed
    //      round trips: 2.59 -> 2.49 ms for the launch at 4K 10 bit) ----
    const pixel* org = ref + tp->refOff + (intptr_t)mny * rs + mnx;
    const int m0 = (int)((uintptr_t)org & 3u);
    {
        constexpr int FQ = PH * (PW / XH_UNITPX) / 256, MQ = (2 * COST_R + 1 + 255) / 256;
        const pixel* src = cur + tp->curOff;
        fquad fq[FQ];
#pragma unroll
        for (int i = 0; i < FQ; i++)
        {
            const int q = tid + 256 * i, y = q / (PW / XH_UNITPX), x = (q % (PW / XH_UNITPX)) * XH_UNITPX;
            fq[i] = ldq(src + (intptr_t)y * cs + x);
        }
        const char* wsrc = (const char*)org - m0;
        const int rowB = (int)rs * (int)sizeof(pixel);
        const int rows = (mxy - mny) + PH, nd = ((((mxx - mnx) + PW) * (int)sizeof(pixel) + m0 + 3) >> 2) + 2;
        const int wq = tid % LOADQ, r0 = tid / LOADQ;
        const bool loader = r0 < LOADROWS && 4 * wq < nd;
        u32x4 wv[LOADPASSES];
#pragma unroll
        for (int p = 0; p < LOADPASSES; p++)
        {
            const int r = min(r0 + p * LOADROWS, rows - 1);
            wv[p] = *(const u32x4a4*)(wsrc + (size_t)r * (size_t)rowB + 16u * (unsigned)(loader ? wq : 0));
        }
        uint16_t mc[MQ];
#pragma unroll
        for (int i = 0; i < MQ; i++) mc[i] = costCentre[min(tid + 256 * i, 2 * COST_R) - COST_R];
#pragma unroll
        for (int i = 0; i < FQ; i++)
        {
            const int q = tid + 256 * i, y = q / (PW / XH_UNITPX), x = (q % (PW / XH_UNITPX)) * XH_UNITPX;
            *(lu2*)((lpixel*)s.fenc + y * PW + x) = fq[i];
        }
        for (int k = tid; k < NJ * NIC; k += 256) s.psum[k] = 0;
#pragma unroll
        for (int i = 0; i < MQ; i++) if (tid + 256 * i < 2 * COST_R + 1) s.mvc[tid + 256 * i] = mc[i];
        if (loader)
        {
#pragma unroll
            for (int p = 0; p < LOADPASSES; p++)
            {
                const int r = r0 + p * LOADROWS;
                if (r < rows)
                {
                    lu32* dst = (lu32*)s.band + (4 * wq) * BRP + r;
                    dst[0] = wv[p].x;
                    if (4 * wq + 1 < BCDW) dst[BRP] = wv[p].y;
                    if (4 * wq + 2 < BCDW) dst[2 * BRP] = wv[p].z;
                    if (4 * wq + 3 < BCDW) dst[3 * BRP] = wv[p].w;
                }
            }
        }
    }
    __syncthreads();
    // the wavefront's half of the source PU, one unit column per lane, for the whole kernel
    const int h = wave & 1, u = lane % LPI;
    fquad f[HR];
    {
        const lpixel* fp = (const lpixel*)s.fenc + (h * HR) * PW + u * XH_UNITPX;
#pragma unroll
        for (int y = 0; y < HR; y++) f[y] = ldf(fp + y * PW);
    }

    int bx = st.mv[0], by = st.mv[1], bcost = st.cost, bPointNr = 0, bDistance = 0;
    if (dbg == 5) return;                                                           // timing experiments: window load only

    // motion.cpp:387-629 with all points of the pass costed up front; the first pass (3 idle rounds end it) costs distances 1-4 first
    auto star_pass = [&](int earlyExitIters) {
        const int ox = bx, oy = by;
        int rounds = 0, dist = 1;
        while (dist <= 8 || dist <= (int)(int16_t)merange)
        {
            // this batch: rounds [dist, dEnd)
            int dEnd = dist, ncand = 0;
            while ((dEnd <= 8 || dEnd <= (int)(int16_t)merange) && ncand + round_slots(dEnd) <= MAXCAND && !(earlyExitIters <= 3 && dist == 1 && dEnd > 4)) { ncand += round_slots(dEnd); dEnd <<= 1; }
            if (tid < ncand)
            {
                int d = dist, k = tid;
                while (k >= round_slots(d)) { k -= round_slots(d); d <<= 1; }
                int x, y, pd;
                const bool ok = star_slot(d == 1 ? 0 : (d <= 8 ? 1 : 2), d, ox, oy, k, mnx, mny, mxx, mxy, x, y, pd);
                s.cx[tid] = ok ? x : ox; s.cy[tid] = ok ? y : oy; s.cpd[tid] = ok ? pd : -1;
            }
            const uint32_t key = cost_batch(s, f, mv, ncand, mnx, mny, m0, tid);
            int c0 = 0;
            for (int d = dist; d < dEnd; d <<= 1)
            {   // a round's strict-`<` updates in point order = its first minimum, if that beats the best so far
                const int ns = round_slots(d);
                const uint32_t m = wave_umin((lane >= c0 && lane < c0 + ns) ? key : 0xFFFFFFFFu);
                c0 += ns;
                if (m != 0xFFFFFFFFu && (int)(m >> 8) < bcost)
                {
                    const int c = (int)(m & 0xFF), pd = s.cpd[c];
                    bcost = (int)(m >> 8); bx = s.cx[c]; by = s.cy[c]; bPointNr = pd & 0xFF; bDistance = pd >> 8;
                    rounds = 0;
                }
                else if (++rounds >= earlyExitIters) return;
            }
            dist = dEnd;
            __syncthreads();                                                         // everybody has read the batch before the next one is written
        }
    };
    auto two_point = [&]() {
        if (tid < 2)
        {
            const int x = bx + k_offsets((bPointNr - 1) * 2 + tid, 0), y = by + k_offsets((bPointNr - 1) * 2 + tid, 1);
            const bool ok = x >= mnx && x <= mxx && y >= mny && y <= mxy;
            s.cx[tid] = ok ? x : bx; s.cy[tid] = ok ? y : by; s.cpd[tid] = ok ? 0 : -1;
        }
        // both neighbours are offsets of the SAME centre (mv1, mv2 are computed before either COST_MV runs): two strict-`<` updates in order
        const uint32_t m = wave_umin(cost_batch(s, f, mv, 2, mnx, mny, m0, tid));
        if (m != 0xFFFFFFFFu && (int)(m >> 8) < bcost) { const int c = (int)(m & 0xFF); bcost = (int)(m >> 8); bx = s.cx[c]; by = s.cy[c]; }
        __syncthreads();
    };
    auto raster = [&]() {
        const int NX = (mxx - mnx) / RD + 1, NY = (mxy - mny) / RD + 1;
        const int groupEnd = NX >= 4 ? ((NX - 4) & ~3) + 4 : 0;                     // placements [0, groupEnd) of a row are costed four at a time (:1372-1392)
        if (tid == 0) s.best = ~0ull;
        const int ngroups = (NX + IPW - 1) / IPW;
        for (int g = wave >> 1; g < ngroups; g += 2)
            raster_task((const lu32*)s.band, f, (lu32*)s.psum, g, h, lane, m0);
        __syncthreads();
        unsigned long long best = ~0ull;
        for (int k = tid; k < NY * NX; k += 256)
        {
            const int j = k / NX, i = k - j * NX;
            const int tx = mnx + RD * i, ty = mny + RD * j;
            const int sad = (int)s.psum[j * NIC + i];
            const bool quirk = i < groupEnd && (i & 3) == 3;                         // the fourth of a sad_x4 group: mvcost(tmv << 3) (:1392)
            const int cost = sad + (quirk ? mvcost(mv, tx * 8, ty * 8) : mvcost(mv, tx * 4, ty * 4));
            const unsigned long long key = ((unsigned long long)(unsigned)cost << 16) | (unsigned)k;
            best = key < best ? key : best;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
        {
            const unsigned lo = __shfl_xor((unsigned)best, off, 64), hi = __shfl_xor((unsigned)(best >> 32), off, 64);
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            best = o < best ? o : best;
        }
        if (lane == 0) __hip_atomic_fetch_min(&s.best, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        const unsigned long long bb = s.best;
        const int cost = (int)(bb >> 16);
        if (cost < bcost)
        {   // COPY2_IF_LT in raster order
            const int k = (int)(bb & 0xFFFFu), j = k / NX, i = k - j * NX;
            bcost = cost; bx = mnx + RD * i; by = mny + RD * j;
        }
        // (psum is used once per kernel: no re-zeroing)
    };

    // motion.cpp:1328-1436
    bool first = true;
    for (;;)
    {
        if (!first) { bPointNr = 0; bDistance = 0; }
        if (dbg == 4 && !first) break;                                              // timing experiments: no re-centred passes
        star_pass(first ? 3 : 32);
        __syncthreads();
        if (dbg == 1) break;                                                        // timing experiments: the first pass only
        if (first)
        {
            first = false;
            if (bDistance == 1)
            {
                if (!bPointNr) break;
                const int saved = bcost;
                two_point();
                if (bcost == saved) break;
            }
            if (bDistance > RD && dbg != 3) raster();
            if (dbg == 2) break;                                                    // timing experiments: first pass + raster
            if (!(bDistance > 0)) break;
        }
        else
        {
            if (bDistance == 1)
            {
                if (bPointNr) two_point();
                break;
            }
            if (!(bDistance > 0)) break;
        }
    }
    if (tid == 0)
    {
        x265hip_me_result r = st;
        r.mv[0] = (int16_t)bx; r.mv[1] = (int16_t)by; r.cost = bcost; r.mvcost = 1;   // mvcost != 0: the full-pel search is done, the second half resumes behind it
        results[item] = r;
    }
}

// ---- raster mode: windows beyond the band (merange up to 128: 52 x 52 placements in a (256 + 64)^2 window) ---------------------------------
// The split 64x64 kernel parks a PU where the raster refinement is due (me_body.inc, phase 3: mvcost = 2 in its record); here the raster is costed as
// SAD surfaces chunk by chunk -- 23 vertical x 24 horizontal placements per pass, each pass with its own (174 x 179)-pixel piece of the window in the
// band -- and the minimum in raster order over all passes replaces the parked position if it is cheaper (phase 4 resumes behind it).
__global__ __launch_bounds__(256, 2) void star64_raster_kernel(const pixel* __restrict__ cur, intptr_t cs, const pixel* __restrict__ ref, intptr_t rs,
                                                               const x265hip_me_task* __restrict__ tasks, int n, const uint16_t* __restrict__ costCentre, int chr,
                                                               int merange, x265hip_me_result* __restrict__ results, const x265hip_me_result* __restrict__ mvpSource)
{
    __shared__ __attribute__((aligned(16))) Shared s;
    const int item = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (item >= n) return;
    const x265hip_me_result st = results[item];
    if (st.reserved != XH_PARKED || st.mvcost != 2) return;
    const x265hip_me_task* __restrict__ tp = tasks + item;
    if (tp->flags & X265HIP_ME_ROWS) costCentre += (size_t)((tp->flags >> 8) & 0xFF) * (size_t)(2 * chr + 1);      // the task's own MVD cost row (one PU per workgroup: its LDS slice too)
    Mv mv; mv.centre = costCentre; mv.lcentre = (const lu16*)s.mvc + COST_R; mv.chr = chr;
    mv.mvpx = tp->qmvp[0]; mv.mvpy = tp->qmvp[1];
    { const int from = tp->mvpFrom; if (from >= 0 && mvpSource) { mv.mvpx = mvpSource[from].mv[0]; mv.mvpy = mvpSource[from].mv[1]; } }
    int mnx = tp->mvmin[0], mny = tp->mvmin[1], mxx = tp->mvmax[0], mxy = tp->mvmax[1];
    if (tp->flags & X265HIP_ME_WINDOW)
    {
        const int lx0 = mnx, ly0 = mny, lx1 = mxx, ly1 = mxy, d = merange << 2;
        mnx = min(lx1, max(lx0, mv.mvpx - d)) >> 2; mny = min(ly1, max(ly0, mv.mvpy - d)) >> 2;
        mxx = min(lx1, max(lx0, mv.mvpx + d)) >> 2; mxy = min(ly1, max(ly0, mv.mvpy + d)) >> 2;
        mxy = max(mxy, mny);
    }
    const int NX = mxx >= mnx ? (mxx - mnx) / RD + 1 : 0, NY = mxy >= mny ? (mxy - mny) / RD + 1 : 0;
    const int groupEnd = NX >= 4 ? ((NX - 4) & ~3) + 4 : 0;
    {
        const pixel* src = cur + tp->curOff;
        for (int q = tid; q < PH * (PW / XH_UNITPX); q += 256)
        {
            const int y = q / (PW / XH_UNITPX), x = (q % (PW / XH_UNITPX)) * XH_UNITPX;
            *(lu2*)((lpixel*)s.fenc + y * PW + x) = ldq(src + (intptr_t)y * cs + x);
        }
        for (int k = tid; k < 2 * COST_R + 1; k += 256) s.mvc[k] = costCentre[k - COST_R];
        if (tid == 0) s.best = ~0ull;
    }
    __syncthreads();
    const int h = wave & 1, u = lane % LPI;
    fquad f[HR];
    {
        const lpixel* fp = (const lpixel*)s.fenc + (h * HR) * PW + u * XH_UNITPX;
#pragma unroll
        for (int y = 0; y < HR; y++) f[y] = ldf(fp + y * PW);
    }
    const int rowB = (int)rs * (int)sizeof(pixel);
    for (int j0 = 0; j0 < NY; j0 += NJ)
    {
        const int nj = min(NJ, NY - j0);
        for (int i0 = 0; i0 < NX; i0 += NIC)
        {
            const int ni = min(NIC, NX - i0);
            const pixel* org = ref + tp->refOff + (intptr_t)(mny + RD * j0) * rs + (mnx + RD * i0);
            const int m0 = (int)((uintptr_t)org & 3u);
            const char* src = (const char*)org - m0;
            const int rows = RD * (nj - 1) + PH, nd = (((RD * (ni - 1) + PW) * (int)sizeof(pixel) + m0 + 3) >> 2) + 2;
            __syncthreads();                                                        // the previous pass is done with the band and the sums
            for (int k = tid; k < NJ * NIC; k += 256) s.psum[k] = 0;
            {
                // as in star64_kernel: all loads of the chunk's window in flight before the first LDS store
                const int q = tid % LOADQ, r0 = tid / LOADQ;
                const bool loader = r0 < LOADROWS && 4 * q < nd;
                u32x4 wv[LOADPASSES];
#pragma unroll
                for (int p = 0; p < LOADPASSES; p++)
                    wv[p] = *(const u32x4a4*)(src + (size_t)min(r0 + p * LOADROWS, rows - 1) * (size_t)rowB + 16u * (unsigned)(loader ? q : 0));
                if (loader)
                {
#pragma unroll
                    for (int p = 0; p < LOADPASSES; p++)
                    {
                        const int r = r0 + p * LOADROWS;
                        if (r < rows)
                        {
                            lu32* dst = (lu32*)s.band + (4 * q) * BRP + r;
                            dst[0] = wv[p].x;
                            if (4 * q + 1 < BCDW) dst[BRP] = wv[p].y;
                            if (4 * q + 2 < BCDW) dst[2 * BRP] = wv[p].z;
                            if (4 * q + 3 < BCDW) dst[3 * BRP] = wv[p].w;
                        }
                    }
                }
            }
            __syncthreads();
            const int ngroups = (ni + IPW - 1) / IPW;
            for (int g = wave >> 1; g < ngroups; g += 2)
                raster_task((const lu32*)s.band, f, (lu32*)s.psum, g, h, lane, m0);
            __syncthreads();
            unsigned long long best = ~0ull;
            for (int k = tid; k < nj * ni; k += 256)
            {
                const int j = k / ni, i = k - j * ni, gi = i0 + i, gj = j0 + j;
                const int tx = mnx + RD * gi, ty = mny + RD * gj;
                const int sad = (int)s.psum[j * NIC + i];
                const bool quirk = gi < groupEnd && (gi & 3) == 3;                   // the fourth of a sad_x4 group: mvcost(tmv << 3) (:1392)
                const int cost = sad + (quirk ? mvcost(mv, tx * 8, ty * 8) : mvcost(mv, tx * 4, ty * 4));
                const unsigned long long key = ((unsigned long long)(unsigned)cost << 16) | (unsigned)(gj * NX + gi);
                best = key < best ? key : best;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1)
            {
                const unsigned lo = __shfl_xor((unsigned)best, off, 64), hi = __shfl_xor((unsigned)(best >> 32), off, 64);
                const unsigned long long o = ((unsigned long long)hi << 32) | lo;
                best = o < best ? o : best;
            }
            if (lane == 0) __hip_atomic_fetch_min(&s.best, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (tid == 0)
    {
        const unsigned long long bb = s.best;
        x265hip_me_result r = st;
        r.mvcost = 3;                                                              // raster done
        if (bb != ~0ull && (int)(bb >> 16) < st.cost)
        {   // COPY2_IF_LT in raster order against the best of the pattern pass
            const int k = (int)(bb & 0xFFFFu), gj = k / NX, gi = k - gj * NX;
            r.mv[0] = (int16_t)(mnx + RD * gi); r.mv[1] = (int16_t)(mny + RD * gj); r.cost = (int)(bb >> 16);
        }
        results[item] = r;
    }
}

} // namespace

// Can the full-pel STAR search of this call run here?  64x64 PUs, row pitch a multiple of 4 bytes (the band is the dword image of the rows),
// windows of at most (2 * 57 + 64)^2 pixels.
bool xh_star64_ok(intptr_t refStride, int merange)
{
    return ((refStride * (intptr_t)sizeof(pixel)) & 3) == 0 && merange >= 1 && merange <= MAXR;
}

int xh_star64(void* stream, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride, const x265hip_me_task* tasks, int n,
              const uint16_t* costCentre, int costHalfRange, int merange, x265hip_me_result* results, const x265hip_me_result* mvpSource)
{
    static const int dbg = xh_experiment("X265HIP_STAR64_DBG") ? atoi(xh_experiment("X265HIP_STAR64_DBG")) : 0;      // timing experiments only (results are wrong with it)
    hipLaunchKernelGGL(star64_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (const pixel*)curPlane, curStride, (const pixel*)refPlane, refStride,
                       tasks, n, costCentre, costHalfRange, merange, results, mvpSource, dbg);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

// the raster-only mode: any window whose placements per row / column fit the 16-bit order field (merange <= 160)
bool xh_star64_raster_ok(intptr_t refStride, int merange)
{
    return ((refStride * (intptr_t)sizeof(pixel)) & 3) == 0 && merange > MAXR && (2 * merange) / RD + 1 <= 255;
}
int xh_star64_raster(void* stream, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride, const x265hip_me_task* tasks, int n,
                     const uint16_t* costCentre, int costHalfRange, int merange, x265hip_me_result* results, const x265hip_me_result* mvpSource)
{
    hipLaunchKernelGGL(star64_raster_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (const pixel*)curPlane, curStride, (const pixel*)refPlane, refStride,
                       tasks, n, costCentre, costHalfRange, merange, results, mvpSource);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}
