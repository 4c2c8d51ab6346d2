// kern_me.hip -- batched motion estimation: one wavefront runs the reference's complete per-PU search
// (MotionEstimate::motionEstimate, encoder/motion.cpp:923-1773) for one (PU, reference) pair:
//   start point (clipped MVP at sub-pel SAD, rounded MVP, zero MV, candidates)  motion.cpp:955-1012
//   integer pattern search DIA / HEX / STAR                                     motion.cpp:1016-1140,1328-1436
//   sub-pel refinement per workload[subme] with luma_hpp/vpp/hvpp + SATD        motion.cpp:1644-1757
//   final zero-MV check                                                          motion.cpp:1763-1768
// Candidate ORDER and strict `<` comparisons are part of the result (COPYn_IF_LT), so the decision logic
// is restated literally and executed wave-uniformly; the data-parallel work (SAD / interpolation /
// Hadamard over the PU) is spread over the 64 lanes in units of 4 horizontally adjacent pixels
// (v_sad_u8 / v_sad_u16 on packed pixels).  The source PU is cached in LDS (the reference's FENC_STRIDE
// cache, motion.cpp:223-229); interpolated candidates and the 14-bit hv intermediate live in LDS too.
#include "xh_mc.h"
#include "../../include/x265hip_frame.h"
#include <cstdlib>
using namespace xh;

namespace {

__device__ const int8_t k_hex2[8][2] = { {-1,-2}, {-2,0}, {-1,2}, {1,2}, {2,0}, {1,-2}, {-1,-2}, {-2,0} };
__device__ const uint8_t k_mod6m1[8] = { 5, 0, 1, 2, 3, 4, 5, 0 };
__device__ const int8_t k_square1[9][2] = { {0,0}, {0,-1}, {0,1}, {-1,0}, {1,0}, {-1,-1}, {-1,1}, {1,-1}, {1,1} };
__device__ const int8_t k_offsets[16][2] = { {-1,0}, {0,-1}, {-1,-1}, {1,-1}, {-1,0}, {1,0}, {-1,1}, {-1,-1},
                                             {1,-1}, {1,1}, {-1,0}, {0,1}, {-1,1}, {1,1}, {1,0}, {0,1} };
__device__ const int8_t k_workload[8][5] = { {1,4,0,4,0}, {1,4,1,4,0}, {1,4,1,4,1}, {2,4,1,4,1}, {2,4,2,4,1}, {1,8,1,8,1}, {2,8,1,8,1}, {2,8,2,8,1} };

struct Ctx : McCtx
{
    lpixel* fenc; lint* cl;              // per-wave LDS: source PU (stride w), candidate list
    const uint16_t* cost; int mvpx, mvpy;
    int dbg;
};

__device__ __forceinline__ int mvcost(const Ctx& c, int qx, int qy)
{   // bitcost.h:57 -- uint16_t sum
    return (uint16_t)(c.cost[qx - c.mvpx] + c.cost[qy - c.mvpy]);
}

// ---- integer-pel SAD of up to 4 candidates listed in LDS; costs (SAD + mvcost) go back to the list ----
// cl layout: [0..15] x, [16..31] y, [32..47] cost
template<int G> __device__ void eval_list(const Ctx& c, int n)
{
    wave_sync();
    for (int base = 0; base < n; base += 4)
    {
        const int k = n - base < 4 ? n - base : 4;
        intptr_t off[4];
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            int jj = j < k ? base + j : base;
            off[j] = (intptr_t)c.cl[16 + jj] * c.rs + c.cl[jj];
        }
        unsigned p0 = 0, p1 = 0, p2 = 0, p3 = 0;
        QUAD_LOOP(c, q, y, x4)
            const lpixel* f = c.fenc + y * c.w + x4;
            const pixel* r = c.fref + (intptr_t)y * c.rs + x4;
            p0 = sad4(f, r + off[0], p0);
            if (k > 1) p1 = sad4(f, r + off[1], p1);
            if (k > 2) p2 = sad4(f, r + off[2], p2);
            if (k > 3) p3 = sad4(f, r + off[3], p3);
        QUAD_END
        int s0 = group_sum<G>((int)p0), s1 = k > 1 ? group_sum<G>((int)p1) : 0, s2 = k > 2 ? group_sum<G>((int)p2) : 0, s3 = k > 3 ? group_sum<G>((int)p3) : 0;
        if (c.lane == 0)
        {
            c.cl[32 + base] = s0 + mvcost(c, c.cl[base] * 4, c.cl[16 + base] * 4);
            if (k > 1) c.cl[32 + base + 1] = s1 + mvcost(c, c.cl[base + 1] * 4, c.cl[16 + base + 1] * 4);
            if (k > 2) c.cl[32 + base + 2] = s2 + mvcost(c, c.cl[base + 2] * 4, c.cl[16 + base + 2] * 4);
            if (k > 3) c.cl[32 + base + 3] = s3 + mvcost(c, c.cl[base + 3] * 4, c.cl[16 + base + 3] * 4);
        }
    }
    wave_sync();
}
__device__ __forceinline__ void put(const Ctx& c, int i, int x, int y)
{
    c.cl[i] = x; c.cl[16 + i] = y;     // every lane stores the same (wave-uniform) value
}
__device__ __forceinline__ int costOf(const Ctx& c, int i) { return c.cl[32 + i]; }

template<int G> __device__ int sad_fpel(const Ctx& c, int mx, int my)      // plain SAD (no mv cost)
{
    unsigned p = 0;
    const intptr_t off = (intptr_t)my * c.rs + mx;
    QUAD_LOOP(c, q, y, x4)
        p = sad4(c.fenc + y * c.w + x4, c.fref + off + (intptr_t)y * c.rs + x4, p);
    QUAD_END
    return group_sum<G>((int)p);
}

__device__ __forceinline__ void had4(int& a, int& b, int& cc, int& d)
{
    int t0 = a + b, t1 = a - b, t2 = cc + d, t3 = cc - d;
    a = t0 + t2; cc = t0 - t2; b = t1 + t3; d = t1 - t3;
}
__device__ __forceinline__ int had4x4_lds(const lpixel* f, const lpixel* p, int w)
{
    int d[16];
#pragma unroll
    for (int y = 0; y < 4; y++)
    {
        int a[4], b[4]; load4(f + y * w, a); load4(p + y * w, b);
#pragma unroll
        for (int x = 0; x < 4; x++) d[4 * y + x] = a[x] - b[x];
        had4(d[4 * y], d[4 * y + 1], d[4 * y + 2], d[4 * y + 3]);
    }
    int s = 0;
#pragma unroll
    for (int x = 0; x < 4; x++)
    {
        had4(d[x], d[4 + x], d[8 + x], d[12 + x]);
        s += abs(d[x]) + abs(d[4 + x]) + abs(d[8 + x]) + abs(d[12 + x]);
    }
    return s;
}
// SAD or SATD between the cached source PU and the candidate block in LDS (SATD rounding per pixel.cpp:1148-1172)
template<int G> __device__ int cmp_pred(const Ctx& c, bool satd)
{
    int s = 0;
    if (!satd)
    {
        unsigned p = 0;
        QUAD_LOOP(c, q, y, x4)
#if X265_DEPTH == 8
            p = __builtin_amdgcn_sad_u8(*(const lu32*)(c.fenc + y * c.w + x4), *(const lu32*)(c.pred + y * c.w + x4), p);
#else
            u32x2 a = *(const lu2*)(c.fenc + y * c.w + x4), b = *(const lu2*)(c.pred + y * c.w + x4);
            p = __builtin_amdgcn_sad_u16(a.x, b.x, p); p = __builtin_amdgcn_sad_u16(a.y, b.y, p);
#endif
        QUAD_END
        s = (int)p;
    }
    else
    {
        const bool use4 = (c.w == 4) || (c.w == 12);
        const int uw = use4 ? 4 : 8, ux = c.w / uw, nunits = ux * (c.h >> 2);
        for (int u = c.lane; u < nunits; u += G)
        {
            int uy = u / ux, x0 = (u - uy * ux) * uw, y0 = uy * 4;
            const lpixel* f = c.fenc + y0 * c.w + x0; const lpixel* p = c.pred + y0 * c.w + x0;
            int v = had4x4_lds(f, p, c.w);
            if (!use4) v += had4x4_lds(f + 4, p + 4, c.w);
            s += v >> 1;
        }
    }
    return group_sum<G>(s);
}
// motion.cpp:1775-1803 subpelCompare (luma)
template<int G> __device__ int subpel_cost(const Ctx& c, int qx, int qy, bool satd)
{
    if (!satd && !((qx | qy) & 3)) return sad_fpel<G>(c, qx >> 2, qy >> 2);
    build_pred(c, qx, qy);
    return cmp_pred<G>(c, satd);
}

struct Star { int bx, by, bcost, bPointNr, bDistance; };

// Evaluate the listed points in order with strict-< updates (COST_MV_PT_DIST semantics); pt/dist packed in cl[48..63]
template<int G> __device__ void star_apply(const Ctx& c, Star& s, int n)
{
    eval_list<G>(c, n);
    for (int i = 0; i < n; i++)
    {
        int cost = costOf(c, i);
        if (cost < s.bcost)
        {
            s.bcost = cost; s.bx = c.cl[i]; s.by = c.cl[16 + i];
            int pd = c.cl[48 + i]; s.bPointNr = pd & 0xFF; s.bDistance = pd >> 8;
        }
    }
}
#define SPUT(i, X, Y, P, D) do { c.cl[i] = (X); c.cl[16 + (i)] = (Y); c.cl[48 + (i)] = (P) | ((D) << 8); } while (0)

// motion.cpp:387-629 StarPatternSearch
template<int G> __device__ void star_pattern(const Ctx& c, int mnx, int mny, int mxx, int mxy, Star& s, int earlyExitIters, int merange)
{
    const int ox = s.bx, oy = s.by;
    int saved = s.bcost, rounds = 0;
    {
        const int dist = 1, top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        int n = 0;
        if (top >= mny) { SPUT(n, ox, top, 2, dist); n++; }
        if (left >= mnx) { SPUT(n, left, oy, 4, dist); n++; }
        if (right <= mxx) { SPUT(n, right, oy, 5, dist); n++; }
        if (bottom <= mxy) { SPUT(n, ox, bottom, 7, dist); n++; }
        star_apply<G>(c, s, n);
        if (s.bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
    for (int dist = 2; dist <= 8; dist <<= 1)
    {
        const int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        const int top2 = oy - (dist >> 1), bottom2 = oy + (dist >> 1), left2 = ox - (dist >> 1), right2 = ox + (dist >> 1);
        saved = s.bcost;
        int n = 0;
        if (top >= mny) { SPUT(n, ox, top, 2, dist); n++; }
        if (top2 >= mny)
        {
            if (left2 >= mnx) { SPUT(n, left2, top2, 1, dist >> 1); n++; }
            if (right2 <= mxx) { SPUT(n, right2, top2, 3, dist >> 1); n++; }
        }
        if (left >= mnx) { SPUT(n, left, oy, 4, dist); n++; }
        if (right <= mxx) { SPUT(n, right, oy, 5, dist); n++; }
        if (bottom2 <= mxy)
        {
            if (left2 >= mnx) { SPUT(n, left2, bottom2, 6, dist >> 1); n++; }
            if (right2 <= mxx) { SPUT(n, right2, bottom2, 8, dist >> 1); n++; }
        }
        if (bottom <= mxy) { SPUT(n, ox, bottom, 7, dist); n++; }
        star_apply<G>(c, s, n);
        if (s.bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
    for (int dist = 16; dist <= (int)(int16_t)merange; dist <<= 1)
    {
        const int top = oy - dist, bottom = oy + dist, left = ox - dist, right = ox + dist;
        saved = s.bcost;
        int n = 0;
        if (top >= mny) { SPUT(n, ox, top, 0, dist); n++; }
        if (left >= mnx) { SPUT(n, left, oy, 0, dist); n++; }
        if (right <= mxx) { SPUT(n, right, oy, 0, dist); n++; }
        if (bottom <= mxy) { SPUT(n, ox, bottom, 0, dist); n++; }
        for (int index = 1; index < 4; index++)
        {
            const int posYT = top + ((dist >> 2) * index), posYB = bottom - ((dist >> 2) * index);
            const int posXL = ox - ((dist >> 2) * index), posXR = ox + ((dist >> 2) * index);
            if (posYT >= mny)
            {
                if (posXL >= mnx) { SPUT(n, posXL, posYT, 0, dist); n++; }
                if (posXR <= mxx) { SPUT(n, posXR, posYT, 0, dist); n++; }
            }
            if (posYB <= mxy)
            {
                if (posXL >= mnx) { SPUT(n, posXL, posYB, 0, dist); n++; }
                if (posXR <= mxx) { SPUT(n, posXR, posYB, 0, dist); n++; }
            }
        }
        star_apply<G>(c, s, n);
        if (s.bcost < saved) rounds = 0;
        else if (++rounds >= earlyExitIters) return;
    }
}

template<int G, int MAXPIX, int MAXW, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void me_kernel(int w, int h, const pixel* __restrict__ cur, intptr_t cs,
                                                        const pixel* __restrict__ ref, intptr_t rs,
                                                        const x265hip_me_task* __restrict__ tasks, int n,
                                                        const uint16_t* __restrict__ costCentre,
                                                        int merange, int method, int subme, x265hip_me_result* __restrict__ results,
                                                        const x265hip_me_result* __restrict__ mvpSource, int dbg)
{
    constexpr int GROUPS = WAVES * (64 / G);          // PUs per workgroup: 64/G lane groups per wavefront
    __shared__ __attribute__((aligned(16))) pixel s_fenc[GROUPS][MAXPIX];
    __shared__ __attribute__((aligned(16))) pixel s_pred[GROUPS][MAXPIX];
    __shared__ __attribute__((aligned(16))) int16_t s_immed[GROUPS][MAXPIX + 7 * MAXW];
    __shared__ int s_cl[GROUPS][64];

    const int wave = threadIdx.x / G;                 // index of this lane group inside the workgroup
    const int item = blockIdx.x * GROUPS + wave;
    if (item >= n) return;            // group-granular exit; no block barriers are used
    x265hip_me_task tk = tasks[item];
    if (tk.mvpFrom >= 0 && mvpSource) { tk.qmvp[0] = mvpSource[tk.mvpFrom].mv[0]; tk.qmvp[1] = mvpSource[tk.mvpFrom].mv[1]; }

    Ctx c;
    c.setGeometry(w, h, threadIdx.x & (G - 1), G);
    c.fenc = (lpixel*)s_fenc[wave]; c.pred = (lpixel*)s_pred[wave]; c.immed = (lshort*)s_immed[wave]; c.cl = (lint*)s_cl[wave];
    c.dbg = dbg; c.fref = ref + tk.refOff; c.rs = rs; c.cost = costCentre; c.mvpx = tk.qmvp[0]; c.mvpy = tk.qmvp[1];

    // cache the source PU (motion.cpp:223-229)
    {
        const pixel* src = cur + tk.curOff;
        QUAD_LOOP(c, q, y, x4)
            int v[4]; load4u(src + (intptr_t)y * cs + x4, v); store4(c.fenc + y * w + x4, v);
        QUAD_END
        wave_sync();
    }
    if (dbg == 1) return;
    if (dbg == 7)
    {   // debug: dump the interpolated block at the predictor position
        build_pred(c, tk.qmvp[0], tk.qmvp[1]);
        for (int i = c.lane; i < w * h; i += G) ((pixel*)results)[i] = c.pred[i];
        return;
    }
    if (dbg == 5 || dbg == 6)
    {   // debug: raw sub-pel cost of the predictor position (SATD for 5, SAD for 6)
        int v = subpel_cost<G>(c, tk.qmvp[0], tk.qmvp[1], dbg == 5);
        if (c.lane == 0) { x265hip_me_result r; r.mv[0] = tk.qmvp[0]; r.mv[1] = tk.qmvp[1]; r.cost = v; r.mvcost = 0; r.reserved = 0; results[item] = r; }
        return;
    }
    int mnx = tk.mvmin[0], mny = tk.mvmin[1], mxx = tk.mvmax[0], mxy = tk.mvmax[1];
    if (tk.flags & X265HIP_ME_WINDOW)
    {   // search.cpp:4969-5021 setSearchRange: clip mvp -/+ 4*merange to the quarter-pel limits, then to full-pel
        const int lx0 = mnx, ly0 = mny, lx1 = mxx, ly1 = mxy, d = merange << 2;
        mnx = min(lx1, max(lx0, tk.qmvp[0] - d)) >> 2; mny = min(ly1, max(ly0, tk.qmvp[1] - d)) >> 2;
        mxx = min(lx1, max(lx0, tk.qmvp[0] + d)) >> 2; mxy = min(ly1, max(ly0, tk.qmvp[1] + d)) >> 2;
        mxy = max(mxy, mny);
    }
    const int qmnx = mnx * 4, qmny = mny * 4, qmxx = mxx * 4, qmxy = mxy * 4;

    // ---- start point, motion.cpp:955-1012 ----
    int pmx = min(max((int)tk.qmvp[0], qmnx), qmxx), pmy = min(max((int)tk.qmvp[1], qmny), qmxy);
    int bestprex = pmx, bestprey = pmy;
    int bprecost = subpel_cost<G>(c, pmx, pmy, false);
    int bx = (pmx + 2) >> 2, by = (pmy + 2) >> 2;
    int bcost = bprecost;
    if ((pmx | pmy) & 3) bcost = sad_fpel<G>(c, bx, by) + mvcost(c, bx * 4, by * 4);
    if (pmx | pmy)
    {
        int cost = sad_fpel<G>(c, 0, 0) + mvcost(c, 0, 0);
        if (cost < bcost) { bcost = cost; bx = 0; by = max(min(0, mxy), mny); }
    }
    for (int i = 0; i < tk.numCand; i++)
    {
        int mx = min(max((int)tk.mvc[2 * i], qmnx), qmxx), my = min(max((int)tk.mvc[2 * i + 1], qmny), qmxy);
        if ((mx | my) && !(mx == pmx && my == pmy) && !(mx == bestprex && my == bestprey))
        {
            int cost = subpel_cost<G>(c, mx, my, false) + mvcost(c, mx, my);
            if (cost < bprecost) { bprecost = cost; bestprex = mx; bestprey = my; }
        }
    }
    if (dbg == 2) return;
    bool finished = false;
    int outx = 0, outy = 0, outcost = 0;
    if (bcost == 0)
    {
        outx = bx * 4; outy = by * 4; outcost = mvcost(c, outx, outy); finished = true;
    }

    if (!finished)
    {
        if (method == X265HIP_ME_DIA)
        {   // motion.cpp:1016-1039
            int bc = bcost << 4;
            int i = merange;
            do
            {
                put(c, 0, bx, by - 1); put(c, 1, bx, by + 1); put(c, 2, bx - 1, by); put(c, 3, bx + 1, by);
                eval_list<G>(c, 4);
                if ((by - 1 >= mny) & (by - 1 <= mxy)) bc = min(bc, (costOf(c, 0) << 4) + 1);
                if ((by + 1 >= mny) & (by + 1 <= mxy)) bc = min(bc, (costOf(c, 1) << 4) + 3);
                bc = min(bc, (costOf(c, 2) << 4) + 4);
                bc = min(bc, (costOf(c, 3) << 4) + 12);
                if (!(bc & 15)) break;
                bx -= (int32_t)((uint32_t)bc << 28) >> 30;
                by -= (int32_t)((uint32_t)bc << 30) >> 30;
                bc &= ~15;
            }
            while (--i && bx >= mnx && bx <= mxx && by >= mny && by <= mxy);
            bcost = bc >> 4;
        }
        else if (method == X265HIP_ME_HEX)
        {   // motion.cpp:1041-1140
            put(c, 0, bx - 2, by); put(c, 1, bx - 1, by + 2); put(c, 2, bx + 1, by + 2);
            put(c, 3, bx + 2, by); put(c, 4, bx + 1, by - 2); put(c, 5, bx - 1, by - 2);
            eval_list<G>(c, 6);
            bcost <<= 3;
            if ((by >= mny) & (by <= mxy)) bcost = min(bcost, (costOf(c, 0) << 3) + 2);
            if ((by + 2 >= mny) & (by + 2 <= mxy)) { bcost = min(bcost, (costOf(c, 1) << 3) + 3); bcost = min(bcost, (costOf(c, 2) << 3) + 4); }
            if ((by >= mny) & (by <= mxy)) bcost = min(bcost, (costOf(c, 3) << 3) + 5);
            if ((by - 2 >= mny) & (by - 2 <= mxy)) { bcost = min(bcost, (costOf(c, 4) << 3) + 6); bcost = min(bcost, (costOf(c, 5) << 3) + 7); }
            if (bcost & 7)
            {
                int dir = (bcost & 7) - 2;
                if ((by + k_hex2[dir + 1][1] >= mny) & (by + k_hex2[dir + 1][1] <= mxy))
                {
                    bx += k_hex2[dir + 1][0]; by += k_hex2[dir + 1][1];
                    for (int i = (merange >> 1) - 1; i > 0 && bx >= mnx && bx <= mxx && by >= mny && by <= mxy; i--)
                    {
                        put(c, 0, bx + k_hex2[dir][0], by + k_hex2[dir][1]);
                        put(c, 1, bx + k_hex2[dir + 1][0], by + k_hex2[dir + 1][1]);
                        put(c, 2, bx + k_hex2[dir + 2][0], by + k_hex2[dir + 2][1]);
                        eval_list<G>(c, 3);
                        bcost &= ~7;
                        if ((by + k_hex2[dir][1] >= mny) & (by + k_hex2[dir][1] <= mxy)) bcost = min(bcost, (costOf(c, 0) << 3) + 1);
                        if ((by + k_hex2[dir + 1][1] >= mny) & (by + k_hex2[dir + 1][1] <= mxy)) bcost = min(bcost, (costOf(c, 1) << 3) + 2);
                        if ((by + k_hex2[dir + 2][1] >= mny) & (by + k_hex2[dir + 2][1] <= mxy)) bcost = min(bcost, (costOf(c, 2) << 3) + 3);
                        if (!(bcost & 7)) break;
                        dir += (bcost & 7) - 2;
                        dir = k_mod6m1[dir + 1];
                        bx += k_hex2[dir + 1][0]; by += k_hex2[dir + 1][1];
                    }
                }
            }
            bcost >>= 3;
            // square refine
            put(c, 0, bx, by - 1); put(c, 1, bx, by + 1); put(c, 2, bx - 1, by); put(c, 3, bx + 1, by);
            put(c, 4, bx - 1, by - 1); put(c, 5, bx - 1, by + 1); put(c, 6, bx + 1, by - 1); put(c, 7, bx + 1, by + 1);
            eval_list<G>(c, 8);
            const bool upOk = (by - 1 >= mny) & (by - 1 <= mxy), dnOk = (by + 1 >= mny) & (by + 1 <= mxy);
            int dir = 0, cc;
            if (upOk && (cc = costOf(c, 0)) < bcost) { bcost = cc; dir = 1; }
            if (dnOk && (cc = costOf(c, 1)) < bcost) { bcost = cc; dir = 2; }
            if ((cc = costOf(c, 2)) < bcost) { bcost = cc; dir = 3; }
            if ((cc = costOf(c, 3)) < bcost) { bcost = cc; dir = 4; }
            if (upOk && (cc = costOf(c, 4)) < bcost) { bcost = cc; dir = 5; }
            if (dnOk && (cc = costOf(c, 5)) < bcost) { bcost = cc; dir = 6; }
            if (upOk && (cc = costOf(c, 6)) < bcost) { bcost = cc; dir = 7; }
            if (dnOk && (cc = costOf(c, 7)) < bcost) { bcost = cc; dir = 8; }
            bx += k_square1[dir][0]; by += k_square1[dir][1];
        }
        else if (method == X265HIP_ME_STAR)
        {   // motion.cpp:1328-1436
            Star s = { bx, by, bcost, 0, 0 };
            star_pattern<G>(c, mnx, mny, mxx, mxy, s, 3, merange);
            bool done = false;
            if (s.bDistance == 1)
            {
                if (s.bPointNr)
                {
                    const int saved = s.bcost;
                    int n = 0;
                    const int x1 = s.bx + k_offsets[(s.bPointNr - 1) * 2][0], y1 = s.by + k_offsets[(s.bPointNr - 1) * 2][1];
                    const int x2 = s.bx + k_offsets[(s.bPointNr - 1) * 2 + 1][0], y2 = s.by + k_offsets[(s.bPointNr - 1) * 2 + 1][1];
                    if (x1 >= mnx && x1 <= mxx && y1 >= mny && y1 <= mxy) { SPUT(n, x1, y1, s.bPointNr, s.bDistance); n++; }
                    if (x2 >= mnx && x2 <= mxx && y2 >= mny && y2 <= mxy) { SPUT(n, x2, y2, s.bPointNr, s.bDistance); n++; }
                    star_apply<G>(c, s, n);      // COST_MV keeps bPointNr/bDistance: the packed values re-store the current ones
                    if (s.bcost == saved) done = true;
                }
                else done = true;
            }
            if (!done)
            {
                const int RasterDistance = 5;
                if (s.bDistance > RasterDistance)
                {   // raster refinement over the whole window, step 5 (:1365-1399), including the `tmv << 3` quirk (:1392)
                    for (int ty = mny; ty <= mxy; ty += RasterDistance)
                        for (int tx = mnx; tx <= mxx; tx += RasterDistance)
                        {
                            if (tx + RasterDistance * 3 <= mxx)
                            {
                                put(c, 0, tx, ty); put(c, 1, tx + 5, ty); put(c, 2, tx + 10, ty); put(c, 3, tx + 15, ty);
                                eval_list<G>(c, 4);
                                for (int k = 0; k < 4; k++)
                                {
                                    int cost = costOf(c, k);
                                    if (k == 3) cost += mvcost(c, (tx + 15) * 8, ty * 8) - mvcost(c, (tx + 15) * 4, ty * 4);
                                    if (cost < s.bcost) { s.bcost = cost; s.bx = tx + 5 * k; s.by = ty; }
                                }
                                tx += 15;
                            }
                            else
                            {
                                put(c, 0, tx, ty);
                                eval_list<G>(c, 1);
                                int cost = costOf(c, 0);
                                if (cost < s.bcost) { s.bcost = cost; s.bx = tx; s.by = ty; }
                            }
                        }
                }
                int bDistance = s.bDistance;
                while (bDistance > 0)
                {
                    s.bPointNr = 0; s.bDistance = 0;
                    star_pattern<G>(c, mnx, mny, mxx, mxy, s, 32, merange);
                    bDistance = s.bDistance;
                    if (bDistance == 1)
                    {
                        if (!s.bPointNr) break;
                        int n = 0;
                        const int x1 = s.bx + k_offsets[(s.bPointNr - 1) * 2][0], y1 = s.by + k_offsets[(s.bPointNr - 1) * 2][1];
                        const int x2 = s.bx + k_offsets[(s.bPointNr - 1) * 2 + 1][0], y2 = s.by + k_offsets[(s.bPointNr - 1) * 2 + 1][1];
                        if (x1 >= mnx && x1 <= mxx && y1 >= mny && y1 <= mxy) { SPUT(n, x1, y1, 0, 0); n++; }
                        if (x2 >= mnx && x2 <= mxx && y2 >= mny && y2 <= mxy) { SPUT(n, x2, y2, 0, 0); n++; }
                        star_apply<G>(c, s, n);
                        break;
                    }
                }
            }
            bx = s.bx; by = s.by; bcost = s.bcost;
        }

        if (dbg == 3) return;
        // ---- sub-pel refinement, motion.cpp:1644-1768 ----
        int qx, qy;
        if (bprecost < bcost) { qx = bestprex; qy = bestprey; bcost = bprecost; }
        else { qx = bx * 4; qy = by * 4; }
        if (!bcost)
            bcost = mvcost(c, qx, qy);
        else
        {
            const int hpelIters = k_workload[subme][0], hpelDirs = k_workload[subme][1], qpelIters = k_workload[subme][2], qpelDirs = k_workload[subme][3];
            const bool hpelSatd = k_workload[subme][4] != 0;
            if (hpelSatd) bcost = subpel_cost<G>(c, qx, qy, true) + mvcost(c, qx, qy);
            for (int iter = 0; iter < hpelIters; iter++)
            {
                int bdir = 0;
                for (int i = 1; i <= hpelDirs; i++)
                {
                    const int tx = qx + k_square1[i][0] * 2, ty = qy + k_square1[i][1] * 2;
                    if ((ty < qmny) | (ty > qmxy)) continue;
                    int cost = subpel_cost<G>(c, tx, ty, hpelSatd) + mvcost(c, tx, ty);
                    if (cost < bcost) { bcost = cost; bdir = i; }
                }
                if (bdir) { qx += k_square1[bdir][0] * 2; qy += k_square1[bdir][1] * 2; }
                else break;
            }
            if (!hpelSatd) bcost = subpel_cost<G>(c, qx, qy, true) + mvcost(c, qx, qy);
            for (int iter = 0; iter < qpelIters; iter++)
            {
                int bdir = 0;
                for (int i = 1; i <= qpelDirs; i++)
                {
                    const int tx = qx + k_square1[i][0], ty = qy + k_square1[i][1];
                    if ((ty < qmny) | (ty > qmxy)) continue;
                    int cost = subpel_cost<G>(c, tx, ty, true) + mvcost(c, tx, ty);
                    if (cost < bcost) { bcost = cost; bdir = i; }
                }
                if (bdir) { qx += k_square1[bdir][0]; qy += k_square1[bdir][1]; }
                else break;
            }
        }
        if (qx | qy)
        {
            int cost = subpel_cost<G>(c, 0, 0, true) + mvcost(c, 0, 0);
            if (cost <= bcost) { qx = 0; qy = 0; }
        }
        outx = qx; outy = qy; outcost = bcost;
    }
    if (c.lane == 0)
    {
        x265hip_me_result r;
        r.mv[0] = (int16_t)outx; r.mv[1] = (int16_t)outy; r.cost = outcost; r.mvcost = mvcost(c, outx, outy); r.reserved = 0;
        results[item] = r;
    }
}

template<int G, int MAXPIX, int MAXW, int WAVES>
int launch_me(hipStream_t st, int w, int h, const pixel* cur, intptr_t cs, const pixel* ref, intptr_t rs, const x265hip_me_task* tasks, int n,
              const uint16_t* costCentre, int merange, int method, int subme, x265hip_me_result* results, const x265hip_me_result* mvpSource)
{
    constexpr int GROUPS = WAVES * (64 / G);
    hipLaunchKernelGGL((me_kernel<G, MAXPIX, MAXW, WAVES>), dim3((n + GROUPS - 1) / GROUPS), dim3(64 * WAVES), 0, st,
                       w, h, cur, cs, ref, rs, tasks, n, costCentre, merange, method, subme, results, mvpSource, getenv("X265HIP_ME_DBG") ? atoi(getenv("X265HIP_ME_DBG")) : 0);
    XH_LAUNCH_CHECK();
    return X265HIP_OK;
}

} // namespace

extern "C" int x265hip_me_batch(void* stream, int w, int h, const void* curPlane, intptr_t curStride, const void* refPlane, intptr_t refStride,
                                const x265hip_me_task* tasks, int n, const uint16_t* costRow, int costHalfRange,
                                int merange, int method, int subpelRefine, x265hip_me_result* results, const x265hip_me_result* mvpSource)
{
    if (n <= 0) return X265HIP_OK;
    if (w < 4 || h < 4 || w > 64 || h > 64 || ((w | h) & 3) || !tasks || !results || !costRow || costHalfRange < 1)
    { set_error("me_batch: bad arguments"); return X265HIP_EARG; }
    if (method != X265HIP_ME_DIA && method != X265HIP_ME_HEX && method != X265HIP_ME_STAR)
    { set_error("me_batch: search method %d is not offloaded (DIA/HEX/STAR are)", method); return X265HIP_EARG; }
    if (subpelRefine < 0 || subpelRefine > 7 || merange < 1) { set_error("me_batch: bad subme/merange"); return X265HIP_EARG; }
    hipStream_t st = (hipStream_t)stream;
    const pixel* cur = (const pixel*)curPlane; const pixel* ref = (const pixel*)refPlane;
    const uint16_t* centre = costRow + costHalfRange;
    const int area = w * h;
    const int nquads = area / 4;
    if (nquads <= 8) return launch_me<8, 32, 8, 4>(st, w, h, cur, curStride, ref, refStride, tasks, n, centre, merange, method, subpelRefine, results, mvpSource);   // 8x4, 4x8
    if (area <= 64 && w <= 16) return launch_me<16, 64, 16, 4>(st, w, h, cur, curStride, ref, refStride, tasks, n, centre, merange, method, subpelRefine, results, mvpSource);   // 8x8, 16x4, 4x16
    if (nquads <= 32 && w <= 16) return launch_me<32, 128, 16, 4>(st, w, h, cur, curStride, ref, refStride, tasks, n, centre, merange, method, subpelRefine, results, mvpSource);   // 16x8, 8x16
    if (area <= 256 && w <= 32) return launch_me<64, 256, 32, 4>(st, w, h, cur, curStride, ref, refStride, tasks, n, centre, merange, method, subpelRefine, results, mvpSource);
    if (area <= 1024) return launch_me<64, 1024, 64, 4>(st, w, h, cur, curStride, ref, refStride, tasks, n, centre, merange, method, subpelRefine, results, mvpSource);
    return launch_me<64, 4096, 64, 2>(st, w, h, cur, curStride, ref, refStride, tasks, n, centre, merange, method, subpelRefine, results, mvpSource);
}
