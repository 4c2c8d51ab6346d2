// xh_amvp.h -- CUData::getPMV as a device function (see kern_amvp.hip): shared by the batch entry point and the ThreadedME stepping kernels (kern_tme.hip)
#pragma once
#include "xh_common.h"
#include "../../include/x265hip_frame.h"

namespace xh {

struct Mv2 { int x, y; };

__device__ __forceinline__ Mv2 scale_poc(Mv2 in, int curPOC, int curRefPOC, int colPOC, int colRefPOC)
{
    const int diffPocD = colPOC - colRefPOC, diffPocB = curPOC - curRefPOC;
    if (diffPocD == diffPocB) return in;
    const int tdb = clip3(-128, 127, diffPocB), tdd = clip3(-128, 127, diffPocD);
    const int x = (0x4000 + abs(tdd / 2)) / tdd;
    const int scale = clip3(-4096, 4095, (tdb * x + 32) >> 6);
    Mv2 o;
    o.x = clip3(-32768, 32767, (scale * in.x + 127 + (scale * in.x < 0)) >> 8);
    o.y = clip3(-32768, 32767, (scale * in.y + 127 + (scale * in.y < 0)) >> 8);
    return o;
}

// cudata.cpp:1806-1990
__device__ __forceinline__ x265hip_amvp_result get_pmv(const x265hip_amvp_task& t, const x265hip_amvp_params& p)
{
    const int list = t.list, curRefPOC = p.refPOC[list][t.refIdx];
    Mv2 direct[5], indirect[5];
    bool vd[5], vi[5];
#pragma unroll
    for (int d = 0; d < 5; d++)
    {
        vd[d] = vi[d] = false;
#pragma unroll
        for (int k = 0; k < 2; k++)
        {
            const int l = k ? !list : list, r = t.nb[d].refIdx[l];
            if (!vd[d] && r >= 0 && curRefPOC == p.refPOC[l][r]) { direct[d].x = t.nb[d].mv[l][0]; direct[d].y = t.nb[d].mv[l][1]; vd[d] = true; }
        }
#pragma unroll
        for (int k = 0; k < 2; k++)
        {
            const int l = k ? !list : list, r = t.nb[d].refIdx[l];
            if (!vi[d] && r >= 0)
            {
                Mv2 m; m.x = t.nb[d].mv[l][0]; m.y = t.nb[d].mv[l][1];
                indirect[d] = scale_poc(m, p.curPOC, curRefPOC, p.curPOC, p.refPOC[l][r]);
                vi[d] = true;
            }
        }
    }
    enum { LEFT, ABOVE, ABOVE_RIGHT, BELOW_LEFT, ABOVE_LEFT, COLLOCATED };
    Mv2 c[2]; int num = 0;
    c[0].x = c[0].y = c[1].x = c[1].y = 0;
    if (vd[BELOW_LEFT]) c[num++] = direct[BELOW_LEFT];
    else if (vd[LEFT]) c[num++] = direct[LEFT];
    else if (vi[BELOW_LEFT]) c[num++] = indirect[BELOW_LEFT];
    else if (vi[LEFT]) c[num++] = indirect[LEFT];
    const bool addedSmvp = num > 0;
    if (vd[ABOVE_RIGHT]) c[num++] = direct[ABOVE_RIGHT];
    else if (vd[ABOVE]) c[num++] = direct[ABOVE];
    else if (vd[ABOVE_LEFT]) c[num++] = direct[ABOVE_LEFT];
    if (!addedSmvp)
    {
        if (vi[ABOVE_RIGHT]) c[num++] = indirect[ABOVE_RIGHT];
        else if (vi[ABOVE]) c[num++] = indirect[ABOVE];
        else if (vi[ABOVE_LEFT]) c[num++] = indirect[ABOVE_LEFT];
    }
    x265hip_amvp_result r;
    int numMvc = 0;
#pragma unroll
    for (int k = 0; k < 11; k++) { r.mvc[k][0] = 0; r.mvc[k][1] = 0; }
#pragma unroll
    for (int d = LEFT; d <= ABOVE_LEFT; d++)
    {
        if (vd[d] && (direct[d].x | direct[d].y)) { r.mvc[numMvc][0] = (int16_t)direct[d].x; r.mvc[numMvc][1] = (int16_t)direct[d].y; numMvc++; }
        if (vi[d] && (indirect[d].x | indirect[d].y)) { r.mvc[numMvc][0] = (int16_t)indirect[d].x; r.mvc[numMvc][1] = (int16_t)indirect[d].y; numMvc++; }
    }
    if (num == 2 && c[0].x == c[1].x && c[0].y == c[1].y) num = 1;
    if (p.temporalMvp && num < 2 && t.nb[COLLOCATED].refIdx[list] != -1)
    {
        Mv2 m; m.x = t.nb[COLLOCATED].mv[list][0]; m.y = t.nb[COLLOCATED].mv[list][1];
        const Mv2 s = scale_poc(m, p.curPOC, curRefPOC, t.colPOC, t.colRefPOC);
        r.mvc[numMvc][0] = (int16_t)s.x; r.mvc[numMvc][1] = (int16_t)s.y; numMvc++;
        c[num++] = s;
    }
    while (num < 2) { c[num].x = 0; c[num].y = 0; num++; }
    r.amvp[0][0] = (int16_t)c[0].x; r.amvp[0][1] = (int16_t)c[0].y; r.amvp[1][0] = (int16_t)c[1].x; r.amvp[1][1] = (int16_t)c[1].y;
    r.numMvc = (int16_t)numMvc;
    return r;
}

} // namespace xh
