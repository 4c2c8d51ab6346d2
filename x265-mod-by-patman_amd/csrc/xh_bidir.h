// xh_bidir.h -- the bidirectional candidate's distortion (reference encoder/search.cpp:436-446): predInterLumaPixel of two references (blocks of their phase planes),
// pixelavg_pp, SATD against the cached source PU.  One wavefront per PU.
#pragma once
#include "xh_mc.h"

namespace xh {

// SATD of the cached source PU (LDS, stride w) against the average of two predictions (planes of reference a / b at quarter-pel MVs)
__device__ __forceinline__ int bidir_satd_core(int w, int h, intptr_t rs, int64_t planeElems, const lpixel* fenc, lpixel* avg, int refOff, const pixel* pa, int ax, int ay, const pixel* pb, int bx, int by, int lane)
{
    const int qpr = w >> 2, nquads = qpr * h;
    const pixel* s0 = pa + (int64_t)((ay & 3) * 4 + (ax & 3)) * planeElems + refOff + (intptr_t)(ay >> 2) * rs + (ax >> 2);
    const pixel* s1 = pb + (int64_t)((by & 3) * 4 + (bx & 3)) * planeElems + refOff + (intptr_t)(by >> 2) * rs + (bx >> 2);
    for (int q = lane; q < nquads; q += 64)
    {
        const int y = q / qpr, x4 = (q - y * qpr) * 4;
        int u[4], v[4], o[4];
        load4u(s0 + (intptr_t)y * rs + x4, u); load4u(s1 + (intptr_t)y * rs + x4, v);
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = (u[e] + v[e] + 1) >> 1;             // pixelavg_pp (pixel.cpp:375-388)
        store4(avg + y * w + x4, o);
    }
    wave_sync();
    const bool use4 = w == 4 || w == 12;
    const int uw = use4 ? 4 : 8, ux = w / uw, nunits = ux * (h >> 2);
    int s = 0;
    LView pv; pv.p = avg; pv.s = w;
    for (int u = lane; u < nunits; u += 64)
    {
        const int uy = u / ux, x0 = (u - uy * ux) * uw, y0 = uy * 4;
        const lpixel* f = fenc + y0 * w + x0;
        int d[16], t = 0;
#pragma unroll
        for (int half = 0; half < 2; half++)
        {
            if (half && use4) break;
#pragma unroll
            for (int yy = 0; yy < 4; yy++)
            {
                int p[4], r[4]; load4(f + half * 4 + yy * w, p); load4u(pv.at(x0 + half * 4, y0 + yy), r);
                const int a0 = p[0] - r[0], a1 = p[1] - r[1], a2 = p[2] - r[2], a3 = p[3] - r[3];
                const int t0 = a0 + a1, t1 = a0 - a1, t2 = a2 + a3, t3 = a2 - a3;
                d[4 * yy] = t0 + t2; d[4 * yy + 2] = t0 - t2; d[4 * yy + 1] = t1 + t3; d[4 * yy + 3] = t1 - t3;
            }
#pragma unroll
            for (int x = 0; x < 4; x++)
            {
                const int t0 = d[x] + d[4 + x], t1 = d[x] - d[4 + x], t2 = d[8 + x] + d[12 + x], t3 = d[8 + x] - d[12 + x];
                t += abs(t0 + t2) + abs(t0 - t2) + abs(t1 + t3) + abs(t1 - t3);
            }
        }
        s += t >> 1;                                                            // satd4: per 4x4; satd8: per 8x4 (pixel.cpp:262-289)
    }
    wave_sync();
    return wsum_u(s);
}

} // namespace xh
