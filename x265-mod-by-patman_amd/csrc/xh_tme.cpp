// xh_tme.cpp -- host side of the ThreadedME producer: the order in which Analysis::deriveMVsForCTU / computeMVForPUs (reference encoder/analysis.cpp:161-306) visit the
// PUs of a CTU, as a flat schedule.  PU k of every CTU of a picture has the same partition shape, position inside the CTU, slot in the CTU's MEData table
// (slice->m_ctuMV, MAX_NUM_PUS_PER_CTU = 593 slots, threadedme.h) and neighbour slots, so a frame batch can be stepped through the schedule with one launch sequence
// per entry over all CTUs.
#include "xh_common.h"
#include "../../include/x265hip_frame.h"

namespace {

struct PuShape { int w, h, part, amp; };
// g_puLookup (threadedme.h:67-92); part numbers = enum PartSize (common.h): 2Nx2N 0, 2NxN 1, Nx2N 2, NxN 3, 2NxnU 4, 2NxnD 5, nLx2N 6, nRx2N 7
const PuShape k_lookup[24] = {
    { 8, 4, 1, 0 }, { 4, 8, 2, 0 }, { 8, 8, 0, 0 }, { 16, 4, 4, 1 }, { 16, 12, 5, 1 }, { 4, 16, 6, 1 }, { 12, 16, 7, 1 }, { 16, 8, 1, 0 }, { 8, 16, 2, 0 }, { 16, 16, 0, 0 },
    { 32, 8, 4, 1 }, { 32, 24, 5, 1 }, { 8, 32, 6, 1 }, { 24, 32, 7, 1 }, { 32, 16, 1, 0 }, { 16, 32, 2, 0 }, { 32, 32, 0, 0 }, { 64, 16, 4, 1 }, { 64, 48, 5, 1 },
    { 16, 64, 6, 1 }, { 48, 64, 7, 1 }, { 64, 32, 1, 0 }, { 32, 64, 2, 0 }, { 64, 64, 0, 0 } };

struct Builder
{
    int ctu, minCu, rect, amp, n, maxSteps;
    int start[129][8];
    x265hip_tme_step* out;

    void startIndices()
    {   // ThreadedME::initPuStartIdx (threadedme.cpp:86-107)
        int s = 0;
        for (const PuShape& p : k_lookup)
        {
            if (p.w > ctu || p.h > ctu) continue;
            const int iw = p.amp ? (p.w > p.h ? p.w : p.h) : p.w, ih = p.amp ? iw : p.h;
            const int num = (ctu / iw) * (ctu / ih);
            start[p.w + p.h][p.part] = s;
            s += p.amp ? 2 * num : num;
        }
    }
    void visit(int cuX, int cuY, int size)
    {   // Analysis::computeMVForPUs (analysis.cpp:161-246): the four sub-CUs first (z order), then every PU shape of this CU in g_puLookup's order
        if (size > minCu)
            for (int s = 0; s < 4; s++) visit(cuX + (s & 1) * (size >> 1), cuY + (s >> 1) * (size >> 1), size >> 1);
        for (const PuShape& p : k_lookup)
        {
            if (p.w > size || p.h > size || (p.w != size && p.h != size)) continue;
            if (!amp && p.amp) continue;
            if (!rect && p.w != p.h && !p.amp) continue;
            const int bw = p.amp ? (p.w > p.h ? p.w : p.h) : p.w, bh = p.amp ? bw : p.h;
            const int cols = ctu / bw, rows = ctu / bh;
            int puOffset = 0;
            if (p.amp) puOffset = rows * cols;
            else if (p.part == 1) puOffset = cols;
            else if (p.part == 2) puOffset = 1;
            const int col = cuX / bw, row = cuY / bh, st = start[p.w + p.h][p.part];
            if (n < maxSteps)
            {
                x265hip_tme_step& o = out[n];
                o.part = (int16_t)p.part; o.cuSize = (int16_t)size; o.cuX = (int16_t)cuX; o.cuY = (int16_t)cuY; o.puOffset = (int16_t)puOffset;
                o.finalIdx = (int16_t)(st + row * cols + col);
                o.neighbor[0] = (int16_t)(col > 0 ? st + row * cols + col - 1 : -1);                          // MD_LEFT
                o.neighbor[1] = (int16_t)(row > 0 ? st + (row - 1) * cols + col : -1);                        // MD_ABOVE
                o.neighbor[2] = (int16_t)(row > 0 && col < cols - 1 ? st + (row - 1) * cols + col + 1 : -1);  // MD_ABOVE_RIGHT
                o.neighbor[3] = -1;                                                                           // MD_BELOW_LEFT: never
                o.neighbor[4] = (int16_t)(row > 0 && col > 0 ? st + (row - 1) * cols + col - 1 : -1);         // MD_ABOVE_LEFT
                // the partitions (PredictionUnit, CUData::getPartIndexAndSize): 2Nx2N one; 2NxN / Nx2N two halves; AMP a quarter and three quarters
                o.numPart = (int16_t)(p.part == 0 ? 1 : 2);
                for (int k = 0; k < 2; k++) { o.pu[k][0] = (int16_t)cuX; o.pu[k][1] = (int16_t)cuY; o.pu[k][2] = (int16_t)size; o.pu[k][3] = (int16_t)size; }
                const int q = size >> 2, hlf = size >> 1;
                switch (p.part)
                {
                case 1: o.pu[0][3] = o.pu[1][3] = (int16_t)hlf; o.pu[1][1] = (int16_t)(cuY + hlf); break;
                case 2: o.pu[0][2] = o.pu[1][2] = (int16_t)hlf; o.pu[1][0] = (int16_t)(cuX + hlf); break;
                case 4: o.pu[0][3] = (int16_t)q; o.pu[1][3] = (int16_t)(size - q); o.pu[1][1] = (int16_t)(cuY + q); break;
                case 5: o.pu[0][3] = (int16_t)(size - q); o.pu[1][3] = (int16_t)q; o.pu[1][1] = (int16_t)(cuY + size - q); break;
                case 6: o.pu[0][2] = (int16_t)q; o.pu[1][2] = (int16_t)(size - q); o.pu[1][0] = (int16_t)(cuX + q); break;
                case 7: o.pu[0][2] = (int16_t)(size - q); o.pu[1][2] = (int16_t)q; o.pu[1][0] = (int16_t)(cuX + size - q); break;
                default: break;
                }
            }
            n++;
        }
    }
};

} // namespace

extern "C" int x265hip_tme_schedule(int ctuSize, int minCuSize, int rect, int amp, x265hip_tme_step* steps, int maxSteps)
{
    if ((ctuSize != 64 && ctuSize != 32 && ctuSize != 16) || minCuSize < 8 || minCuSize > ctuSize || (minCuSize & (minCuSize - 1)) || (maxSteps > 0 && !steps)) return X265HIP_EARG;
    Builder b{};
    b.ctu = ctuSize; b.minCu = minCuSize; b.rect = rect; b.amp = amp; b.n = 0; b.maxSteps = maxSteps; b.out = steps;
    b.startIndices();
    b.visit(0, 0, ctuSize);
    return b.n;
}
