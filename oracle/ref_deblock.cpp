/*
 * ref_deblock.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * The reference's deblocking filter (Deblock::deblockCTU, common/deblock.cpp:37-497, calling primitives.pelFilterLumaStrong / pelFilterChroma,
 * common/loopfilter.cpp:136-232) compiled from its own sources (oracle/Makefile target "deblock", whole encoder, no asm) and run on its own objects:
 * a Frame whose FrameData holds one CUData per CTU (FrameData::create, CUData::initCTU), the per-partition arrays filled from a file, the reconstructed
 * PicYuv.  All vertical edges of the picture, then all horizontal edges (the order the standard defines; FrameFilter::ParallelFilter::processTasks,
 * encoder/framefilter.cpp:383-443, runs the same two passes CTU by CTU with a lag of one CTU).  4:2:0, one slice.
 *
 * usage: x265deblock_<depth> <width> <height> <ctu> <in.bin> <out.bin> <sliceType 0=B 1=P> <betaOffsetDiv2> <tcOffsetDiv2> <cbQpOffset> <crQpOffset> <tqBypassEnabled>
 *   in.bin : planes Y, Cb, Cr (pixels, tightly packed); then per CTU, in z-scan order of its 4x4 partitions, the arrays
 *            log2CUSize, cuDepth, partSize, tuDepth, predMode, cbf[luma], tqBypass (uint8), qp, refIdx[0], refIdx[1] (int8), mv[0], mv[1] (int32 x, y);
 *            then int32 refPic[2][16] = an identifier of the picture behind every (list, refIdx)
 *   out.bin: planes Y, Cb, Cr after deblocking, uint16 samples
 */
#include "common.h"
#include "primitives.h"
#include "picyuv.h"
#include "frame.h"
#include "framedata.h"
#include "slice.h"
#include "cudata.h"
#include "deblock.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace X265_NS;

template<typename T> static bool rd(FILE* f, T* dst, size_t n) { return fread(dst, sizeof(T), n, f) == n; }

int main(int argc, char** argv)
{
    if (argc < 12) { fprintf(stderr, "usage: %s width height ctu in.bin out.bin sliceP betaDiv2 tcDiv2 cbOff crOff bypass\n", argv[0]); return 2; }
    const int W = atoi(argv[1]), H = atoi(argv[2]), ctu = atoi(argv[3]);
    x265_param* p = x265_param_alloc();
    x265_param_default_preset(p, "medium", NULL);
    const int csp = getenv("X265REF_CSP") ? atoi(getenv("X265REF_CSP")) : X265_CSP_I420;      /* 1 = 4:2:0, 2 = 4:2:2, 3 = 4:4:4 */
    const int hs = csp == X265_CSP_I444 ? 0 : 1, vs = csp == X265_CSP_I420 ? 1 : 0;
    p->sourceWidth = W; p->sourceHeight = H; p->internalCsp = csp; p->maxCUSize = ctu; p->minCUSize = 8;
    p->maxLog2CUSize = ctu == 64 ? 6 : ctu == 32 ? 5 : 4; p->unitSizeDepth = p->maxLog2CUSize - 2;       /* Encoder::configure */
    p->num4x4Partitions = (ctu >> 2) * (ctu >> 2);
    x265_setup_primitives(p);
    static SPS sps; static PPS pps;
    memset(&sps, 0, sizeof(sps)); memset(&pps, 0, sizeof(pps));
    sps.numCuInWidth = (W + ctu - 1) / ctu; sps.numCuInHeight = (H + ctu - 1) / ctu; sps.numCUsInFrame = sps.numCuInWidth * sps.numCuInHeight;
    sps.numPartitions = p->num4x4Partitions; sps.numPartInCUSize = ctu >> 2; sps.chromaFormatIdc = csp;
    pps.deblockingFilterBetaOffsetDiv2 = atoi(argv[7]); pps.deblockingFilterTcOffsetDiv2 = atoi(argv[8]);
    pps.chromaQpOffset[0] = atoi(argv[9]); pps.chromaQpOffset[1] = atoi(argv[10]); pps.bTransquantBypassEnabled = atoi(argv[11]) != 0;

    Frame frame;
    frame.m_param = p;
    frame.m_reconPic[0] = new PicYuv;
    if (!frame.m_reconPic[0]->create(p, true, NULL) || !frame.m_reconPic[0]->createOffsets(sps)) { fprintf(stderr, "PicYuv::create failed\n"); return 2; }
    FILE* in = fopen(argv[4], "rb"); FILE* out = fopen(argv[5], "wb");
    if (!in || !out) { fprintf(stderr, "cannot open files\n"); return 2; }
    PicYuv* pic = frame.m_reconPic[0];
    for (int c = 0; c < 3; c++)
    {
        const int w = c ? W >> hs : W, h = c ? H >> vs : H;
        const intptr_t st = c ? pic->m_strideC : pic->m_stride;
        for (int y = 0; y < h; y++)
            if (!rd(in, pic->m_picOrg[c] + (intptr_t)y * st, w)) { fprintf(stderr, "short input (planes)\n"); return 2; }
    }
    FrameData* fd = frame.m_encData = new FrameData;
    if (!fd->create(*p, sps, p->internalCsp)) { fprintf(stderr, "FrameData::create failed\n"); return 2; }
    fd->m_reconPic[0] = pic;
    Slice* slice = fd->m_slice;
    slice->m_sps = &sps; slice->m_pps = &pps; slice->m_param = p; slice->m_sliceType = atoi(argv[6]) ? P_SLICE : B_SLICE;
    const uint32_t np = p->num4x4Partitions;
    std::vector<char> sliceStart(sps.numCuInHeight + 1, 0);
    if (const char* e = getenv("X265REF_SLICE_ROWS"))
        for (const char* q = e; *q; ) { char* end; long r = strtol(q, &end, 10); if (end == q) break; if (r > 0 && r < (long)sps.numCuInHeight) sliceStart[r] = 1; q = *end ? end + 1 : end; }
    for (uint32_t a = 0; a < sps.numCUsInFrame; a++)
    {
        CUData& c = fd->m_picCTU[a];
        /* X265REF_SLICE_ROWS=r1,r2,...: CTU rows that begin a slice (--slices: FrameEncoder::processRowEncoder hands the same three flags to CUData::initCTU) */
        const uint32_t row = a / sps.numCuInWidth, col = a % sps.numCuInWidth;
        const bool first = row == 0 || sliceStart[row], last = row == sps.numCuInHeight - 1 || sliceStart[row + 1];
        c.initCTU(frame, a, 30, first, last, last && col == sps.numCuInWidth - 1);
        std::vector<int32_t> mv(2 * np);
        bool ok = rd(in, c.m_log2CUSize, np) && rd(in, c.m_cuDepth, np) && rd(in, c.m_partSize, np) && rd(in, c.m_tuDepth, np) && rd(in, c.m_predMode, np) &&
                  rd(in, c.m_cbf[0], np) && rd(in, c.m_tqBypass, np) && rd(in, c.m_qp, np) && rd(in, c.m_refIdx[0], np) && rd(in, c.m_refIdx[1], np);
        for (int l = 0; l < 2 && ok; l++)
        {
            ok = rd(in, mv.data(), 2 * np);
            for (uint32_t i = 0; i < np; i++) { c.m_mv[l][i].x = mv[2 * i]; c.m_mv[l][i].y = mv[2 * i + 1]; }
        }
        if (!ok) { fprintf(stderr, "short input (CTU %u)\n", a); return 2; }
    }
    int32_t refPic[2][16];
    if (!rd(in, &refPic[0][0], 32)) { fprintf(stderr, "short input (refPic)\n"); return 2; }
    static char anchors[4096];                                     /* only compared, never dereferenced */
    for (int l = 0; l < 2; l++) for (int i = 0; i < 16; i++) slice->m_refFrameList[l][i] = (Frame*)(void*)&anchors[refPic[l][i] & 4095];

    std::vector<CUGeom> geoms((size_t)sps.numCUsInFrame * CUGeom::MAX_GEOMS);
    for (uint32_t a = 0; a < sps.numCUsInFrame; a++)
    {
        const uint32_t x = (a % sps.numCuInWidth) * ctu, y = (a / sps.numCuInWidth) * ctu;
        CUData::calcCTUGeoms(X265_MIN((uint32_t)ctu, W - x), X265_MIN((uint32_t)ctu, H - y), ctu, 8, &geoms[(size_t)a * CUGeom::MAX_GEOMS]);
    }
    for (int dir = 0; dir < 2; dir++)
        for (uint32_t a = 0; a < sps.numCUsInFrame; a++)
            Deblock::deblockCTU(&fd->m_picCTU[a], geoms[(size_t)a * CUGeom::MAX_GEOMS], dir);

    std::vector<uint16_t> line;
    for (int c = 0; c < 3; c++)
    {
        const int w = c ? W >> hs : W, h = c ? H >> vs : H;
        const intptr_t st = c ? pic->m_strideC : pic->m_stride;
        line.resize(w);
        for (int y = 0; y < h; y++) { for (int x = 0; x < w; x++) line[x] = pic->m_picOrg[c][(intptr_t)y * st + x]; fwrite(line.data(), 2, w, out); }
    }
    fclose(out); fclose(in);
    return 0;
}
