/*
 * ref_sao.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * The reference's per-CTU SAO statistics (SAO::calcSaoStatsCTU, encoder/sao.cpp:729-905), compiled from its own sources (oracle/Makefile
 * target "sao", whole encoder, no asm) and driven on its own objects: a Frame with source / reconstructed PicYuv and the CTU records
 * (position, first / last row flags) the function reads.  It pins the frame-level restatement xo_sao_stats_frame (region rules of
 * every offset class) and through it the HIP batch x265hip_sao_stats_frame.  Luma plane, one slice, deblocked statistics (the default).
 *
 * usage: x265sao_<depth> <width> <height> <ctu> <in.raw> <out.bin> [sao-non-deblock 0|1|2] [planes 1|3]
 *   sao-non-deblock 2: the statistics SAO::calcSaoStatsCu_BeforeDblk collects (sao.cpp:908-1207) -- the bottom / right border of every CTU, on the picture as it is BEFORE
 *   deblocking (in.raw's reconstructed planes are taken as that picture); same output layout
 *   in.raw  : per plane (Y, then Cb, Cr of a 4:2:0 picture when planes = 3) the source plane then the reconstructed plane, tightly packed
 *   out.bin : per plane, per CTU 5 x 32 int32 offsetOrg then 5 x 32 int32 count (types SAO_EO_0..3, SAO_BO as in sao.h:43-50)
 * apply mode (argv[8] = params.bin: per plane, per CTU 6 int32 = typeIdx (-1 = off), bandPos, offset[4]): the SAO of the picture (luma, and Cb / Cr through
 *   SAO::generateChromaOffsets, sao.cpp:626-730, when planes = 3) the way the frame filter runs
 *   it -- SAO::generateLumaOffsets CTU by CTU in raster order (sao.cpp:566-623 -> applyPixelOffsets :268-563), m_tmpU holding the unmodified last row
 *   of the CTU row above (FrameFilter::ParallelFilter::copySaoAboveRef, framefilter.cpp:303-311); out.bin = the reconstructed plane(s) afterwards (int32 per pixel)
 */
#include "common.h"
#include "primitives.h"
#include "picyuv.h"
#include "frame.h"
#include "framedata.h"
#include "slice.h"
#include "cudata.h"
#include "sao.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace X265_NS;

struct SaoX : public SAO
{
    void clear() { memset(m_count, 0, sizeof(m_count)); memset(m_offsetOrg, 0, sizeof(m_offsetOrg)); }
    const int32_t* counts() const { return &m_count[0][0][0]; }
    const int32_t* sums() const { return &m_offsetOrg[0][0][0]; }
    const int32_t* preCounts(int addr) const { return &m_countPreDblk[addr][0][0][0]; }
    const int32_t* preSums(int addr) const { return &m_offsetOrgPreDblk[addr][0][0][0]; }
    pixel* aboveRow(int plane = 0) { return m_tmpU[plane]; }
};

int main(int argc, char** argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s width height ctu in.raw out.bin [sao-non-deblock]\n", argv[0]); return 2; }
    const int W = atoi(argv[1]), H = atoi(argv[2]), ctu = atoi(argv[3]);
    x265_param* p = x265_param_alloc();
    x265_param_default_preset(p, "medium", NULL);
    const int nplanes = argc > 7 ? atoi(argv[7]) : 1;
    p->sourceWidth = W; p->sourceHeight = H; p->internalCsp = nplanes == 3 ? (getenv("X265REF_CSP") ? atoi(getenv("X265REF_CSP")) : X265_CSP_I420) : X265_CSP_I400;      /* X265REF_CSP: 1 = 4:2:0, 2 = 4:2:2, 3 = 4:4:4 */ p->maxCUSize = ctu;
    p->maxLog2CUSize = ctu == 64 ? 6 : ctu == 32 ? 5 : 4; p->unitSizeDepth = p->maxLog2CUSize - 2;       /* Encoder::configure */
    const int statMode = argc > 6 ? atoi(argv[6]) : 0;
    p->bSaoNonDeblocked = statMode ? 1 : 0; p->bLimitSAO = 0;
    x265_setup_primitives(p);
    SPS sps; memset(&sps, 0, sizeof(sps));
    sps.numCuInWidth = (W + ctu - 1) / ctu; sps.numCuInHeight = (H + ctu - 1) / ctu; sps.numCUsInFrame = sps.numCuInWidth * sps.numCuInHeight;

    Frame frame;
    frame.m_param = p;
    frame.m_fencPic = new PicYuv; frame.m_reconPic[0] = new PicYuv;
    if (!frame.m_fencPic->create(p, true, NULL) || !frame.m_reconPic[0]->create(p, true, NULL)) { fprintf(stderr, "PicYuv::create failed\n"); return 2; }
    if (!frame.m_fencPic->createOffsets(sps) || !frame.m_reconPic[0]->createOffsets(sps)) { fprintf(stderr, "createOffsets failed\n"); return 2; }
    FILE* in = fopen(argv[4], "rb"); FILE* out = fopen(argv[5], "wb");
    if (!in || !out) { fprintf(stderr, "cannot open files\n"); return 2; }
    PicYuv* pics[2] = { frame.m_fencPic, frame.m_reconPic[0] };
    for (int k = 0; k < 2; k++)
    {   /* the allocations are larger than the picture: clear them first */
        PicYuv* pic = pics[k];
        memset(pic->m_picBuf[0], 0, sizeof(pixel) * pic->m_stride * (sps.numCuInHeight * ctu + 2 * pic->m_lumaMarginY));
        for (int c = 1; c < nplanes; c++)
            memset(pic->m_picBuf[c], 0, sizeof(pixel) * pic->m_strideC * (((sps.numCuInHeight * ctu) >> pic->m_vChromaShift) + 2 * pic->m_chromaMarginY));
    }
    for (int plane = 0; plane < nplanes; plane++)
        for (int k = 0; k < 2; k++)
        {
            PicYuv* pic = pics[k];
            const int w = plane ? W >> pic->m_hChromaShift : W, h = plane ? H >> pic->m_vChromaShift : H;
            const intptr_t st = plane ? pic->m_strideC : pic->m_stride;
            for (int y = 0; y < h; y++)
                if (fread(pic->m_picOrg[plane] + (intptr_t)y * st, sizeof(pixel), w, in) != (size_t)w) { fprintf(stderr, "short input\n"); return 2; }
        }
    FrameData* fd = frame.m_encData = new FrameData;
    fd->m_slice = new Slice; fd->m_slice->m_sliceType = P_SLICE;
    fd->m_picCTU = new CUData[sps.numCUsInFrame];
    /* X265REF_SLICE_ROWS=r1,r2,...: CTU rows that begin a slice (--slices; CUData::initCTU's firstRowInSlice / lastRowInSlice, cudata.cpp:290-291) */
    std::vector<char> sliceStart(sps.numCuInHeight + 1, 0);
    if (const char* e = getenv("X265REF_SLICE_ROWS"))
        for (const char* q = e; *q; ) { char* end; long r = strtol(q, &end, 10); if (end == q) break; if (r > 0 && r < (long)sps.numCuInHeight) sliceStart[r] = 1; q = *end ? end + 1 : end; }
    for (uint32_t a = 0; a < sps.numCUsInFrame; a++)
    {
        CUData& c = fd->m_picCTU[a];
        c.m_cuPelX = (a % sps.numCuInWidth) * ctu; c.m_cuPelY = (a / sps.numCuInWidth) * ctu;
        const uint32_t row = a / sps.numCuInWidth;
        c.m_bFirstRowInSlice = row == 0 || sliceStart[row]; c.m_bLastRowInSlice = row == sps.numCuInHeight - 1 || sliceStart[row + 1];
    }
    SaoX sao;
    if (!sao.create(p, 1)) { fprintf(stderr, "SAO::create failed\n"); return 2; }
    sao.m_frame = &frame;
    if (argc > 8)
    {
        FILE* pf = fopen(argv[8], "rb");
        std::vector<int32_t> raw(6 * (size_t)sps.numCUsInFrame * nplanes);
        if (!pf || fread(raw.data(), 4, raw.size(), pf) != raw.size()) { fprintf(stderr, "bad params file\n"); return 2; }
        fclose(pf);
        std::vector<SaoCtuParam> prm[3];
        for (int c = 0; c < nplanes; c++)
        {
            prm[c].resize(sps.numCUsInFrame);
            for (uint32_t a = 0; a < sps.numCUsInFrame; a++)
            {
                const int32_t* r = &raw[6 * ((size_t)c * sps.numCUsInFrame + a)];
                prm[c][a].reset(); prm[c][a].typeIdx = r[0]; prm[c][a].bandPos = (uint32_t)r[1];
                for (int i = 0; i < 4; i++) prm[c][a].offset[i] = r[2 + i];
            }
        }
        PicYuv* rp = frame.m_reconPic[0];
        std::vector<pixel> pristine[3];
        for (int c = 0; c < nplanes; c++)
        {
            const int w = c ? W >> rp->m_hChromaShift : W, h = c ? H >> rp->m_vChromaShift : H;
            const intptr_t st = c ? rp->m_strideC : rp->m_stride;
            pristine[c].resize((size_t)st * h);
            for (int y = 0; y < h; y++) memcpy(&pristine[c][(size_t)y * st], rp->m_picOrg[c] + (intptr_t)y * st, w * sizeof(pixel));
        }
        SaoCtuParam* prm3[3] = { prm[0].data(), nplanes == 3 ? prm[1].data() : NULL, nplanes == 3 ? prm[2].data() : NULL };
        for (uint32_t row = 0; row < sps.numCuInHeight; row++)
        {
            /* copySaoAboveRef: the unmodified row above the CTU row -- for the first CTU row its own first row (framefilter.cpp:303-325) */
            for (int c = 0; c < nplanes; c++)
            {
                const int w = c ? W >> rp->m_hChromaShift : W, ch = c ? ctu >> rp->m_vChromaShift : ctu;
                const intptr_t st = c ? rp->m_strideC : rp->m_stride;
                memcpy(sao.aboveRow(c), &pristine[c][(size_t)(row ? row * ch - 1 : 0) * st], w * sizeof(pixel));
            }
            for (uint32_t col = 0; col < sps.numCuInWidth; col++)
            {
                sao.generateLumaOffsets(prm3[0], (int)row, (int)col);
                if (nplanes == 3) sao.generateChromaOffsets(prm3, (int)row, (int)col);
            }
        }
        for (int c = 0; c < nplanes; c++)
        {
            const int w = c ? W >> rp->m_hChromaShift : W, h = c ? H >> rp->m_vChromaShift : H;
            const intptr_t st = c ? rp->m_strideC : rp->m_stride;
            std::vector<int32_t> o((size_t)w * h);
            for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) o[(size_t)y * w + x] = rp->m_picOrg[c][(intptr_t)y * st + x];
            fwrite(o.data(), 4, o.size(), out);
        }
        fclose(out); fclose(in);
        return 0;
    }
    if (statMode == 2)
    {
        for (uint32_t a = 0; a < sps.numCUsInFrame; a++) sao.calcSaoStatsCu_BeforeDblk(&frame, (int)(a % sps.numCuInWidth), (int)(a / sps.numCuInWidth));
        for (int plane = 0; plane < nplanes; plane++)
            for (uint32_t a = 0; a < sps.numCUsInFrame; a++)
            {
                fwrite(sao.preSums((int)a) + plane * 5 * 32, 4, 5 * 32, out);
                fwrite(sao.preCounts((int)a) + plane * 5 * 32, 4, 5 * 32, out);
            }
        fclose(out); fclose(in);
        return 0;
    }
    for (int plane = 0; plane < nplanes; plane++)
        for (uint32_t a = 0; a < sps.numCUsInFrame; a++)
        {
            sao.clear();
            sao.calcSaoStatsCTU((int)a, plane);
            fwrite(sao.sums() + plane * 5 * 32, 4, 5 * 32, out);
            fwrite(sao.counts() + plane * 5 * 32, 4, 5 * 32, out);
        }
    fclose(out); fclose(in);
    return 0;
}
