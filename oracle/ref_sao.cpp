/*
 * ref_sao.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * The reference's per-CTU SAO statistics (SAO::calcSaoStatsCTU, encoder/sao.cpp:729-905), compiled from its own sources (oracle/Makefile
 * target "sao", whole encoder, no asm) and driven on its own objects: a Frame with source / reconstructed PicYuv and the CTU records
 * (position, first / last row flags) the function reads.  It pins the frame-level restatement xo_sao_stats_frame (region rules of
 * every offset class) and through it the HIP batch x265hip_sao_stats_frame.  Luma plane, one slice, deblocked statistics (the default).
 *
 * usage: x265sao_<depth> <width> <height> <ctu> <in.raw> <out.bin> [sao-non-deblock 0|1] [planes 1|3]
 *   in.raw  : per plane (Y, then Cb, Cr of a 4:2:0 picture when planes = 3) the source plane then the reconstructed plane, tightly packed
 *   out.bin : per plane, per CTU 5 x 32 int32 offsetOrg then 5 x 32 int32 count (types SAO_EO_0..3, SAO_BO as in sao.h:43-50)
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
};

int main(int argc, char** argv)
{
    if (argc < 6) { fprintf(stderr, "usage: %s width height ctu in.raw out.bin [sao-non-deblock]\n", argv[0]); return 2; }
    const int W = atoi(argv[1]), H = atoi(argv[2]), ctu = atoi(argv[3]);
    x265_param* p = x265_param_alloc();
    x265_param_default_preset(p, "medium", NULL);
    const int nplanes = argc > 7 ? atoi(argv[7]) : 1;
    p->sourceWidth = W; p->sourceHeight = H; p->internalCsp = nplanes == 3 ? X265_CSP_I420 : X265_CSP_I400; p->maxCUSize = ctu;
    p->maxLog2CUSize = ctu == 64 ? 6 : ctu == 32 ? 5 : 4; p->unitSizeDepth = p->maxLog2CUSize - 2;       /* Encoder::configure */
    p->bSaoNonDeblocked = argc > 6 ? atoi(argv[6]) : 0; p->bLimitSAO = 0;
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
            const int w = plane ? W >> 1 : W, h = plane ? H >> 1 : H;
            const intptr_t st = plane ? pic->m_strideC : pic->m_stride;
            for (int y = 0; y < h; y++)
                if (fread(pic->m_picOrg[plane] + (intptr_t)y * st, sizeof(pixel), w, in) != (size_t)w) { fprintf(stderr, "short input\n"); return 2; }
        }
    FrameData* fd = frame.m_encData = new FrameData;
    fd->m_slice = new Slice; fd->m_slice->m_sliceType = P_SLICE;
    fd->m_picCTU = new CUData[sps.numCUsInFrame];
    for (uint32_t a = 0; a < sps.numCUsInFrame; a++)
    {
        CUData& c = fd->m_picCTU[a];
        c.m_cuPelX = (a % sps.numCuInWidth) * ctu; c.m_cuPelY = (a / sps.numCuInWidth) * ctu;
        c.m_bFirstRowInSlice = a < sps.numCuInWidth; c.m_bLastRowInSlice = a >= sps.numCUsInFrame - sps.numCuInWidth;
    }
    SaoX sao;
    if (!sao.create(p, 1)) { fprintf(stderr, "SAO::create failed\n"); return 2; }
    sao.m_frame = &frame;
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
